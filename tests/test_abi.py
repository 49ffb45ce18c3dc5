"""CPU tests of the product library: it loads, exports every symbol the header declares, and its host-side
geometry agrees with the oracle.  No compute entry point is called (there is no GPU here and no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mi355slam.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", hdr)))


def test_library_loads_and_exports_every_declared_symbol():
    import mi355slam
    L = mi355slam.lib()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libmi355slam.so does not export %s" % n
    assert b"gfx950" in L.ms_version()


def test_native_c5_driver_is_built_on_the_c_abi_only():
    """tools/c5_native.cpp (bench.py --c5-native): built next to the library, exports its two entry points, and names no symbol outside the C ABI."""
    path = os.path.join(ROOT, "slam-module_amd", "lib", "libc5native.so")
    assert os.path.exists(path), "make -C slam-module_amd/csrc builds it"
    L = ctypes.CDLL(path)
    assert hasattr(L, "c5_prepare") and hasattr(L, "c5_go")
    src = open(os.path.join(ROOT, "tools", "c5_native.cpp")).read()
    assert '#include "mi355slam.h"' in src and "hip" not in src.lower().replace("mi355slam", "") and "oracle" not in src


def test_host_geometry_equals_oracle(oracle):
    import mi355slam
    for levels, f, w, h, k in [(8, 1.2, 1280, 720, 2000), (8, 1.2, 640, 480, 2000), (5, 1.5, 333, 222, 777), (1, 1.2, 100, 100, 50), (12, 1.1, 1920, 1080, 5000)]:
        assert np.array_equal(mi355slam.scale_factors(levels, f), oracle.scale_factors(levels, f))
        assert np.array_equal(mi355slam.level_sigma_sq(levels, f), oracle.level_sigma_sq(levels, f))
        assert np.array_equal(mi355slam.level_quotas(levels, f, k), oracle.level_quotas(levels, f, k))
        a, b = mi355slam.level_sizes(levels, f, w, h); c, d = oracle.level_sizes(levels, f, w, h)
        assert np.array_equal(a, c) and np.array_equal(b, d)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a gfx950 device the context cannot be created; nothing silently runs on the CPU."""
    import mi355slam
    import torch
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(mi355slam.MsError):
        mi355slam.Context(0)
    h = ctypes.c_void_p()
    assert mi355slam.lib().ms_ctx_create(0, ctypes.byref(h)) != 0 and not h.value


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "slam-module_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "mso.h" not in txt and "import mso" not in txt and "libmso" not in txt, os.path.join(dirpath, fn)


def test_bench_and_tools_touch_the_oracle_only_as_the_cpu_baseline():
    """bench.py may import oracle/ in its cpu_baseline leg only (its inputs come from the neutral tools/synth.py and tests/ba_synth.py);
    the input generators themselves never do."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("import mso") == 1 and "def _oracle():" in src
    body = src[src.index("def _oracle():"):]
    assert body.index("import mso") < body.index("def host_cores")                   # the one import sits inside _oracle()
    callers = [ln for ln in src.splitlines() if "_oracle()" in ln and "def " not in ln]
    assert callers and all("mso = _oracle()" in ln for ln in callers)
    for fn in ("cpu_baseline_frames", "cpu_baseline_ba", "cpu_baseline_greedy", "cpu_baseline_c3"):
        seg = src[src.index("def %s(" % fn):]
        assert "_oracle()" in seg[:seg.index("\ndef ", 5)]
    assert len(callers) == 4
    for rel in ("tools/synth.py", "tests/ba_synth.py"):
        txt = open(os.path.join(ROOT, rel)).read()
        assert "import mso" not in txt and "libmso" not in txt


def test_synthetic_frames_of_the_bench_equal_the_oracles_generator(oracle):
    """tools/synth.py (numpy, used by bench.py and the GPU tests) and oracle/frontend.c hold the same integer generator."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    for (w, h, seed, sx, sy) in [(640, 480, 1000, 0, 0), (1280, 720, 1003, 14, 7), (331, 257, 5, 2, 1), (200, 120, 77, 30, 40)]:
        assert np.array_equal(synth.synth_frame(w, h, seed, sx, sy), oracle.synth_frame(w, h, seed, sx, sy))
    seq = synth.synth_sequences(24, 320, 200, 50, n_seq=3)
    for i in (0, 7, 8, 23):
        assert np.array_equal(seq[i], oracle.synth_frame(320, 200, 50 + i // 8, 2 * (i % 8), i % 8))
    sparse = synth.synth_frame(640, 480, 1000, sparse=True)
    dense = synth.synth_frame(640, 480, 1000)
    assert (oracle.fast_score_map(sparse, 20) > 20).mean() < 0.4 * (oracle.fast_score_map(dense, 20) > 20).mean()


def test_angle_check_host_function(oracle):
    """A1 (match_angle_checker.h:60-134) is host arithmetic in the product too: compare with the oracle on the CPU."""
    import mi355slam
    rng = np.random.default_rng(4)
    for n in (0, 1, 50, 2000):
        d = np.concatenate([rng.normal(40, 10, n // 2), rng.uniform(-360, 720, n - n // 2)]).astype(np.float32)
        ids = rng.permutation(n).astype(np.int32)
        assert np.array_equal(mi355slam.angle_check(d, ids), oracle.angle_check(d, ids))


def test_keypoint_records_unpack_in_serialize_order():
    """KeyPoint::serialize (key_point.hpp:22-25): x, y, angle, octave, octave, bearing, descriptor = 76 packed bytes; host half of N4."""
    import struct
    import mi355slam
    rec = np.zeros((3, 76), np.uint8)
    for k in range(3):
        rec[k] = np.frombuffer(struct.pack("<fffii3d8I", 1.5 + k, 2.5, 33.25, 4 + k, 4 + k, 0.1, 0.2, 0.3 + k, *range(k, k + 8)), np.uint8)
    o = mi355slam.unpack_keypoints(rec)
    assert o["x"].tolist() == [1.5, 2.5, 3.5] and o["octave"].tolist() == [4, 5, 6] and o["bearing"][2].tolist() == [0.1, 0.2, 2.3]
    assert o["desc"][1].tolist() == list(range(1, 9)) and o["angle"][0] == 33.25
    rec[1, 16] ^= 1                                                    # second copy of octave differs: corrupt record
    with pytest.raises(mi355slam.MsError):
        mi355slam.unpack_keypoints(rec)


def test_prepare_process_sets_the_queue_count_once():
    """ms_prepare_process: GPU_MAX_HW_QUEUES = clamp(2 x contexts, 4, 32) unless the caller's environment already has a value; no HIP call involved."""
    import ctypes as C
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%r, "slam-module_amd"))
import mi355slam
L = mi355slam.lib()
libc = C.CDLL("libc.so.6"); libc.getenv.restype = C.c_char_p
assert L.ms_prepare_process(0) != 0
want = os.environ.get("GPU_MAX_HW_QUEUES")
assert L.ms_prepare_process(int(sys.argv[1])) == 0
print(libc.getenv(b"GPU_MAX_HW_QUEUES").decode())
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    run = lambda n, e: subprocess.run([sys.executable, "-c", code, str(n)], env=e, capture_output=True, text=True, check=True).stdout.strip()
    assert run(8, env) == "16" and run(2, env) == "4" and run(1, env) == "4" and run(40, env) == "32"
    assert run(8, dict(env, GPU_MAX_HW_QUEUES="2")) == "2"               # the caller's own setting wins


def test_prepare_process_says_when_it_is_too_late():
    """Once the GPU runtime of the process is up (the driver's device node is open) the variable has been read: MS_ERR_TOO_LATE, environment untouched.
    The child opens /dev/kfd itself, as the runtime would (no HIP call): where the node exists (the GPU box) the late half is checked, elsewhere the early half."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%r, "slam-module_amd"))
import mi355slam
L = mi355slam.lib()
fd = None
try:
    fd = os.open("/dev/kfd", os.O_RDWR)
except OSError:
    pass
rc = L.ms_prepare_process(8)
libc = C.CDLL("libc.so.6"); libc.getenv.restype = C.c_char_p
print(rc, libc.getenv(b"GPU_MAX_HW_QUEUES") is not None, fd is not None)
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    rc, was_set, opened = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    if opened == "True":
        assert int(rc) == mi355slam.MS_ERR_TOO_LATE and was_set == "False"
    else:
        assert int(rc) == 0 and was_set == "True"
