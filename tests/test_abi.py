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
    if torch.cuda.is_available():
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


def test_angle_check_host_function(oracle):
    """A1 (match_angle_checker.h:60-134) is host arithmetic in the product too: compare with the oracle on the CPU."""
    import mi355slam
    rng = np.random.default_rng(4)
    for n in (0, 1, 50, 2000):
        d = np.concatenate([rng.normal(40, 10, n // 2), rng.uniform(-360, 720, n - n // 2)]).astype(np.float32)
        ids = rng.permutation(n).astype(np.int32)
        assert np.array_equal(mi355slam.angle_check(d, ids), oracle.angle_check(d, ids))
