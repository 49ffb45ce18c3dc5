"""Config C5 on one GPU: the per-frame pipeline of ONE sequence exactly as bench.py's C5 leg drives it (bench.SequenceRunner: extract ->
match against the previous frame -> ratio test, a local BA of a new window on every keyframe) against the CPU oracle, frame by
frame: keypoints / descriptors bit-exact, best / second / accepted match indices exact, BA residuals within 1e-5."""
import threading

import numpy as np
import pytest

import ba_synth

pytestmark = pytest.mark.gpu


def test_eight_frame_sequence_matches_oracle(oracle):
    import bench
    import synth
    F = 8
    g = synth.SequenceSynth(bench.W, bench.H, 2003, 2 * (F - 1), F - 1)
    frames = np.stack([g.frame(2 * i, i) for i in range(F)])
    windows = [ba_synth.make_problem_fast(50, 2000, 10, seed=9000 + k) for k in range(2)]      # C4 windows, the ones bench_c5 solves
    ocfg = oracle.cfg(levels=bench.LEVELS, scale_factor=bench.SCALE, max_kpts=bench.MAX_KPTS, fast_threshold=bench.FAST_THR)
    want_kp = [oracle.orb_extract(ocfg, frames[i]) for i in range(F)]
    seen = {"frames": 0, "ba": 0, "matches": 0}

    def on_frame(i, ex, bufs, ba_out):
        got = ex.download(0)
        want = want_kp[i]
        assert len(got["x"]) == len(want["x"]) > 1000
        for k in ("x", "y", "angle"):
            assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (i, k)
        assert np.array_equal(got["octave"], want["octave"]) and np.array_equal(got["desc"], want["desc"]), i
        if bufs is not None:
            n, cap = len(want["x"]), ex.capacity
            bi, bd, sd, match = bufs
            wi, wd, ws = oracle.hamming_best2(want["desc"], want_kp[i - 1]["desc"])
            assert np.array_equal(bi.download(np.int32, (cap,))[:n], wi), i
            assert np.array_equal(bd.download(np.uint16, (cap,))[:n], wd) and np.array_equal(sd.download(np.uint16, (cap,))[:n], ws), i
            # accept rule of keyframe_matcher.cpp:115-122: best <= 50 and ratio * second >= best (float32 product)
            ok = (wd <= 50) & (np.float32(bench.LOWE_RATIO) * ws.astype(np.float32) >= wd.astype(np.float32))
            m = match.download(np.int32, (cap,))
            assert np.array_equal(m[:n], np.where(ok, wi, -1)), i
            assert (m[n:] == -1).all()
            seen["matches"] += int(ok.sum())
        if ba_out is not None:
            p = windows[(i // 5) % 2]
            want_ba = oracle.ba_solve(p, 10, False)
            rg = ba_synth.residuals_fast(p, ba_out["pose"], ba_out["point"]); rw = ba_synth.residuals_fast(p, want_ba["pose"], want_ba["point"])
            assert np.abs(rg - rw).max() < 1e-5
            assert ba_out["stats"]["iters"] == want_ba["stats"]["iters"]
            seen["ba"] += 1
        seen["frames"] += 1

    start = threading.Event(); start.set()
    r = bench.SequenceRunner(0, 3, frames, windows, 5, start, on_frame=on_frame)
    r.start(); r.join()
    if r.error is not None:
        raise r.error
    assert seen["frames"] == F and r.frames_done == F and seen["ba"] == 2 and r.ba_done == 2        # keyframes 0 and 5
    assert seen["matches"] > 3000                                    # the sequence really matches frame to frame


def test_native_driver_counts_what_the_python_driver_counts():
    """bench.py --c5-native (tools/c5_native.cpp, C++ threads on the C ABI) against bench.SequenceRunner on the same two sequences: frames,
    bundle adjustments and the last frame's accepted matches."""
    import ctypes as C
    import bench
    import mi355slam
    import synth
    lib = bench._c5_native_lib()
    assert lib is not None, "slam-module_amd/lib/libc5native.so is missing (make -C slam-module_amd/csrc)"
    F, S, TOTAL = 6, 2, 9                  # 9 frames over 6 images: 0 1 2 3 4 5 4 3 2 (both drivers walk back)
    seqs = []
    for s in range(S):
        g = synth.SequenceSynth(bench.W, bench.H, 2100 + s, 2 * (F - 1), F - 1)
        seqs.append(np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(F)])))
    windows = [ba_synth.make_problem_fast(20, 500, 8, seed=80 + k) for k in range(2)]
    start = threading.Event(); start.set()
    runners = [bench.SequenceRunner(0, s, seqs[s], windows, 5, start, n_total=TOTAL) for s in range(S)]
    for r in runners:
        r.start()
    for r in runners:
        r.join()
        if r.error is not None:
            raise r.error
    structs, keep = zip(*[mi355slam._ba_struct(w, 10) for w in windows])
    warr = (mi355slam.BaProblemC * len(structs))(*structs)
    fptr = (C.c_void_p * S)(*[f.ctypes.data for f in seqs])
    lib.c5_prepare.restype = C.c_void_p
    job = lib.c5_prepare(0, S, TOTAL, F, bench.W, bench.H, fptr, warr, len(structs), 5, bench.LEVELS, C.c_float(bench.SCALE), bench.MAX_KPTS, bench.FAST_THR, C.c_float(bench.LOWE_RATIO))
    assert job
    secs, seq_s = C.c_double(), (C.c_double * S)()
    fd, bd, lm = (C.c_int32 * S)(), (C.c_int32 * S)(), (C.c_int32 * S)()
    err = C.create_string_buffer(512)
    rc = lib.c5_go(C.c_void_p(job), C.byref(secs), seq_s, fd, bd, lm, err, 512)
    assert rc == 0, err.value
    del keep
    assert list(fd) == [r.frames_done for r in runners] == [TOTAL] * S
    assert list(bd) == [r.ba_done for r in runners] == [2] * S
    assert list(lm) == [r.matches for r in runners] and min(lm) > 500
