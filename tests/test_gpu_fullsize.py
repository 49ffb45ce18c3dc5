"""Parity at the sizes bench.py times (BASELINE configs C2 and C3), on bench.py's own inputs and call sequence: the oracle checks
sampled units bit for bit, size-independent properties cover every unit of the batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _assert_same_keypoints(got, want):
    assert len(got["x"]) == len(want["x"])
    for k in ("x", "y", "angle"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert np.array_equal(got["octave"], want["octave"]) and np.array_equal(got["track_id"], want["track_id"]) and np.array_equal(got["desc"], want["desc"])


def test_c2_batch_of_256_frames_and_its_256_pairs(oracle, ctx):
    """256 x 720p frames resident on the device and used in place (a 2.3 GB slab): frames 0, 127, 255 against the oracle; the
    pairing bench.py times (frame f against f-1, frame 0 against 255) for pairs 0<-255 and 128<-127; every frame of the batch
    against the same frame extracted alone; index / distance invariants on all 256 x capacity rows."""
    import bench
    import mi355slam
    import synth
    B, W, H = bench.BATCH, bench.W, bench.H
    frames = synth.synth_sequences(B, W, H, 1000)
    buf = ctx.upload(frames)
    ex = mi355slam.OrbExtractor(ctx, W, H, levels=bench.LEVELS, scale_factor=bench.SCALE, max_kpts=bench.MAX_KPTS, fast_threshold=bench.FAST_THR, max_batch=B)
    ex.extract(buf, n_frames=B, frame_stride=W * H, row_stride=W)
    cap, v = ex.capacity, ex.device_view()
    ocfg = oracle.cfg(levels=bench.LEVELS, scale_factor=bench.SCALE, max_kpts=bench.MAX_KPTS, fast_threshold=bench.FAST_THR)
    want = {f: oracle.orb_extract(ocfg, frames[f]) for f in (0, 127, 128, 255)}
    got_all = [ex.download(f) for f in range(B)]
    for f in (0, 127, 128, 255):
        _assert_same_keypoints(got_all[f], want[f])
    counts = np.array([len(g["x"]) for g in got_all])
    assert (counts > 1000).all() and (counts <= cap).all()
    # the batch is 256 independent units: frame f alone gives the same bits (every 16th frame + the last of each sequence)
    one = mi355slam.OrbExtractor(ctx, W, H, levels=bench.LEVELS, scale_factor=bench.SCALE, max_kpts=bench.MAX_KPTS, fast_threshold=bench.FAST_THR, max_batch=1)
    for f in sorted(set(range(0, B, 16)) | set(range(31, B, 32))):
        one.extract(buf.ptr + f * W * H, n_frames=1, frame_stride=W * H, row_stride=W)
        _assert_same_keypoints(one.download(0), got_all[f])
    # matching exactly as bench.Headline.step does it
    pq = np.arange(B, dtype=np.int32); pt = np.roll(pq, 1)
    dq, dt = ctx.upload(pq), ctx.upload(pt)
    bi, bd, sd, match = ctx.alloc(4 * B * cap), ctx.alloc(2 * B * cap), ctx.alloc(2 * B * cap), ctx.alloc(4 * B * cap)
    mi355slam.hamming_best2_sets(ctx, v.desc, cap, v.count, v.desc, cap, v.count, dq, dt, B, bi, bd, sd)
    mi355slam.ratio_test_device(ctx, bi, bd, sd, B * cap, bench.LOWE_RATIO, 50, match)
    ctx.sync()
    BI, BD, SD, M = (bi.download(np.int32, (B, cap)), bd.download(np.uint16, (B, cap)), sd.download(np.uint16, (B, cap)), match.download(np.int32, (B, cap)))
    for f in (0, 128):
        q, t = want[f], want[(f - 1) % B]
        wi, wd, ws = oracle.hamming_best2(q["desc"], t["desc"])
        n = len(q["x"])
        assert np.array_equal(BI[f, :n], wi) and np.array_equal(BD[f, :n], wd) and np.array_equal(SD[f, :n], ws), f
        ok = (wd <= 50) & (np.float32(bench.LOWE_RATIO) * ws.astype(np.float32) >= wd.astype(np.float32))
        assert np.array_equal(M[f, :n], np.where(ok, wi, -1)), f
    for f in range(B):                                               # invariants on every pair of the batch
        n, nt = counts[f], counts[(f - 1) % B]
        assert (BI[f, :n] >= 0).all() and (BI[f, :n] < nt).all() and (BD[f, :n] <= SD[f, :n]).all() and (SD[f, :n] <= 256).all()
        assert (BI[f, n:] == -1).all() and (BD[f, n:] == 256).all() and (SD[f, n:] == 256).all() and (M[f, n:] == -1).all()
        # a reported best distance is the distance to the reported index (spot check by popcount on the host)
        d = np.unpackbits((got_all[f]["desc"] ^ got_all[(f - 1) % B]["desc"][BI[f, :n]]).view(np.uint8), axis=1).sum(1)
        assert np.array_equal(d, BD[f, :n]), f
    inside = np.array([(M[f] >= 0).sum() for f in range(B) if f % 32])         # frames that follow a frame of the same sequence
    assert inside.min() > 300


def test_c3_thousand_pairs_of_2000_by_2000(oracle, ctx):
    """BASELINE config C3: 1000 independent (2000 x 2000) searches in one launch.  Oracle on 8 sampled pairs (bit-exact best index,
    best and second distance, ratio rule); on all 1000: the matrix-core kernel and the popcount kernel agree bit for bit, invariants
    hold, and the planted inliers (8 % flipped bits) are found."""
    import mi355slam
    import synth
    P, NQ, NT = 1000, 2000, 2000
    Q = np.empty((P, NQ, 8), np.uint32); T = np.empty((P, NT, 8), np.uint32); perms = []
    for p in range(P):
        Q[p], T[p], perm = synth.synth_descriptor_pair(p, NQ, NT)
        perms.append(perm)
    dq, dt = ctx.upload(Q), ctx.upload(T)
    outs = []
    for path in (0, 1):
        ctx.set_hamming_path(path)
        bi, bd, sd = ctx.alloc(4 * P * NQ), ctx.alloc(2 * P * NQ), ctx.alloc(2 * P * NQ)
        ctx.check(mi355slam.lib().ms_hamming_best2(ctx._h, mi355slam._vp(dq), NQ, mi355slam._vp(dt), NT, P, None, None, None,
                                                   mi355slam._vp(bi), mi355slam._vp(bd), mi355slam._vp(sd)), "ms_hamming_best2")
        ctx.sync()
        outs.append((bi.download(np.int32, (P, NQ)), bd.download(np.uint16, (P, NQ)), sd.download(np.uint16, (P, NQ))))
    ctx.set_hamming_path(0)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    BI, BD, SD = outs[0]
    assert (BI >= 0).all() and (BI < NT).all() and (BD <= SD).all() and (SD <= 256).all()
    m = mi355slam.ratio_test(ctx, BI.reshape(-1), BD.reshape(-1), SD.reshape(-1), 0.75).reshape(P, NQ)
    for p in (0, 1, 127, 128, 499, 500, 998, 999):
        wi, wd, ws = oracle.hamming_best2(Q[p], T[p])
        assert np.array_equal(BI[p], wi) and np.array_equal(BD[p], wd) and np.array_equal(SD[p], ws), p
        ok = (wd <= 50) & (np.float32(0.75) * ws.astype(np.float32) >= wd.astype(np.float32))
        assert np.array_equal(m[p], np.where(ok, wi, -1)), p
    found = 0
    for p in range(P):                                               # target row j is a noisy copy of query perm[j]
        found += int((BI[p, perms[p]] == np.arange(len(perms[p]))).sum())
        assert (m[p, perms[p]] >= 0).mean() > 0.95
    assert found >= 0.999 * P * 1400
