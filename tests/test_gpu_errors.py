"""Error convention of the C ABI: bad arguments give a negative status + message, never a crash or a silent fallback."""
import ctypes as C

import numpy as np
import pytest

import ba_synth

pytestmark = pytest.mark.gpu


def test_orb_create_rejects_unsupported_configs(ctx):
    import mi355slam
    for kw in (dict(levels=0), dict(levels=17), dict(scale_factor=1.0), dict(fast_threshold=0), dict(max_kpts=0),
               dict(lk_track_level=8), dict(width=60, height=60, levels=4),            # level 3 would be 35x35 < 40x40
               dict(max_kpts=30000, levels=1),                                          # quota above the 4096-key LDS sort
               dict(width=5000, height=4000)):                                          # > 2^24 pixels: raster index no longer fits the key
        base = dict(width=640, height=480)
        base.update(kw)
        with pytest.raises(mi355slam.MsError) as e:
            mi355slam.OrbExtractor(ctx, **base)
        assert "ms_orb_create" in str(e.value)


def test_extract_argument_checks(ctx, oracle):
    import mi355slam
    ex = mi355slam.OrbExtractor(ctx, 640, 480, max_batch=2)
    imgs = np.stack([oracle.synth_frame(640, 480, 5 + i) for i in range(3)])
    with pytest.raises(mi355slam.MsError, match="n_frames"):
        ex.extract(imgs)                                                     # 3 frames > max_batch 2
    buf = ctx.upload(imgs[:1])
    with pytest.raises(mi355slam.MsError, match="strides"):
        ex.extract(buf, n_frames=1, frame_stride=640 * 480, row_stride=600)  # row stride shorter than the row
    with pytest.raises(mi355slam.MsError):
        ex.download(0)                                                       # nothing extracted yet
    ex.extract(imgs[:2])
    assert len(ex.download(1)["x"]) > 100
    with pytest.raises(mi355slam.MsError):
        ex.download(2)


def test_hamming_argument_checks(ctx):
    import mi355slam
    L = mi355slam.lib()
    buf = ctx.alloc(4096)
    rc = L.ms_hamming_best2(ctx._h, C.c_void_p(buf.ptr + 4), 4, C.c_void_p(buf.ptr), 4, 1, None, None, None,
                            C.c_void_p(buf.ptr), C.c_void_p(buf.ptr), C.c_void_p(buf.ptr))
    assert rc == -1 and b"aligned" in L.ms_last_error(ctx._h)
    rc = L.ms_hamming_best2(ctx._h, C.c_void_p(buf.ptr), 4, C.c_void_p(buf.ptr), 4, 1, C.c_void_p(buf.ptr), None, None,
                            C.c_void_p(buf.ptr), C.c_void_p(buf.ptr), C.c_void_p(buf.ptr))
    assert rc == -1 and b"bucket" in L.ms_last_error(ctx._h)
    assert L.ms_hamming_best2(ctx._h, None, 4, None, 4, 1, None, None, None, None, None, None) == -1


def test_ba_capacity_and_index_checks(ctx):
    import mi355slam
    big = ba_synth.make_problem(2060, 30, 3, seed=1, yaw_total=0.005, z_drift=0.0005)                         # 2060 free poses > 2048
    with pytest.raises(mi355slam.MsError, match="free poses"):
        mi355slam.BundleAdjuster(ctx, [big])
    bad = ba_synth.make_problem(5, 20, 3, seed=2)
    bad["obs_point"] = bad["obs_point"].copy(); bad["obs_point"][3] = 999
    with pytest.raises(mi355slam.MsError, match="outside"):
        mi355slam.BundleAdjuster(ctx, [bad])
    small = mi355slam.BundleAdjuster(ctx, [bad | dict(obs_point=ba_synth.make_problem(5, 20, 3, seed=2)["obs_point"])])
    with pytest.raises(mi355slam.MsError):
        small.set_factor_team(65)                                            # more workgroups than a team can have
    small.set_factor_team(3); small.set_team(2); small.solve()               # a factor team on a small system is ignored
    assert small.download(0)["stats"]["chi2_final"] <= small.download(0)["stats"]["chi2_init"]
    small.close()
    ok = ba_synth.make_problem(176, 400, 8, seed=3)                          # exactly at the limit works
    ba = mi355slam.BundleAdjuster(ctx, [ok], max_iters=3); ba.solve()
    out = ba.download(0)
    assert out["stats"]["chi2_final"] < out["stats"]["chi2_init"]


def test_ba_degenerate_problems(ctx, oracle):
    """No observations / every pose fixed / zero iterations: defined results, no hang."""
    import mi355slam
    p = ba_synth.make_problem(4, 10, 3, seed=4)
    allfixed = dict(p); allfixed["pose_fixed"] = np.ones(4, np.uint8)
    none = dict(p); none["obs_pose"] = p["obs_pose"][:0]; none["obs_point"] = p["obs_point"][:0]; none["obs_uv"] = p["obs_uv"][:0]; none["obs_info"] = p["obs_info"][:0]
    ba = mi355slam.BundleAdjuster(ctx, [allfixed, none], max_iters=5); ba.solve()
    a, b = ba.download(0), ba.download(1)
    wa = oracle.ba_solve(allfixed, 5, False)
    assert np.array_equal(a["pose"], p["pose"])                               # fixed poses untouched, points still refined
    assert abs(a["stats"]["chi2_final"] - wa["stats"]["chi2_final"]) <= 1e-8 * max(wa["stats"]["chi2_final"], 1.0)
    assert np.isfinite(b["stats"]["chi2_final"])
    z = mi355slam.BundleAdjuster(ctx, [p], max_iters=0); z.solve()
    out = z.download(0)
    assert out["stats"]["iters"] == 0 and np.allclose(out["pose"], p["pose"]) and out["stats"]["chi2_final"] == pytest.approx(out["stats"]["chi2_init"])


def test_round2_entry_points_check_their_arguments(ctx, oracle):
    """ms_keypoints_pack, ms_hamming_set_path (per context since round 2), the team test hooks: bad arguments give a status, nothing is written."""
    import mi355slam
    L = mi355slam.lib()
    assert L.ms_hamming_set_path(ctx._h, 2) == -1 and L.ms_hamming_set_path(None, 0) == -1 and L.ms_hamming_set_path(ctx._h, 0) == 0
    other = mi355slam.Context(0)
    other.set_hamming_path(1)                                                # one context's choice does not leak into another
    q = np.random.default_rng(1).integers(0, 2**32, (300, 8), dtype=np.uint64).astype(np.uint32)
    a, b = mi355slam.hamming_best2(ctx, q, q[::-1].copy()), mi355slam.hamming_best2(other, q, q[::-1].copy())
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    other.close()
    ex = mi355slam.OrbExtractor(ctx, 640, 480, max_batch=1)
    ex.extract(oracle.synth_frame(640, 480, 9)[None])
    v = ex.device_view()
    out = np.full((4, 76), 0xAB, np.uint8)
    assert L.ms_keypoints_pack(ctx._h, C.byref(v), 0, ex.capacity + 1, None, out.ctypes.data_as(C.c_void_p)) == -1      # more than the view holds
    assert L.ms_keypoints_pack(ctx._h, C.byref(v), -1, 4, None, out.ctypes.data_as(C.c_void_p)) == -1
    assert L.ms_keypoints_pack(ctx._h, C.byref(v), 0, 4, None, None) == -1
    assert (out == 0xAB).all()
    assert L.ms_keypoints_pack(ctx._h, C.byref(v), 0, 0, None, None) == 0                                                  # nothing to pack is fine
    assert L.ms_keypoints_unpack(None, 3, None, None, None, None, None, None) == -1
    assert L.ms_ba_team_fallbacks(None) == -1 and L.ms_ba_debug_fail_team_barriers(None, 1) == -1
