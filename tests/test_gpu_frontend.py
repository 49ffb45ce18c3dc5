"""GPU parity: pyramid, detector, orientation and descriptors through the C ABI vs the CPU oracle.

Bar: bit-exact pixels, keypoint coordinates/order, angles (float32 bit patterns) and descriptor bits.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(oracle, **kw):
    d = dict(levels=8, scale_factor=1.2, max_kpts=2000, lk_track_level=0, fast_threshold=20, min_distance=0.0)
    d.update(kw)
    return oracle.cfg(**d), d


def _extract_both(oracle, ctx, imgs, mask=None, tracks=None, track_ids=None, max_tracks=0, **kw):
    import mi355slam
    ocfg, d = _cfg(oracle, **kw)
    n, h, w = imgs.shape
    ex = mi355slam.OrbExtractor(ctx, w, h, levels=d["levels"], scale_factor=d["scale_factor"], max_kpts=d["max_kpts"],
                                lk_track_level=d["lk_track_level"], fast_threshold=d["fast_threshold"],
                                max_tracks=max_tracks, max_batch=n, min_distance=d["min_distance"])
    if mask is not None:
        ex.set_valid_mask(mask)
    ex.extract(imgs, track_xy=tracks, track_id=track_ids)
    got = [ex.download(f) for f in range(n)]
    want = [oracle.orb_extract(ocfg, imgs[f], valid_mask=mask,
                               track_xy=None if tracks is None else np.asarray(tracks[f], np.float32).reshape(-1, 2),
                               track_id=None if track_ids is None else track_ids[f]) for f in range(n)]
    return ex, got, want


def _assert_same_keypoints(got, want):
    assert len(got["x"]) == len(want["x"])
    for k in ("x", "y", "angle"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k      # float32 bit patterns
    assert np.array_equal(got["octave"], want["octave"])
    assert np.array_equal(got["track_id"], want["track_id"])
    assert np.array_equal(got["desc"], want["desc"])


def test_pyramid_pixels_bit_exact(oracle, ctx):
    img = oracle.synth_frame(640, 480, 1000)
    ex, got, want = _extract_both(oracle, ctx, img[None])
    ocfg, _ = _cfg(oracle)
    levels, blurs = oracle.build_pyramid(ocfg, img)
    for l in range(8):
        assert ex.level_size(l) == (levels[l].shape[1], levels[l].shape[0])
        assert np.array_equal(ex.download_level(0, l, False), levels[l]), "level %d" % l
        assert np.array_equal(ex.download_level(0, l, True), blurs[l]), "blurred level %d" % l


def test_detections_match_oracle_per_level(oracle, ctx):
    img = oracle.synth_frame(640, 480, 1001)
    ex, _, _ = _extract_both(oracle, ctx, img[None])
    ocfg, _ = _cfg(oracle)
    levels, _ = oracle.build_pyramid(ocfg, img)
    quotas = oracle.level_quotas(8, 1.2, 2000)
    for l in range(8):
        xs, ys, sc = oracle.detect_level(levels[l], 20, int(quotas[l]))
        gx, gy, gs = ex.download_detections(0, l)
        assert np.array_equal(gx, xs) and np.array_equal(gy, ys) and np.array_equal(gs, sc), "level %d" % l


def test_c1_vga_frame_bit_exact(oracle, ctx):
    """BASELINE config C1: one 640x480 synthetic frame, 8 levels, 2000 keypoints."""
    img = oracle.synth_frame(640, 480, 1000)
    _, got, want = _extract_both(oracle, ctx, img[None])
    assert len(want[0]["x"]) > 1000
    _assert_same_keypoints(got[0], want[0])


def test_720p_batch_bit_exact(oracle, ctx):
    imgs = np.stack([oracle.synth_frame(1280, 720, 1000 + i, 2 * i, i) for i in range(3)])
    _, got, want = _extract_both(oracle, ctx, imgs)
    for f in range(3):
        _assert_same_keypoints(got[f], want[f])
    assert not np.array_equal(got[0]["desc"][:100], got[1]["desc"][:100])


def test_pure_ramp_gives_no_keypoints(oracle, ctx):
    """A pure gradient has no corners: the early return of orb_extractor.cpp:137."""
    y, x = np.mgrid[0:480, 0:640]
    img = ((x * 96) // 639 + (y * 64) // 479).astype(np.uint8)
    _, got, want = _extract_both(oracle, ctx, img[None])
    assert len(want[0]["x"]) == 0 and len(got[0]["x"]) == 0


def test_random_noise_saturates_quota(oracle, ctx):
    """Noise floods every level with corners: exercises the radix select / quota cut and score ties."""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (300, 400), dtype=np.uint8)
    _, got, want = _extract_both(oracle, ctx, img[None], levels=4, max_kpts=1500)
    _assert_same_keypoints(got[0], want[0])


def test_quotas_that_sum_past_max_kpts(oracle, ctx):
    """The per-level quotas are rounded one by one (static_settings.cpp:52): 15 levels / 160 keypoints add up to 161.  The reference
    keeps per-level vectors, so a saturated frame returns all 161; here that needs the per-frame slot stride to be the quota sum
    (frame f's last level must not land in frame f+1's first slots)."""
    import mi355slam
    q = mi355slam.level_quotas(15, 1.2, 160)
    assert int(q.sum()) == 161
    def multiscale(seed, w=1600, h=1300):         # random blocks of 64 and 16 pixels: corners on every level of a 15-level pyramid
        r = np.random.default_rng(seed)
        a = np.kron(r.integers(0, 2, (h // 64 + 1, w // 64 + 1)), np.ones((64, 64), np.int64))[:h, :w] * 120 + 30
        b = np.kron(r.integers(0, 2, (h // 16 + 1, w // 16 + 1)), np.ones((16, 16), np.int64))[:h, :w] * 60
        return np.clip(a + b + r.integers(0, 20, (h, w)), 0, 255).astype(np.uint8)
    imgs = np.stack([multiscale(s) for s in range(3)])
    ex, got, want = _extract_both(oracle, ctx, imgs, levels=15, max_kpts=160)
    assert ex.capacity == 161
    for f in range(3):
        assert len(want[f]["x"]) > 140
        _assert_same_keypoints(got[f], want[f])
    assert any((w["octave"] == 13).any() for w in want)              # level 13 sits at slots 158..160
    # 8 levels / 7 keypoints: quotas 2 1 1 1 1 1 1 0 = 8; level 6 is slot 7 = the old stride: it landed in the next frame's slot 0
    import synth
    imgs = np.stack([synth.synth_frame(720, 600, s) for s in (24, 20, 24, 28)])
    ex, got, want = _extract_both(oracle, ctx, imgs, levels=8, max_kpts=7, max_tracks=3, tracks=[[(100.5, 80.25)], [(300.0, 200.0), (50.0, 60.0)], [], [(10.0, 10.0)]],
                                 track_ids=[[7], [8, 9], [], [3]])
    assert ex.capacity == 8 + 3
    assert [len(w["x"]) for w in want] == [8 + 1, 7 + 2, 8, 6]       # the track at (10, 10) lies inside the 19 px border
    for f in range(4):
        assert (want[f]["octave"] == 6).any()
        _assert_same_keypoints(got[f], want[f])


def test_odd_sizes_and_thresholds(oracle, ctx):
    for (w, h, thr, levels, sf) in [(331, 257, 10, 5, 1.2), (200, 120, 35, 3, 1.5), (97, 83, 20, 2, 1.1), (400, 300, 20, 2, 2.5), (250, 250, 15, 3, 2.0)]:
        img = oracle.synth_frame(w, h, 5 + w)
        _, got, want = _extract_both(oracle, ctx, img[None], levels=levels, scale_factor=sf, fast_threshold=thr, max_kpts=700)
        _assert_same_keypoints(got[0], want[0])


def test_tracks_and_valid_mask(oracle, ctx):
    rng = np.random.default_rng(3)
    img = oracle.synth_frame(640, 480, 1002)
    mask = np.ones((480, 640), np.uint8)
    mask[:, :60] = 0
    mask[400:, :] = 0
    tracks = [np.stack([rng.uniform(-5, 645, 150), rng.uniform(-5, 485, 150)], 1).astype(np.float32)]
    ids = [np.arange(1000, 1150, dtype=np.int32)]
    for lk in (0, 2):
        _, got, want = _extract_both(oracle, ctx, img[None], mask=mask, tracks=tracks, track_ids=ids, max_tracks=200, lk_track_level=lk)
        assert (want[0]["track_id"] >= 0).sum() > 50
        _assert_same_keypoints(got[0], want[0])


def test_unaligned_device_input_is_copied(oracle, ctx):
    """Device frames with an odd stride cannot be used in place; the device-side copy path must agree."""
    import mi355slam
    img = oracle.synth_frame(333, 222, 77)
    padded = np.zeros((222, 341), np.uint8)
    padded[:, :333] = img
    buf = ctx.upload(padded)
    ex = mi355slam.OrbExtractor(ctx, 333, 222, levels=4, max_kpts=500)
    ex.extract(buf, n_frames=1, frame_stride=padded.size, row_stride=341)
    got = ex.download(0)
    want = oracle.orb_extract(oracle.cfg(levels=4, max_kpts=500), img)
    _assert_same_keypoints(got, want)


def test_keypoints_on_the_minimum_margin(oracle, ctx):
    """Regression for the round-1 fault: a keypoint whose patch starts at the first byte of a plane.  Isolated bright pixels
    at (19, 19), (w-20, 19), (19, h-20), (w-20, h-20) are FAST corners exactly on the 19-pixel margin (feature_detector.cpp:106-123): the
    orientation / descriptor patches then begin at column / row 0 (negative offsets from the centre would address ~4 GiB past the slab).
    Level 0 in the slab (host frames), used in place (aligned device frames) and through the device-side copy (odd stride)."""
    import mi355slam
    w, h = 320, 256
    img = np.full((h, w), 30, np.uint8)
    corners = [(19, 19), (w - 20, 19), (19, h - 20), (w - 20, h - 20), (100, 100), (160, 19)]
    for x, y in corners:
        img[y, x] = 250
    cfg = oracle.cfg(levels=3, max_kpts=200)
    want = oracle.orb_extract(cfg, img)
    lvl0 = {(int(x), int(y)) for x, y, o in zip(want["x"], want["y"], want["octave"]) if o == 0}
    assert set(corners) <= lvl0
    ex = mi355slam.OrbExtractor(ctx, w, h, levels=3, max_kpts=200, max_batch=2)
    ex.extract(np.stack([img, img]))                                  # host frames: level 0 copied into the slab
    _assert_same_keypoints(ex.download(0), want); _assert_same_keypoints(ex.download(1), want)
    buf = ctx.upload(np.stack([img, img]))
    ex.extract(buf, n_frames=2, frame_stride=w * h, row_stride=w)     # aligned device frames: used in place as level 0
    _assert_same_keypoints(ex.download(0), want); _assert_same_keypoints(ex.download(1), want)
    padded = np.zeros((h, w + 5), np.uint8); padded[:, :w] = img
    pb = ctx.upload(padded)
    ex.extract(pb, n_frames=1, frame_stride=padded.size, row_stride=w + 5)     # odd stride: device-side copy
    _assert_same_keypoints(ex.download(0), want)


def test_extract_is_deterministic_across_runs(oracle, ctx):
    import mi355slam
    imgs = np.stack([oracle.synth_frame(640, 480, 2000 + i) for i in range(4)])
    ex = mi355slam.OrbExtractor(ctx, 640, 480, max_batch=4)
    ex.extract(imgs)
    a = [ex.download(f) for f in range(4)]
    ex.extract(imgs)
    b = [ex.download(f) for f in range(4)]
    for f in range(4):
        _assert_same_keypoints(a[f], b[f])


def test_min_distance_suppression(oracle, ctx):
    """gfttMinDistance (feature_detector.cpp:79-82): greedy spacing in key order, solved as a parallel fixed point on the GPU."""
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    for img, kw in [(oracle.synth_frame(1280, 720, 1005), dict(min_distance=12.0)),
                    (oracle.synth_frame(640, 480, 1006), dict(min_distance=30.0)),
                    (noise, dict(min_distance=20.0, levels=4, max_kpts=1200)),          # saturated levels: long dependency chains
                    (oracle.synth_frame(640, 480, 1007), dict(min_distance=3.0))]:     # tiny distance: coarser grid cells than min_dist
        ex, got, want = _extract_both(oracle, ctx, img[None], **kw)
        _assert_same_keypoints(got[0], want[0])
        # the spacing really holds per level, in level coordinates
        cfgd = dict(levels=8, scale_factor=1.2); cfgd.update({k: v for k, v in kw.items() if k in ("levels", "scale_factor")})
        for l in range(cfgd["levels"]):
            x, y, _ = ex.download_detections(0, l)
            w, h = ex.level_size(l)
            md = oracle.level_min_dist(kw["min_distance"], w, h)
            if md >= 2 and len(x) > 1:
                d2 = (x[:, None] - x[None, :]) ** 2 + (y[:, None] - y[None, :]) ** 2 + np.eye(len(x), dtype=np.int64) * 10**9
                assert d2.min() >= md * md, (l, md)


def test_tile_edge_geometries(oracle, ctx):
    """Widths / heights around the 248 x 30 detector tiles, the 248 x 72 blur tiles and the 5-row resize groups, on noise
    (corners everywhere, including the first and last columns and rows, where the border waves clamp, shift or reflect their loads)."""
    rng = np.random.default_rng(31)
    for (w, h) in [(248, 56), (249, 57), (252, 49), (496, 70), (497, 71), (500, 50), (744, 53), (745, 85), (1000, 59),
                   (247, 60), (251, 61), (253, 89), (254, 90), (255, 91), (495, 72), (499, 73), (743, 144), (992, 145)]:
        img = rng.integers(0, 256, (1, h, w), dtype=np.uint8)
        img[0, :, : w // 2] = (img[0, :, : w // 2] // 64) * 64                      # flat patches with sharp steps on one half
        _, got, want = _extract_both(oracle, ctx, img, levels=2, scale_factor=1.2, fast_threshold=12, max_kpts=3000)
        _assert_same_keypoints(got[0], want[0])


def test_full_hd_frame_and_batch_larger_than_one_launch_slot(oracle, ctx):
    """1920 x 1080 (more tiles per level than 720p) and a batch that reuses an extractor across calls of different size."""
    import mi355slam
    img = oracle.synth_frame(1920, 1080, 77)
    _, got, want = _extract_both(oracle, ctx, img[None], max_kpts=3000)
    _assert_same_keypoints(got[0], want[0])
    frames = np.stack([oracle.synth_frame(320, 240, 400 + i, 2 * i, i) for i in range(5)])
    ocfg, _ = _cfg(oracle, levels=4, max_kpts=500)
    ex = mi355slam.OrbExtractor(ctx, 320, 240, levels=4, max_kpts=500, max_batch=5)
    for n in (5, 2, 1, 4):                                                          # shrinking and growing the batch on one extractor
        ex.extract(frames[:n])
        for f in range(n):
            _assert_same_keypoints(ex.download(f), oracle.orb_extract(ocfg, frames[f]))


def test_randomised_configurations(oracle, ctx):
    """Differential fuzz: 40 random (size, levels, scale factor, threshold, quota, image kind) configurations, bit-exact each."""
    rng = np.random.default_rng(2024)
    done = 0
    while done < 40:
        w, h = int(rng.integers(60, 700)), int(rng.integers(60, 420))
        levels = int(rng.integers(1, 7)); sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
        if min(w, h) / sf ** (levels - 1) < 41: continue                      # every level must hold a 40 x 40 patch area
        thr = int(rng.integers(5, 60)); kp = int(rng.integers(50, 3000))
        kind = int(rng.integers(0, 3))
        if kind == 0: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1: img = oracle.synth_frame(w, h, int(rng.integers(0, 1000)))
        else:
            img = (rng.integers(0, 4, (h // 8 + 1, w // 8 + 1), dtype=np.uint8) * 80).repeat(8, 0).repeat(8, 1)[:h, :w].copy()   # blocky
        _, got, want = _extract_both(oracle, ctx, np.ascontiguousarray(img)[None], levels=levels, scale_factor=sf, fast_threshold=thr, max_kpts=kp)
        _assert_same_keypoints(got[0], want[0])
        done += 1


def test_keypoint_records_round_trip(oracle, ctx):
    """N4: the device SoA packed into KeyPoint::serialize records (key_point.hpp:22-25) and read back: every field survives, the
    byte layout is x, y, angle (f32), octave twice (i32), bearing (3 x f64), descriptor (8 x u32) = 76 bytes."""
    import struct
    import mi355slam
    imgs = np.stack([oracle.synth_frame(640, 480, 1000 + i) for i in range(2)])
    ex, got, want = _extract_both(oracle, ctx, imgs)
    for f in range(2):
        n = len(want[f]["x"])
        bearing = np.random.default_rng(f).normal(size=(n, 3))
        rec = ex.pack_keypoints(f, bearing)
        assert rec.shape == (n, 76)
        back = mi355slam.unpack_keypoints(rec)
        for k in ("x", "y", "angle"):
            assert np.array_equal(back[k].view(np.uint32), want[f][k].view(np.uint32))
        assert np.array_equal(back["octave"], want[f]["octave"]) and np.array_equal(back["desc"], want[f]["desc"]) and np.array_equal(back["bearing"], bearing)
        i = n // 2                                                     # one record decoded independently of the library
        x, y, a, o1, o2, b0, b1, b2, *d = struct.unpack("<fffii3d8I", rec[i].tobytes())
        assert (x, y, a, o1, o2) == (want[f]["x"][i], want[f]["y"][i], want[f]["angle"][i], want[f]["octave"][i], want[f]["octave"][i])
        assert (b0, b1, b2) == tuple(bearing[i]) and d == want[f]["desc"][i].tolist()
    assert np.array_equal(mi355slam.unpack_keypoints(ex.pack_keypoints(0))["bearing"], np.zeros((len(want[0]["x"]), 3)))     # no bearing given: zeros


def test_host_batches_in_pieces_and_asynchronous_downloads(oracle, ctx):
    """A batch of >= 32 host frames goes in as four pieces on a copy stream, each piece's kernels under the next piece's copy (frames whose rows are
    contiguous move as ONE 2-D copy per piece, others frame by frame); ms_dev_download_async takes the outputs out on a third stream while the
    next batch is already enqueued.  Every frame must equal what the same frames give when they are resident on the device -- and the oracle."""
    import ctypes as C
    import synth
    import mi355slam
    for (w, h, n, pad) in [(320, 192, 45, 0), (333, 222, 37, 5)]:          # 320 = slab pitch (whole-frame copies); 333: pitched 2-D copies per frame
        frames = np.stack([synth.synth_frame(w, h, 300 + i, i % 5, i % 3) for i in range(n)])
        host = np.zeros((n, h, w + pad), np.uint8); host[:, :, :w] = frames
        ex = mi355slam.OrbExtractor(ctx, w, h, levels=5, max_kpts=600, max_batch=n)
        cap = ex.capacity
        lib = mi355slam.lib()
        sizes = {"count": 4 * n, "x": 4 * n * cap, "y": 4 * n * cap, "angle": 4 * n * cap, "octave": 4 * n * cap, "desc": 32 * n * cap}
        rounds = []
        for rep in range(3):                                             # three batches back to back: the pieces of batch k+1 wait for batch k's kernels and downloads
            src = ctx.pinned(host.shape)                                 # page-locked: the copies really are asynchronous
            src[...] = np.roll(host, rep, axis=0)                        # a different frame order per round
            ctx.check(lib.ms_orb_extract(ex._h, C.c_void_p(src.ctypes.data), 0, n, C.c_size_t(h * (w + pad)), C.c_size_t(w + pad), None, None, None), "ms_orb_extract")
            v = ex.device_view()
            outs = {k: ctx.pinned((b,)) for k, b in sizes.items()}
            ptr = {"count": v.count, "x": v.x, "y": v.y, "angle": v.angle, "octave": v.octave, "desc": v.desc}
            for k, b in sizes.items():
                ctx.check(lib.ms_dev_download_async(ctx._h, C.c_void_p(outs[k].ctypes.data), C.c_void_p(ptr[k]), C.c_size_t(b)), "ms_dev_download_async")
            rounds.append((src, outs))                                   # keep both alive until the copies are done
        ctx.check(lib.ms_dev_download_wait(ctx._h), "ms_dev_download_wait")
        ocfg = oracle.cfg(levels=5, max_kpts=600)
        for rep, (src, outs) in enumerate(rounds):
            cnt = outs["count"].view(np.int32)
            for f in (0, 1, n // 4, n // 4 + 1, n // 2, n - 2, n - 1):  # frames on both sides of every piece boundary
                want = oracle.orb_extract(ocfg, frames[(f - rep) % n])
                k = cnt[f]
                assert k == len(want["x"]) > 100
                assert np.array_equal(outs["x"].view(np.float32).reshape(n, cap)[f, :k], want["x"])
                assert np.array_equal(outs["y"].view(np.float32).reshape(n, cap)[f, :k], want["y"])
                assert np.array_equal(outs["angle"].view(np.uint32).reshape(n, cap)[f, :k], want["angle"].view(np.uint32))
                assert np.array_equal(outs["octave"].view(np.int32).reshape(n, cap)[f, :k], want["octave"])
                assert np.array_equal(outs["desc"].view(np.uint32).reshape(n, cap, 8)[f, :k], want["desc"])
        # the same frames resident on the device (one launch over the whole batch): identical counts for every frame
        dev = ctx.upload(np.ascontiguousarray(frames))
        if w % 16 == 0:
            ex.extract(dev, n_frames=n, frame_stride=w * h, row_stride=w)
            cnt_dev = np.array([len(ex.download(f)["x"]) for f in range(n)])
            assert np.array_equal(cnt_dev, rounds[0][1]["count"].view(np.int32))
        ex.close()
        ctx.free_pinned()
