"""GPU parity: Hamming search and the greedy BoW matchers through the C ABI vs the CPU oracle (bit-exact indices)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_pair(seed, nq=2000, nt=2000, inliers=1400, flip=0.08):
    """SURVEY 8d C3 generator: targets = permuted noisy copies of queries + random outliers."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32)
    perm = rng.permutation(nq)[:inliers]
    noise = np.packbits(rng.random((inliers, 256)) < flip, axis=1, bitorder="little").view(np.uint32)
    t = np.concatenate([q[perm] ^ noise, rng.integers(0, 2**32, (nt - inliers, 8), dtype=np.uint64).astype(np.uint32)])
    t = t[rng.permutation(nt)]
    return q, t


def test_hamming_best2_c3_pairs(oracle, ctx):
    import mi355slam
    pairs = [make_pair(s) for s in range(4)]
    q = np.concatenate([p[0] for p in pairs]); t = np.concatenate([p[1] for p in pairs])
    bi, bd, sd = mi355slam.hamming_best2(ctx, q, t, n_pairs=4)
    for p in range(4):
        wi, wd, ws = oracle.hamming_best2(pairs[p][0], pairs[p][1])
        sl = slice(p * 2000, (p + 1) * 2000)
        assert np.array_equal(bi[sl], wi) and np.array_equal(bd[sl], wd) and np.array_equal(sd[sl], ws)
    assert (bd < 50).mean() > 0.6


def test_hamming_ties_lowest_index_and_ragged(oracle, ctx):
    import mi355slam
    rng = np.random.default_rng(1)
    for nq, nt in [(1, 1), (3, 700), (257, 255), (513, 1025), (64, 0)]:
        q = rng.integers(0, 4, (nq, 8)).astype(np.uint32)        # few distinct values -> many exact ties
        t = rng.integers(0, 4, (nt, 8)).astype(np.uint32)
        bi, bd, sd = mi355slam.hamming_best2(ctx, q, t)
        wi, wd, ws = oracle.hamming_best2(q, t)
        assert np.array_equal(bi, wi) and np.array_equal(bd, wd) and np.array_equal(sd, ws)
        if nt == 0:
            assert (bi == -1).all() and (bd == 256).all()


def test_hamming_matrix_core_tile_edges_and_extremes(oracle, ctx):
    """The unmasked search runs as an i8 matrix product (32-row target tiles, 128-target stages, 64 queries per wave):
    sizes around every tile edge, distance 0 and 256, and per-set counts through the pool API."""
    import mi355slam
    rng = np.random.default_rng(9)
    for nq, nt in [(1, 31), (31, 32), (33, 33), (63, 127), (65, 128), (64, 129), (255, 160), (256, 161), (300, 2), (2, 257)]:
        q = rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32)
        t = rng.integers(0, 2**32, (nt, 8), dtype=np.uint64).astype(np.uint32)
        t[rng.integers(0, nt)] = q[0]                                   # distance 0
        q[nq - 1] = 0; t[nt - 1] = 0xFFFFFFFF                           # distance 256 between these two
        if nt > 1: t[0] = q[0] ^ np.uint32(1)                           # a near tie next to the exact copy
        bi, bd, sd = mi355slam.hamming_best2(ctx, q, t)
        wi, wd, ws = oracle.hamming_best2(q, t)
        assert np.array_equal(bi, wi) and np.array_equal(bd, wd) and np.array_equal(sd, ws), (nq, nt)
    q = np.zeros((5, 8), np.uint32); t = np.full((3, 8), 0xFFFFFFFF, np.uint32)
    bi, bd, sd = mi355slam.hamming_best2(ctx, q, t)
    assert (bi == 0).all() and (bd == 256).all() and (sd == 256).all()
    # pools with per-set counts: 3 sets of stride 200, pair k = (set k) x (set (k+1) % 3)
    stride, counts = 200, np.array([200, 37, 129], np.int32)
    pool = rng.integers(0, 2**32, (3 * stride, 8), dtype=np.uint64).astype(np.uint32)
    pq, pt = np.array([0, 1, 2], np.int32), np.array([1, 2, 0], np.int32)
    dp, dc, dq, dt = ctx.upload(pool), ctx.upload(counts), ctx.upload(pq), ctx.upload(pt)
    obi, obd, osd = ctx.alloc(4 * 3 * stride), ctx.alloc(2 * 3 * stride), ctx.alloc(2 * 3 * stride)
    mi355slam.hamming_best2_sets(ctx, dp, stride, dc, dp, stride, dc, dq, dt, 3, obi, obd, osd)
    ctx.sync()
    bi, bd, sd = obi.download(np.int32, (3, stride)), obd.download(np.uint16, (3, stride)), osd.download(np.uint16, (3, stride))
    for k in range(3):
        a, b = pq[k], pt[k]
        wi, wd, ws = oracle.hamming_best2(pool[a * stride:a * stride + counts[a]], pool[b * stride:b * stride + counts[b]])
        n = counts[a]
        assert np.array_equal(bi[k, :n], wi) and np.array_equal(bd[k, :n], wd) and np.array_equal(sd[k, :n], ws), k
        assert (bi[k, n:] == -1).all() and (bd[k, n:] == 256).all()


def test_hamming_masks(oracle, ctx):
    import mi355slam
    rng = np.random.default_rng(2)
    q, t = make_pair(11, 600, 900, 400)
    qb = rng.integers(0, 20, 600).astype(np.int32); tb = rng.integers(0, 20, 900).astype(np.int32)
    tv = (rng.random(900) < 0.7).astype(np.uint8)
    for kw in (dict(q_bucket=qb, t_bucket=tb), dict(t_valid=tv), dict(q_bucket=qb, t_bucket=tb, t_valid=tv)):
        bi, bd, sd = mi355slam.hamming_best2(ctx, q, t, **kw)
        wi, wd, ws = oracle.hamming_best2(q, t, **kw)
        assert np.array_equal(bi, wi) and np.array_equal(bd, wd) and np.array_equal(sd, ws)


def test_ratio_test_rule(oracle, ctx):
    import mi355slam
    q, t = make_pair(5)
    bi, bd, sd = mi355slam.hamming_best2(ctx, q, t)
    for ratio in (0.75, 0.9, 1.0):
        m = mi355slam.ratio_test(ctx, bi, bd, sd, ratio)
        want = np.where((bd <= 50) & ~(np.float32(ratio) * sd.astype(np.float32) < bd.astype(np.float32)), bi, -1)
        assert np.array_equal(m, want)


def _frames(seed, n1=2000, n2=2000, buckets=100):
    """Two keyframes whose true correspondences share a vocabulary node (as DBoW2 would give) most of the time."""
    rng = np.random.default_rng(seed)
    inl = int(0.7 * min(n1, n2))
    q = rng.integers(0, 2**32, (n1, 8), dtype=np.uint64).astype(np.uint32)
    b1 = rng.integers(0, buckets, n1).astype(np.int32)
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    src = rng.permutation(n1)[:inl]
    noise = np.packbits(rng.random((inl, 256)) < 0.06, axis=1, bitorder="little").view(np.uint32)
    t = np.concatenate([q[src] ^ noise, rng.integers(0, 2**32, (n2 - inl, 8), dtype=np.uint64).astype(np.uint32)])
    b2 = np.concatenate([np.where(rng.random(inl) < 0.9, b1[src], rng.integers(0, buckets, inl)), rng.integers(0, buckets, n2 - inl)]).astype(np.int32)
    a2 = np.concatenate([(a1[src] + 40 + rng.normal(0, 6, inl)) % 360, rng.uniform(0, 360, n2 - inl)]).astype(np.float32)
    sh = rng.permutation(n2)
    t, b2, a2 = t[sh], b2[sh], a2[sh]
    u1 = (rng.random(n1) < 0.8).astype(np.uint8); u2 = (rng.random(n2) < 0.8).astype(np.uint8)
    return q, t, b1, b2, a1, a2, u1, u2


def test_match_loop_closure_exact_greedy(oracle, ctx):
    import mi355slam
    f1s, f2s, wants = [], [], []
    for seed, (n1, n2, nb) in enumerate([(2000, 2000, 100), (500, 1500, 7), (300, 200, 1), (50, 60, 400)]):
        q, t, b1, b2, a1, a2, u1, u2 = _frames(100 + seed, n1, n2, nb)
        # low-entropy descriptors in one case to force ties and contention for the same target
        if seed == 2:
            q &= 0x3; t &= 0x3
        f1s.append(mi355slam.FrameOnDevice(ctx, q, a1, u1, b1)); f2s.append(mi355slam.FrameOnDevice(ctx, t, a2, u2, b2))
        wants.append(oracle.match_loop_closure(q, a1, u1, b1, t, a2, u2, b2, 0.75, True))
    counts, matched = mi355slam.match_loop_closure(ctx, f1s, f2s, 0.75, True)
    for i, (wn, wm) in enumerate(wants):
        assert counts[i] == wn and np.array_equal(matched[i], wm), i
    assert wants[0][0] > 100
    # orientation check off
    counts, matched = mi355slam.match_loop_closure(ctx, f1s[:1], f2s[:1], 0.9, False)
    q, t, b1, b2, a1, a2, u1, u2 = _frames(100, 2000, 2000, 100)
    wn, wm = oracle.match_loop_closure(q, a1, u1, b1, t, a2, u2, b2, 0.9, False)
    assert counts[0] == wn and np.array_equal(matched[0], wm)


def _rot(rng):
    a = rng.normal(size=3) * 0.1
    th = np.linalg.norm(a); k = a / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def test_match_triangulation_exact_greedy(oracle, ctx):
    import mi355slam
    rng = np.random.default_rng(9)
    sf = oracle.scale_factors(8, 1.2)
    f1s, f2s, wants, Es = [], [], [], []
    for seed, (n1, n2, nb) in enumerate([(1500, 1600, 100), (400, 300, 5)]):
        q, t, b1, b2, a1, a2, u1, u2 = _frames(300 + seed, n1, n2, nb)
        R1, R2 = _rot(rng), _rot(rng); t1, t2 = rng.normal(size=3), rng.normal(size=3)
        # bearings: 3-D points seen from both cameras so that many pairs satisfy the epipolar gate
        X = rng.uniform(-3, 3, (max(n1, n2), 3)) + np.array([0, 0, 8.0])
        be1 = (R1 @ X[:n1].T).T + t1; be1 /= np.linalg.norm(be1, axis=1, keepdims=True)
        idx = rng.integers(0, n1, n2)
        be2 = (R2 @ X[idx].T).T + t2; be2 /= np.linalg.norm(be2, axis=1, keepdims=True)
        o1 = rng.integers(0, 8, n1).astype(np.int32)
        E = oracle.create_E21(R2, t2, R1, t1)            # call order of keyframe_matcher.cpp:171-175
        f1s.append(mi355slam.FrameOnDevice(ctx, q, a1, u1, b1, octave=o1, bearing=be1))
        f2s.append(mi355slam.FrameOnDevice(ctx, t, a2, u2, b2, bearing=be2))
        Es.append(E)
        wants.append(oracle.match_triangulation(q, a1, o1, be1, u1, b1, t, a2, be2, u2, b2, E, sf, 2.0, True))
    counts, matched = mi355slam.match_triangulation(ctx, f1s, f2s, np.stack(Es), sf, 2.0, True)
    for i, (wn, wm) in enumerate(wants):
        assert counts[i] == wn and np.array_equal(matched[i], wm), i


def test_greedy_matchers_node_parallel_equals_sequential_walk(oracle, ctx):
    """Both execution paths of M1 / M2 (ms_match_set_path) against the oracle: nodes of 1..600 candidates (registers + LDS bitset),
    query lists longer than a wave, keyframes with nodes the other one lacks, empty keyframes."""
    import mi355slam
    rng = np.random.default_rng(77)
    sf = oracle.scale_factors(8, 1.2)
    shapes = [(2000, 2000, 100), (900, 1300, 3), (700, 650, 1), (130, 90, 40), (64, 64, 1), (65, 129, 2), (5, 0, 3), (0, 7, 2), (1500, 1400, 5), (300, 2500, 1)]
    f1m, f2m, f1t, f2t, Es, want_m, want_t = [], [], [], [], [], [], []
    for seed, (n1, n2, nb) in enumerate(shapes):
        q, t, b1, b2, a1, a2, u1, u2 = _frames(900 + seed, max(n1, 1), max(n2, 1), nb)
        q, t, b1, b2, a1, a2, u1, u2 = q[:n1], t[:n2], b1[:n1], b2[:n2], a1[:n1], a2[:n2], u1[:n1], u2[:n2]
        if seed in (2, 8):
            q &= 0x7; t &= 0x7                                         # ties everywhere: first / last minimum decide
        if seed == 3:
            b2 = b2 + 1000 * (b2 % 3 == 0)                             # a third of kf2's nodes do not exist in kf1
        be1 = rng.normal(size=(n1, 3)); be1 /= np.linalg.norm(be1, axis=1, keepdims=True) if n1 else 1
        be2 = be1[rng.integers(0, max(n1, 1), n2)] + 0.01 * rng.normal(size=(n2, 3)) if n1 else rng.normal(size=(n2, 3))
        if n2: be2 /= np.linalg.norm(be2, axis=1, keepdims=True)
        o1 = rng.integers(0, 8, n1).astype(np.int32)
        E = oracle.create_E21(_rot(rng), rng.normal(size=3), _rot(rng), rng.normal(size=3))
        f1m.append(mi355slam.FrameOnDevice(ctx, q.reshape(-1, 8), a1, u1, b1)); f2m.append(mi355slam.FrameOnDevice(ctx, t.reshape(-1, 8), a2, u2, b2))
        f1t.append(mi355slam.FrameOnDevice(ctx, q.reshape(-1, 8), a1, u1, b1, octave=o1, bearing=be1)); f2t.append(mi355slam.FrameOnDevice(ctx, t.reshape(-1, 8), a2, u2, b2, bearing=be2))
        Es.append(E)
        want_m.append(oracle.match_loop_closure(q, a1, u1, b1, t, a2, u2, b2, 0.75, True))
        want_t.append(oracle.match_triangulation(q, a1, o1, be1, u1, b1, t, a2, be2, u2, b2, E, sf, 25.0, True))
    assert want_m[1][0] > 50 and want_t[1][0] > 20 and want_m[2][0] > 20
    try:
        for path in (0, 1, 2):                                         # 2: one workgroup walks all large nodes, so its LDS holds the previous node's candidates
            ctx.set_match_path(path)
            counts, matched = mi355slam.match_loop_closure(ctx, f1m, f2m, 0.75, True)
            for i, (wn, wm) in enumerate(want_m):
                assert counts[i] == wn and np.array_equal(matched[i], wm), ("loop closure", path, i)
            counts, matched = mi355slam.match_triangulation(ctx, f1t, f2t, np.stack(Es), sf, 25.0, True)
            for i, (wn, wm) in enumerate(want_t):
                assert counts[i] == wn and np.array_equal(matched[i], wm), ("triangulation", path, i)
    finally:
        ctx.set_match_path(0)


def test_greedy_matchers_keypoint_listed_in_two_nodes_falls_back_to_the_exact_walk(oracle, ctx):
    """A DBoW2 FeatureVector never names a keypoint twice; node lists that do make the result depend on the order of the NODES, which
    only the sequential walk reproduces -- the node-parallel pass must notice and hand the pair over (the pair next to it stays parallel)."""
    import mi355slam
    q, t, b1, b2, a1, a2, u1, u2 = _frames(4242, 400, 420, 6)
    q &= 0x3; t &= 0x3                    # low-entropy descriptors + ratio 1.0 below: every query takes some free target, so consumption decides

    def csr(b, extra):
        order = np.argsort(b, kind="stable").astype(np.int32)
        ids, counts = np.unique(b, return_counts=True)
        lists = [list(order[np.cumsum(counts)[i] - counts[i]:np.cumsum(counts)[i]]) for i in range(len(ids))]
        for node, kp in extra:
            lists[node].insert(0, kp)                 # in front: a free copy would win every tie
        start = np.zeros(len(ids) + 1, np.int32); start[1:] = np.cumsum([len(l) for l in lists])
        return ids.astype(np.int32), start, np.concatenate(lists).astype(np.int32)
    # kf2: the keypoints of node 0 are also offered in node 3 (a consumed target comes back as a candidate); kf1: one query listed twice
    c2 = csr(b2, [(3, int(k)) for k in np.flatnonzero(b2 == 0)[:25]])
    c1 = csr(b1, [(4, int(np.flatnonzero(b1 == 1)[0]))])
    wn, wm = oracle.match_loop_closure(q, a1, u1, c1, t, a2, u2, c2, 1.0, True)
    wn0, wm0 = oracle.match_loop_closure(q, a1, u1, b1, t, a2, u2, b2, 1.0, True)
    fa, fb = mi355slam.FrameOnDevice(ctx, q, a1, u1, None, csr=c1), mi355slam.FrameOnDevice(ctx, t, a2, u2, None, csr=c2)
    ga, gb = mi355slam.FrameOnDevice(ctx, q, a1, u1, b1), mi355slam.FrameOnDevice(ctx, t, a2, u2, b2)
    counts, matched = mi355slam.match_loop_closure(ctx, [ga, fa, ga], [gb, fb, gb], 1.0, True)
    assert counts[1] == wn and np.array_equal(matched[1], wm)
    for i in (0, 2):
        assert counts[i] == wn0 and np.array_equal(matched[i], wm0)
    assert wn > 20


def test_hamming_candidates_projection_core(oracle, ctx):
    """Scoring core of searchByProjection / replaceDuplication: each map point against its own radius-query candidates."""
    import mi355slam
    rng = np.random.default_rng(21)
    q, t = make_pair(33, 500, 1800, 400)
    t_oct = rng.integers(0, 8, 1800).astype(np.int32)
    skip = (rng.random(1800) < 0.2).astype(np.uint8)
    cands = [rng.choice(1800, size=int(k), replace=False).astype(np.int32) for k in rng.integers(0, 130, 500)]
    cands[3] = np.zeros(0, np.int32)
    t2 = t.copy(); t2[cands[5][:3]] = q[5]                     # exact ties: first in list order must win, the next is the second
    for sk in (None, skip):
        bi, bd, sd, bo, so = mi355slam.hamming_candidates(ctx, q, t2, cands, t_skip=sk, t_octave=t_oct)
        for i in range(500):
            w = oracle.best2_candidates(q[i], t2, cands[i], skip=sk, t_octave=t_oct)
            assert (bi[i], bd[i], sd[i], bo[i], so[i]) == w, i
    assert bi[3] == -1 and bd[3] == 256


def test_descriptor_medoid_equals_update_descriptor(oracle, ctx):
    """MapPoint::updateDescriptor (map_point.cpp:75-116) for a batch of map points: median-Hamming medoid, first index on ties."""
    import mi355slam
    rng = np.random.default_rng(77)
    pool = rng.integers(0, 2**32, (6000, 8), dtype=np.uint64).astype(np.uint32)
    # observations of one map point look alike: a base descriptor with a few flipped bits, so ties and small medians occur
    for b in range(0, 6000, 12):
        base = pool[b].copy()
        for k in range(12):
            d = base.copy()
            for bit in rng.integers(0, 256, rng.integers(0, 20)): d[bit >> 5] ^= np.uint32(1 << (bit & 31))
            pool[b + k] = d
    sizes = list(rng.integers(1, 40, 700)) + [0, 1, 2, 3, 64, 65, 129, 256]
    lists = []
    for n in sizes:
        if n <= 12 and n > 0:
            b = 12 * int(rng.integers(0, 500)); lists.append((b + rng.permutation(12)[:n]).astype(np.int32))
        else:
            lists.append(rng.choice(6000, size=int(n), replace=False).astype(np.int32))
    lists[5] = np.array([7, 7, 7, 7], np.int32)                    # identical observations: index 0
    bl, bp = mi355slam.descriptor_medoid(ctx, pool, lists)
    for p, obs in enumerate(lists):
        want = oracle.descriptor_medoid(pool[obs]) if len(obs) else -1
        assert bl[p] == want, (p, len(obs))
        assert bp[p] == (obs[want] if want >= 0 else -1), p
    with pytest.raises(mi355slam.MsError):
        mi355slam.descriptor_medoid(ctx, pool, [np.arange(257, dtype=np.int32)])


def test_projection_candidates_radius_query_and_scan(oracle, ctx):
    """getFeaturesAround + the candidate scan of searchByProjection (feature_search.cpp:33-48, keyframe_matcher.cpp:349-378)."""
    import mi355slam
    rng = np.random.default_rng(123)
    n, nq = 1900, 700
    kx = np.round(rng.uniform(0, 1280, n), 0).astype(np.float32) * np.float32(1.2)         # coordinates are level pixels x scale: many equal y
    ky = np.round(rng.uniform(0, 720, n), 0).astype(np.float32) * np.float32(1.2)
    toct = rng.integers(0, 8, n).astype(np.int32)
    q, t = make_pair(7, nq, n, 500)
    skip = (rng.random(n) < 0.25).astype(np.uint8)
    qx = rng.uniform(-20, 1300, nq).astype(np.float32); qy = rng.uniform(-20, 740, nq).astype(np.float32)
    qr = rng.uniform(0.5, 60, nq).astype(np.float32); qr[:5] = [0.0, 1e-3, 2000.0, 15.0, 15.0]
    qx[3], qy[3] = kx[10], ky[10]                                                           # dead centre on a keypoint
    t[11] = q[3]; kx[11], ky[11] = kx[10], ky[10]                                           # two keypoints at the same place, one an exact match
    lo = rng.integers(0, 6, nq).astype(np.int32); hi = lo + rng.integers(0, 3, nq).astype(np.int32)
    sx, sy, si = mi355slam.feature_search_sort(kx, ky)
    assert np.all(np.diff(sy) >= 0) and np.array_equal(np.sort(si), np.arange(n))
    for (sk, use_oct) in ((None, False), (skip, False), (skip, True)):
        got = mi355slam.projection_candidates(ctx, kx, ky, t, qx, qy, qr, q, t_octave=toct, t_skip=sk,
                                              q_min_octave=lo if use_oct else None, q_max_octave=hi if use_oct else None)
        for i in range(nq):
            pos = oracle.features_around(sx, sy, qx[i], qy[i], qr[i])
            cand = si[pos]
            assert got[6][i] == len(cand), i
            if use_oct: cand = cand[(toct[cand] >= lo[i]) & (toct[cand] <= hi[i])]
            w = oracle.best2_candidates(q[i], t, cand, skip=sk, t_octave=toct)
            assert (got[0][i], got[1][i], got[2][i], got[3][i], got[4][i]) == w, (i, sk is not None, use_oct)
    assert got[6][0] == 0 and got[0][0] == -1                                               # radius 0: nothing is strictly inside


def test_matrix_core_search_equals_popcount_search(ctx):
    """The two kernels behind the unmasked search (i8 matrix cores vs v_xor / v_bcnt popcount) agree bit for bit."""
    import mi355slam
    L = mi355slam.lib()
    rng = np.random.default_rng(99)
    try:
        for nq, nt in [(2000, 2000), (1837, 1911), (77, 3000)]:
            q = rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32)
            t = rng.integers(0, 2**32, (nt, 8), dtype=np.uint64).astype(np.uint32)
            t[: nq // 2] = q[: nq // 2] ^ np.uint32(1 << 7)              # near copies: small distances and ties
            assert L.ms_hamming_set_path(ctx._h, 0) == 0
            a = mi355slam.hamming_best2(ctx, q, t)
            assert L.ms_hamming_set_path(ctx._h, 1) == 0
            b = mi355slam.hamming_best2(ctx, q, t)
            for x, y in zip(a, b): assert np.array_equal(x, y)
    finally:
        L.ms_hamming_set_path(ctx._h, 0)


def test_top4_candidate_lists(oracle, ctx):
    """ms_projection_topk / ms_hamming_candidates_topk: per query the four smallest (distance, scan position) keys among the candidates that are
    not skipped and inside the octave window -- the order the reference's strict-less scan induces (keyframe_matcher.cpp:356-378) -- and the
    number of candidates scored.  The first two entries are exactly ms_projection_candidates' best / second."""
    import mi355slam
    rng = np.random.default_rng(5)
    n, nq = 1500, 700
    kx = rng.uniform(0, 640, n).astype(np.float32); ky = np.round(rng.uniform(0, 480, n)).astype(np.float32)      # rounded y: ties in the sort key
    t = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
    src = rng.integers(0, n, nq)
    q = t[src] ^ np.packbits(rng.random((nq, 256)) < 0.05, axis=1, bitorder="little").view(np.uint32)
    q[::5] &= 0xF; t[::3] &= 0xF                                                              # low-entropy rows: equal distances, position decides
    qx = (kx[src] + rng.uniform(-6, 6, nq)).astype(np.float32); qy = (ky[src] + rng.uniform(-6, 6, nq)).astype(np.float32)
    qr = rng.choice([0.5, 4.0, 12.0, 40.0, 90.0], nq).astype(np.float32)
    toct = rng.integers(0, 8, n).astype(np.int32); skip = (rng.random(n) < 0.2).astype(np.uint8)
    lo = rng.integers(0, 4, nq).astype(np.int32); hi = (lo + rng.integers(0, 5, nq)).astype(np.int32)
    ti, td, to, ns, nc = mi355slam.projection_topk(ctx, kx, ky, t, qx, qy, qr, q, toct, skip, lo, hi)
    bi, bd, sd, bo, so, si, nc2 = mi355slam.projection_candidates(ctx, kx, ky, t, qx, qy, qr, q, toct, skip, lo, hi)
    assert np.array_equal(nc, nc2)
    assert np.array_equal(ti[:, 0], bi) and np.array_equal(td[:, 0], bd) and np.array_equal(ti[:, 1], si) and np.array_equal(td[:, 1], sd)
    sx, sy, sidx = mi355slam.feature_search_sort(kx, ky)
    lists = []
    for i in range(nq):
        cand = oracle.features_around(sx, sy, qx[i], qy[i], qr[i])                            # positions in the sorted array, ascending
        keep = [p for p in cand if not skip[sidx[p]] and lo[i] <= toct[sidx[p]] <= hi[i]]
        d = [oracle.hamming256(q[i], t[sidx[p]]) for p in keep]
        order = sorted(range(len(keep)), key=lambda k: (d[k], keep[k]))[:4]
        assert ns[i] == len(keep) and nc[i] == len(cand)
        want_i = [int(sidx[keep[k]]) for k in order] + [-1] * (4 - len(order)); want_d = [d[k] for k in order] + [256] * (4 - len(order))
        assert list(ti[i]) == want_i and list(td[i]) == want_d, i
        assert list(to[i]) == [int(toct[j]) if j >= 0 else -1 for j in want_i]
        lists.append([int(sidx[p]) for p in cand])
    assert (ns > 4).sum() > 100 and (ns == 0).sum() > 5
    # the same lists handed over explicitly (the host's getFeaturesAround route)
    ci, cd, co, cn = mi355slam.hamming_candidates_topk(ctx, q, t, lists, skip, toct)
    for i in range(nq):
        keep = [j for j in lists[i] if not skip[j]]
        d = [oracle.hamming256(q[i], t[j]) for j in keep]
        order = sorted(range(len(keep)), key=lambda k: (d[k], k))[:4]
        assert cn[i] == len(keep)
        assert list(ci[i]) == [keep[k] for k in order] + [-1] * (4 - len(order)) and list(cd[i]) == [d[k] for k in order] + [256] * (4 - len(order)), i


def test_match_fuzz_tool_on_random_keyframe_pairs():
    """tools/match_fuzz.py on 40 random keyframe pairs (0..3000 keypoints, 1..600 vocabulary nodes, node sizes across the register / LDS / beyond-LDS regimes,
    usable masks, low-entropy descriptors, ratios, thresholds), M1 and M2, every execution path: zero mismatches against the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "match_fuzz.py"), "40", "2025"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
