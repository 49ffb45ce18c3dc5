"""CPU tests: Hamming distance, rotation histogram and matcher policies of the oracle."""
import numpy as np


def test_hamming_known_answers(oracle):
    z = np.zeros(8, np.uint32); f = np.full(8, 0xFFFFFFFF, np.uint32)
    assert oracle.hamming256(z, z) == 0 and oracle.hamming256(z, f) == 256
    one = z.copy(); one[3] = 1 << 17
    assert oracle.hamming256(z, one) == 1
    rng = np.random.default_rng(1)
    a = rng.integers(0, 2**32, (500, 8), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, (500, 8), dtype=np.uint64).astype(np.uint32)
    want = np.unpackbits((a ^ b).view(np.uint8), axis=1).sum(1)
    got = np.array([oracle.hamming256(a[i], b[i]) for i in range(500)])
    assert np.array_equal(got, want)


def _best2_py(q, t, qb=None, tb=None, tv=None):
    bi, bd, sd = [], [], []
    for i in range(len(q)):
        best, second, idx = 256, 256, -1
        for j in range(len(t)):
            if tv is not None and not tv[j]: continue
            if qb is not None and qb[i] != tb[j]: continue
            d = int(np.unpackbits((q[i] ^ t[j]).view(np.uint8)).sum())
            if d < best: second, best, idx = best, d, j
            elif d < second: second = d
        bi.append(idx); bd.append(best); sd.append(second)
    return np.array(bi), np.array(bd), np.array(sd)


def test_best2_update_rule(oracle):
    rng = np.random.default_rng(2)
    q = rng.integers(0, 8, (40, 8)).astype(np.uint32); t = rng.integers(0, 8, (60, 8)).astype(np.uint32)
    qb = rng.integers(0, 3, 40).astype(np.int32); tb = rng.integers(0, 3, 60).astype(np.int32); tv = (rng.random(60) < 0.6).astype(np.uint8)
    for kw, py in ((dict(), _best2_py(q, t)), (dict(q_bucket=qb, t_bucket=tb, t_valid=tv), _best2_py(q, t, qb, tb, tv))):
        bi, bd, sd = oracle.hamming_best2(q, t, **kw)
        assert np.array_equal(bi, py[0]) and np.array_equal(bd, py[1]) and np.array_equal(sd, py[2])
    bi, bd, sd = oracle.hamming_best2(q, np.zeros((0, 8), np.uint32))
    assert (bi == -1).all() and (bd == 256).all() and (sd == 256).all()


def test_angle_checker_bins_and_top3(oracle):
    # cvRound(delta / 30) with wrap to [0, 360): match_angle_checker.h:72-83
    deltas = np.array([0, 14.9, 15.1, 44.9, -10, 359.9, 360.0, 725.0 - 360, 200, 200, 200, 100, 100, 40], np.float32)
    ids = np.arange(len(deltas), dtype=np.int32)
    inv = oracle.angle_check(deltas, ids)
    # bins: 0:{0,1,6? ...}; compute independently
    def bin_of(d):
        d = np.float32(d)
        if d < 0: d = np.float32(np.float64(d) + 360.0)
        if d >= 360.0: d = np.float32(np.float64(d) - 360.0)
        return int(np.rint(np.float32(d * np.float32(1.0 / 30))))
    bins = [bin_of(d) for d in deltas]
    counts = np.bincount(bins, minlength=30)
    order = sorted(range(30), key=lambda b: (-counts[b], b))[:3]
    want = [i for b in range(30) if b not in order for i in range(len(bins)) if bins[i] == b]
    assert inv.tolist() == want
    assert oracle.angle_check(np.zeros(0, np.float32), np.zeros(0, np.int32)).tolist() == []


def _m1_py(d1, a1, u1, b1, d2, a2, u2, b2, ratio):
    n1, n2 = len(d1), len(d2)
    matched = -np.ones(n1, np.int64); used = np.zeros(n2, bool); recs = []
    for node in sorted(set(b1.tolist()) & set(b2.tolist())):
        for i1 in np.nonzero(b1 == node)[0]:
            if not u1[i1]: continue
            best, second, bi = 256, 256, -1
            for i2 in np.nonzero(b2 == node)[0]:
                if not u2[i2] or used[i2]: continue
                d = int(np.unpackbits((d1[i1] ^ d2[i2]).view(np.uint8)).sum())
                if d < best: second, best, bi = best, d, i2
                elif d < second: second = d
            if best > 50: continue
            if np.float32(ratio) * np.float32(second) < np.float32(best): continue
            matched[i1] = bi; used[bi] = True; recs.append((a1[i1] - a2[bi], i1))
    return matched, recs


def test_match_loop_closure_vs_python_restatement(oracle):
    rng = np.random.default_rng(5)
    n1, n2 = 120, 140
    d1 = rng.integers(0, 2**32, (n1, 8), dtype=np.uint64).astype(np.uint32)
    src = rng.integers(0, n1, n2)
    noise = np.packbits(rng.random((n2, 256)) < 0.05, axis=1, bitorder="little").view(np.uint32)
    d2 = d1[src] ^ noise
    b1 = rng.integers(0, 6, n1).astype(np.int32); b2 = b1[src].copy(); b2[::7] = 99
    a1 = rng.uniform(0, 360, n1).astype(np.float32); a2 = ((a1[src] + 50) % 360).astype(np.float32)
    u1 = (rng.random(n1) < 0.9).astype(np.uint8); u2 = (rng.random(n2) < 0.9).astype(np.uint8)
    n, m = oracle.match_loop_closure(d1, a1, u1, b1, d2, a2, u2, b2, 0.8, False)
    pm, recs = _m1_py(d1, a1, u1, b1, d2, a2, u2, b2, 0.8)
    assert np.array_equal(m, pm) and n == (pm >= 0).sum() and n > 30
    # each kf2 keypoint is consumed at most once (keyframe_matcher.cpp:128)
    mm = m[m >= 0]; assert len(set(mm.tolist())) == len(mm)
    # with the orientation check: the invalid set is exactly what the histogram says
    n2_, m2_ = oracle.match_loop_closure(d1, a1, u1, b1, d2, a2, u2, b2, 0.8, True)
    inv = oracle.angle_check(np.array([r[0] for r in recs], np.float32), np.array([r[1] for r in recs], np.int32))
    want = pm.copy(); want[inv] = -1
    assert np.array_equal(m2_, want) and n2_ == (want >= 0).sum()


def test_create_E21_and_epipolar_gate(oracle):
    rng = np.random.default_rng(6)
    def rot(v):
        th = np.linalg.norm(v); k = v / th; K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    R1, R2 = rot(rng.normal(size=3) * 0.2), rot(rng.normal(size=3) * 0.2); t1, t2 = rng.normal(size=3), rng.normal(size=3)
    E = oracle.create_E21(R1, t1, R2, t2)
    R21 = R2 @ R1.T; t21 = -R21 @ t1 + t2
    S = np.array([[0, -t21[2], t21[1]], [t21[2], 0, -t21[0]], [-t21[1], t21[0], 0]])
    assert np.allclose(E, S @ R21, atol=1e-14)
    # a true correspondence satisfies x2^T E x1 = 0
    X = rng.uniform(-2, 2, (50, 3)) + np.array([0, 0, 6.0])
    x1 = (R1 @ X.T).T + t1; x2 = (R2 @ X.T).T + t2
    assert np.abs(np.einsum("ni,ij,nj->n", x2, E, x1)).max() < 1e-10


def test_match_triangulation_last_wins_ties(oracle):
    """d > best -> continue (keyframe_matcher.cpp:231): among equal distances the LAST candidate of the node wins."""
    d1 = np.zeros((1, 8), np.uint32); d2 = np.zeros((3, 8), np.uint32)
    a = np.zeros(1, np.float32); a2 = np.zeros(3, np.float32)
    be1 = np.array([[0, 0, 1.0]]); be2 = np.tile(np.array([[0, 0, 1.0]]), (3, 1))
    E = oracle.create_E21(np.eye(3), np.zeros(3), np.eye(3), np.array([1.0, 0, 0]))
    sf = oracle.scale_factors(8, 1.2)
    n, m = oracle.match_triangulation(d1, a, np.zeros(1, np.int32), be1, np.ones(1, np.uint8), np.zeros(1, np.int32),
                                      d2, a2, be2, np.ones(3, np.uint8), np.zeros(3, np.int32), E, sf, 1.0, False)
    assert n == 1 and m[0] == 2
    # the epipolar gate rejects a bearing far off the epipolar plane
    be2b = np.array([[0, 0.5, 1.0]] * 3); be2b /= np.linalg.norm(be2b, axis=1, keepdims=True)
    n, m = oracle.match_triangulation(d1, a, np.zeros(1, np.int32), be1, np.ones(1, np.uint8), np.zeros(1, np.int32),
                                      d2, a2, be2b, np.ones(3, np.uint8), np.zeros(3, np.int32), E, sf, 1.0, False)
    assert n == 0 and m[0] == -1


def test_descriptor_medoid_follows_update_descriptor(oracle):
    """map_point.cpp:75-116 in numpy: sort each row of the distance matrix, take element (n-1)//2, first strict minimum."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 7, 16, 33):
        d = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32)
        d[n // 2] = d[0]                                            # a duplicate, so medians tie
        bits = np.unpackbits(d.view(np.uint8), axis=1).astype(np.int32)
        mat = (bits[:, None, :] != bits[None, :, :]).sum(-1)
        med = np.sort(mat, axis=1)[:, int(0.5 * (n - 1))]
        assert oracle.descriptor_medoid(d) == int(np.argmin(med)), n
    assert oracle.descriptor_medoid(np.zeros((0, 8), np.uint32)) == -1


def test_features_around_is_the_sorted_radius_query(oracle):
    """feature_search.cpp:33-48 in numpy: points sorted by y (stable), those with y in [qy - r, qy + r] and dx^2 + dy^2 < r^2 (float32)."""
    rng = np.random.default_rng(8)
    x = np.round(rng.uniform(0, 640, 500)).astype(np.float32); y = np.round(rng.uniform(0, 480, 500)).astype(np.float32)
    order = np.argsort(y, kind="stable"); sx, sy = x[order], y[order]
    for _ in range(200):
        qx, qy, r = np.float32(rng.uniform(-10, 650)), np.float32(rng.uniform(-10, 490)), np.float32(rng.uniform(0, 80))
        dx, dy = qx - sx, qy - sy
        inside = (sy >= qy - r) & (sy <= qy + r) & ((dx * dx + dy * dy).astype(np.float32) < r * r)
        assert np.array_equal(oracle.features_around(sx, sy, qx, qy, r), np.nonzero(inside)[0])
