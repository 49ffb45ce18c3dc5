"""Differential fuzz of the two BoW-guided greedy matchers (M1 matchForLoopClosures, M2 matchForTriangulationDBoW) against the CPU oracle: random keyframe
sizes, vocabularies from 1 to 600 nodes (node sizes from 0 to the whole keyframe: registers, LDS staging and the beyond-LDS tail of k_greedy_big_nodes), usable
masks, descriptor entropy (ties), ratios and thresholds; every execution path of ms_match_set_path.  usage: python tools/match_fuzz.py [N] [seed]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
ctx = mi355slam.Context(0)
sf = mso.scale_factors(8, 1.2)
bad = 0
for case in range(N):
    n1 = int(rng.choice([0, 1, 63, 64, 65, 300, 1000, 2000, 3000])) if rng.random() < 0.5 else int(rng.integers(1, 2600))
    n2 = int(rng.choice([0, 1, 64, 65, 256, 257, 2048, 2049, 2600])) if rng.random() < 0.5 else int(rng.integers(1, 2600))
    nb = int(rng.choice([1, 2, 5, 30, 100, 600]))
    q = rng.integers(0, 2**32, (n1, 8), dtype=np.uint64).astype(np.uint32)
    t = rng.integers(0, 2**32, (n2, 8), dtype=np.uint64).astype(np.uint32)
    if n1 and n2:
        src = rng.integers(0, n1, n2)
        close = rng.random(n2) < 0.6
        noise = np.packbits(rng.random((n2, 256)) < rng.choice([0.02, 0.08, 0.15]), axis=1, bitorder="little").view(np.uint32)
        t = np.where(close[:, None], q[src] ^ noise, t)
    if rng.random() < 0.3:
        m = np.uint32(rng.choice([0x1, 0x7, 0xFF])); q &= m; t &= m                       # low entropy: ties decide
    b1 = rng.integers(0, nb, n1).astype(np.int32); b2 = rng.integers(0, nb, n2).astype(np.int32)
    if n1 and n2 and rng.random() < 0.7: b2 = np.where(rng.random(n2) < 0.85, b1[src], b2).astype(np.int32)
    a1 = rng.uniform(0, 360, n1).astype(np.float32); a2 = rng.uniform(0, 360, n2).astype(np.float32)
    if n1 and n2: a2 = np.where(rng.random(n2) < 0.7, (a1[src] + 50 + rng.normal(0, 5, n2)) % 360, a2).astype(np.float32)
    u1 = (rng.random(n1) < rng.choice([0.3, 0.8, 1.0])).astype(np.uint8); u2 = (rng.random(n2) < rng.choice([0.3, 0.8, 1.0])).astype(np.uint8)
    be1 = rng.normal(size=(n1, 3)); be1 /= np.maximum(np.linalg.norm(be1, axis=1, keepdims=True), 1e-9)
    be2 = (be1[src] + 0.02 * rng.normal(size=(n2, 3))) if n1 and n2 else rng.normal(size=(n2, 3))
    be2 /= np.maximum(np.linalg.norm(be2, axis=1, keepdims=True), 1e-9)
    o1 = rng.integers(0, 8, n1).astype(np.int32)
    E = rng.normal(size=(3, 3))
    ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0])); thr = float(rng.choice([1.0, 10.0, 40.0])); chk = bool(rng.random() < 0.8)
    wm1 = mso.match_loop_closure(q, a1, u1, b1, t, a2, u2, b2, ratio, chk)
    wm2 = mso.match_triangulation(q, a1, o1, be1, u1, b1, t, a2, be2, u2, b2, E, sf, thr, chk)
    f1m = mi355slam.FrameOnDevice(ctx, q.reshape(-1, 8), a1, u1, b1); f2m = mi355slam.FrameOnDevice(ctx, t.reshape(-1, 8), a2, u2, b2)
    f1t = mi355slam.FrameOnDevice(ctx, q.reshape(-1, 8), a1, u1, b1, octave=o1, bearing=be1); f2t = mi355slam.FrameOnDevice(ctx, t.reshape(-1, 8), a2, u2, b2, bearing=be2)
    for path in (0, 2, 1):
        ctx.set_match_path(path)
        c1, m1 = mi355slam.match_loop_closure(ctx, [f1m], [f2m], ratio, chk)
        c2, m2 = mi355slam.match_triangulation(ctx, [f1t], [f2t], E.reshape(1, 3, 3), sf, thr, chk)
        ok = c1[0] == wm1[0] and np.array_equal(m1[0], wm1[1]) and c2[0] == wm2[0] and np.array_equal(m2[0], wm2[1])
        if not ok:
            bad += 1
            print("MISMATCH case", case, "path", path, dict(n1=n1, n2=n2, nb=nb, ratio=ratio, thr=thr, chk=chk), "M1", c1[0], wm1[0], "M2", c2[0], wm2[0], flush=True)
    ctx.set_match_path(0)
    if case % 10 == 9: print("case", case + 1, "of", N, "mismatches so far", bad, "(last: n1 %d n2 %d nodes %d, M1 %d / M2 %d matches)" % (n1, n2, nb, wm1[0], wm2[0]), flush=True)
print("match fuzz:", N, "cases x 3 paths x 2 matchers,", bad, "mismatches")
sys.exit(1 if bad else 0)
