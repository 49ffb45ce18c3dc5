"""Throughput of the widened rows: projection matcher core (radius query + scan), descriptor medoid, pose-only BA."""
import os, sys, ctypes as C
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam
ctx = mi355slam.Context(0)
L = mi355slam.lib()
rng = np.random.default_rng(0)
# --- projection candidates: 2000 keypoints, 2000 map points, radius ~ 15 px (searchByProjection per keyframe), 256 keyframes' worth of queries in one launch
n, nq = 2000, 2000 * 256
kx = rng.uniform(0, 1280, n).astype(np.float32); ky = rng.uniform(0, 720, n).astype(np.float32)
sx, sy, si = mi355slam.feature_search_sort(kx, ky)
t = rng.integers(0, 2**32, (n, 8), dtype=np.uint64).astype(np.uint32); q = rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32)
qx = rng.uniform(0, 1280, nq).astype(np.float32); qy = rng.uniform(0, 720, nq).astype(np.float32); qr = np.full(nq, 15.0, np.float32)
up = ctx.upload
d = dict(sx=up(sx), sy=up(sy), si=up(si), t=up(t), oct=up(rng.integers(0, 8, n).astype(np.int32)), qx=up(qx), qy=up(qy), qr=up(qr), q=up(q))
outs = [ctx.alloc(4 * nq + 16), ctx.alloc(2 * nq + 16), ctx.alloc(2 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16), ctx.alloc(4 * nq + 16)]
vp = mi355slam._vp
def proj():
    ctx.check(L.ms_projection_candidates(ctx._h, vp(d["sx"]), vp(d["sy"]), vp(d["si"]), n, vp(d["t"]), vp(d["oct"]), None, vp(d["qx"]), vp(d["qy"]), vp(d["qr"]), None, None,
                                         vp(d["q"]), nq, *[vp(o) for o in outs]), "proj")
proj(); ctx.sync()
ctx.timer_start(); proj(); ms = ctx.timer_stop_ms()
nc = outs[6].download(np.int32, (nq,)).mean()
print("projection candidates: %d queries x %d keypoints, radius 15 px (%.1f candidates each): %.3f ms = %.1f M queries/s" % (nq, n, nc, ms, nq / ms / 1e3))
# --- descriptor medoid: 100k map points x 8 observations
npts, k = 100000, 8
pool = rng.integers(0, 2**32, (npts * k, 8), dtype=np.uint64).astype(np.uint32)
start = (np.arange(npts + 1) * k).astype(np.int32); idx = np.arange(npts * k, dtype=np.int32)
dp, ds, di = up(pool), up(start), up(idx); bl, bp = ctx.alloc(4 * npts + 16), ctx.alloc(4 * npts + 16)
def med(): ctx.check(L.ms_descriptor_medoid(ctx._h, vp(dp), vp(ds), vp(di), npts, k, vp(bl), vp(bp)), "medoid")
med(); ctx.sync()
ctx.timer_start(); med(); ms = ctx.timer_stop_ms()
print("descriptor medoid: %d map points x %d observations: %.3f ms = %.1f M map points/s" % (npts, k, ms, npts / ms / 1e3))
# --- vocabulary-tree descent (N3): k = 10, L = 6 (1 111 111 nodes, 10^6 words -- the shape of the ORB vocabularies), 256 frames x 2000 descriptors
import time, mso, bow_synth
k, Lv = 10, 6
par, desc_l = [np.zeros(1, np.int32)], [np.zeros((1, 8), np.uint32)]
first = 0
for lv in range(1, Lv + 1):
    npar = len(desc_l[-1]); ids = first + np.arange(npar, dtype=np.int32)
    par.append(np.repeat(ids, k))
    base = rng.integers(0, 2**32, (npar * k, 8), dtype=np.uint64).astype(np.uint32) if lv == 1 else np.repeat(desc_l[-1], k, axis=0)
    flips = rng.integers(0, 2**32, base.shape, dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, base.shape, dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, base.shape, dtype=np.uint64).astype(np.uint32)
    desc_l.append(base ^ (flips if lv > 1 else 0))       # ~12 % of the bits differ from the parent
    first += npar
par = np.concatenate(par); nd = np.concatenate(desc_l); nn = len(par)
word = np.full(nn, -1, np.int32); word[nn - k**Lv:] = np.arange(k**Lv, dtype=np.int32)
wt = np.zeros(nn); wt[nn - k**Lv:] = rng.random(k**Lv) + 0.1
V = mi355slam.BowVocabulary(ctx, par, nd, wt, word, Lv)
nq = 2000 * 256
q = nd[rng.integers(1, nn, nq)] ^ (rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32) & rng.integers(0, 2**32, (nq, 8), dtype=np.uint64).astype(np.uint32))
dq = up(q); ow, owt, ond = ctx.alloc(4 * nq + 16), ctx.alloc(8 * nq + 16), ctx.alloc(4 * nq + 16)
def bow(): ctx.check(L.ms_bow_transform(ctx._h, V._h, vp(dq), nq, 4, vp(ow), vp(owt), vp(ond)), "bow")
bow(); ctx.sync()
ctx.timer_start(); bow(); ms = ctx.timer_stop_ms()
vocab = dict(parent=par, desc=nd, weight=wt, word=word, depth_levels=Lv)
ns = 20000
t0 = time.perf_counter(); cw, cwt, cnd = mso.bow_transform(vocab, q[:ns], 4); cpu = time.perf_counter() - t0
assert np.array_equal(ow.download(np.int32, (nq,))[:ns], cw) and np.array_equal(ond.download(np.int32, (nq,))[:ns], cnd)
print("vocabulary descent: %d descriptors down a k=10 L=6 tree (%d nodes): %.3f ms = %.1f M descriptors/s (%.0f frames of 2000/s); CPU oracle 1 thread %.2f M descriptors/s"
      % (nq, nn, ms, nq / ms / 1e3, 256 / ms * 1e3, ns / cpu / 1e6))
