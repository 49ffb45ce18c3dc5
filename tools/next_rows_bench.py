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
