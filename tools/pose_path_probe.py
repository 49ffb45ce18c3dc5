"""The per-frame cost of poseBundleAdjust as a NEW problem (what every non-keyframe pays, mapper_helpers.cpp:1043-1050): where the time outside the kernel goes.
Times the Python wrapper's argument marshalling, ms_ba_create, ms_ba_solve (asynchronous), ms_ba_download (waits) and ms_ba_destroy separately."""
import sys, os, time, ctypes as C
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests", "tools"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
from mi355slam import lib, _ba_struct, BaProblemC, BaResultC, _vp
ctx = mi355slam.Context(0)
w = ba_synth.make_problem_fast(50, 2000, 10, seed=5)
p = ba_synth.pose_only_from_window(w, 25)
N = 200
t = dict(marshal=0.0, create=0.0, solve=0.0, download=0.0, destroy=0.0)
for rep in range(N + 20):
    if rep == 20: t = {k: 0.0 for k in t}
    t0 = time.perf_counter()
    s, keep = _ba_struct(p, 10); arr = (BaProblemC * 1)(s); h = C.c_void_p()
    t1 = time.perf_counter()
    ctx.check(lib().ms_ba_create(ctx._h, arr, 1, C.byref(h)), "create")
    t2 = time.perf_counter()
    ctx.check(lib().ms_ba_solve(h), "solve")
    t3 = time.perf_counter()
    pose, point, chi2 = np.zeros((s.n_pose, 7)), np.zeros((s.n_point, 3)), np.zeros(s.n_obs); r = BaResultC()
    ctx.check(lib().ms_ba_download(h, 0, _vp(pose), _vp(point), _vp(chi2), C.byref(r)), "download")
    t4 = time.perf_counter()
    lib().ms_ba_destroy(h)
    t5 = time.perf_counter()
    for k, d in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): t[k] += d
print("pose-only new problem, %d observations: per call, microseconds: %s; total %.1f" % (s.n_obs, {k: round(v / N * 1e6, 1) for k, v in t.items()}, sum(t.values()) / N * 1e6))
