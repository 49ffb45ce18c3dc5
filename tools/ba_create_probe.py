import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests", "tools"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
probs = [ba_synth.make_problem_fast(50, 2000, 10, seed=42 + i) for i in range(8)]
for i in range(3):
    b = mi355slam.BundleAdjuster(ctx, [probs[i]], max_iters=10); b.solve(); b.download(0); b.close()
os.environ["MS_BA_TIMING"] = "1"
tc = ts = td = tx = 0.0
N = 8
for i in range(N):
    t0 = time.perf_counter(); b = mi355slam.BundleAdjuster(ctx, [probs[i]], max_iters=10)
    t1 = time.perf_counter(); b.solve(); ctx.sync()
    t2 = time.perf_counter(); b.download(0)
    t3 = time.perf_counter(); b.close()
    t4 = time.perf_counter()
    tc += t1 - t0; ts += t2 - t1; td += t3 - t2; tx += t4 - t3
print("per new window: create %.3f ms, solve + sync %.3f ms, download %.3f ms, destroy %.3f ms" % (tc / N * 1e3, ts / N * 1e3, td / N * 1e3, tx / N * 1e3))
