import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
probs = [ba_synth.make_problem_fast(seed=42 + i) for i in range(4)]
for i in range(6):
    t0 = time.perf_counter(); b = mi355slam.BundleAdjuster(ctx, [probs[i % 4]], max_iters=10); t1 = time.perf_counter()
    b.solve(); ctx.sync(); t2 = time.perf_counter(); b.download(0); t3 = time.perf_counter(); b.close(); t4 = time.perf_counter()
    print("create %.2f solve %.2f download %.2f close %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
