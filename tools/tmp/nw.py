import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
probs = [ba_synth.make_problem_fast(seed=42 + i) for i in range(4)]
for i in range(4):
    b = mi355slam.BundleAdjuster(ctx, [probs[i % 4]], max_iters=10)
    for rep in range(3):
        t1 = time.perf_counter(); b.solve(); ctx.sync(); t2 = time.perf_counter()
        st = b.download(0)["stats"]
        print(i, rep, "solve %.2f ms" % ((t2 - t1) * 1e3), "fallbacks", b.team_fallbacks(), "trials", st["trials"], {k: int(v) // 1000 for k, v in st["phase_cycles"].items()}, flush=True)
    b.close()
