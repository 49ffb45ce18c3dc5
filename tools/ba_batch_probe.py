"""Local-BA throughput against the number of windows per launch (do two problems per CU pay?)."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
probs = [ba_synth.make_problem(seed=42 + i) for i in range(4)]
for nb in (1, 64, 256, 512, 768):
    ba = mi355slam.BundleAdjuster(ctx, [probs[i % 4] for i in range(nb)], max_iters=10)
    ba.solve(); ctx.sync()
    ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
    print("windows %4d : %8.2f ms per launch, %8.1f solves/s" % (nb, ms, nb / ms * 1e3), flush=True)
    ba.close()
