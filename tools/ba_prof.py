"""Phase shares of the local-BA kernel (in-kernel clock64 stamps of workgroup 0's problem): one window on one workgroup, one window on
a team, and a 256-window launch (distinct windows)."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
probs = [ba_synth.make_problem_fast(seed=42 + i) for i in range(256)]
for nb, team in ((1, 1), (1, 0), (256, 0)):
    ba = mi355slam.BundleAdjuster(ctx, probs[:nb], max_iters=10)
    ba.set_team(team)
    ba.solve(); ctx.sync()
    ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
    st = ba.download(0)["stats"]
    pc = st["phase_cycles"]; tot = pc["total"]
    print("windows %3d team %s: %7.3f ms  iters %d trials %d  total %.1f Mcyc  " % (nb, team or "auto", ms, st["iters"], st["trials"], tot / 1e6) +
          "  ".join("%s %.1f%%" % (k, 100 * v / tot) for k, v in pc.items() if k != "total"), flush=True)
    ba.close()
