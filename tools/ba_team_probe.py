"""One C4 window solved by 1, 2, 4, ... workgroups: agreement with the single-workgroup result and latency."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
prob = ba_synth.make_problem()
ref = None
for team in (1, 2, 4, 8, 16, 32, 64):
    ba = mi355slam.BundleAdjuster(ctx, [prob], max_iters=10)
    ba.set_team(team)
    ba.solve(); ctx.sync()
    ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
    out = ba.download(0)
    st = out["stats"]
    if ref is None: ref = out
    dp = np.abs(out["pose"] - ref["pose"]).max(); dx = np.abs(out["point"] - ref["point"]).max()
    pc = st["phase_cycles"]; tot = pc["total"]
    print("team %2d : %7.3f ms  iters %d trials %d  chi2 %.6f  max|dpose| %.2e  max|dpoint| %.2e  %s" % (team, ms, st["iters"], st["trials"], st["chi2_final"], dp, dx, {k: round(v / 1e6, 2) for k, v in pc.items()}), flush=True)
    ba.close()
