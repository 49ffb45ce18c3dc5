"""One C4 window on a team, timed alone: milliseconds per solve and the in-kernel phase cycles of the lead workgroup."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
probs = [ba_synth.make_problem_fast(seed=42)]
ba = mi355slam.BundleAdjuster(ctx, probs, max_iters=10); ba.set_team(0)
ba.solve(); ctx.sync()
ts = []
for _ in range(20):
    ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ts.append(ctx.event_elapsed_ms(0, 1))
r = ba.download(0); st = r["stats"]
print("one window: min %.3f median %.3f ms  trials %d chi2 %.17g" % (min(ts), sorted(ts)[10], st["trials"], st["chi2_final"]), {k: int(v) for k, v in st["phase_cycles"].items()})
