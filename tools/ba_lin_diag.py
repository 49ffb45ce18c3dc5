import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in ("slam-module_amd","oracle","tests"): sys.path.insert(0, os.path.join(R,p))
import numpy as np, mi355slam, ba_synth
ctx=mi355slam.Context(0)
for team in (1, 32):
    ba=mi355slam.BundleAdjuster(ctx,[ba_synth.make_problem()],max_iters=10); ba.set_team(team)
    ba.solve(); ctx.sync()
    pc=ba.download(0)["stats"]["phase_cycles"]
    print(team, {k: round(v/1e6,3) for k,v in pc.items()})
