import sys, os, time
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in ("slam-module_amd","oracle","tests"): sys.path.insert(0, os.path.join(R,p))
import numpy as np, mi355slam, ba_synth
ctx=mi355slam.Context(0)
p=ba_synth.make_problem()
for nb in (1, 64):
    ba=mi355slam.BundleAdjuster(ctx,[p]*nb,max_iters=10)
    ba.solve(); ctx.sync()
    ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms=ctx.event_elapsed_ms(0,1)
    st=ba.download(0)["stats"]
    pc=st["phase_cycles"]; tot=pc["total"]
    print(nb, "ms", round(ms,2), "iters", st["iters"], "trials", st["trials"], {k: round(v/tot,3) for k,v in pc.items()}, "total Mcyc", round(tot/1e6,1))
