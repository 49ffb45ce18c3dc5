"""Latency of the small BA problems: pose-only BA (every non-keyframe, mapper_helpers.cpp:1043-1050) and stage 1 of the local BA."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
full = ba_synth.make_problem()
cases = {"C4 local BA (50 free poses)": full}
p1 = dict(full); p1["pose_fixed"] = np.ones(50, np.uint8); p1["pose_fixed"][49] = 0
cases["stage 1 (one free pose, points free)"] = p1
p2 = dict(p1); p2["point_fixed"] = np.ones(2000, np.uint8)
cases["pose BA (one free pose, points fixed)"] = p2
for name, p in cases.items():
    for nb, team in ((1, 1), (1, 0), (256, 0)):
        ba = mi355slam.BundleAdjuster(ctx, [p] * nb, max_iters=10)
        ba.set_team(team)
        ba.solve(); ctx.sync()
        ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
        st = ba.download(0)["stats"]
        pc = st["phase_cycles"]; tot = pc["total"]
        print("%-40s x%-3d team %d %8.3f ms  iters %d trials %d  %s" % (name, nb, team, ms, st["iters"], st["trials"], {k: round(v / tot, 2) for k, v in pc.items() if k != "total"}), flush=True)
        ba.close()
