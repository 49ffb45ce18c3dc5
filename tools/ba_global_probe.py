"""Global bundle adjustment beyond the LDS panel (cholesky_factor_team): latency and phase shares by problem size and team."""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
for n_pose, n_point, run in ((220, 2200, 12), (500, 20000, 12), (1000, 40000, 12), (2000, 60000, 10)):
    t0 = time.time()
    prob = ba_synth.make_problem(n_pose, n_point, run, seed=5, fix_first=True, yaw_total=10.0 / n_pose if n_pose > 300 else 0.05, z_drift=1.0 / n_pose)
    tg = time.time() - t0
    for team, ft in ((64, 0), (64, 1), (64, 4), (64, 16), (64, 64)) if n_pose > 220 else ((1, 0), (32, 0), (32, 32)):
        t0 = time.time(); ba = mi355slam.BundleAdjuster(ctx, [prob], max_iters=10); tc = time.time() - t0
        ba.set_team(team); ba.set_factor_team(ft)
        ba.solve(); ctx.sync()
        ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
        st = ba.download(0)["stats"]; pc = st["phase_cycles"]
        print("%4d poses %6d points %7d obs (gen %.0f s, create %.1f s) team %2d factor team %2d : %9.2f ms  iters %d trials %d  chi2 %.3e -> %.3e  Mcycles %s"
              % (n_pose, n_point, len(prob["obs_pose"]), tg, tc, team, ft, ms, st["iters"], st["trials"], st["chi2_init"], st["chi2_final"], {k: round(v / 1e6, 1) for k, v in pc.items()}), flush=True)
        ba.close()
