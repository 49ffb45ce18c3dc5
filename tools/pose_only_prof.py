"""Where k_ba_pose_only's time goes (poseBundleAdjust, bundle_adjuster.cpp:396-491): the deployment-shape problem (one free keyframe, its fixed map points, the
odometry edge to the fixed previous keyframe) with and without the edge; cycles of thread 0 per phase from the kernel's own stamps."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests", "tools"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
w = ba_synth.make_problem_fast(50, 2000, 10, seed=5)
full = ba_synth.pose_only_from_window(w, 25)
noedge = dict(full); noedge["edge_i"] = np.zeros(0, np.int32); noedge["edge_j"] = np.zeros(0, np.int32); noedge["edge_meas"] = np.zeros((0, 7)); noedge["edge_info"] = np.zeros((0, 36))
for name, p in (("with the odometry edge", full), ("without it", noedge)):
    for nb in (1, 256):
        ba = mi355slam.BundleAdjuster(ctx, [p] * nb, max_iters=10)
        for _ in range(3): ba.solve()
        ctx.sync(); ctx.event_mark(0)
        for _ in range(20): ba.solve()
        ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1) / 20
        st = ba.download(0)["stats"]; pc = st["phase_cycles"]
        print("%-24s x%-3d %.4f ms per launch, %d observations, %d iterations / %d trials; cycles of thread 0: observations %d, edges %d, reductions %d, solve + exp %d, total %d (+ %d before the first sweep)"
              % (name, nb, ms, len(p["obs_pose"]), st["iters"], st["trials"], pc["eval"], pc["linearise"], pc["schur"], pc["cholesky"], pc["total"], pc["schur_init"]), flush=True)
        ba.close()
