"""Latency of the small BA problems of the per-frame path: poseBundleAdjust (bundle_adjuster.cpp:396-491: ONE free pose, every point fixed, the keypoints of the
current frame as observations) and stage 1 of localBundleAdjust (one free pose + all points of the window free), per team size."""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests", "tools"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)


def timed(ba, reps=30):
    for _ in range(3): ba.solve()
    ctx.sync(); ctx.event_mark(0)
    for _ in range(reps): ba.solve()
    ctx.event_mark(1)
    return ctx.event_elapsed_ms(0, 1) / reps


# pose-only: the newest keyframe of a C4-like window against 1200 of its map points (all fixed), 12 iterations
w = ba_synth.make_problem_fast(50, 2000, 10, seed=5)
cur = 25
sel = np.flatnonzero(w["obs_pose"] == cur)
pts = np.unique(w["obs_point"][sel])
remap = -np.ones(len(w["point"]), np.int64); remap[pts] = np.arange(len(pts))
pose_only = dict(pose=w["pose"][cur:cur + 1].copy(), pose_fixed=np.zeros(1, np.uint8), point=w["point"][pts].copy(), point_fixed=np.ones(len(pts), np.uint8),
                 obs_pose=np.zeros(len(sel), np.int32), obs_point=remap[w["obs_point"][sel]].astype(np.int32), obs_uv=w["obs_uv"][sel].copy(), obs_info=w["obs_info"][sel].copy(),
                 huber_delta=w["huber_delta"], edge_i=np.zeros(0, np.int32), edge_j=np.zeros(0, np.int32), edge_meas=np.zeros((0, 7)), edge_info=np.zeros((0, 36)))
print("pose-only problem: %d observations of %d fixed points" % (len(sel), len(pts)))
for team in (0, 1, 2, 4, 8):
    ba = mi355slam.BundleAdjuster(ctx, [pose_only], max_iters=12); ba.set_team(team)
    ms = timed(ba); st = ba.download(0)["stats"]
    t0 = time.perf_counter()
    for _ in range(10):
        b = mi355slam.BundleAdjuster(ctx, [pose_only], max_iters=12); b.set_team(team); b.solve(); b.download(0); b.close()
    new_ms = (time.perf_counter() - t0) / 10 * 1e3
    print("  pose-only, team %d (0 / 1 = k_ba_pose_only unless MS_BA_NO_POSE_KERNEL; >= 2 = the general kernel): %.3f ms per solve (%d iterations, %d trials); create + solve + download + destroy %.3f ms" % (team, ms, st["iters"], st["trials"], new_ms), flush=True)
    ba.close()
# stage 1 of the two-stage schedule: one free pose, all points free
s1 = dict(w); s1["pose_fixed"] = np.ones(50, np.uint8); s1["pose_fixed"][cur] = 0
for team in (0, 1, 4, 8, 16, 32):
    ba = mi355slam.BundleAdjuster(ctx, [s1], max_iters=8); ba.set_team(team)
    print("  stage 1 (1 free pose, 2000 free points, 20 k observations), team %d: %.3f ms per solve" % (team, timed(ba)), flush=True)
    ba.close()
s2 = dict(w)
for team in (0, 32):
    ba = mi355slam.BundleAdjuster(ctx, [s2], max_iters=8); ba.set_team(team)
    print("  all poses free, 8 iterations, team %d: %.3f ms per solve" % (team, timed(ba)), flush=True)
    ba.close()
