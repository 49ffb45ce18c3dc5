"""Times stage 1 of localBundleAdjust (one free keyframe + all points free, 8 iterations) on the C4 window: one window with several team sizes and lanes per
point, the general kernel beside it (MS_BA_NO_ONE_POSE_KERNEL), and a batch of 256 windows.  Usage: python tools/ba_stage1_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "slam-module_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import mi355slam, ba_synth


def stage1(p, cur):
    s = dict(p); s["pose_fixed"] = np.ones(len(p["pose"]), np.uint8); s["pose_fixed"][cur] = 0
    return s


def timed(ctx, ba, reps=20):
    ba.solve(); ctx.sync()
    ctx.event_mark(10)
    for _ in range(reps): ba.solve()
    ctx.event_mark(11); ctx.sync()
    return ctx.event_elapsed_ms(10, 11) / reps


def main():
    ctx = mi355slam.Context(0)
    iters = 8
    p = stage1(ba_synth.make_problem(), 49)
    one = mi355slam.BundleAdjuster(ctx, [p], max_iters=iters)
    os.environ["MS_BA_NO_ONE_POSE_KERNEL"] = "1"
    one.set_team(0); print("general kernel, automatic team: %.3f ms" % timed(ctx, one), one.download(0)["stats"])
    one.set_team(1); print("general kernel, one workgroup: %.3f ms" % timed(ctx, one, 5))
    del os.environ["MS_BA_NO_ONE_POSE_KERNEL"]
    one.set_team(0); print("one-pose kernel, automatic: %.3f ms" % timed(ctx, one), one.download(0)["stats"])
    for team in (1, 2, 4, 8, 16, 32):
        for lanes in (1, 2, 4, 8):
            os.environ["MS_BA_ONE_POSE_LANES"] = str(lanes)
            one.set_team(team)
            ms = timed(ctx, one)
            ph = one.download(0)["stats"]["phase_cycles"]      # k_ba_one_pose: chi2 sweeps | linearise (+ Schur terms) | reduction | 6 x 6 solve | points + trial chi2 | total | reduction
            print("one-pose kernel, team %2d, %d lanes per point: %.3f ms   kcycles: chi2 %d, linearise %d, reduce %d, solve %d, trial %d, reduce %d, total %d" %
                  (team, lanes, ms, ph["eval"] / 1e3, ph["linearise"] / 1e3, ph["schur"] / 1e3, ph["cholesky"] / 1e3, ph["points_update"] / 1e3, ph["schur_init"] / 1e3, ph["total"] / 1e3))
    del os.environ["MS_BA_ONE_POSE_LANES"]
    t0 = time.perf_counter()
    for _ in range(10):
        h = mi355slam.BundleAdjuster(ctx, [p], max_iters=iters); h.solve(); h.download(0); h.close()
    print("new stage-1 window (create + solve + download + destroy): %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    one.close()
    probs = [stage1(ba_synth.make_problem_fast(seed=100 + i), 49) for i in range(256)]
    b = mi355slam.BundleAdjuster(ctx, probs, max_iters=iters)
    print("256 windows, one-pose kernel: %.3f ms per launch" % timed(ctx, b, 5))
    os.environ["MS_BA_NO_ONE_POSE_KERNEL"] = "1"
    print("256 windows, general kernel: %.3f ms per launch" % timed(ctx, b, 3))
    b.close()


if __name__ == "__main__":
    main()
