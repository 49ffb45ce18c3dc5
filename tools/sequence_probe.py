"""One SLAM sequence as a sequential pipeline sees it (config C5, one sequence per GPU): per frame extract -> match against the
previous frame -> ratio test, and every 5th frame (a keyframe) one local bundle adjustment of a 50-keyframe window.  Nothing is
batched across frames; the numbers are what a single backend thread driving the library would get."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "tools", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, synth, ba_synth
ctx = mi355slam.Context(0)
W, H, N = 1280, 720, 64
g = synth.SequenceSynth(W, H, 1000, 2 * (N - 1), N - 1)
frames = np.stack([g.frame(2 * i, i) for i in range(N)])
buf = ctx.upload(frames)
ex = [mi355slam.OrbExtractor(ctx, W, H, max_batch=1) for _ in range(2)]           # ping-pong: the previous frame's keypoints stay on the device
views = [None, None]
cap = ex[0].capacity
bi, bd, sd, match = ctx.alloc(4 * cap + 16), ctx.alloc(2 * cap + 16), ctx.alloc(2 * cap + 16), ctx.alloc(4 * cap + 16)
ba = mi355slam.BundleAdjuster(ctx, [ba_synth.make_problem()], max_iters=10)           # C4 window; solve() restarts from its initial estimates
def frame(i, with_ba):
    e = ex[i & 1]
    e.extract(buf.ptr + i * W * H, n_frames=1, frame_stride=W * H, row_stride=W)
    if views[i & 1] is None: views[i & 1] = e.device_view()
    if i and views[(i - 1) & 1] is not None:
        q, t = views[i & 1], views[(i - 1) & 1]
        mi355slam.hamming_best2_sets(ctx, q.desc, cap, q.count, t.desc, cap, t.count, None, None, 1, bi, bd, sd)
        mi355slam.ratio_test_device(ctx, bi, bd, sd, cap, 0.75, 50, match)
    if with_ba and i % 5 == 0: ba.solve()
for i in range(8): frame(i, True)
ctx.sync()
for with_ba in (False, True):
    t0 = time.perf_counter(); ctx.timer_start()
    reps = 4
    for r in range(reps):
        for i in range(N): frame(i, with_ba)
    gpu_ms = ctx.timer_stop_ms(); wall = (time.perf_counter() - t0) * 1e3
    n = reps * N
    m = match.download(np.int32, (cap,))
    print("sequence of 720p frames, extract + match%s: %.3f ms per frame on the GPU (%.3f ms wall incl. Python) = %.0f frames/s for ONE sequence; last frame: %d matches"
          % (" + a C4 local BA every 5th frame" if with_ba else "", gpu_ms / n, wall / n, n / (gpu_ms * 1e-3), int((m >= 0).sum())), flush=True)
