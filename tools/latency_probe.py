"""Latency of ONE frame through the front end (extract, then match against the previous frame), as a sequential pipeline sees it."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
ctx = mi355slam.Context(0)
for (w, h) in ((640, 480), (1280, 720)):
    frames = np.stack([mso.synth_frame(w, h, 1000 + i, 2 * i, i) for i in range(8)])
    buf = ctx.upload(frames)
    ex = mi355slam.OrbExtractor(ctx, w, h, max_batch=1)
    ex.set_profiling(True)
    for f in range(8): ex.extract(buf.ptr + f * w * h, n_frames=1, frame_stride=w * h, row_stride=w)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.timer_start()
    N = 200
    for i in range(N): ex.extract(buf.ptr + (i % 8) * w * h, n_frames=1, frame_stride=w * h, row_stride=w)
    gpu_ms = ctx.timer_stop_ms() / N
    wall_ms = (time.perf_counter() - t0) * 1e3 / N
    print("%dx%d one frame: %.3f ms GPU time per extract (%.3f ms wall incl. Python), stages %s" % (w, h, gpu_ms, wall_ms, {k: round(v, 4) for k, v in ex.stage_ms().items()}), flush=True)
