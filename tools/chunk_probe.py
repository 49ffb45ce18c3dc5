"""How does the front end's time per frame depend on the frames per launch (does a cache-resident chunk pay for its launch gaps)?"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
ctx = mi355slam.Context(0)
N = 256
base = np.stack([mso.synth_frame(1280, 720, 1000 + i, 2 * (i % 8), i % 8) for i in range(8)])
frames = np.concatenate([base] * (N // 8))
buf = ctx.upload(frames)
for B in (8, 16, 32, 64, 128, 256):
    ex = mi355slam.OrbExtractor(ctx, 1280, 720, max_batch=B)
    def run():
        for c in range(N // B):
            ex.extract(buf.ptr + c * B * 1280 * 720, n_frames=B, frame_stride=1280 * 720, row_stride=1280)
    run(); ctx.sync()
    ctx.timer_start()
    for _ in range(3): run()
    ms = ctx.timer_stop_ms() / 3
    print("frames per launch %4d : %.3f ms per 256 frames (%.1f us per frame)" % (B, ms, 1e3 * ms / N), flush=True)
    del ex
