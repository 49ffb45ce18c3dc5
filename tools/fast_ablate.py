"""Ablation of k_fast: with threshold 254 nothing survives the compass pre-test, so the stage time is phase A1 + tile overhead."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
ctx = mi355slam.Context(0)
B = 64
frames = np.stack([mso.synth_frame(1280, 720, 1000 + i // 8, 2 * (i % 8), i % 8) for i in range(B)])
buf = ctx.upload(frames)
for thr in (20, 60, 254):
    ex = mi355slam.OrbExtractor(ctx, 1280, 720, fast_threshold=thr, max_batch=B)
    ex.set_profiling(True)
    for _ in range(3):
        ex.extract(buf, n_frames=B, frame_stride=1280 * 720, row_stride=1280)
        t = ex.stage_ms()
    n = np.mean([len(ex.download(f)["x"]) for f in range(4)])
    print("thr", thr, "kpts/frame", n, {k: round(v * 256 / B, 3) for k, v in t.items()}, "(ms scaled to 256 frames)")
    ex.close()
