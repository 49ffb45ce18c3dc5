"""In-kernel s_memtime stamps of k_fast (diagnostic build): where do the waves spend their cycles?"""
import ctypes as C, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
ctx = mi355slam.Context(0)
B = 64
frames = np.stack([mso.synth_frame(1280, 720, 1000 + i // 8, 2 * (i % 8), i % 8) for i in range(B)])
buf = ctx.upload(frames)
ex = mi355slam.OrbExtractor(ctx, 1280, 720, max_batch=B)
L = mi355slam.lib()
ex.extract(buf, n_frames=B, frame_stride=1280 * 720, row_stride=1280); ctx.sync()
assert L.ms_orb_fast_phase_cycles(ex._h, 1, None) == 0
ex.extract(buf, n_frames=B, frame_stride=1280 * 720, row_stride=1280); ctx.sync()
cyc = (C.c_double * 8)()
assert L.ms_orb_fast_phase_cycles(ex._h, 0, cyc) == 0
names = ["setup+clear", "A1 work", "A1 barrier", "A2 work", "A2 barrier", "NMS work", "NMS barrier", "atomic+barrier"]
tot = sum(cyc)
for n, v in zip(names, cyc): print("%-16s %6.1f %%" % (n, 100 * v / tot))
print("wave-cycles per wave:", tot / (452 * B * 4))
