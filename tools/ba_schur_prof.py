"""Where the fused Schur pass of a 256-window launch spends its cycles, per wave of workgroup 0 (schur_fused in ba.hip): needs a library built with the stamps,
   bash tools/ba_variants.sh fsprof "-DMS_FS_PROF"   ->  python tools/ba_schur_prof.py tools/variants/lib_fsprof.so [windows]"""
import sys, os, ctypes
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
mi355slam.LIB_PATH = os.path.join(R, sys.argv[1])
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = mi355slam.Context(0)
lib = mi355slam.lib()
probs = [ba_synth.make_problem_fast(50, 2000, 10, seed=42 + i) for i in range(nb)]
ba = mi355slam.BundleAdjuster(ctx, probs, max_iters=10)
ba.solve(); ctx.sync()
a = (ctypes.c_longlong * 64)(); lib.ms_debug_fsprof(a); b0 = list(a)
ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
lib.ms_debug_fsprof(a); d = [x - y for x, y in zip(list(a), b0)]
st = ba.download(0)["stats"]
print("windows %d: %.3f ms per launch, trials %d, phase cycles %s" % (nb, ms, st["trials"], st["phase_cycles"]))
names = {0: "hand-out", 1: "batch top", 2: "jacobians+slab", 6: "prefetch issue", 7: "single/enumerated", 3: "chunk loop", 4: "last flush"}
for w in range(8):
    v = d[8 * w: 8 * w + 8]; tot = sum(v[i] for i in names)
    print("wave %d: %6.0f batches, %8.0f cycles per batch | " % (w, v[5], tot / max(v[5], 1)) + "  ".join("%s %.0f (%.1f%%)" % (names[i], v[i] / max(v[5], 1), 100 * v[i] / max(tot, 1)) for i in (0, 1, 2, 6, 7, 3, 4)))
