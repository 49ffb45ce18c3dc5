"""Where the linearisation (build_system in ba.hip) of a 256-window launch spends its cycles, per wave of workgroup 0: needs a library built with the stamps,
   bash tools/ba_variants.sh linprof "-DMS_LIN_PROF"   ->  python tools/ba_lin_prof.py tools/variants/lib_linprof.so [windows]
Stamps (cycles since the start of the LAST call): 0 Hpp / bp zeroed, 1 barrier, 2 point table zeroed + observations of fixed poses, 3 per-pose pass, 4 SE3 edges, 5 end"""
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
ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
a = (ctypes.c_longlong * (32 * 8 * 6))(); lib.ms_debug_linprof(a); d = list(a)
st = ba.download(0)["stats"]
print("windows %d: %.3f ms per launch, phase cycles %s" % (nb, ms, st["phase_cycles"]))
names = ["zero Hpp", "barrier", "fixed-pose obs", "per-pose pass", "SE3 edges", "end barrier"]
for w in range(8):
    v = d[6 * w: 6 * w + 6]
    print("wave %d: " % w + "  ".join("%s %d" % (n, x - (v[i - 1] if i else 0)) for i, (n, x) in enumerate(zip(names, v))) + "   total %d" % v[5])
