"""Where one factorisation of the windowed Cholesky spends its cycles (cholesky_window in ba.hip): build the library with the stamps first,
   touch slam-module_amd/csrc/ba.hip && make -C slam-module_amd/csrc BA_EXTRA=-DMS_CW_PROF   (then touch and rebuild without it).  Prints per-call cycle sums of thread 0 (the factoring wave) and thread 64."""
import sys, os, ctypes
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
ctx = mi355slam.Context(0)
lib = ctypes.CDLL(os.path.join(R, "slam-module_amd/lib/libmi355slam.so"))
probs = [ba_synth.make_problem_fast(seed=42)]
ba = mi355slam.BundleAdjuster(ctx, probs, max_iters=10); ba.set_team(0)
ba.solve(); ctx.sync()
a = (ctypes.c_longlong * 48)(); lib.ms_debug_cwprof(a); b0 = list(a)
ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1)
lib.ms_debug_cwprof(a); d = [x - y for x, y in zip(list(a), b0)]
w1raw = d[24:48]
st = ba.download(0)["stats"]
print("ms", ms, "trials", st["trials"], "calls", d[7], st["phase_cycles"])
names = ["init", "A", "A-wait", "B-update-next", "B-factor", "B-wait", "bs-init", "", "bs-part", "bs-wait1", "bs-recur", "bs-store", "bs-wait2", "bs-out", "factor-core", "A-load", "A-solve", "A-store"]
d7 = d[7]; d[7] = 0
tot = sum(d[:18])
d = d[:7] + [0] + d[8:]
d.insert(0, 0); d.pop(0)
for n, v in zip(names, d[:18]): print("%-14s %9.0f cyc/call  %5.1f%%" % (n, v / max(d7, 1), 100 * v / tot))
print("total per call", tot / max(d7, 1))

w1 = d[24:48] if len(d) >= 48 else None
for n, i in (("w1 prefetch-issue", 18), ("w1 update pairs", 19), ("w1 z update", 20), ("w1 write-back", 21), ("w1 pf store (load wait)", 22), ("w1 rest", 23), ("w1 barrier wait", 5), ("w1 A", 1), ("w1 A-wait", 2)):
    print("%-26s %9.0f cyc/call" % (n, w1raw[i] / max(d7, 1)))
