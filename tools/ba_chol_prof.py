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
names = {0: "init", 1: "seg1 work", 2: "barrier 1", 4: "seg2 work", 5: "barrier 2", 6: "bs init", 8: "bs prefetch issue", 10: "bs chain", 11: "bs store", 12: "bs barrier", 13: "bs out",
         19: "w pairs", 20: "w z update", 21: "w write-back", 22: "w pf store"}
d7 = max(d[7], 1)
for who, v in (("wave 0", d[:24]), ("wave 1", w1raw)):
    tot = sum(x for i, x in enumerate(v) if i != 7)
    print(who, "total per call %.0f" % (tot / d7))
    for i, x in enumerate(v):
        if i != 7 and x: print("   %-20s %9.0f cyc/call %5.1f%%" % (names.get(i, str(i)), x / d7, 100 * x / tot))
