"""Local BA, batch throughput against the launch shape: windows per launch x library variant (tools/ba_variants.sh: LDS budget, observations per
Schur batch, register budget).  Every variant runs in a process of its own (one copy of the library per process); the first windows' final chi2
and LM trajectory are printed so that a variant that computes something else shows at once.
usage: python tools/ba_occupancy_probe.py [lib.so ...]      (no argument: the tree's library)"""
import os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for L in sys.argv[1:]:
        print("==", L, flush=True)
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", L], timeout=600)
        if rc != 0: print("   variant failed with", rc, flush=True)
    sys.exit(0)
for p in ("slam-module_amd", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, ba_synth
if len(sys.argv) > 2: mi355slam.LIB_PATH = os.path.join(R, sys.argv[2])
ctx = mi355slam.Context(0)
sizes = [int(x) for x in os.environ.get("BA_PROBE_SIZES", "256,512,1024").split(",")]
probs = [ba_synth.make_problem_fast(50, 2000, 10, seed=42 + i) for i in range(max(sizes))]
for nb in sizes:
    ba = mi355slam.BundleAdjuster(ctx, probs[:nb], max_iters=10)
    ba.solve(); ctx.sync()
    ctx.event_mark(0)
    for _ in range(2): ba.solve()
    ctx.event_mark(1); ms = ctx.event_elapsed_ms(0, 1) / 2
    st = [ba.download(i)["stats"] for i in (0, nb - 1)]
    pc = st[0]["phase_cycles"]; tot = max(pc["total"], 1)
    print("windows %4d : %8.3f ms per launch, %8.1f solves/s   chi2 %.6f / %.6f  trials %d / %d   " % (nb, ms, nb / ms * 1e3, st[0]["chi2_final"], st[1]["chi2_final"], st[0]["trials"], st[1]["trials"]) +
          "  ".join("%s %.1f%%" % (k, 100 * v / tot) for k, v in pc.items() if k != "total") + "  total %.2f Mcyc" % (tot / 1e6), flush=True)
    ba.close()
