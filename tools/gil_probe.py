"""Is the slowdown of tools/hog_probe.py on the host?  The front end of one sequence (Python thread, ~15 ctypes calls per frame) beside N Python threads that make
ctypes calls in a loop WITHOUT any GPU work (a) cheap calls back to back, (b) a call + hipStreamSynchronize on an idle context."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "slam-module_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")): sys.path.insert(0, p)
import numpy as np
import bench, synth
import mi355slam
F = 600
mi355slam.prepare_process(16)
FD = 40
g = synth.SequenceSynth(bench.W, bench.H, 2000, 2 * (FD - 1), FD - 1)
frames = np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(FD)]))


def run(kind, n):
    start, stop = threading.Event(), threading.Event()
    seq = bench.SequenceRunner(0, 0, frames, [], 5, start, n_total=F)

    def loop():
        ctx = mi355slam.Context(0) if kind == "sync" else None
        L = mi355slam.lib()
        start.wait()
        while not stop.is_set():
            if kind == "sync": ctx.sync()
            else: L.ms_version()
        if ctx: ctx.close()
    ths = [threading.Thread(target=loop) for _ in range(n)]
    for t in ths: t.start()
    seq.start(); seq.ready.wait(); time.sleep(0.2)
    t0 = time.perf_counter(); start.set(); seq.join(); dt = time.perf_counter() - t0
    stop.set()
    for t in ths: t.join()
    return dt / F * 1e3


base = run("none", 0)
print("front end alone: %.3f ms per frame" % base)
for n in (1, 2, 4, 7):
    print("beside %d host-only threads: cheap ctypes calls %.3f ms per frame, ms_ctx_sync on idle contexts %.3f ms per frame" % (n, run("cheap", n), run("sync", n)), flush=True)
