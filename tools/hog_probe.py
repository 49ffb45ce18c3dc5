"""(Round 4; the CU-partition column of profiles/r04_hog_probe.txt came from a build with ms_ctx_create_partitioned -- CU-masked streams, removed again: no gain.)
Why do the front end's small kernels run slower beside local-BA teams?  The front end of ONE sequence (bench.SequenceRunner without BA, per-frame rate) beside
N streams that keep (a) a real C4 window on a team of 32 workgroups, (b) 32 workgroups of tools/cu_hog.hip -- the same residency footprint, no memory traffic, no
fences, no barriers -- in flight back to back.   python tools/hog_probe.py [frames]"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "slam-module_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")): sys.path.insert(0, p)
import numpy as np
import bench, synth, ba_synth
import mi355slam

F = int(sys.argv[1]) if len(sys.argv) > 1 else 600
mi355slam.prepare_process(16)
hog = C.CDLL(os.path.join(ROOT, "tools", "variants", "libcuhog.so"))
FD = 40
g = synth.SequenceSynth(bench.W, bench.H, 2000, 2 * (FD - 1), FD - 1)
frames = np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(FD)]))
probs = [ba_synth.make_problem_fast(50, 2000, 10, seed=42 + i) for i in range(8)]


def run(kind, n_streams):
    start, stop = threading.Event(), threading.Event()
    seq = bench.SequenceRunner(0, 0, frames, [], 5, start, n_total=F)
    ready = []

    def loop(i):
        ctx = mi355slam.Context(0)
        sink = ctx.alloc(64)
        ba = mi355slam.BundleAdjuster(ctx, [probs[i]], max_iters=10) if kind == "team" else None
        if ba: ba.solve()
        ctx.sync(); ready.append(i)
        start.wait()
        while not stop.is_set():
            for _ in range(DEPTH):                          # DEPTH launches ahead: with 4 there is no gap on the stream while the host comes round again
                if ba: ba.solve()
                else: hog.hog_launch(32, -1750 if kind == "silent" else 1750, C.c_void_p(mi355slam.lib().ms_ctx_stream(ctx._h)), C.c_void_p(sink.ptr))
            if SLEEPWAIT: time.sleep(SLEEPWAIT * 1e-3)      # the host thread sleeps through the launch instead of waiting in hipStreamSynchronize
            ctx.sync()
        if ba: ba.close()
        ctx.close()
    ths = [threading.Thread(target=loop, args=(i,)) for i in range(n_streams)]
    for t in ths: t.start()
    while len(ready) < n_streams: time.sleep(0.01)
    if os.environ.get("HOG_FRONT_PRIORITY"): os.environ["MS_STREAM_PRIORITY"] = os.environ["HOG_FRONT_PRIORITY"]      # the front end's context stream alone gets the priority
    seq.start(); seq.ready.wait()
    os.environ.pop("MS_STREAM_PRIORITY", None)
    t0 = time.perf_counter(); start.set(); seq.join(); dt = time.perf_counter() - t0
    stop.set()
    for t in ths: t.join()
    if seq.error: raise seq.error
    return dt / F * 1e3


DEPTH = int(os.environ.get("HOG_DEPTH", "4"))
SLEEPWAIT = float(os.environ.get("HOG_SLEEPWAIT_MS", "0"))
print("launches enqueued ahead per stream: %d" % DEPTH)
base = run("none", 0)
print("front end alone: %.3f ms per frame" % base, flush=True)
for n in [int(x) for x in os.environ.get("HOG_N", "1,2,4,7").split(",")]:
    a = run("team", n); b = run("hog", n)
    d = run("silent", n)
    print("beside %d streams (%3d of 256 CUs taken): real BA teams %.3f ms per frame (x %.2f)   hogs polling the clock %.3f (x %.2f)   hogs that only sleep %.3f (x %.2f)" %
          (n, 32 * n, a, a / base, b, b / base, d, d / base), flush=True)
