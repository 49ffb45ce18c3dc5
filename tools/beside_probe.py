"""The front end of ONE sequence (extract + match + ratio test per frame, bench.SequenceRunner without BA) alone and beside a bundle adjuster that another
context keeps busy: how much slower do the front end's small kernels run, and does it depend on what the BA launch does at its team barriers?
   python tools/beside_probe.py <mode> [frames] [lib.so]     mode: alone | team (one C4 window on the library's team of 32, solved over and over)
                                                                 | batch (256 C4 windows, one workgroup each: no team barrier in the launch)
Under `GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/beside_probe.py ...` the kernel trace has the per-kernel durations
(tools/c5_trace_summary.py); without the profiler the script prints the sequence's frame rate and the BA launches' duration."""
import os, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "slam-module_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")): sys.path.insert(0, p)
import numpy as np
import bench, synth, ba_synth
import mi355slam


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    if len(sys.argv) > 3: mi355slam.LIB_PATH = os.path.join(ROOT, sys.argv[3])
    mi355slam.prepare_process(8)
    FD = 40
    g = synth.SequenceSynth(bench.W, bench.H, 2000, 2 * (FD - 1), FD - 1)
    frames = np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(FD)]))
    start, stop = threading.Event(), threading.Event()
    seq = bench.SequenceRunner(0, 0, frames, [], 5, start, n_total=F)
    ba_ms = []

    def ba_loop():
        ctx = mi355slam.Context(0)
        nb = 1 if mode == "team" else 256
        probs = [ba_synth.make_problem_fast(50, 2000, 10, seed=42 + i) for i in range(nb)]
        ba = mi355slam.BundleAdjuster(ctx, probs, max_iters=10)
        ba.solve(); ctx.sync()
        start.wait()
        while not stop.is_set():
            ctx.event_mark(0); ba.solve(); ctx.event_mark(1); ba_ms.append(ctx.event_elapsed_ms(0, 1))
        ba.close(); ctx.close()
    th = None
    if mode != "alone":
        th = threading.Thread(target=ba_loop); th.start()
    seq.start(); seq.ready.wait()
    time.sleep(0.5 if mode != "alone" else 0.0)          # (the BA thread has built its handle by now)
    t0 = time.perf_counter(); start.set(); seq.join(); dt = time.perf_counter() - t0
    stop.set()
    if th: th.join()
    if seq.error: raise seq.error
    print("front end %s: %d frames in %.3f s = %.0f frames/s (%.3f ms per frame)%s" % (
        mode, F, dt, F / dt, dt / F * 1e3, "; %d BA launches beside it, %.3f ms each" % (len(ba_ms), float(np.mean(ba_ms))) if ba_ms else ""), flush=True)


if __name__ == "__main__":
    main()
