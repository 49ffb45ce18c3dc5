"""C5 on one GPU outside bench.py: N sequences (host thread + context each, bench.py's SequenceRunner) for F frames, a new-window BA every 5th frame.
   python tools/c5_probe.py [n_seq] [frames] [ba_team] [no_ba]
Under `GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace -d DIR -- python3 tools/c5_probe.py ...` the kernel trace shows how the sequences' kernels overlap (tools/c5_trace_summary.py);
the variable has to be in the environment there: the profiler's preloaded tool initialises the GPU before ms_prepare_process can set it."""
import os, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "slam-module_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bench, synth, ba_synth


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    team = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    no_ba = len(sys.argv) > 4 and sys.argv[4] == "no_ba"
    FD = min(40, F)
    if os.environ.get("C5_PREPARE"):
        import mi355slam
        mi355slam.prepare_process(n_seq)                          # in-process: before the first HIP call
    start = threading.Event()
    runners = []
    for s in range(n_seq):
        g = synth.SequenceSynth(bench.W, bench.H, 2000 + s, 2 * (FD - 1), FD - 1)
        frames = np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(FD)]))
        windows = [] if no_ba else [ba_synth.make_problem_fast(50, 2000, 10, seed=9000 + 16 * s + k) for k in range(4)]
        runners.append(bench.SequenceRunner(0, s, frames, windows, 5, start, n_total=F, ba_team=team))
    for r in runners: r.start()
    for r in runners: r.ready.wait()
    t0 = time.perf_counter(); start.set()
    for r in runners: r.join()
    dt = time.perf_counter() - t0
    for r in runners:
        if r.error: raise r.error
    nf, nb = sum(r.frames_done for r in runners), sum(r.ba_done for r in runners)
    print("%d sequences x %d frames, BA team %d: %.0f frames/s + %.0f BA/s over %.3f s (per sequence %.3f ms per frame)" % (n_seq, F, team, nf / dt, nb / dt, dt, dt / F * 1e3))


if __name__ == "__main__":
    main()
