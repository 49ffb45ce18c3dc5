#!/usr/bin/env python3
"""bench.py -- ORB extract + match throughput on synthetic 720p frames (BASELINE.json configs C2 + C3 policy), with the local-BA
(C4) and independent-sequences (C5) legs beside it.

One "step" = one pass of the hot path over one batch of 256 synthetic 1280x720 frames that are already resident in HBM: pyramid
(8 levels x1.2) -> FAST -> top-2000 selection -> orientation -> 256-bit steered BRIEF, then brute-force Hamming best/second-best +
ratio test of every frame's descriptors against the previous frame's (256 pairs, up to 2000x2000 each).

Multi-GPU: one process per GPU, every rank runs its own batch (independent units, no data-path collective); RCCL is
used for the barriers and the MAX / SUM of (time, units) only.
  * `python bench.py --gpus N` with no RANK in the environment is the LAUNCHER: it starts N child processes of this file (one per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything touches the GPU (torch is not even imported in the parent),
    relays rank 0's JSON line and exits non-zero if any child fails.  Children are started with subprocess, never exec.
  * under `python -m torch.distributed.run ... bench.py --gpus N` (RANK set) the process is a rank.
  * `--plumbing` replaces the GPU work by a sleep and the backend by gloo: the launcher / rendezvous / aggregation path on a CPU box
    (tests/test_bench_launcher.py).  Its line carries "plumbing": true and is not a measurement.

Prints ONE JSON line (contract in the task statement).  `roofline` is computed for the dominant kernel from HIP-event durations
taken inside the timed region on the stream the kernels run on; `cpu_baseline` times the CPU oracle (port of the reference
algorithm, test infrastructure) on a bounded sample of the same workload on rank 0 at N=1, with 1 thread and with all cores.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in ("slam-module_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))

W, H, LEVELS, SCALE, MAX_KPTS, FAST_THR, BATCH = 1280, 720, 8, 1.2, 2000, 20, 256
LOWE_RATIO = 0.75
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
PMC_FILE = os.path.join("profiles", "r04_pmc_traffic.json")      # HBM bytes / instruction counts per launch of the committed kernel sources
N_SEQ = int(os.environ.get("BENCH_N_SEQ", "8"))      # C5: independent sequences of the whole job (8; the variable is for experiments)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--no-greedy", action="store_true", help="skip the M1 / M2 greedy matcher leg")
    ap.add_argument("--only-greedy", action="store_true", help="M1 / M2 greedy matcher leg only (profiling runs): prints its object as the line")
    ap.add_argument("--no-extra", action="store_true", help="skip the sparse-corner input and the PCIe-inclusive legs")
    ap.add_argument("--only-headline", action="store_true", help="C2+C3 leg only (profiling runs)")
    ap.add_argument("--only-ba", action="store_true", help="local-BA leg only (profiling runs): prints its object as the line")
    ap.add_argument("--ba-batch", type=int, default=256)
    ap.add_argument("--ba-steps", type=int, default=3)
    ap.add_argument("--no-ba-two-stage", action="store_true", help="skip the two-stage schedule of the local-BA leg (counter passes: only the single-stage 256-window launches)")
    ap.add_argument("--c5-frames", type=int, default=320, help="frames per sequence in the C5 leg")
    ap.add_argument("--c5-distinct", type=int, default=40, help="distinct images per sequence (walked forwards and backwards)")
    ap.add_argument("--c5-keyframe-every", type=int, default=5)
    ap.add_argument("--c5-team", type=int, default=0, help="workgroups per BA window in the C5 leg (0 = by the number of sequences sharing the GPU)")
    ap.add_argument("--c5-native", action="store_true", help="drive the C5 sequences from C++ threads (tools/c5_native.cpp) instead of Python threads")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="every rank binds device 0 and the ranks' few numbers travel over gloo (RCCL refuses two ranks on one device): the N > 1 code -- launcher, one "
                         "context / slab set per rank, ms_prepare_process per rank, the C5 partition s mod N, the ending -- on a one-GPU lease.  Not a scaling measurement")
    ap.add_argument("--only-c5", action="store_true", help="C5 leg only (after a 2-step headline)")
    ap.add_argument("--plumbing", action="store_true", help="no GPU work: launcher / rendezvous / aggregation only (gloo)")
    ap.add_argument("--plumbing-fail-rank", type=int, default=-1, help="with --plumbing: this rank exits with code 3 (launcher test)")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--launcher-timeout", type=int, default=1500, help="wall-clock limit of the N-rank launcher in seconds: past it every rank is stopped")
    a = ap.parse_args(argv)
    if a.only_headline:
        a.no_ba = a.no_c5 = a.no_extra = a.no_cpu_baseline = a.no_greedy = True
    a.no_ba_leg = a.no_ba
    if a.only_c5:
        a.no_ba_leg = a.no_extra = a.no_cpu_baseline = a.no_greedy = True      # (the C5 leg keeps its own new-window BA every 5th frame)
        a.steps, a.warmup = min(a.steps, 2), min(a.warmup, 1)
    return a


# ------------------------------------------------------------------------------------------------------------------ launcher
def _child_setup():
    """In the child, before exec of the interpreter: die with the launcher (PR_SET_PDEATHSIG), whatever kills it."""
    import ctypes
    import signal
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM, 0, 0, 0)      # PR_SET_PDEATHSIG = 1
    except Exception:
        pass


def launch(args, argv):
    """Parent of an N-rank run: no GPU call, no torch import.  One child per GPU, each in a session of its own (so the whole rank, helper threads and
    all, can be signalled as a group) and bound to the launcher's life (PDEATHSIG); rank 0's stdout carries the JSON line, every rank's stderr is kept
    and its tail shown when the run fails.  SIGTERM / SIGINT to the launcher and the wall-clock deadline end every rank."""
    import signal
    import socket
    import tempfile
    port = args.master_port
    if not port:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        errs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True, preexec_fn=_child_setup,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)                      # the rank's own session = its process group
                except (ProcessLookupError, PermissionError):
                    pass

    def reap(grace):
        stop_all(signal.SIGTERM)
        t_end = time.time() + grace
        for p in procs:
            try:
                p.wait(max(t_end - time.time(), 0.1))
            except subprocess.TimeoutExpired:
                pass
        stop_all(signal.SIGKILL)
        for p in procs:
            p.wait()
    stopped = []

    def on_signal(signum, _frame):
        stopped.append(signum)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)      # drain rank 0's pipe while the ranks run
    reader.start()
    deadline = time.time() + args.launcher_timeout
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):            # one rank failed: the others would wait for it in a barrier forever
            why = "a rank failed"
        elif stopped:
            why = "launcher received signal %d" % stopped[0]
        elif time.time() > deadline:
            why = "launcher deadline of %d s passed" % args.launcher_timeout
        if why:
            reap(20)
            break
        time.sleep(0.05)
    for sg, h in old.items():
        signal.signal(sg, h)
    codes = [p.wait() for p in procs]
    reader.join(5)
    out0 = (chunks[0] if chunks else b"").decode()
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if why or any(c != 0 for c in codes) or line is None:
        sys.stderr.write("bench.py launcher: %srank exit codes %s%s\n" % ((why + "; ") if why else "", codes, "" if line else "; rank 0 printed no result line"))
        for r, f in enumerate(errs):
            if codes[r] != 0 or line is None:
                f.seek(0)
                tail = f.read().decode(errors="replace").splitlines()[-15:]
                if tail:
                    sys.stderr.write("---- rank %d stderr (last lines) ----\n%s\n" % (r, "\n".join(tail)))
        if line:
            sys.stderr.write("rank 0 said: %s\n" % line)
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mso                                   # the checker: imported only here, for the cpu_baseline leg
    mso.build()
    return mso


def host_cores():
    """Host threads this process may really use: CPU affinity, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(n, 1)


def _pool_rate(fn, items, threads):
    """items/s of fn over items with `threads` host threads (ctypes releases the GIL inside the oracle's C code)."""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    if threads == 1:
        for it in items:
            fn(it)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(fn, items))
    return len(items) / (time.perf_counter() - t0)


def cpu_baseline_frames(frames):
    mso = _oracle()
    cfg = mso.cfg(levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR)
    cores = host_cores()

    RUN = 8                                      # a task = a run of 8 consecutive frames: extract each, match it against the previous one

    def one(f0):
        prev = None
        for f in range(f0, f0 + RUN):
            kp = mso.orb_extract(cfg, frames[f])
            mso.hamming_best2(kp["desc"], (prev if prev is not None else kp)["desc"])
            prev = kp
    n1 = 96
    r1 = _pool_rate(one, list(range(0, n1, RUN)), 1) * RUN
    nall = min(len(frames), max(cores * 2 * RUN, 64)) // RUN * RUN
    rall = _pool_rate(one, list(range(0, nall, RUN)), cores) * RUN
    return {"value": round(r1, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d synthetic 720p frames (extract + 2000x2000 Hamming best2 per frame), oracle/libmso.so (gcc -O2, scalar), 1 thread" % (n1, BATCH),
            "all_cores": {"value": round(rall, 2), "unit": "frames/s", "cores": cores,
                          "sample": "%d frames in runs of 8 over %d host threads (the oracle's C code runs outside the GIL)" % (nall, cores)}}


def cpu_baseline_ba(problems, pose_problems=None):
    mso = _oracle()
    cores = host_cores()
    n1 = 32
    r1 = _pool_rate(lambda p: mso.ba_solve(p, 10, False), [problems[i % len(problems)] for i in range(n1)], 1)
    nall = max(cores * 4, 32)
    rall = _pool_rate(lambda p: mso.ba_solve(p, 10, False), [problems[i % len(problems)] for i in range(nall)], cores)
    return {"value": round(r1, 2), "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": "%d solves of distinct C4 windows, oracle/libmso.so (Schur + dense Cholesky, fp64), 1 thread" % n1,
            "all_cores": {"value": round(rall, 2), "unit": "solves/s", "cores": cores, "sample": "%d solves over %d host threads" % (nall, cores)},
            "pose_only_ms_per_solve_1_thread": round(1e3 / _pool_rate(lambda p: mso.ba_solve(p, 10, False), pose_problems, 1), 4) if pose_problems else None}


# ------------------------------------------------------------------------------------------------------------------ rank
class Rank:
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        import torch                                   # plumbing: device memory for the inputs + torch.distributed
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.deferred = []                             # rank 0: (object, fn) -- CPU baselines, run after every GPU leg and after the process group is gone
        self.share = bool(getattr(args, "ranks_share_gpu", False))
        self.gpu = 0 if self.share else self.local_rank       # the device this rank's context, tensors and streams live on
        self.device = "cpu" if (args.plumbing or self.share) else "cuda"      # where the few numbers the ranks exchange live (gloo: host tensors)
        if not args.plumbing:
            # before the first HIP call of this rank (and AFTER `import torch`, whose own copy of the HIP runtime has to be the one the process loads): one
            # hardware queue per sequence that will share the GPU (C5: up to 8 contexts on one device; with the runtime's default of 4 queues a sequence's 10 us
            # front-end kernels wait behind another sequence's 1.8 ms BA launch -- 4 300 against 6 100 frames/s)
            import mi355slam
            # (round 4: ms_prepare_process asks for TWO queues per context: the sequences of THIS rank, 16 queues for the 8 of one rank,
            #  plus the rank's own: 18 queues for the 8 sequences of one rank, 4 - 6 at N = 8 or for six ranks on one GPU; the pipelined leg's 2 x 8 contexts run on the same 18)
            n_mine = len([q for q in range(N_SEQ) if q % max(self.world, 1) == self.rank])
            self.queues_prepared = mi355slam.prepare_process(n_mine + 1)       # (+ the rank's own context)  False: the runtime was up already (a profiler's preloaded tool): the queues are what the environment said then
            self.hw_queues = mi355slam.hw_queues()
            torch.cuda.set_device(self.gpu)
        self.backend = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import datetime
            if args.plumbing or self.share:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=datetime.timedelta(minutes=3))
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank),
                                        timeout=datetime.timedelta(minutes=3))
            self.backend = dist.get_backend()
            self.group_world = dist.get_world_size()          # what the collective library itself says (RCCL on the real N-GPU run)

    def barrier(self):
        if not self.args.plumbing:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def aggregate(self, units, seconds):
        from mi355slam import shard
        return shard.aggregate(self.dist if self.world > 1 else None, self.torch, units, seconds, device=self.device)

    def gather(self, value):
        """Every rank's float, in rank order (reported per GPU; a few bytes)."""
        if self.world == 1:
            return [float(value)]
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        outs = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [float(o.item()) for o in outs]

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()


def base_line(R, args, value, dt):
    line = {"metric": "frames/sec ORB extract+match (720p)", "value": round(value, 1), "unit": "frames/s", "n_gpus": R.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic"}
    if R.world > 1:
        line["collective_backend"] = getattr(R, "backend", None)             # "nccl" = RCCL over xGMI on the N-GPU run
        line["rccl_world"] = getattr(R, "group_world", None) if getattr(R, "backend", None) == "nccl" else None
        line["collective_world"] = getattr(R, "group_world", None)
    if getattr(R, "share", False):
        line["ranks_share_gpu"] = True
        line["note"] = ("--ranks-share-gpu: %d ranks (processes) on ONE device, their totals over gloo -- a run of the N > 1 code on a one-GPU lease, not a scaling "
                        "measurement: `value` is the sum of ranks that share the same CUs") % R.world
    return line


def run_plumbing(R, args):
    """The N-rank skeleton without a GPU: same barriers, same aggregation, a sleep for the hot path."""
    if args.plumbing_fail_rank == R.rank:
        sys.exit(3)
    for _ in range(args.warmup):
        time.sleep(0.001)
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + R.rank))            # uneven ranks: the reported time must be the slowest rank's
    R.barrier()
    dt = time.perf_counter() - t0
    units, dt = R.aggregate(BATCH * args.steps, dt)
    from mi355slam import shard
    out = base_line(R, args, units / dt, dt)
    out.update({"plumbing": True, "data": "none (plumbing run: no GPU work)", "config": {"workload": "launcher / rendezvous / aggregation only"},
                "units_total": units, "per_gpu_units": R.gather(BATCH * args.steps),
                "c5": {"sequences_of_rank": [[s for s in range(N_SEQ) if shard.sequence_of(s, R.world) == r] for r in range(R.world)]}})
    # the same ending as a GPU run: the ranks meet, the process group goes down, rank 0 alone does the (here: pretended) CPU baseline and prints
    def fake_baseline():
        time.sleep(0.3)                                   # the other ranks are gone by now; nothing may wait for them
        return {"value": 0.0, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "none (plumbing run)"}
    finish(R, args, out, [(out, fake_baseline)])


class Headline:
    """C2 + C3: the 256-frame batch step and its per-kernel HIP-event times."""

    def __init__(self, R, ctx, frames_np):
        import mi355slam
        torch = R.torch
        self.R, self.ctx, self.ms = R, ctx, mi355slam
        self.h, self.w = frames_np.shape[1:]
        self.frames = torch.from_numpy(frames_np).cuda()        # inputs resident in HBM before any timed region
        self.ex = mi355slam.OrbExtractor(ctx, self.w, self.h, levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR, max_batch=BATCH)
        cap = self.cap = self.ex.capacity
        self.view = None
        self.pair_q = torch.arange(BATCH, dtype=torch.int32, device="cuda")
        self.pair_t = torch.roll(self.pair_q, 1)                # frame f against frame f-1 (frame 0 against the last one)
        self.best_idx = torch.empty(BATCH * cap, dtype=torch.int32, device="cuda")
        self.best_dist = torch.empty(BATCH * cap, dtype=torch.int16, device="cuda")
        self.second_dist = torch.empty(BATCH * cap, dtype=torch.int16, device="cuda")
        self.match = torch.empty(BATCH * cap, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()

    def set_frames(self, frames_np):
        self.frames.copy_(self.R.torch.from_numpy(frames_np))
        self.R.torch.cuda.synchronize()

    def search(self):
        v, cap = self.view, self.cap
        self.ms.hamming_best2_sets(self.ctx, v.desc, cap, v.count, v.desc, cap, v.count, self.pair_q.data_ptr(), self.pair_t.data_ptr(),
                                   BATCH, self.best_idx.data_ptr(), self.best_dist.data_ptr(), self.second_dist.data_ptr())

    def step(self, images=None, mark=None):
        if images is None:
            self.ex.extract(self.frames.data_ptr(), n_frames=BATCH, frame_stride=self.w * self.h, row_stride=self.w)
        else:
            self.ex.extract(images)                             # host frames: the H2D copies are part of the call
        if self.view is None:
            self.view = self.ex.device_view()
        if mark is not None:
            self.ctx.event_mark(mark)
        self.search()
        if mark is not None:
            self.ctx.event_mark(mark + 1)
        self.ms.ratio_test_device(self.ctx, self.best_idx.data_ptr(), self.best_dist.data_ptr(), self.second_dist.data_ptr(), BATCH * self.cap,
                                  LOWE_RATIO, 50, self.match.data_ptr())

    def timed(self, steps, warmup):
        """warmup untimed steps, then exactly `steps` steps between barrier + synchronize; returns (seconds, per-kernel ms).  The steps are enqueued back to
        back -- no host wait inside the timed region: the extractor records every step's stage events into a ring of its own and the search is bracketed by
        the context's event slots, all of them read after the final synchronize (up to 120 steps are read; beyond that the last 120)."""
        R, ctx = self.R, self.ctx
        for _ in range(warmup):
            self.step()
        ctx.sync()
        R.barrier()
        self.ex.set_profiling(True)
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(mark=16 + 2 * (i % 500))
        ctx.sync()
        R.barrier()
        dt = time.perf_counter() - t0
        self.ex.set_profiling(False)
        read = min(steps, 120)
        stage_sum, match_ms = {}, 0.0
        for back in range(read):
            for k, v in self.ex.stage_ms_back(back).items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
            i = steps - 1 - back
            match_ms += ctx.event_elapsed_ms(16 + 2 * (i % 500), 17 + 2 * (i % 500))
        avg = {k: v / read for k, v in stage_sum.items()}
        avg["hamming"] = match_ms / read
        return dt, avg

    def close(self):
        self.ex.close()
        del self.frames, self.best_idx, self.best_dist, self.second_dist, self.match

    def counts(self):
        import numpy as np
        n_kp = np.frombuffer(ctx_download(self.ctx, self.view.count, 4 * BATCH), dtype=np.int32)
        n_match = int((self.match.view(BATCH, self.cap) >= 0).sum().item())
        return n_kp, n_match


def kernel_table(avg_ms, n_kp, w=W, h=H):
    """Per kernel: ms per launch and algorithmic GB/s (bytes per frame from SURVEY 8d / DESIGN.md 5, x 256 frames per launch)."""
    import numpy as np
    import mi355slam
    ws, hs = mi355slam.level_sizes(LEVELS, SCALE, w, h)
    P = int((ws.astype(np.int64) * hs).sum()); N0 = w * h; K = float(n_kp.mean())
    alg = {"resize": (P - int(ws[-1]) * int(hs[-1])) + (P - N0),      # read levels 0..n-2, write levels 1..n-1
           "blur": 0,                                            # the blurred pyramid is not materialised any more (k_describe blurs its own patches); the stage slot stays
           "fast": P, "select": 4 * K, "tracks": 0, "describe": 1821 * K, "hamming": 32 * 2 * K + 8 * K}
    kernels = {k: {"ms_per_launch": round(avg_ms[k], 4), "alg_GBs": round(alg[k] * BATCH / (avg_ms[k] * 1e-3) / 1e9, 1) if avg_ms[k] > 0 and alg[k] > 0 else None}
               for k in avg_ms}
    return kernels, alg, P, K


def pmc_of(kernel):
    """HBM bytes and instruction counts per launch from the committed PMC passes (tools/profile_set.sh), or None: absent, or taken on OTHER kernel
    sources than the ones in this tree (the summary carries the SHA-256 of slam-module_amd/csrc + include/; tools/build_id.py)."""
    try:
        import build_id
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
        k = d["kernels"][kernel]
        sha = k.get("src_sha256") or d.get("src_sha256")
        if sha != build_id.source_hash():
            return {"stale": True}
        return k
    except Exception:
        return None


def run_gpu(R, args):
    import numpy as np
    import mi355slam                                # after torch: both must share one HIP runtime
    import synth
    torch = R.torch
    ctx = mi355slam.Context(R.gpu)
    if args.only_ba:
        args.no_cpu_baseline = True
        res = bench_ba(R, ctx, args)
        ctx.close()
        finish(R, args, res)
        return
    if args.only_greedy:
        res = bench_greedy(R, ctx, args)
        ctx.close()
        finish(R, args, res)
        return
    frames_np = synth.synth_sequences(BATCH, W, H, 1000 + N_SEQ * R.rank)
    hl = Headline(R, ctx, frames_np)

    # ---- headline: exactly K steps between barrier + synchronize, MAX over ranks ----
    dt, avg_ms = hl.timed(args.steps, args.warmup)
    frames_total, dt = R.aggregate(BATCH * args.steps, dt)
    value = frames_total / dt
    n_kp, n_match = hl.counts()
    kernels, alg, P, K = kernel_table(avg_ms, n_kp)

    # outside the timed region: the same 256 searches on the popcount kernel (v_xor / v_bcnt), for comparison with the matrix-core one
    ctx.set_hamming_path(1)
    for rep in range(3):
        if rep == 1:
            ctx.event_mark(2)
        hl.search()
    ctx.event_mark(3)
    kernels["hamming"]["popcount_kernel_ms"] = round(ctx.event_elapsed_ms(2, 3) / 2, 4)
    ctx.set_hamming_path(0)
    # the unmasked search is an i8 matrix product (1 multiply-add per descriptor bit and pair): its own roofline is the dense i8 MFMA peak
    pair_ops = 2.0 * 256 * float((n_kp.astype(np.float64) * np.roll(n_kp, 1).astype(np.float64)).sum())
    kernels["hamming"]["mfma"] = {"bound": "mfma", "achieved": round(pair_ops / (avg_ms["hamming"] * 1e-3) / 1e12, 1), "peak": 5000.0, "unit": "Top/s (i8)",
                                  "frac": round(pair_ops / (avg_ms["hamming"] * 1e-3) / 1e12 / 5000.0, 4)}
    # ---- roofline of the dominant kernel: algorithmic bytes per launch / HIP-event duration measured above ----
    dom = max(avg_ms, key=avg_ms.get)
    achieved = alg[dom] * BATCH / (avg_ms[dom] * 1e-3) / 1e9
    pmc = pmc_of({"hamming": "k_hamming_mfma"}.get(dom, "k_" + dom))
    stale = bool(pmc and pmc.get("stale"))
    traffic = pmc.get("hbm_bytes_per_step") if pmc and not stale else None
    valu = pmc.get("SQ_INSTS_VALU_per_step") if pmc and not stale else None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": ((PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of these kernel sources, tools/profile_set.sh; not measured in this run)") if not stale
                                   else (PMC_FILE + " was taken on other kernel sources than this tree's (src_sha256 differs): traffic withheld, re-run tools/profile_set.sh")) if pmc else None,
                # the front-end kernels are bound by VALU issue (VOP3 / packed forms issue once per ~4 cycles per SIMD, profiles/r01_e_valu_issue_rates.txt)
                "valu_insts": valu, "valu_issue_frac": round(valu / 1024 * 4 / 2.4e9 / (avg_ms[dom] * 1e-3), 3) if valu else None,
                # bytes the step has in its contract NOW: the blurred pyramid is no longer written or read (k_describe blurs its own patches), so
                # input N0 + levels written (P - N0) + levels read for detection P = 2P, + 1821 K (patches, outputs) + 72 K (match pair);
                # the SURVEY 8d contract figure (4P + ..., with the blur round trip) is kept beside it
                "whole_step_alg_GBs": round((2 * P + 1821 * K + 72 * K) * BATCH * args.steps / dt / 1e9, 1),
                "contract_alg_GBs": round((4 * P + 1821 * K + 72 * K) * BATCH * args.steps / dt / 1e9, 1)}
    out = base_line(R, args, value, dt)
    out["config"] = {"workload": "C2+C3: ORB extract 1280x720, 8 levels x1.2, 2000 kpts/frame, FAST thr 20, batch 256 synthetic frames/GPU, "
                                 "+ Hamming brute-force best2 + ratio 0.75 of each frame vs the previous (256 pairs, <=2000x2000)",
                     "batch_per_gpu": BATCH, "keypoints_per_frame": round(K, 1), "ratio_matches_per_frame": round(n_match / BATCH, 1),
                     "corner_density": "7-9 % of pixels (tools/synth.py defaults)"}
    out["per_gpu_frames_per_s"] = [round(v, 1) for v in R.gather(BATCH * args.steps / dt)]
    out["roofline"] = roofline
    out["kernels"] = kernels

    # The legs below are secondary.  Should one of them raise, the headline above must still reach the driver: the error goes into the
    # line and the remaining legs are skipped (they share collectives; every rank runs the same code on the same shapes, so a
    # failure is the same on all ranks; an asymmetric one ends at the process group's 3-minute timeout instead of hanging).
    failed = []

    def leg(name, fn):
        if failed:
            return
        try:
            out[name] = fn()
        except Exception as e:                                      # noqa: BLE001 -- reported in the line
            import traceback
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            failed.append(name)
            traceback.print_exc(file=sys.stderr)

    if not args.no_extra:
        def sparse_leg():
            # the same step on frames with the corner density of camera images (1-2 % instead of 7-9 %)
            sparse_np = synth.synth_sequences(BATCH, W, H, 1000 + N_SEQ * R.rank, sparse=True)
            hl.set_frames(sparse_np)
            sdt, savg = hl.timed(max(args.steps // 2, 1), 1)
            s_total, sdt = R.aggregate(BATCH * max(args.steps // 2, 1), sdt)
            skp, smatch = hl.counts()
            return {"value": round(s_total / sdt, 1), "unit": "frames/s", "corner_density": "~2 % of pixels (tools/synth.py sparse=True)",
                    "keypoints_per_frame": round(float(skp.mean()), 1), "ratio_matches_per_frame": round(smatch / BATCH, 1),
                    "ms_per_launch": {k: round(v, 4) for k, v in savg.items()}}

        def pcie_leg():
            # PCIe-inclusive: frames start in pinned HOST memory, keypoints + descriptors end in host memory
            hl.set_frames(frames_np)
            return pcie_inclusive(R, hl, frames_np, max(min(args.steps // 4, 12), 2))
        def vga_leg():
            # north_star: "synthetic VGA/720p frames" -- the same step on 256 frames of 640x480 (same quotas, 2000 keypoints)
            vga_np = synth.synth_sequences(BATCH, 640, 480, 1000 + N_SEQ * R.rank)
            hv = Headline(R, ctx, vga_np)
            vdt, vavg = hv.timed(max(args.steps // 2, 1), 2)
            v_total, vdt_max = R.aggregate(BATCH * max(args.steps // 2, 1), vdt)
            vkp, vmatch = hv.counts()
            vk, valg, vP, vK = kernel_table(vavg, vkp, 640, 480)
            hv.close()
            return {"value": round(v_total / vdt_max, 1), "unit": "frames/s", "ms_per_step": round(vdt / max(args.steps // 2, 1) * 1e3, 3),
                    "workload": "the C2+C3 step on 256 synthetic 640x480 frames per GPU (P = %d pyramid pixels per frame)" % vP,
                    "keypoints_per_frame": round(vK, 1), "ratio_matches_per_frame": round(vmatch / BATCH, 1),
                    "whole_step_alg_GBs": round((2 * vP + 1821 * vK + 72 * vK) * BATCH * max(args.steps // 2, 1) / vdt / 1e9, 1),
                    "contract_alg_GBs": round((4 * vP + 1821 * vK + 72 * vK) * BATCH * max(args.steps // 2, 1) / vdt / 1e9, 1), "kernels": vk}
        leg("sparse_input", sparse_leg)
        leg("vga", vga_leg)
        leg("value_pcie_inclusive", pcie_leg)
        hl.close()
        leg("c3", lambda: bench_c3(R, ctx, args))
    # ---- secondary metric: local-BA solves/s (BASELINE config C4), 256 distinct windows per launch, device-resident ----
    if not args.no_ba_leg:
        leg("local_ba", lambda: bench_ba(R, ctx, args))
    # ---- the reference's own matchers M1 / M2 on a new keyframe's pairs ----
    if not args.no_greedy:
        leg("greedy_match", lambda: bench_greedy(R, ctx, args))
    # ---- C5: 8 independent sequences, sequence s on GPU s mod N, frame by frame ----
    if not args.no_c5:
        leg("pipelined_sequence", lambda: bench_pipelined(R, args))
        leg("c5", lambda: bench_c5(R, args))
    if failed:
        out["failed_legs"] = failed
    ctx.close()
    finish(R, args, out, [(out, lambda: cpu_baseline_frames(frames_np))] if not args.no_cpu_baseline else [])


def finish(R, args, out, first=()):
    """Every GPU leg is done: the ranks meet once more and take the process group down; then rank 0 -- alone, the other ranks have nothing left to do --
    times the CPU baselines (at every N, on this rank's host cores) and prints the line."""
    R.barrier()
    R.close()
    if R.rank != 0:
        return
    for obj, fn in list(first) + R.deferred:
        try:
            obj["cpu_baseline"] = fn()
        except Exception as e:                                      # noqa: BLE001 -- reported in the line
            obj["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    print(json.dumps(out), flush=True)


def pcie_inclusive(R, hl, frames_np, steps):
    """frames/s with the H2D copy of the 256 input frames and the D2H copy of every frame's keypoints, descriptors and matches inside the timed region,
    from ONE context: ms_orb_extract takes the pinned host batch in four pieces on its copy stream (piece k+1 copies under piece k's kernels),
    ms_dev_download_async takes the outputs out on a third stream while the next step is already enqueued."""
    import ctypes as C
    import numpy as np
    torch, ctx = R.torch, hl.ctx
    import mi355slam
    L = mi355slam.lib()
    host = torch.from_numpy(frames_np).pin_memory()
    host_np = host.numpy()
    cap = hl.cap
    hl.step(images=host_np); ctx.sync()
    v = hl.view
    sizes = {"count": 4 * BATCH, "x": 4 * BATCH * cap, "y": 4 * BATCH * cap, "angle": 4 * BATCH * cap, "octave": 4 * BATCH * cap, "desc": 32 * BATCH * cap,
             "match": 4 * BATCH * cap}
    outs = {k: torch.empty(n, dtype=torch.uint8).pin_memory() for k, n in sizes.items()}
    ptr = {"count": v.count, "x": v.x, "y": v.y, "angle": v.angle, "octave": v.octave, "desc": v.desc, "match": hl.match.data_ptr()}

    def d2h_async():
        for k, n in sizes.items():
            ctx.check(L.ms_dev_download_async(ctx._h, C.c_void_p(outs[k].data_ptr()), C.c_void_p(ptr[k]), C.c_size_t(n)), "ms_dev_download_async")

    def d2h_blocking():
        for k, n in sizes.items():
            ctx.check(L.ms_dev_download(ctx._h, C.c_void_p(outs[k].data_ptr()), C.c_void_p(ptr[k]), C.c_size_t(n)), "ms_dev_download")
    bytes_step = frames_np.nbytes + sum(sizes.values())
    res = {}
    for name, d2h, note in (("pipelined", d2h_async, "steps enqueued back to back: the outputs of step k leave (ms_dev_download_async) while step k + 1's frames come in; one host wait at the end"),
                            ("step_by_step", d2h_blocking, "every step waits for its own outputs on the host before the next one starts (blocking ms_dev_download per array)")):
        hl.step(images=host_np); d2h(); ctx.sync()
        R.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            hl.step(images=host_np)
            d2h()
        ctx.sync()
        R.barrier()
        dt = time.perf_counter() - t0
        total, dt = R.aggregate(BATCH * steps, dt)
        res[name] = {"value": round(total / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
                     "host_GBs": round(bytes_step * steps * R.world / dt / 1e9, 1), "note": note}
    n_kp = outs["count"].numpy().view(np.int32)
    out = dict(res["pipelined"])
    out.update({"host_bytes_per_step": int(bytes_step), "keypoints_per_frame": round(float(n_kp.mean()), 1), "step_by_step": res["step_by_step"],
                "note": "ONE context: 256 pinned host frames -> device in four pieces on the extractor's copy stream (one 2-D copy per piece), each piece's kernels under the next piece's "
                        "copy; all SoA outputs + matches -> pinned host on a third stream.  " + res["pipelined"]["note"]})
    out["overlapped"] = pcie_overlapped(R, host, cap, steps)
    return out


class PcieWorker(threading.Thread):
    """Half of the batch on a context (stream) of its own: host frames in, extract + match + ratio test, all outputs back to the host.  Two of
    them side by side let the copies of one stream run under the kernels of the other (the design rule: copies overlap compute on
    separate HIP streams); nothing else changes -- same library calls, same bytes."""

    def __init__(self, device, host_frames, steps, start_evt):
        super().__init__()
        self.device, self.host, self.steps, self.start_evt = device, host_frames, steps, start_evt
        self.ready = threading.Event()
        self.error = None

    def run(self):
        try:
            import ctypes as C
            import numpy as np
            import torch
            import mi355slam
            n = self.host.shape[0]
            ctx = mi355slam.Context(self.device)
            ex = mi355slam.OrbExtractor(ctx, W, H, levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR, max_batch=n)
            cap = ex.capacity
            pq = ctx.upload(np.arange(n, dtype=np.int32)); pt = ctx.upload(np.roll(np.arange(n, dtype=np.int32), 1))
            bi, bd, sd, match = ctx.alloc(4 * n * cap), ctx.alloc(2 * n * cap), ctx.alloc(2 * n * cap), ctx.alloc(4 * n * cap)
            frames = self.host.numpy()
            ex.extract(frames); ctx.sync()
            v = ex.device_view()
            sizes = {"count": 4 * n, "x": 4 * n * cap, "y": 4 * n * cap, "angle": 4 * n * cap, "octave": 4 * n * cap, "desc": 32 * n * cap, "match": 4 * n * cap}
            ptr = {"count": v.count, "x": v.x, "y": v.y, "angle": v.angle, "octave": v.octave, "desc": v.desc, "match": match.ptr}
            outs = {k: torch.empty(b, dtype=torch.uint8).pin_memory() for k, b in sizes.items()}

            def step():
                ex.extract(frames)
                mi355slam.hamming_best2_sets(ctx, v.desc, cap, v.count, v.desc, cap, v.count, pq, pt, n, bi, bd, sd)
                mi355slam.ratio_test_device(ctx, bi, bd, sd, n * cap, LOWE_RATIO, 50, match)
                for k, b in sizes.items():
                    ctx.check(mi355slam.lib().ms_dev_download(ctx._h, C.c_void_p(outs[k].data_ptr()), C.c_void_p(ptr[k]), C.c_size_t(b)), "ms_dev_download")
            step()
            self.ready.set()
            self.start_evt.wait()
            for _ in range(self.steps):
                step()
            ctx.sync()
            self.bytes_step = self.host.numel() + sum(sizes.values())
            ctx.close()
        except Exception as e:                                   # noqa: BLE001 -- reported by the parent
            self.error = e
            self.ready.set()


def pcie_overlapped(R, host, cap, steps):
    """The same PCIe-inclusive step as two half-batches on two contexts (two host threads, two streams)."""
    start = threading.Event()
    half = BATCH // 2
    workers = [PcieWorker(R.gpu, host[i * half:(i + 1) * half], steps, start) for i in range(2)]
    for wk in workers:
        wk.start()
    for wk in workers:
        wk.ready.wait()
    R.barrier()
    t0 = time.perf_counter()
    start.set()
    for wk in workers:
        wk.join()
    R.barrier()
    dt = time.perf_counter() - t0
    errs = [wk.error for wk in workers if wk.error]
    if errs:
        raise errs[0]
    total, dt = R.aggregate(BATCH * steps, dt)
    return {"value": round(total / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "host_GBs": round(sum(wk.bytes_step for wk in workers) * steps * R.world / dt / 1e9, 1),
            "note": "two half-batches of 128 frames on two contexts (streams) driven by two host threads: one stream's copies run under the other's kernels"}


def bench_ba(R, ctx, args):
    """C4: 50 keyframes x 2000 points x 20000 observations, 10 LM iterations, Huber sqrt(5.991), 49 odometry edges; 256 DISTINCT windows."""
    import ba_synth
    import mi355slam
    probs = [ba_synth.make_problem_fast(50, 2000, 10, seed=42 + 1000 * R.rank + i) for i in range(args.ba_batch)]
    ba = mi355slam.BundleAdjuster(ctx, probs, max_iters=10)
    ba.solve(); ctx.sync()                                   # warm-up
    R.barrier()
    t0 = time.perf_counter()
    ctx.event_mark(2)
    for _ in range(args.ba_steps):
        ba.solve()
    ctx.event_mark(3)
    ctx.sync()
    R.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ctx.event_elapsed_ms(2, 3) / args.ba_steps
    solves_total, dt = R.aggregate(args.ba_batch * args.ba_steps, dt)
    stats = [ba.download(i)["stats"] for i in range(0, args.ba_batch, max(args.ba_batch // 16, 1))]
    iters = sum(s["iters"] for s in stats) / len(stats)
    trials = sum(s["trials"] for s in stats) / len(stats)
    # latency of ONE window (what a sequential SLAM pipeline sees): a team of workgroups shares the problem (automatic size)
    one = mi355slam.BundleAdjuster(ctx, probs[:1], max_iters=10)
    one.solve(); ctx.sync()
    ctx.event_mark(4); one.solve(); ctx.event_mark(5)
    single_ms = ctx.event_elapsed_ms(4, 5)
    # what a keyframe pays when the window is new: create (host index build + upload) + solve + download + destroy
    n_new = 8
    for i in range(n_new + 2):                                 # (two untimed: the first window of a size takes its device block and the host scratch their final size)
        if i == 2:
            t1 = time.perf_counter()
        b = mi355slam.BundleAdjuster(ctx, [probs[i % len(probs)]], max_iters=10); b.solve(); b.download(0); b.close()
    new_window_ms = (time.perf_counter() - t1) / n_new * 1e3
    two_stage = None if args.no_ba_two_stage else bench_ba_two_stage(R, ctx, args, probs)
    # poseBundleAdjust (bundle_adjuster.cpp:396-491; runs on every non-keyframe, mapper_helpers.cpp:1043-1050): one free keyframe, its map points fixed, the odometry
    # edge to the fixed previous keyframe -- k_ba_pose_only.  One problem per call (what a frame pays) and 256 of them in one launch
    pose_probs = [ba_synth.pose_only_from_window(probs[i % len(probs)], 25) for i in range(256)]
    pb = mi355slam.BundleAdjuster(ctx, pose_probs[:1], max_iters=10)
    pb.solve(); ctx.sync(); ctx.event_mark(4)
    for _ in range(10):
        pb.solve()
    ctx.event_mark(5)
    pose_one_ms = ctx.event_elapsed_ms(4, 5) / 10
    pstats = pb.download(0)["stats"]; pb.close()
    t1 = time.perf_counter()
    for i in range(10):
        b = mi355slam.BundleAdjuster(ctx, [pose_probs[i]], max_iters=10); b.solve(); b.download(0); b.close()
    pose_new_ms = (time.perf_counter() - t1) / 10 * 1e3
    pbb = mi355slam.BundleAdjuster(ctx, pose_probs, max_iters=10)
    pbb.solve(); ctx.sync(); ctx.event_mark(4); pbb.solve(); ctx.event_mark(5)
    pose_batch_ms = ctx.event_elapsed_ms(4, 5); pbb.close()
    pose_only = {"workload": "one free keyframe, %d fixed map points / observations, odometry edge to the fixed previous keyframe, 10 iterations" % len(pose_probs[0]["obs_pose"]),
                 "ms_per_solve": round(pose_one_ms, 4), "new_problem_ms": round(pose_new_ms, 4), "iterations_trials": [pstats["iters"], pstats["trials"]],
                 "batch_of_256_ms": round(pose_batch_ms, 4), "batch_solves_per_s": round(256 / pose_batch_ms * 1e3, 1)}
    alg_bytes_per_launch = 6.61e6 * trials * args.ba_batch    # 6.61 MB per LM iteration (= per damped solve) at C4
    achieved = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    pmc = pmc_of("k_ba_lm")
    res = {"metric": "local-BA solves/sec (50 KF x 2000 pts x 20k obs, 10 LM iters)", "value": round(solves_total / dt, 1),
           "unit": "solves/s", "windows_per_launch": args.ba_batch, "distinct_windows": len(probs), "ms_per_launch": round(kernel_ms, 3),
           "lm_iterations": round(iters, 2), "lm_trials": round(trials, 2),
           "single_window_ms": round(single_ms, 3), "single_window_solves_per_s": round(1e3 / single_ms, 1),
           "new_window_ms": round(new_window_ms, 3), "dtype": "f64", "two_stage": two_stage, "pose_only": pose_only,
           "roofline": {"kernel": "k_ba_lm", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": pmc.get("hbm_bytes_per_launch") if pmc and not pmc.get("stale") else None,
                        # fp64 issue: VALU wave-instructions of the launch (PMC, ~95 % of them fp64) x the 6.5 cycles at which ONE wave issues dependent-free
                        # v_fma_f64 back to back (tools/valu_rates.hip), over the SIMD-cycles of the launch
                        "fp64_issue_frac": round(pmc["SQ_INSTS_VALU_per_launch"] * (args.ba_batch / 256.0) * 6.5 / (1024 * 2.4e9 * kernel_ms * 1e-3), 3)
                                           if pmc and not pmc.get("stale") and pmc.get("SQ_INSTS_VALU_per_launch") else None,
                        "traffic_calibration": pmc.get("fetch_calibration") if pmc and not pmc.get("stale") else None,
                        "traffic_source": ((PMC_FILE + " (rocprofv3 --pmc passes of these kernel sources over the 256-window launch, tools/pmc_ba.sh)") if not pmc.get("stale")
                                           else (PMC_FILE + " was taken on other kernel sources than this tree's: traffic withheld")) if pmc else None}}
    if R.rank == 0 and not args.no_cpu_baseline:
        R.deferred.append((res, lambda: cpu_baseline_ba(probs[:32], pose_probs[:32])))
    ba.close(); one.close()
    return res


def two_stage_problems(p, cur):
    """The reference's localBundleAdjust schedule on one window (bundle_adjuster.cpp:245-373): stage 1 frees only the current keyframe (+ all points),
    stage 2 frees every keyframe and adds the orientation prior Omega = diag((100 r)^2 I3, 0) against a fixed copy of the stage-1 pose."""
    import numpy as np
    n = len(p["pose"])
    s1 = dict(p); s1["pose_fixed"] = np.ones(n, np.uint8); s1["pose_fixed"][cur] = 0
    s2 = dict(p); s2["pose"] = np.vstack([p["pose"], p["pose"][cur:cur + 1]])
    s2["pose_fixed"] = np.concatenate([np.zeros(n, np.uint8), [1]]).astype(np.uint8)
    Wm = np.zeros((6, 6)); Wm[:3, :3] = np.eye(3) * (100 * 100.0) ** 2
    s2["edge_i"] = np.concatenate([p["edge_i"], [n]]).astype(np.int32); s2["edge_j"] = np.concatenate([p["edge_j"], [cur]]).astype(np.int32)
    s2["edge_meas"] = np.vstack([p["edge_meas"], [[0, 0, 0, 1, 0, 0, 0]]]); s2["edge_info"] = np.vstack([p["edge_info"], Wm.reshape(1, 36)])
    return s1, s2


def bench_ba_two_stage(R, ctx, args, probs):
    """C4 under the reference's two-stage schedule, iterations = int(1 + sqrt(50)) = 8 per stage (bundle_adjuster.cpp:156,322-373): both handles built once,
    stage 1 -> ms_ba_copy_state -> stage 2 chained on the device."""
    import numpy as np
    import mi355slam
    iters = int(1 + np.sqrt(50.0))
    cur = len(probs[0]["pose"]) - 1                           # the newest keyframe of the window
    st = [two_stage_problems(p, cur) for p in probs]
    b1 = mi355slam.BundleAdjuster(ctx, [a for a, _ in st], max_iters=iters)
    b2 = mi355slam.BundleAdjuster(ctx, [b for _, b in st], max_iters=iters)
    extra = np.full(len(probs), cur, np.int32)

    def run():
        b1.solve(); b2.copy_state_from(b1, extra); b2.solve()
    run(); ctx.sync()
    ctx.event_mark(10); b1.solve(); ctx.event_mark(11); b2.copy_state_from(b1, extra); b2.solve(); ctx.event_mark(12)
    s1_ms, s2_ms = ctx.event_elapsed_ms(10, 11), ctx.event_elapsed_ms(11, 12)
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(args.ba_steps):
        run()
    ctx.sync()
    R.barrier()
    dt = time.perf_counter() - t0
    total, dt = R.aggregate(len(probs) * args.ba_steps, dt)
    r1, r2 = b1.download(0)["stats"], b2.download(0)["stats"]
    # one window, the way the mapper pays for it: both creates + stage 1 + copy + stage 2 + one download + destroys
    s1, s2 = st[0]
    o1 = mi355slam.BundleAdjuster(ctx, [s1], max_iters=iters); o2 = mi355slam.BundleAdjuster(ctx, [s2], max_iters=iters)
    o1.solve(); o2.copy_state_from(o1, extra[:1]); o2.solve(); ctx.sync()
    ctx.event_mark(10); o1.solve(); o2.copy_state_from(o1, extra[:1]); o2.solve(); ctx.event_mark(11)
    one_ms = ctx.event_elapsed_ms(10, 11)
    o1.close(); o2.close()
    n_new = 6
    for i in range(n_new + 2):
        if i == 2:
            t1 = time.perf_counter()
        a, b = st[i % len(st)]
        h1 = mi355slam.BundleAdjuster(ctx, [a], max_iters=iters); h1.solve()                  # stage 1 runs while the host builds stage 2's index structures (the order of the host mirror)
        h2 = mi355slam.BundleAdjuster(ctx, [b], max_iters=iters)
        h2.copy_state_from(h1, extra[:1]); h2.solve(); h2.download(0); h1.close(); h2.close()
    new_ms = (time.perf_counter() - t1) / n_new * 1e3
    b1.close(); b2.close()
    return {"schedule": "stage 1: current keyframe + all points free, %d iterations; stage 2: every keyframe free + orientation prior edge, %d iterations (bundle_adjuster.cpp:156,322-373)" % (iters, iters),
            "value": round(total / dt, 1), "unit": "two-stage solves/s", "windows_per_launch": len(probs),
            "stage1_ms_per_launch": round(s1_ms, 3), "stage2_ms_per_launch": round(s2_ms, 3),
            "stage1_iterations_trials": [r1["iters"], r1["trials"]], "stage2_iterations_trials": [r2["iters"], r2["trials"]],
            "chi2": [round(r1["chi2_init"], 1), round(r1["chi2_final"], 1), round(r2["chi2_final"], 1)],
            "single_window_ms": round(one_ms, 3), "new_window_ms": round(new_ms, 3)}


def bench_c3(R, ctx, args, n_pairs=1000, nq=2000, nt=2000, n_inlier=1400, flip=0.08, buckets=100):
    """BASELINE config 3 on its own workload: 1000 pairs of 2000 x 2000 descriptors -- target rows 0..1399 are rows of the query set,
    permuted, each bit flipped with probability 0.08 (mean distance ~20), the other 600 random; policy = M1 core (best / second, best <= 50,
    0.75 second >= best).  Timed: the unmasked search on the matrix cores and on the popcount kernel, the 100-bucket masked variant, the ratio
    test, and the greedy-unique variant (matchForLoopClosures' consumption of targets, keyframe_matcher.cpp:98-100,:128) with one bucket and with
    100.  Descriptors are generated on the device with torch (plumbing); same recipe as tools/synth.py synth_descriptor_pair."""
    import ctypes as C
    import numpy as np
    import mi355slam
    torch = R.torch
    L, vp = mi355slam.lib(), mi355slam._vp
    g = torch.Generator(device="cuda"); g.manual_seed(4242 + R.rank)
    rbits = lambda *shape: torch.randint(-2 ** 31, 2 ** 31, shape, dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    q = rbits(n_pairs, nq, 8)
    t = torch.empty(n_pairs, nt, 8, dtype=torch.int32, device="cuda")
    perm = torch.argsort(torch.rand(n_pairs, nq, device="cuda", generator=g), dim=1)[:, :n_inlier]
    w32 = (torch.ones(32, dtype=torch.int64, device="cuda") << torch.arange(32, device="cuda")).view(1, 1, 1, 32)
    for p0 in range(0, n_pairs, 50):
        sl = slice(p0, min(p0 + 50, n_pairs))
        src = torch.gather(q[sl], 1, perm[sl].unsqueeze(-1).expand(-1, -1, 8))
        fl = ((torch.rand(src.shape[0], n_inlier, 8, 32, device="cuda", generator=g) < flip).to(torch.int64) * w32).sum(-1).to(torch.int32)
        t[sl, :n_inlier] = src ^ fl
    t[:, n_inlier:] = rbits(n_pairs, nt - n_inlier, 8)
    # vocabulary nodes for the bucketed variants: a target that is a noisy copy of a query sits in the query's node (what a vocabulary does most of the time)
    qb = torch.randint(0, buckets, (n_pairs, nq), dtype=torch.int32, device="cuda", generator=g)
    tb = torch.randint(0, buckets, (n_pairs, nt), dtype=torch.int32, device="cuda", generator=g)
    tb[:, :n_inlier] = torch.gather(qb, 1, perm)
    bi = torch.empty(n_pairs * nq, dtype=torch.int32, device="cuda"); bd = torch.empty(n_pairs * nq, dtype=torch.int16, device="cuda")
    sd = torch.empty_like(bd); match = torch.empty_like(bi)
    torch.cuda.synchronize()

    def best2(masked):
        ctx.check(L.ms_hamming_best2(ctx._h, vp(q.data_ptr()), nq, vp(t.data_ptr()), nt, n_pairs, vp(qb.data_ptr()) if masked else None, vp(tb.data_ptr()) if masked else None, None,
                                     vp(bi.data_ptr()), vp(bd.data_ptr()), vp(sd.data_ptr())), "ms_hamming_best2")

    def ratio():
        mi355slam.ratio_test_device(ctx, bi.data_ptr(), bd.data_ptr(), sd.data_ptr(), n_pairs * nq, LOWE_RATIO, 50, match.data_ptr())

    def timed(fn, reps=3):
        fn(); ctx.sync()
        ctx.event_mark(8)
        for _ in range(reps):
            fn()
        ctx.event_mark(9)
        return ctx.event_elapsed_ms(8, 9) / reps
    ops = 2.0 * 256 * nq * nt * n_pairs                       # one multiply-add per descriptor bit and (query, target)
    res = {"workload": "%d pairs x %d x %d descriptors per GPU, %d inliers at bit-flip probability %.2f, Lowe ratio %.2f, max distance 50" % (n_pairs, nq, nt, n_inlier, flip, LOWE_RATIO),
           "alg_bytes_per_pair": 32 * (nq + nt) + 8 * nq, "unit": "pairs/s"}
    ms = timed(lambda: best2(False))
    ratio(); ctx.sync()
    res["ratio_matches_per_pair"] = round(float((match >= 0).sum().item()) / n_pairs, 1)
    res["matrix_core_search"] = {"ms_per_launch": round(ms, 3), "pairs_per_s": round(n_pairs / ms * 1e3, 1), "alg_GBs": round((32 * (nq + nt) + 8 * nq) * n_pairs / ms / 1e6, 1),
                                 "roofline": {"bound": "mfma", "achieved": round(ops / ms / 1e9, 1), "peak": 5000.0, "unit": "Top/s (i8)", "frac": round(ops / ms / 1e9 / 5000.0, 4)}}
    ctx.set_hamming_path(1)
    try:
        ms = timed(lambda: best2(False))
    finally:
        ctx.set_hamming_path(0)
    # v_xor + v_bcnt per descriptor word and pair: 16 lane-operations per pair against the chip's 1024 SIMDs x 16 lanes x 2.4 GHz
    res["popcount_search"] = {"ms_per_launch": round(ms, 3), "pairs_per_s": round(n_pairs / ms * 1e3, 1),
                              "valu_lane_ops_frac": round(16.0 * nq * nt * n_pairs / (ms * 1e-3) / (1024 * 64 * 2.4e9 / 4), 3)}
    ms = timed(lambda: best2(True))
    res["masked_%d_buckets" % buckets] = {"ms_per_launch": round(ms, 3), "pairs_per_s": round(n_pairs / ms * 1e3, 1)}
    res["ratio_test_ms"] = round(timed(ratio), 4)
    # greedy-unique variant: M1 with every keypoint usable, no rotation histogram
    ones = torch.ones(max(nq, nt), dtype=torch.uint8, device="cuda")
    outs = torch.empty(n_pairs, nq, dtype=torch.int32, device="cuda"); nm = torch.empty(n_pairs, dtype=torch.int32, device="cuda")
    ptrs = (C.c_void_p * n_pairs)(*[outs.data_ptr() + 4 * nq * p for p in range(n_pairs)])
    iota = torch.arange(max(nq, nt), dtype=torch.int32, device="cuda")
    node1 = torch.zeros(1, dtype=torch.int32, device="cuda")
    st_q1 = torch.tensor([0, nq], dtype=torch.int32, device="cuda"); st_t1 = torch.tensor([0, nt], dtype=torch.int32, device="cuda")
    nodeB = torch.arange(buckets, dtype=torch.int32, device="cuda")

    def csr(b):                                                # per pair: keypoints of a node in index order (stable sort), node_start by counting
        order = torch.argsort(b, dim=1, stable=True).to(torch.int32).contiguous()
        cnt = torch.zeros(b.shape[0], buckets, dtype=torch.int64, device="cuda").scatter_add_(1, b.to(torch.int64), torch.ones_like(b, dtype=torch.int64))
        start = torch.zeros(b.shape[0], buckets + 1, dtype=torch.int32, device="cuda"); start[:, 1:] = torch.cumsum(cnt, 1).to(torch.int32)
        return order, start.contiguous()
    oq, sq = csr(qb); ot, st = csr(tb)
    torch.cuda.synchronize()

    def frames(bucketed):
        F1, F2 = (mi355slam.MatchFrame * n_pairs)(), (mi355slam.MatchFrame * n_pairs)()
        for p in range(n_pairs):
            b1 = mi355slam.Bow(buckets, nodeB.data_ptr(), sq.data_ptr() + 4 * (buckets + 1) * p, oq.data_ptr() + 4 * nq * p) if bucketed else mi355slam.Bow(1, node1.data_ptr(), st_q1.data_ptr(), iota.data_ptr())
            b2 = mi355slam.Bow(buckets, nodeB.data_ptr(), st.data_ptr() + 4 * (buckets + 1) * p, ot.data_ptr() + 4 * nt * p) if bucketed else mi355slam.Bow(1, node1.data_ptr(), st_t1.data_ptr(), iota.data_ptr())
            F1[p] = mi355slam.MatchFrame(nq, q.data_ptr() + 32 * nq * p, 0, 0, 0, ones.data_ptr(), b1)
            F2[p] = mi355slam.MatchFrame(nt, t.data_ptr() + 32 * nt * p, 0, 0, 0, ones.data_ptr(), b2)
        return F1, F2
    for name, bucketed in (("greedy_unique_single_bucket", False), ("greedy_unique_%d_buckets" % buckets, True)):
        F1, F2 = frames(bucketed)
        ms = timed(lambda: ctx.check(L.ms_match_loop_closure(ctx._h, F1, F2, n_pairs, C.c_float(LOWE_RATIO), 0, ptrs, vp(nm.data_ptr())), "ms_match_loop_closure"), reps=2)
        res[name] = {"ms_per_launch": round(ms, 3), "pairs_per_s": round(n_pairs / ms * 1e3, 1), "matches_per_pair": round(float(nm.sum().item()) / n_pairs, 1)}
        ms1 = timed(lambda: ctx.check(L.ms_match_loop_closure(ctx._h, F1, F2, 1, C.c_float(LOWE_RATIO), 0, ptrs, vp(nm.data_ptr())), "ms_match_loop_closure"), reps=3)
        res[name]["one_pair_ms"] = round(ms1, 4)
    if R.rank == 0 and not args.no_cpu_baseline:
        host = [x[:4].cpu().numpy().view(np.uint32) for x in (q, t)] + [x[:4].cpu().numpy() for x in (qb, tb)]
        R.deferred.append((res, lambda: cpu_baseline_c3(host, buckets)))
    return res


def cpu_baseline_c3(host, buckets):
    """The oracle on 4 of the leg's pairs, 1 thread: the brute-force best / second search, its 100-bucket form, and M1's greedy walk with one bucket and with 100."""
    mso = _oracle()
    import numpy as np
    q, t, qb, tb = host
    n = len(q)
    ones_q, ones_t, zq, zt = np.ones(q.shape[1], np.uint8), np.ones(t.shape[1], np.uint8), np.zeros(q.shape[1], np.float32), np.zeros(t.shape[1], np.float32)
    legs = {"search": lambda i: mso.hamming_best2(q[i], t[i]),
            "masked_%d_buckets" % buckets: lambda i: mso.hamming_best2(q[i], t[i], q_bucket=qb[i], t_bucket=tb[i]),
            "greedy_unique_single_bucket": lambda i: mso.match_loop_closure(q[i], zq, ones_q, np.zeros(q.shape[1], np.int32), t[i], zt, ones_t, np.zeros(t.shape[1], np.int32), LOWE_RATIO, False),
            "greedy_unique_%d_buckets" % buckets: lambda i: mso.match_loop_closure(q[i], zq, ones_q, qb[i], t[i], zt, ones_t, tb[i], LOWE_RATIO, False)}
    out = {"kind": "port", "cores": 1, "unit": "pairs/s", "sample": "%d of the leg's pairs, oracle/libmso.so (gcc -O2, scalar), 1 thread" % n}
    for name, fn in legs.items():
        out[name] = round(_pool_rate(fn, list(range(n)), 1), 2)
    return out


def cpu_baseline_greedy(wl):
    """The oracle's M1 / M2 (plain C restatement of keyframe_matcher.cpp:50-293) on the same keyframe pairs: 1 thread and all cores."""
    mso = _oracle()
    cores = host_cores()
    k0, sf = wl["kfs"][0], wl["scale_factors"]
    pairs = list(range(1, wl["n_adj"] + 1))

    def m1(i):
        k = wl["kfs"][i]
        return mso.match_loop_closure(k0["desc"], k0["angle"], k0["has_mp"], k0["node"], k["desc"], k["angle"], k["has_mp"], k["node"], wl["lowe_ratio"], True)

    def m2(i):
        k = wl["kfs"][i]
        return mso.match_triangulation(k0["desc"], k0["angle"], k0["octave"], k0["bearing"], 1 - k0["has_mp"], k0["node"],
                                       k["desc"], k["angle"], k["bearing"], 1 - k["has_mp"], k["node"], wl["E"][i - 1], sf, wl["thr_deg"], True)
    out = {"kind": "port", "unit": "ms per keyframe pair", "cores_all": cores,
           "sample": "the leg's %d keyframe pairs x 10 repetitions, oracle/libmso.so (gcc -O2, scalar; includes building the two CSR node tables per call)" % len(pairs)}
    for name, fn in (("loop_closure", m1), ("triangulation", m2)):
        fn(1)
        r1 = _pool_rate(fn, pairs * 10, 1)
        rall = _pool_rate(fn, pairs * 10 * max(cores // 4, 1), cores)
        out[name] = {"ms_per_pair_1_thread": round(1e3 / r1, 4), "ms_per_pair_all_cores": round(1e3 / rall, 4)}
    return out


def bench_greedy(R, ctx, args):
    """M1 matchForLoopClosures / M2 matchForTriangulationDBoW (keyframe_matcher.cpp:50-293) as the mapper calls them: one new keyframe against
    each of its adjacent keyframes (mapper_helpers.cpp:280-293).  ms per pair for a call with ONE pair and for a call with all 20."""
    import ctypes as C
    import numpy as np
    import mi355slam
    import greedy_workload
    wl = greedy_workload.build(ctx, mi355slam, n_adj=20, seed=3000 + R.rank)
    kfs, n_adj = wl["kfs"], wl["n_adj"]
    L = mi355slam.lib()
    res = {"workload": "1 new 720p keyframe vs its %d adjacent keyframes (shift 6 px / 3 px per keyframe), %.0f keypoints and %.0f vocabulary nodes per keyframe "
                       "(k=10, L=6 synthetic vocabulary, levelsUp 4, ms_bow_transform), usable masks at 50 %%, Lowe ratio %.2f, epipolar threshold %.1f deg"
                       % (n_adj, wl["keypoints_per_kf"], wl["nodes_per_kf"], wl["lowe_ratio"], wl["thr_deg"]), "unit": "ms per keyframe pair"}
    dE, dsf = ctx.upload(wl["E"].reshape(-1, 9)), ctx.upload(np.asarray(wl["scale_factors"], np.float32))
    for name, tri in (("loop_closure", False), ("triangulation", True)):
        fr = [mi355slam.FrameOnDevice(ctx, k["desc"], k["angle"], (1 - k["has_mp"]) if tri else k["has_mp"], k["node"],
                                      octave=k["octave"] if tri else None, bearing=k["bearing"] if tri else None) for k in kfs]
        outs = [ctx.alloc(4 * fr[0].n + 16) for _ in range(n_adj)]
        nm = ctx.alloc(4 * n_adj)
        A1 = (mi355slam.MatchFrame * n_adj)(*[fr[0].struct] * n_adj)
        A2 = (mi355slam.MatchFrame * n_adj)(*[f.struct for f in fr[1:]])
        ptrs = (C.c_void_p * n_adj)(*[o.ptr for o in outs])

        def call(n):
            if tri:
                ctx.check(L.ms_match_triangulation(ctx._h, A1, A2, n, mi355slam._vp(dE), mi355slam._vp(dsf), C.c_float(wl["thr_deg"]), 1, ptrs, mi355slam._vp(nm)), "ms_match_triangulation")
            else:
                ctx.check(L.ms_match_loop_closure(ctx._h, A1, A2, n, C.c_float(wl["lowe_ratio"]), 1, ptrs, mi355slam._vp(nm)), "ms_match_loop_closure")
        leg = {}
        for n, reps in ((1, 20), (n_adj, 10)):
            call(n); ctx.sync()
            ctx.event_mark(6)
            for _ in range(reps):
                call(n)
            ctx.event_mark(7)
            ctx.sync()
            dev_ms = ctx.event_elapsed_ms(6, 7) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                call(n); ctx.sync()                               # what a sequential caller sees: launch + wait
            wall_ms = (time.perf_counter() - t0) / reps * 1e3
            leg["batch_%d" % n] = {"ms_per_call_device": round(dev_ms, 4), "ms_per_call_host_synchronous": round(wall_ms, 4), "ms_per_pair": round(dev_ms / n, 4)}
        leg["matches_per_pair"] = round(float(nm.download(np.int32, (n_adj,)).mean()), 1)
        res[name] = leg
    if R.rank == 0 and not args.no_cpu_baseline:
        R.deferred.append((res, lambda: cpu_baseline_greedy(wl)))
    return res


def image_of(i, n_distinct):
    """Frame i of a sequence that walks n_distinct images forwards and backwards (0, 1, .., n-1, n-2, .., 1, 0, 1, ..): consecutive frames always overlap."""
    if n_distinct < 2:
        return 0
    j = i % (2 * n_distinct - 2)
    return j if j < n_distinct else 2 * n_distinct - 2 - j


class SequenceRunner(threading.Thread):
    """One SLAM sequence as its backend thread drives it (mapper_helpers.cpp:1011-1131 order): per frame extract -> match against the
    previous frame -> ratio test; on every k-th frame (a keyframe) one local BA of a NEW C4-shaped window (create + solve + download).
    Own context (stream) and handles; nothing is batched across frames."""

    def __init__(self, device, seq_id, frames_np, windows, kf_every, start_evt, on_frame=None, n_total=None, ba_team=0):
        super().__init__()
        self.ba_team = ba_team                                   # workgroups per BA window (0 = the library's choice: up to 32)
        self.device, self.seq_id, self.frames_np, self.windows, self.kf_every, self.start_evt = device, seq_id, frames_np, windows, kf_every, start_evt
        self.n_total = n_total or len(frames_np)                 # frames of the sequence: the images are walked forwards and backwards (image_of)
        self.on_frame = on_frame                                 # tests only: called after every frame with the device results (synchronises)
        self.frames_done = self.ba_done = self.matches = 0
        self.error = None
        self.seconds = 0.0
        self.ready = threading.Event()

    def run(self):
        try:
            import mi355slam
            ctx = mi355slam.Context(self.device)
            n, nd = self.n_total, len(self.frames_np)
            buf = ctx.upload(self.frames_np)
            ex = [mi355slam.OrbExtractor(ctx, W, H, levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR, max_batch=1) for _ in range(2)]
            cap = ex[0].capacity
            bi, bd, sd, match = ctx.alloc(4 * cap + 16), ctx.alloc(2 * cap + 16), ctx.alloc(2 * cap + 16), ctx.alloc(4 * cap + 16)
            views = [None, None]

            def frame(i, count):
                e = ex[i & 1]
                e.extract(buf.ptr + image_of(i, nd) * W * H, n_frames=1, frame_stride=W * H, row_stride=W)
                if views[i & 1] is None:
                    views[i & 1] = e.device_view()
                if i:
                    q, t = views[i & 1], views[(i - 1) & 1]
                    mi355slam.hamming_best2_sets(ctx, q.desc, cap, q.count, t.desc, cap, t.count, None, None, 1, bi, bd, sd)
                    mi355slam.ratio_test_device(ctx, bi, bd, sd, cap, LOWE_RATIO, 50, match)
                ba_out = None
                if i % self.kf_every == 0 and self.windows:
                    b = mi355slam.BundleAdjuster(ctx, [self.windows[(i // self.kf_every) % len(self.windows)]], max_iters=10)
                    if self.ba_team:
                        b.set_team(self.ba_team)
                    b.solve(); ba_out = b.download(0); b.close()
                    if count:
                        self.ba_done += 1
                if self.on_frame is not None and count:
                    ctx.sync()
                    self.on_frame(i, e, (bi, bd, sd, match) if i else None, ba_out)
            for i in range(min(2, n)):
                frame(i, False)                                  # warm-up: lazily built state, first launches
            ctx.sync()
            self.ready.set()
            self.start_evt.wait()
            t0 = time.perf_counter()
            for i in range(n):
                frame(i, True)
                self.frames_done += 1
            ctx.sync()
            self.seconds = time.perf_counter() - t0
            import numpy as np
            self.matches = int((match.download(np.int32, (cap,)) >= 0).sum())
            ctx.close()
        except Exception as e:                                   # noqa: BLE001 -- reported by the parent
            self.error = e
            self.ready.set()


def bench_pipelined(R, args):
    """One sequence the way the reference runs it on one device: the FRONT END on frame k + 1 (mapper.cpp:356-393: detectAndExtract, match against the previous
    frame, poseBundleAdjust) beside the BACK END on keyframe k (mapper.cpp:229-279, mapper_helpers.cpp:1079-1081: localBundleAdjust, two-stage, a NEW C4 window),
    two host threads, two contexts.  Measured three ways on the same inputs: the front end alone, the back end alone (windows back to back), and both together
    with the back end fed one keyframe per `--c5-keyframe-every` frames (a keyframe that arrives while the back end is busy waits; the front end never does)."""
    import numpy as np
    import ba_synth
    import mi355slam
    import synth
    F, FD, KF = args.c5_frames, min(args.c5_distinct, args.c5_frames), args.c5_keyframe_every
    g = synth.SequenceSynth(W, H, 2000 + 100 * R.rank, 2 * (FD - 1), FD - 1)
    frames = np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(FD)]))
    wins = [ba_synth.make_problem_fast(50, 2000, 10, seed=9000 + 16 * R.rank + k) for k in range(4)]
    cur = len(wins[0]["pose"]) - 1
    stages = [two_stage_problems(p, cur) for p in wins]
    pose_probs = [ba_synth.pose_only_from_window(p, 25) for p in wins]
    iters = int(1 + np.sqrt(50.0))
    extra = np.full(1, cur, np.int32)

    class Front(threading.Thread):
        def __init__(self, start_evt, on_keyframe):
            super().__init__()
            self.start_evt, self.on_keyframe = start_evt, on_keyframe
            self.ready, self.error, self.seconds, self.stage, self.pose_ms = threading.Event(), None, 0.0, {}, 0.0

        def run(self):
            try:
                ctx = mi355slam.Context(R.gpu)
                buf = ctx.upload(frames)
                ex = [mi355slam.OrbExtractor(ctx, W, H, levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR, max_batch=1) for _ in range(2)]
                for e in ex:
                    e.set_profiling(True)
                cap = ex[0].capacity
                bi, bd, sd, match = ctx.alloc(4 * cap + 16), ctx.alloc(2 * cap + 16), ctx.alloc(2 * cap + 16), ctx.alloc(4 * cap + 16)
                views = [None, None]
                pose_t = [0.0]

                def frame(i, count):
                    e = ex[i & 1]
                    e.extract(buf.ptr + image_of(i, FD) * W * H, n_frames=1, frame_stride=W * H, row_stride=W)
                    if views[i & 1] is None:
                        views[i & 1] = e.device_view()
                    if i:
                        q, t = views[i & 1], views[(i - 1) & 1]
                        mi355slam.hamming_best2_sets(ctx, q.desc, cap, q.count, t.desc, cap, t.count, None, None, 1, bi, bd, sd)
                        mi355slam.ratio_test_device(ctx, bi, bd, sd, cap, LOWE_RATIO, 50, match)
                    t0 = time.perf_counter()
                    pb = mi355slam.BundleAdjuster(ctx, [pose_probs[i % len(pose_probs)]], max_iters=10)      # poseBundleAdjust of this frame (bundle_adjuster.cpp:396-491)
                    pb.solve(); pb.download(0); pb.close()
                    if count:
                        pose_t[0] += time.perf_counter() - t0
                        if i % KF == 0 and self.on_keyframe:
                            self.on_keyframe(i // KF)
                for i in range(2):
                    frame(i, False)
                ctx.sync()
                self.ready.set()
                self.start_evt.wait()
                t0 = time.perf_counter()
                for i in range(F):
                    frame(i, True)
                ctx.sync()
                self.seconds = time.perf_counter() - t0
                self.pose_ms = pose_t[0] / F * 1e3
                acc = {}
                for back in range(max(1, min(16, F // 2 - 1))):            # the kernels of the last (up to) 32 frames (HIP events around every stage of ms_orb_extract)
                    for e in ex:
                        for k, v in e.stage_ms_back(back).items():
                            acc.setdefault(k, []).append(v)
                self.stage = {k: round(float(np.mean(v)) * 1e3, 2) for k, v in acc.items() if k not in ("blur",)}
                for e in ex:
                    e.close()
                ctx.close()
            except Exception as e:                                       # noqa: BLE001 -- reported by the caller
                self.error = e
                self.ready.set()

    class Back(threading.Thread):
        def __init__(self, start_evt, n_windows=None):
            super().__init__()
            self.start_evt, self.n_windows = start_evt, n_windows            # n_windows: back to back (alone); None: fed by keyframes until stop()
            self.ready, self.error = threading.Event(), None
            self.pending, self.cv, self.stopping = [], threading.Condition(), False
            self.lat_ms, self.wait_ms = [], []

        def keyframe(self, k):
            with self.cv:
                self.pending.append((k, time.perf_counter())); self.cv.notify()

        def stop(self):
            with self.cv:
                self.stopping = True; self.cv.notify()

        def run(self):
            try:
                ctx = mi355slam.Context(R.gpu)

                def window(k):
                    a, b = stages[k % len(stages)]
                    h1 = mi355slam.BundleAdjuster(ctx, [a], max_iters=iters); h1.solve()      # stage 1 runs while the host builds stage 2's index structures
                    h2 = mi355slam.BundleAdjuster(ctx, [b], max_iters=iters)
                    if args.c5_team:
                        h2.set_team(args.c5_team)
                    h2.copy_state_from(h1, extra); h2.solve(); h2.download(0); h1.close(); h2.close()
                window(0); ctx.sync()
                self.ready.set()
                self.start_evt.wait()
                if self.n_windows is not None:
                    for k in range(self.n_windows):
                        t0 = time.perf_counter(); window(k); self.lat_ms.append((time.perf_counter() - t0) * 1e3)
                else:
                    while True:
                        with self.cv:
                            while not self.pending and not self.stopping:
                                self.cv.wait()
                            if not self.pending:
                                break
                            k, t_arr = self.pending.pop(0)
                        t0 = time.perf_counter(); window(k); t1 = time.perf_counter()
                        self.lat_ms.append((t1 - t0) * 1e3); self.wait_ms.append((t0 - t_arr) * 1e3)
                ctx.close()
            except Exception as e:                                       # noqa: BLE001
                self.error = e
                self.ready.set()

    def run_pair(with_front, back_mode):
        start = threading.Event()
        back = Back(start, n_windows=24) if back_mode == "alone" else (Back(start) if back_mode == "fed" else None)
        front = Front(start, back.keyframe if back_mode == "fed" else None) if with_front else None
        for t in (front, back):
            if t:
                t.start()
        for t in (front, back):
            if t:
                t.ready.wait()
        start.set()
        if front:
            front.join()
        if back and back_mode == "fed":
            back.stop()
        if back:
            back.join()
        for t in (front, back):
            if t and t.error:
                raise t.error
        return front, back
    f_alone, _ = run_pair(True, None)
    _, b_alone = run_pair(False, "alone")
    f_both, b_both = run_pair(True, "fed")
    med = lambda v: round(float(np.median(v)), 3) if len(v) else None

    def run_many(n_seq):
        """n_seq sequences side by side, each with its front end and its back end (2 n_seq host threads and contexts): C5's sequences in the deployment shape."""
        start = threading.Event()
        backs = [Back(start) for _ in range(n_seq)]
        fronts = [Front(start, b.keyframe) for b in backs]
        for t in fronts + backs:
            t.start()
        for t in fronts + backs:
            t.ready.wait()
        start.set()
        for t in fronts:
            t.join()
        for b in backs:
            b.stop()
        for b in backs:
            b.join()
        for t in fronts + backs:
            if t.error:
                raise t.error
        return fronts, backs
    from mi355slam import shard
    n_mine = len([q for q in range(N_SEQ) if shard.sequence_of(q, R.world) == R.rank])
    many = None
    if n_mine > 1:
        fs, bs = run_many(n_mine)
        secs = max(f.seconds for f in fs)
        lat = [x for b in bs for x in b.lat_ms]
        many = {"sequences": n_mine, "frames_per_s": round(n_mine * F / secs, 1), "per_sequence_frames_per_s": [round(F / f.seconds, 1) for f in fs],
                "pose_ba_ms_per_frame": round(float(np.mean([f.pose_ms for f in fs])), 4), "keyframes_handled": len(lat), "keyframes_per_s": round(len(lat) / secs, 1),
                "two_stage_new_window_ms_median": med(lat), "keyframe_wait_ms_median": med([x for b in bs for x in b.wait_ms])}
    # ... and the same with C++ threads on the C ABI (tools/c5_native.cpp: c5p_prepare / c5p_go), so that the interpreter's share of the figure above is known
    native, many_native = _c5_native_lib(), None
    if n_mine > 1 and native is not None and hasattr(native, "c5p_prepare"):
        import ctypes as C
        keep = []
        def structs(probs, it):
            st, kp = zip(*[mi355slam._ba_struct(q, it) for q in probs]); keep.append(kp)
            return (mi355slam.BaProblemC * len(st))(*st)
        parr, s1arr, s2arr = structs(pose_probs, 10), structs([a for a, _ in stages], iters), structs([b for _, b in stages], iters)
        fptr = (C.c_void_p * n_mine)(*[frames.ctypes.data for _ in range(n_mine)])
        native.c5p_prepare.restype = C.c_void_p
        job = native.c5p_prepare(R.gpu, n_mine, F, FD, W, H, fptr, parr, len(pose_probs), s1arr, s2arr, len(stages), cur, KF, LEVELS, C.c_float(SCALE), MAX_KPTS, FAST_THR, C.c_float(LOWE_RATIO))
        if not job:
            raise RuntimeError("c5p_prepare failed")
        secs, seq_s, pose_ms, handled, win_ms, wait_ms = C.c_double(), (C.c_double * n_mine)(), C.c_double(), C.c_int32(), C.c_double(), C.c_double()
        err = C.create_string_buffer(512)
        rc = native.c5p_go(C.c_void_p(job), C.byref(secs), seq_s, C.byref(pose_ms), C.byref(handled), C.byref(win_ms), C.byref(wait_ms), err, 512)
        if rc != 0:
            raise RuntimeError("c5p native driver: %s (status %d)" % (err.value.decode(), rc))
        smax = max(seq_s)
        many_native = {"sequences": n_mine, "driver": "tools/c5_native.cpp (a front-end and a back-end C++ thread per sequence on the C ABI)", "frames_per_s": round(n_mine * F / smax, 1),
                       "per_sequence_frames_per_s": [round(F / x, 1) for x in seq_s], "pose_ba_ms_per_frame": round(pose_ms.value, 4), "keyframes_handled": handled.value,
                       "keyframes_per_s": round(handled.value / smax, 1), "two_stage_new_window_ms_median": round(win_ms.value, 3), "keyframe_wait_ms_median": round(wait_ms.value, 3)}
        del keep
    return {"sequences_side_by_side": many, "sequences_side_by_side_native": many_native, "workload": "one 720p sequence x %d frames; front end per frame: extract -> match vs previous -> ratio test -> poseBundleAdjust (new problem: create + solve + download); "
                        "back end per keyframe (every %d-th frame): localBundleAdjust of a NEW C4 window, two-stage, %d + %d iterations; two host threads, two contexts" % (F, KF, iters, iters),
            "reference": "mapper.cpp:356-393 beside mapper.cpp:229-279 (mapper_helpers.cpp:1043-1050, :1079-1081)",
            "front_end_alone": {"frames_per_s": round(F / f_alone.seconds, 1), "ms_per_frame": round(f_alone.seconds / F * 1e3, 4), "pose_ba_ms_per_frame": round(f_alone.pose_ms, 4),
                                "extract_stage_us": f_alone.stage},
            "back_end_alone": {"two_stage_new_window_ms_median": med(b_alone.lat_ms), "windows": len(b_alone.lat_ms)},
            "together": {"frames_per_s": round(F / f_both.seconds, 1), "ms_per_frame": round(f_both.seconds / F * 1e3, 4), "pose_ba_ms_per_frame": round(f_both.pose_ms, 4),
                         "extract_stage_us": f_both.stage, "two_stage_new_window_ms_median": med(b_both.lat_ms), "keyframes_handled": len(b_both.lat_ms),
                         "keyframe_wait_ms_median": med(b_both.wait_ms), "keyframes_per_s": round(len(b_both.lat_ms) / f_both.seconds, 1)}}


def bench_c5(R, args):
    """BASELINE config 5: 8 independent synthetic sequences, sequence s on GPU s mod N (shard.sequence_of), one host thread + context per
    sequence.  The job's total work is fixed (8 sequences), so across N this leg is STRONG scaling; `value` above stays weak scaling."""
    import ba_synth
    import synth
    from mi355slam import shard
    mine = [s for s in range(N_SEQ) if shard.sequence_of(s, R.world) == R.rank]
    F, FD = args.c5_frames, min(args.c5_distinct, args.c5_frames)
    import numpy as np
    seq_frames, seq_windows = [], []
    for s in mine:
        g = synth.SequenceSynth(W, H, 2000 + s, 2 * (FD - 1), FD - 1)
        seq_frames.append(np.ascontiguousarray(np.stack([g.frame(2 * i, i) for i in range(FD)])))
        seq_windows.append([] if args.no_ba else [ba_synth.make_problem_fast(50, 2000, 10, seed=9000 + 16 * s + k) for k in range(4)])
    # (measured with 8 sequences on one GPU: teams of 32 / 16 / 8 / 4 workgroups per window give 4.4 k / 4.0 k / 3.2 k / 2.3 k frames/s -- a sequence waits for
    # its own window, so the window's latency decides, not the CUs it leaves to the others)
    ba_team = args.c5_team                     # 0 = the library's choice (up to 32 workgroups per window)
    native = _c5_native_lib() if args.c5_native else None
    if native is not None and mine:
        # the sequences' threads are C++ (tools/c5_native.cpp, the C ABI and nothing else).  Measured equal to the Python threads below (3.4 k frames/s + 690 BA/s
        # for 8 sequences on one GPU either way): the leg is bound by how many small kernels of 8 streams the GPU runs side by side, not by the interpreter
        import ctypes as C
        import mi355slam
        windows = [w for ws in seq_windows[:1] for w in ws]       # the same four windows for every sequence of this rank (distinct per rank)
        structs, keep = zip(*[mi355slam._ba_struct(w, 10) for w in windows]) if windows else ((), ())
        warr = (mi355slam.BaProblemC * max(len(structs), 1))(*structs)
        fptr = (C.c_void_p * len(mine))(*[f.ctypes.data for f in seq_frames])
        native.c5_prepare.restype = C.c_void_p
        job = native.c5_prepare(R.gpu, len(mine), F, FD, W, H, fptr, warr, len(structs), args.c5_keyframe_every, LEVELS, C.c_float(SCALE), MAX_KPTS, FAST_THR, C.c_float(LOWE_RATIO))
        if not job:
            raise RuntimeError("c5_prepare failed")
        secs, seq_s = C.c_double(), (C.c_double * len(mine))()
        fd, bd_, lm = (C.c_int32 * len(mine))(), (C.c_int32 * len(mine))(), (C.c_int32 * len(mine))()
        err = C.create_string_buffer(512)
        R.barrier()
        t0 = time.perf_counter()
        rc = native.c5_go(C.c_void_p(job), C.byref(secs), seq_s, fd, bd_, lm, err, 512)
        R.barrier()
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("c5 native driver: %s (status %d)" % (err.value.decode(), rc))
        frames_mine, ba_mine, matches, driver = sum(fd), sum(bd_), list(lm), "native (tools/c5_native.cpp: one C++ thread per sequence on the C ABI)"
        del keep
    else:
        start = threading.Event()
        runners = [SequenceRunner(R.gpu, s, seq_frames[k], seq_windows[k], args.c5_keyframe_every, start, n_total=F, ba_team=ba_team) for k, s in enumerate(mine)]
        for r in runners:
            r.start()
        for r in runners:
            r.ready.wait()
        R.barrier()
        t0 = time.perf_counter()
        start.set()
        for r in runners:
            r.join()
        R.barrier()
        dt = time.perf_counter() - t0
        errs = [r.error for r in runners if r.error]
        if errs:
            raise errs[0]
        frames_mine, ba_mine, matches, driver = sum(r.frames_done for r in runners), sum(r.ba_done for r in runners), [r.matches for r in runners], "python threads (SequenceRunner)"
    frames_total, dt_max = R.aggregate(frames_mine, dt)
    ba_total, _ = R.aggregate(ba_mine, dt)
    return {"workload": "8 independent 720p sequences x %d frames (%d distinct images each, walked forwards and backwards); per frame extract -> match vs previous -> ratio test; "
                        "every %d-th frame a local BA of a new C4 window (create + solve + download); sequence s on GPU s mod N, one host thread + context per sequence" % (F, FD, args.c5_keyframe_every),
            "driver": driver, "ba_team": ba_team or "library default (up to 32)",
            "scaling": "strong (8 sequences in total)", "frames_per_s": round(frames_total / dt_max, 1), "ba_per_s": round(ba_total / dt_max, 1),
            "seconds": round(dt_max, 4), "per_gpu": [round(v, 1) for v in R.gather(frames_mine / dt)],
            "sequences_per_gpu": R.gather(len(mine)), "last_frame_matches": matches,
            # the hardware queues this rank's streams were mapped onto (ms_prepare_process; under a profiler that initialises the GPU first it is what the environment said)
            "hw_queues": getattr(R, "hw_queues", None), "hw_queues_prepared_by_library": getattr(R, "queues_prepared", None)}


def _c5_native_lib():
    """The native C5 driver built next to the library (make -C slam-module_amd/csrc), or None."""
    import ctypes as C
    path = os.path.join(ROOT, "slam-module_amd", "lib", "libc5native.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path)
    except OSError:
        return None


def ctx_download(ctx, dev_ptr, nbytes):
    import ctypes as C
    import mi355slam
    buf = (C.c_char * nbytes)()
    ctx.check(mi355slam.lib().ms_dev_download(ctx._h, buf, C.c_void_p(dev_ptr), C.c_size_t(nbytes)), "ms_dev_download")
    return bytes(buf)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch(args, argv))                  # the parent never touches the GPU
    R = Rank(args)
    if R.world != args.gpus and R.rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d; reporting n_gpus = WORLD_SIZE\n" % (args.gpus, R.world))
    try:
        if args.plumbing:
            run_plumbing(R, args)
        else:
            run_gpu(R, args)
    finally:
        R.close()


if __name__ == "__main__":
    main()
