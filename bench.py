#!/usr/bin/env python3
"""bench.py -- ORB extract + match throughput on synthetic 720p frames (BASELINE.json config C2 + C3 policy).

One "step" = one pass of the hot path over one batch of 256 synthetic 1280x720 frames that are already
resident in HBM: pyramid (8 levels x1.2) -> FAST -> top-2000 selection -> orientation -> 256-bit steered
BRIEF, then brute-force Hamming best/second-best + ratio test of every frame's descriptors against the
previous frame's (256 pairs, up to 2000x2000 each).  Multi-GPU: one process per GPU, every rank runs its
own batch (independent sequences, no data-path collective); RCCL is used only to agree on the timing.

Prints ONE JSON line (see the contract in the task statement).  `roofline` is computed for the dominant
kernel from HIP-event durations taken inside the timed region; `cpu_baseline` times the CPU oracle (port of
the reference algorithm, test infrastructure) on a bounded sample of the same workload on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "slam-module_amd"))

W, H, LEVELS, SCALE, MAX_KPTS, FAST_THR, BATCH = 1280, 720, 8, 1.2, 2000, 20, 256
LOWE_RATIO = 0.75
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def synth_batch(n, base_seed):
    """8 sequences of n/8 frames: frame i = synth(seed, shift=(2k, k)) so consecutive frames really match."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mso                                  # generator only (integer synthetic frames)
    import numpy as np
    per = max(n // 8, 1)
    return np.stack([mso.synth_frame(W, H, base_seed + i // per, 2 * (i % per), i % per) for i in range(n)])


def cpu_baseline(frames, n_sample):
    """CPU oracle (port) on the first n_sample frames: extract each + match against the previous one; 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mso
    cfg = mso.cfg(levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR)
    t0 = time.perf_counter()
    prev = None
    for f in range(n_sample):
        kp = mso.orb_extract(cfg, frames[f])
        if prev is not None:
            mso.hamming_best2(kp["desc"], prev["desc"])
        else:
            mso.hamming_best2(kp["desc"], kp["desc"])
        prev = kp
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d synthetic 720p frames (extract + 2000x2000 Hamming best2), oracle/libmso.so, 1 thread" % (n_sample, BATCH)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--ba-batch", type=int, default=256)
    ap.add_argument("--ba-steps", type=int, default=3)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch                                   # plumbing: device memory for the inputs + torch.distributed
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import mi355slam                                # after torch: both must share one HIP runtime

    ctx = mi355slam.Context(local_rank)
    frames_np = synth_batch(BATCH, 1000 + 8 * rank)
    frames = torch.from_numpy(frames_np).cuda()     # inputs resident in HBM before the timed region
    ex = mi355slam.OrbExtractor(ctx, W, H, levels=LEVELS, scale_factor=SCALE, max_kpts=MAX_KPTS, fast_threshold=FAST_THR, max_batch=BATCH)
    cap = ex.capacity
    view = None
    pair_q = torch.arange(BATCH, dtype=torch.int32, device="cuda")
    pair_t = torch.roll(pair_q, 1)                  # frame f against frame f-1 (frame 0 against the last one)
    best_idx = torch.empty(BATCH * cap, dtype=torch.int32, device="cuda")
    best_dist = torch.empty(BATCH * cap, dtype=torch.int16, device="cuda")
    second_dist = torch.empty(BATCH * cap, dtype=torch.int16, device="cuda")
    match = torch.empty(BATCH * cap, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()

    def step(profile_match=False):
        nonlocal view
        ex.extract(frames.data_ptr(), n_frames=BATCH, frame_stride=W * H, row_stride=W)
        if view is None:
            view = ex.device_view()
        if profile_match:
            ctx.event_mark(0)
        mi355slam.hamming_best2_sets(ctx, view.desc, cap, view.count, view.desc, cap, view.count, pair_q.data_ptr(), pair_t.data_ptr(),
                                     BATCH, best_idx.data_ptr(), best_dist.data_ptr(), second_dist.data_ptr())
        if profile_match:
            ctx.event_mark(1)
        mi355slam.ratio_test_device(ctx, best_idx.data_ptr(), best_dist.data_ptr(), second_dist.data_ptr(), BATCH * cap, LOWE_RATIO, 50, match.data_ptr())

    for _ in range(args.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # ---- timed region: exactly K steps; per-kernel HIP events ride along on the context stream ----
    ex.set_profiling(True)
    stage_sum = {}
    match_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(profile_match=True)
        for k, v in ex.stage_ms().items():          # reading the events waits for this step (steps are serial anyway)
            stage_sum[k] = stage_sum.get(k, 0.0) + v
        match_ms += ctx.event_elapsed_ms(0, 1)
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    from mi355slam import shard
    frames_total, dt = shard.aggregate(dist if world > 1 else None, torch, BATCH * args.steps, dt, device="cuda")

    # outside the timed region: the same 256 searches on the popcount kernel (v_xor / v_bcnt), for comparison with the matrix-core one
    mi355slam.lib().ms_hamming_set_path(1)
    for rep in range(3):
        if rep == 1:
            ctx.event_mark(2)
        mi355slam.hamming_best2_sets(ctx, view.desc, cap, view.count, view.desc, cap, view.count, pair_q.data_ptr(), pair_t.data_ptr(),
                                     BATCH, best_idx.data_ptr(), best_dist.data_ptr(), second_dist.data_ptr())
    ctx.event_mark(3)
    popcount_ms = ctx.event_elapsed_ms(2, 3) / 2
    mi355slam.lib().ms_hamming_set_path(0)
    n_kp = np.frombuffer(ctx_download(ctx, view.count, 4 * BATCH), dtype=np.int32)
    n_match = int((match.view(BATCH, cap) >= 0).sum().item())
    value = frames_total / dt

    # ---- roofline of the dominant kernel (algorithmic bytes per launch, SURVEY 8d / DESIGN.md) ----
    ws, hs = mi355slam.level_sizes(LEVELS, SCALE, W, H)
    P = int((ws.astype(np.int64) * hs).sum()); N0 = W * H; K = float(n_kp.mean())
    alg = {   # bytes per frame
        "resize": (P - int(ws[-1]) * int(hs[-1])) + (P - N0),      # read levels 0..n-2, write levels 1..n-1
        "blur": 2 * P, "fast": P, "select": 4 * K, "tracks": 0, "describe": 1821 * K,
        "hamming": 32 * 2 * K + 8 * K,
    }
    avg_ms = {k: v / args.steps for k, v in stage_sum.items()}
    avg_ms["hamming"] = match_ms / args.steps
    dom = max(avg_ms, key=avg_ms.get)
    kernels = {k: {"ms_per_launch": round(avg_ms[k], 4), "alg_GBs": round(alg[k] * BATCH / (avg_ms[k] * 1e-3) / 1e9, 1) if avg_ms[k] > 0 else None}
               for k in avg_ms}
    kernels["hamming"]["popcount_kernel_ms"] = round(popcount_ms, 4)
    # the unmasked search is an i8 matrix product (1 multiply-add per descriptor bit and pair): its own roofline is the dense i8 MFMA peak
    pair_ops = 2.0 * 256 * float((n_kp.astype(np.float64) * np.roll(n_kp, 1).astype(np.float64)).sum())
    kernels["hamming"]["mfma"] = {"bound": "mfma", "achieved": round(pair_ops / (avg_ms["hamming"] * 1e-3) / 1e12, 1), "peak": 5000.0, "unit": "Top/s (i8)",
                                  "frac": round(pair_ops / (avg_ms["hamming"] * 1e-3) / 1e12 / 5000.0, 4)}
    achieved = alg[dom] * BATCH / (avg_ms[dom] * 1e-3) / 1e9
    # From the committed PMC passes (profiles/r01_pmc_traffic.json, tools/pmc_summary.py): HBM bytes per launch and the
    # wave-level VALU instruction count.  The front-end kernels are bound by VALU ISSUE, not by HBM: almost all their
    # instructions are VOP3 / packed forms that issue once per ~4 cycles per SIMD (profiles/r01_e_valu_issue_rates.txt), so
    # valu_issue_frac = count / 1024 SIMDs x 4 cycles / 2.4 GHz / (measured launch time) is the fraction of that limit in use.
    traffic = valu = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
        kname = {"hamming": "k_hamming_mfma"}.get(dom, "k_" + dom)
        traffic = pmc[kname]["hbm_bytes_per_step"]
        valu = pmc[kname]["SQ_INSTS_VALU_per_step"]
    except Exception:
        pass
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "valu_insts": valu, "valu_issue_frac": round(valu / 1024 * 4 / 2.4e9 / (avg_ms[dom] * 1e-3), 3) if valu else None,
                "whole_step_alg_GBs": round((4 * P + 1821 * K + 72 * K) * BATCH * args.steps / dt / 1e9, 1)}
    out = {
        "metric": "frames/sec ORB extract+match (720p)", "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "C2+C3: ORB extract 1280x720, 8 levels x1.2, 2000 kpts/frame, FAST thr 20, batch 256 synthetic frames/GPU, "
                               "+ Hamming brute-force best2 + ratio 0.75 of each frame vs the previous (256 pairs, <=2000x2000)",
                   "batch_per_gpu": BATCH, "keypoints_per_frame": round(K, 1), "ratio_matches_per_frame": round(n_match / BATCH, 1)},
        "roofline": roofline, "kernels": kernels,
    }
    # ---- secondary metric: local-BA solves/s (BASELINE config C4), 256 windows per launch, device-resident ----
    if not args.no_ba:
        out["local_ba"] = bench_ba(ctx, args, world, rank, dist if world > 1 else None, torch)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_np, 128)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


def bench_ba(ctx, args, world, rank, dist, torch):
    """C4: 50 keyframes x 2000 points x 20000 observations, 10 LM iterations, Huber sqrt(5.991), 49 odometry edges."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ba_synth
    import mi355slam
    distinct = [ba_synth.make_problem(50, 2000, 10, seed=42 + 8 * rank + i) for i in range(4)]
    probs = [distinct[i % len(distinct)] for i in range(args.ba_batch)]
    ba = mi355slam.BundleAdjuster(ctx, probs, max_iters=10)
    ba.solve(); ctx.sync()                                   # warm-up
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    ctx.event_mark(2)
    for _ in range(args.ba_steps):
        ba.solve()
    ctx.event_mark(3)
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ctx.event_elapsed_ms(2, 3) / args.ba_steps
    from mi355slam import shard
    solves_total, dt = shard.aggregate(dist, torch, args.ba_batch * args.ba_steps, dt, device="cuda")
    st = ba.download(0)["stats"]
    # single-window latency (one workgroup on one CU)
    # latency of ONE window (what a sequential SLAM pipeline sees): a team of workgroups shares the problem (automatic size);
    # one_cu = the same solve confined to a single workgroup, as every window of the 256-window launch above runs
    one = mi355slam.BundleAdjuster(ctx, probs[:1], max_iters=10)
    one.solve(); ctx.sync()
    ctx.event_mark(4); one.solve(); ctx.event_mark(5)
    single_ms = ctx.event_elapsed_ms(4, 5)
    one.set_team(1); one.solve(); ctx.sync()
    ctx.event_mark(4); one.solve(); ctx.event_mark(5)
    single_one_cu_ms = ctx.event_elapsed_ms(4, 5)
    alg_bytes_per_solve = 6.61e6 * st["iters"]               # 6.61 MB per LM iteration at C4
    res = {"metric": "local-BA solves/sec (50 KF x 2000 pts x 20k obs, 10 LM iters)", "value": round(solves_total / dt, 1),
           "unit": "solves/s", "windows_per_launch": args.ba_batch, "ms_per_launch": round(kernel_ms, 3), "lm_iterations": st["iters"],
           "lm_trials": st["trials"], "single_window_ms": round(single_ms, 3), "single_window_solves_per_s": round(1e3 / single_ms, 1),
           "single_window_one_cu_ms": round(single_one_cu_ms, 3),
           "alg_GBs": round(alg_bytes_per_solve * args.ba_batch / (kernel_ms * 1e-3) / 1e9, 1), "dtype": "f64"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mso
        t1 = time.perf_counter()
        n_cpu = 48
        for i in range(n_cpu):
            mso.ba_solve(distinct[i % 4], 10, False)
        res["cpu_baseline"] = {"value": round(n_cpu / (time.perf_counter() - t1), 2), "unit": "solves/s", "cores": 1, "kind": "port",
                               "sample": "%d solves of C4 windows (4 distinct), oracle/libmso.so (Schur + dense Cholesky), 1 thread" % n_cpu}
    ba.close(); one.close()
    return res


def ctx_download(ctx, dev_ptr, nbytes):
    import ctypes as C
    import mi355slam
    buf = (C.c_char * nbytes)()
    ctx.check(mi355slam.lib().ms_dev_download(ctx._h, buf, C.c_void_p(dev_ptr), C.c_size_t(nbytes)), "ms_dev_download")
    return bytes(buf)


if __name__ == "__main__":
    main()
