"""Do two batches in flight on two streams (each with its own extractor and scratch) beat one after the other?
The front-end kernels have complementary limits (k_fast: VALU issue; k_resize / k_describe: memory latency)."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
W, H, B = 1280, 720, 256
base = np.stack([mso.synth_frame(W, H, 1000 + i // 8, 2 * (i % 8), i % 8) for i in range(B)])
def make(n_pipes):
    pipes = []
    for _ in range(n_pipes):
        ctx = mi355slam.Context(0)
        buf = ctx.upload(base)
        ex = mi355slam.OrbExtractor(ctx, W, H, max_batch=B)
        cap = ex.capacity
        pq = ctx.upload(np.arange(B, dtype=np.int32)); pt = ctx.upload(np.roll(np.arange(B, dtype=np.int32), 1))
        bi, bd, sd, m = ctx.alloc(4 * B * cap), ctx.alloc(2 * B * cap), ctx.alloc(2 * B * cap), ctx.alloc(4 * B * cap)
        pipes.append(dict(ctx=ctx, buf=buf, ex=ex, cap=cap, pq=pq, pt=pt, bi=bi, bd=bd, sd=sd, m=m, view=None))
    return pipes
def step(p):
    p["ex"].extract(p["buf"], n_frames=B, frame_stride=W * H, row_stride=W)
    if p["view"] is None: p["view"] = p["ex"].device_view()
    v = p["view"]
    mi355slam.hamming_best2_sets(p["ctx"], v.desc, p["cap"], v.count, v.desc, p["cap"], v.count, p["pq"], p["pt"], B, p["bi"], p["bd"], p["sd"])
    mi355slam.ratio_test_device(p["ctx"], p["bi"], p["bd"], p["sd"], B * p["cap"], 0.75, 50, p["m"])
for n_pipes in (1, 2, 3):
    pipes = make(n_pipes)
    for p in pipes: step(p); step(p)
    for p in pipes: p["ctx"].sync()
    K = 24
    t0 = time.perf_counter()
    for i in range(K): step(pipes[i % n_pipes])
    for p in pipes: p["ctx"].sync()
    dt = time.perf_counter() - t0
    print("%d batch(es) in flight: %.3f ms per 256-frame step, %.0f frames/s" % (n_pipes, dt / K * 1e3, K * B / dt), flush=True)
    for p in pipes: p["ctx"].close()
