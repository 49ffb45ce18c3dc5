"""One NEW C4 window under the two-stage schedule, the way the mapper pays for it (bundle_adjuster.cpp:245-373): wall-clock of every call of the sequence
create(stage 1) -> solve -> create(stage 2) -> copy_state -> solve -> download -> destroys.  MS_BA_TIMING=1 adds the library's own split of each create."""
import sys, os, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in ("slam-module_amd", "tests", "tools", R): sys.path.insert(0, os.path.join(R, p) if p != R else R)
import numpy as np, mi355slam, ba_synth
import bench
ctx = mi355slam.Context(0)
wins = [ba_synth.make_problem_fast(50, 2000, 10, seed=5 + i) for i in range(4)]
st = [bench.two_stage_problems(w, 49) for w in wins]
extra = np.array([49], np.int32)
names = ("create1", "solve1", "create2", "copy", "solve2", "download", "close")
acc = {k: 0.0 for k in names}; N = 12
for i in range(N + 3):
    if i == 3: acc = {k: 0.0 for k in names}; t_all = time.perf_counter()
    a, b = st[i % len(st)]
    t = [time.perf_counter()]
    h1 = mi355slam.BundleAdjuster(ctx, [a], max_iters=8); t.append(time.perf_counter())
    h1.solve(); t.append(time.perf_counter())
    h2 = mi355slam.BundleAdjuster(ctx, [b], max_iters=8); t.append(time.perf_counter())
    h2.copy_state_from(h1, extra); t.append(time.perf_counter())
    h2.solve(); t.append(time.perf_counter())
    h2.download(0); t.append(time.perf_counter())
    h1.close(); h2.close(); t.append(time.perf_counter())
    for k, (x, y) in zip(names, zip(t[:-1], t[1:])): acc[k] += y - x
print("new two-stage window: %.3f ms; per call (ms): %s" % ((time.perf_counter() - t_all) / N * 1e3, {k: round(v / N * 1e3, 3) for k, v in acc.items()}))
# the device side alone
h1 = mi355slam.BundleAdjuster(ctx, [st[0][0]], max_iters=8); h2 = mi355slam.BundleAdjuster(ctx, [st[0][1]], max_iters=8)
for rep in range(3):
    ctx.sync(); ctx.event_mark(0); h1.solve(); ctx.event_mark(1); h2.copy_state_from(h1, extra); h2.solve(); ctx.event_mark(2); ctx.sync()
print("device: stage 1 %.3f ms, copy + stage 2 %.3f ms" % (ctx.event_elapsed_ms(0, 1), ctx.event_elapsed_ms(1, 2)))
