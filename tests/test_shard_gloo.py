"""N>1 path on CPU: two gloo ranks shard independent units with no data-path collective and agree on totals / max time."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-module_amd"))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "slam-module_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from mi355slam import shard
    import mso
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(7, rank, world)             # 7 descriptor pairs over 2 ranks: 4 + 3
    digest = 0
    for p in range(lo, hi):                                # each rank works on its own units only (CPU oracle stands in for the GPU)
        rng = np.random.default_rng(p)
        q_ = rng.integers(0, 2**32, (50, 8), dtype=np.uint64).astype(np.uint32); t_ = rng.integers(0, 2**32, (60, 8), dtype=np.uint64).astype(np.uint32)
        bi, bd, sd = mso.hamming_best2(q_, t_)
        digest += int(bd.sum())
    dist.barrier()
    units, seconds = shard.aggregate(dist, torch, hi - lo, 1.0 + rank)
    q.put((rank, lo, hi, units, seconds, digest))
    dist.destroy_process_group()


def test_two_ranks_shard_and_aggregate():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(60)
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]                  # disjoint, complete cover
    assert all(r[3] == 7.0 and r[4] == 2.0 for r in res)                     # units summed, time = max over ranks
    # the sharded work equals the unsharded work
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mso, numpy as np
    total = 0
    for p in range(7):
        rng = np.random.default_rng(p)
        q_ = rng.integers(0, 2**32, (50, 8), dtype=np.uint64).astype(np.uint32); t_ = rng.integers(0, 2**32, (60, 8), dtype=np.uint64).astype(np.uint32)
        total += int(mso.hamming_best2(q_, t_)[1].sum())
    assert res[0][5] + res[1][5] == total


def test_shard_range_properties():
    from mi355slam import shard
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert [shard.sequence_of(s, 8) for s in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
