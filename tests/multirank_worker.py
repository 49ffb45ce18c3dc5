"""One rank of tests/test_gpu_multirank.py: bench.py's own Rank (device binding, ms_prepare_process, process group, aggregation, the ending) with every rank on
device 0 (--ranks-share-gpu), a small extraction + match + local BA per rank on rank-specific inputs, each checked against the CPU oracle IN the rank.
Started once per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set; rank 0 prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "slam-module_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import bench


def main():
    args = bench.parse_args(["--gpus", os.environ["WORLD_SIZE"], "--ranks-share-gpu", "--no-cpu-baseline"])
    R = bench.Rank(args)                                   # torch first, then ms_prepare_process, then the device, then the (gloo) group
    import ba_synth
    import mi355slam
    import mso
    ctx = mi355slam.Context(R.gpu)
    ok = {}
    imgs = np.stack([mso.synth_frame(640, 480, 1000 + 8 * R.rank, 2 * i, i) for i in range(2)])      # a 2-frame sequence of this rank's own
    ex = mi355slam.OrbExtractor(ctx, 640, 480, max_batch=2)
    ex.extract(imgs)
    kps = [ex.download(f) for f in range(2)]
    want = [mso.orb_extract(mso.cfg(), imgs[f]) for f in range(2)]
    ok["keypoints"] = all(len(k["x"]) == len(w["x"]) > 300 and np.array_equal(k["desc"], w["desc"]) and
                          np.array_equal(k["angle"].view(np.uint32), w["angle"].view(np.uint32)) for k, w in zip(kps, want))
    bi, bd, sd = mi355slam.hamming_best2(ctx, kps[1]["desc"], kps[0]["desc"])
    wi, wd, ws = mso.hamming_best2(kps[1]["desc"], kps[0]["desc"])
    ok["matches"] = bool(np.array_equal(bi, wi) and np.array_equal(bd, wd) and np.array_equal(sd, ws))
    prob = ba_synth.make_problem(8, 150, 5, seed=3 + R.rank)
    probs = [prob, ba_synth.make_problem(10, 200, 6, seed=40 + R.rank)]
    for name, team in (("ba_batch", 1), ("ba_team", 4)):
        ba = mi355slam.BundleAdjuster(ctx, probs if team == 1 else probs[:1], max_iters=6)
        ba.set_team(team)
        ba.solve()
        good = True
        for i in range(2 if team == 1 else 1):
            got, w = ba.download(i), mso.ba_solve(probs[i], 6, False)
            d = np.abs(ba_synth.residuals(probs[i], got["pose"], got["point"]) - ba_synth.residuals(probs[i], w["pose"], w["point"])).max()
            good = good and d < 1e-7 and got["stats"]["trials"] == w["stats"]["trials"]
        ok[name] = bool(good)
        ba.close()
    ex.close()
    ctx.close()
    units, seconds = R.aggregate(len(kps[0]["x"]) + len(kps[1]["x"]), 1.0 + R.rank)      # summed units, the slowest rank's time
    all_ok = R.gather(1.0 if all(ok.values()) else 0.0)
    n_kp = R.gather(len(kps[0]["x"]))
    R.barrier()
    R.close()
    if R.rank == 0:
        print(json.dumps({"world": R.world, "backend": R.backend, "device_of_rank": R.gpu, "units": units, "seconds": seconds, "ranks_ok": all_ok,
                          "keypoints_of_rank": n_kp, "rank0": ok, "hw_queues": getattr(R, "hw_queues", None)}), flush=True)
    if not all(ok.values()):
        sys.stderr.write("rank %d: %s\n" % (R.rank, ok))
        sys.exit(4)


if __name__ == "__main__":
    main()
