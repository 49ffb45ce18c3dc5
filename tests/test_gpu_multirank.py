"""The N > 1 code of bench.py on ONE GPU: `--ranks-share-gpu` binds every rank to device 0 and moves the ranks' few numbers over gloo (RCCL refuses two ranks on a
device).  Everything else is the real multi-GPU path: the launcher, one process / context / slab set per rank, ms_prepare_process per rank, the C5 partition
s mod N, the ending (ranks meet, group goes down, rank 0 prints).  What cannot be seen here is RCCL itself and xGMI."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_ranks_on_one_gpu_check_their_own_outputs_against_the_oracle():
    """Two processes built on bench.Rank, each with its own context on device 0 and its own inputs: keypoints / descriptors / matches bit-exact and BA residuals
    within 1e-7 of the oracle IN BOTH ranks; units summed, time = the slowest rank's."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("GPU_MAX_HW_QUEUES", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py")], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-2000:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["world"] == 2 and line["backend"] == "gloo" and line["device_of_rank"] == 0
    assert line["ranks_ok"] == [1.0, 1.0] and all(line["rank0"].values())
    assert line["units"] > sum(line["keypoints_of_rank"]) > 600 and line["seconds"] == 2.0      # SUM of the units, MAX of the seconds
    assert line["keypoints_of_rank"][0] != line["keypoints_of_rank"][1]                        # the ranks really worked on different inputs
    assert line["hw_queues"] == 10                                                             # ms_prepare_process took effect in the rank (before its first HIP call)


def test_bench_launcher_runs_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --ranks-share-gpu` as the driver would start it (small legs): one line, n_gpus 2, both ranks' rates, BA and C5 on both ranks."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--ranks-share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-greedy",
           "--ba-batch", "4", "--ba-steps", "1", "--c5-frames", "10", "--c5-distinct", "5", "--launcher-timeout", "400"]
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=480, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_share_gpu"] is True and line["collective_backend"] == "gloo" and line["collective_world"] == 2
    assert len(line["per_gpu_frames_per_s"]) == 2 and min(line["per_gpu_frames_per_s"]) > 1000
    assert abs(line["value"] - 2 * 256 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 0.02      # both ranks' frames over the slowest rank's time
    assert "failed_legs" not in line, {k: line[k] for k in line.get("failed_legs", [])}
    assert line["pipelined_sequence"]["together"]["keyframes_handled"] >= 1
    assert line["local_ba"]["windows_per_launch"] == 4 and line["local_ba"]["value"] > 100
    assert line["c5"]["sequences_per_gpu"] == [4.0, 4.0] and line["c5"]["frames_per_s"] > 100
    assert line["c5"]["hw_queues"] == 10
