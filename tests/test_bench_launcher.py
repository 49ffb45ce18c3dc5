"""bench.py --gpus N is its own launcher: N child ranks started before anything touches a GPU, rank 0's line relayed, a failed rank
fails the run.  Driven here exactly as the driver would (`python bench.py --gpus 2 ...`) with the GPU work replaced by a sleep
(--plumbing, gloo): the rendezvous, the barriers, the MAX-over-ranks time and the unit totals are the real code path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*extra, env=None):
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH, "--plumbing", "--steps", "4", "--warmup", "1", *extra], capture_output=True, text=True, env=e, timeout=300)


def test_launcher_starts_two_ranks_and_reports_the_whole_job():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                           # ONE JSON line, printed by the parent
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["plumbing"] is True
    assert out["units_total"] == 2 * 256 * 4 and out["per_gpu_units"] == [1024.0, 1024.0]      # both ranks' units, summed
    # rank 1 sleeps twice as long per step as rank 0: the job's time is the slowest rank's
    assert out["ms_per_step"] >= 3.9
    assert abs(out["value"] - out["units_total"] / (out["ms_per_step"] * 4e-3)) / out["value"] < 0.01
    assert out["c5"]["sequences_of_rank"] == [[0, 2, 4, 6], [1, 3, 5, 7]]                      # sequence s -> GPU s mod N
    assert out["cpu_baseline"]["kind"] == "port"                                                # rank 0 timed it alone, after the process group was gone -- at N = 2 too
    for k in ("metric", "unit", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out


def test_single_rank_needs_no_launcher():
    r = _run()
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["units_total"] == 1024.0


def test_a_failing_rank_fails_the_run():
    r = _run("--gpus", "2", "--plumbing-fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]                      # no result line from a broken run
    assert "rank exit codes" in r.stderr


def test_rank_mode_under_an_external_launcher():
    """`python -m torch.distributed.run ... bench.py --gpus 2` sets RANK itself: then bench.py must not spawn anything."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", BENCH, "--gpus", "2", "--plumbing", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_parent_imports_no_gpu_library():
    """The launcher branch runs before `import torch` / the HIP library: asserted on the source (the parent must never touch the GPU)."""
    src = open(BENCH).read()
    head = src[:src.index("def launch(")]
    assert "import torch" not in head and "import mi355slam" not in head
    body = src[src.index("def launch("):src.index("# ------------------------------------------------------------------------------------------------------------------ CPU baseline")]
    assert "import torch" not in body and "import mi355slam" not in body and "os.exec" not in src


def _long_launcher(*extra):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    # 30000 steps of 2-4 ms: the ranks would run for a minute or two if nothing stopped them
    return subprocess.Popen([sys.executable, BENCH, "--plumbing", "--gpus", "2", "--steps", "30000", "--warmup", "0", *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def _ranks_of(launcher, wait_s=60):
    import time
    psutil = pytest.importorskip("psutil")      # (in this image's wheelhouse; not a dependency of the product)
    t_end = time.time() + wait_s
    while time.time() < t_end:
        kids = [c for c in psutil.Process(launcher.pid).children(recursive=True)]
        if len(kids) >= 2:
            return kids
        time.sleep(0.1)
    return []


def _all_gone(procs, wait_s=40):
    psutil = pytest.importorskip("psutil")      # (in this image's wheelhouse; not a dependency of the product)
    gone, alive = psutil.wait_procs(procs, timeout=wait_s)
    return not alive


def test_launcher_deadline_stops_every_rank():
    """a launcher that is past its wall-clock limit ends its ranks (own sessions, TERM then KILL) and says why."""
    p = _long_launcher("--launcher-timeout", "4")
    kids = _ranks_of(p)
    assert len(kids) >= 2
    out, err = p.communicate(timeout=120)
    assert p.returncode != 0 and "deadline" in err and not [ln for ln in out.splitlines() if ln.startswith("{")]
    assert _all_gone(kids)


def test_launcher_killed_or_terminated_takes_its_ranks_along():
    import signal
    import time
    p = _long_launcher()
    kids = _ranks_of(p)
    assert len(kids) >= 2
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)                  # handled: ranks are stopped, exit code non-zero, the reason is printed
    out, err = p.communicate(timeout=120)
    assert p.returncode != 0 and "signal" in err
    assert _all_gone(kids)
    p = _long_launcher()
    kids = _ranks_of(p)
    assert len(kids) >= 2
    p.kill()                                       # not handled by anything: PR_SET_PDEATHSIG ends the ranks
    p.wait()
    assert _all_gone(kids)
