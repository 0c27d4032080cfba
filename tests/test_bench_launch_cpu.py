"""`python bench.py --gpus N` from a bare shell must bring up its own N ranks (the driver calls it exactly like that),
as child processes and before anything touches a GPU; under an existing launch WORLD_SIZE must agree with --gpus.
The GPU work itself is covered by tests/test_gpu_bench.py; here: the command line, the guard, and a real 2-rank
rendezvous over gloo in --dry-run mode."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launch_command_is_the_drivers_launch_line():
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--mode", "shard"]
    args = bench.parse_args(argv)
    cmd = bench.launch_command(args, argv, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv                                # every flag reaches the ranks unchanged


def test_self_launch_never_execs_and_comes_before_any_gpu_call():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
    body = src[src.index("def main("):]
    assert body.index("self_launch(args, argv)") < body.index("torch.cuda.is_available()")
    assert body.index("self_launch(args, argv)") < body.index("import torch")


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in r.stderr


@pytest.mark.timeout(600)
def test_plain_command_brings_up_two_ranks_and_rank0_prints_one_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=_env(), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True


def test_single_rank_dry_run_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


@pytest.mark.timeout(300)
def test_a_failing_rank_ends_the_launch_with_FAIL_and_a_nonzero_code_not_a_signal_or_a_hang():
    """rank 1 fails before its first collective while rank 0 already waits in the barrier.  The
    failing rank must say FAIL and leave with a non-zero code at once (no collective to share the failure: the other
    rank is in a different one), the launcher then ends rank 0; nothing aborts, nothing hangs."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo",
                        "--inject-failure", "1"], env=_env(), capture_output=True, text=True, timeout=250)
    dt = time.time() - t0
    assert r.returncode != 0, (r.stdout, r.stderr[-2000:])
    assert "FAIL: RuntimeError('injected failure on rank 1')  [rank 1 of 2]" in r.stderr, r.stderr
    assert "Signal 6" not in r.stderr and "SIGABRT" not in r.stderr and "SIGSEGV" not in r.stderr, r.stderr
    assert "exitcode  : 1" in r.stderr                       # the launcher's summary: an exit code, not a signal
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no bench line from a failed run
    assert dt < 120, dt


def test_single_rank_failure_is_a_plain_nonzero_exit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--inject-failure", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "FAIL: RuntimeError('injected failure on rank 0')" in r.stderr
