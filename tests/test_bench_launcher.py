"""`python bench.py --gpus N` invoked bare: the parent makes no GPU call, starts one fresh rank process per GPU through
torch.distributed.run, relays rank 0's JSON line and passes the ranks' status on.  BENCH_LAUNCH_ONLY=1 replaces the GPU
work of the ranks by a gloo rendezvous + one all-reduce, so the launcher itself is covered on a CPU-only box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *argv):
    env = dict(os.environ, BENCH_LAUNCH_ONLY="1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bare_multi_gpu_invocation_spawns_the_ranks_and_relays_one_line():
    p = run_bench({}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_sum"] == 1.0 and line["launched_by_parent"] is True


def test_a_failing_rank_fails_the_bare_invocation():
    p = run_bench({"BENCH_FAIL_RANK": "1", "BENCH_TEST_HOOKS": "1"}, "--gpus", "2")
    assert p.returncode != 0   # (rank 0 may still have printed its line: the status is what tells the driver)


def test_a_stray_hook_variable_without_the_switch_changes_nothing():
    p = run_bench({"BENCH_FAIL_RANK": "1"}, "--gpus", "2")          # no BENCH_TEST_HOOKS=1 beside it: read as unset
    assert p.returncode == 0, p.stderr[-2000:]


def test_single_gpu_invocation_stays_in_process():
    p = run_bench({}, "--gpus", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["launched_by_parent"] is False


def test_the_parent_imports_no_gpu_runtime_before_spawning():
    # the launcher path must not import torch (or anything that could initialise HIP) in the parent process
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2']\n"
            "import bench\n"
            "bench.launch_children = lambda a, v: (print('torch' in sys.modules), 0)[1]\n"
            "sys.exit(bench.main(['--gpus', '2']))\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "False", (p.stdout, p.stderr[-1000:])
