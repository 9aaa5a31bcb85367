"""`python bench.py --gpus N` as typed (no launcher): the parent starts the ranks itself, never touches a GPU, passes
rank 0's single JSON line through and returns the children's exit status.  Checked here on CPU with `--dry-run`
(gloo process group, barrier, all-gather of per-rank rows, max-over-ranks reduction; no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_self_launch_two_ranks_prints_one_json_line():
    r = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["gathered_rows"] == 6 and rec["gather_ok"] is True


def test_single_rank_needs_no_launcher_and_a_failing_child_fails_the_parent():
    r = run_bench("--dry-run")
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # a rank that dies makes the parent exit non-zero: without a GPU the real (non dry-run) ranks refuse to start
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline")
    assert r.returncode != 0
