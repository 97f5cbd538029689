"""`python bench.py --gpus N` starts its own ranks (reference: train_classification.py:8152-8169
mp.spawn(train, nprocs=world_size)).  CPU rehearsal of that launch path with gloo and bench.py's stub step (BENCH_STUB=1):
rendezvous on 127.0.0.1, barrier + max-over-ranks timing, rank 0's single JSON line relayed by the parent, and a rank that
dies must end the job with a non-zero status instead of leaving the others waiting in a collective."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_STUB="1", **extra_env)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout)
    return p, time.monotonic() - t0


def test_self_launch_two_ranks_relays_rank0_line():
    p, _ = _run({}, "--gpus", "2", "--steps", "5", "--warmup", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["warmup"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["allreduce_check"] == 3.0  # (rank 0: 1) + (rank 1: 2): both ranks were in the collective
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    # the self-verifying keys of an N > 1 line (main_pretrain.py:201-214 prints the same facts at start-up):
    # built by bench.dist_config(), the function the GPU path calls, from torch.distributed itself and an all_gather_object
    c = rec["config"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["ranks_seen"] == 2
    assert c["local_ranks"] == "0,1" and c["distinct_local_devices"] == 2 and c["devices"] == "0:cpu; 1:cpu"
    assert c["sync_collectives_per_step"] == 1 and c["sync_allreduce_bytes_per_step"] == 4 * 4096
    assert c["sync_exposed_wait_ms"] >= 0 and c["sync_trainable_bytes"] == 4 * 4096


def test_single_rank_needs_no_launcher():
    p, _ = _run({}, "--gpus", "1", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_library_banners_on_fd1_do_not_reach_stdout():
    """RCCL prints a version banner on file descriptor 1 when its communicator comes up (seen on the GPU box under --force-sync);
    bench.py hands fd 1 to stderr at start-up and writes its one JSON line to the saved descriptor: stdout stays ONE line."""
    p, _ = _run({"BENCH_STUB_NOISE": "1"}, "--gpus", "1", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1, p.stdout
    assert "stub banner on fd 1" in p.stderr


def test_a_dead_rank_fails_the_job_and_frees_the_others():
    # rank 1 exits before the rendezvous; rank 0 would wait in init_process_group for its peer (default timeout: minutes)
    p, dt = _run({"BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1", timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert "rank 1 exited with status 3" in p.stderr
    assert p.stdout.strip() == ""  # no line from a failed job
    assert dt < 60, dt


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", BENCH_STUB="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
