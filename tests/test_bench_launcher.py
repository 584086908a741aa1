"""`python bench.py --gpus N` must START N ranks: without a torchrun environment the command re-launches itself
through torch.distributed.run, one process per GPU, and rank 0 prints the one JSON line with n_gpus = N. Here on CPU with the
launcher's self-test workload (gloo rendezvous, closed-form records gathered through torch.distributed and through the C ABI's
gather over TCP): the process tree, the rendezvous, the gather order and the line are what is under test — no circuit work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_gathers():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "0", "--launcher-self-test")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["records_gathered"] == 2 * 5 and out["records_match"] is True
    assert "SELF-TEST" in out["metric"] and out["value"] is None  # can never be read as a measurement
    assert "starting 2 ranks" in r.stderr
    # the fields VERDICT r5 item 5 asks of the line: the cold-slot figure next to `value`, the byte bases, and a block leg that leads with the
    # sharded blocks (round-robin owners, ranks in the collective)
    assert "value_cold_slots" in out and "bytes_basis" in out["roofline"] and "bytes_basis" in out["synthesis"]
    b = out["full_block"]["batched"]
    assert b["n_gpus"] == 2 and "zkw_blocks_run_sharded" in b["sharding"] and "rccl_ranks" in b and "per_rank_blocks_per_s" in b
    assert b["block_owners"] == [0, 1, 0, 1, 0] and b["record_words"] == 73
    assert "does not scale" in out["full_block"]["scaling_note"]


def test_refuses_more_ranks_than_gpus():
    """no GPU in this container: the real workload at --gpus 2 must fail loudly instead of running one rank"""
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    """a torchrun environment that disagrees with --gpus is an error, not a silent one-rank run"""
    r = _run("--gpus", "2", "--steps", "1", "--launcher-self-test", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr
