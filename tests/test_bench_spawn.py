"""bench.py's multi-rank contract on CPU: `python bench.py --gpus 2` with NO launcher around it must start two ranks
itself (re-exec under torch.distributed.run), time exactly K steps between barriers, take the max over ranks, gather
to rank 0, and print ONE JSON line whose n_gpus == --gpus == ranks_seen.  The compute is stubbed
(MBHIP_BENCH_STUB=1, gloo): this checks the plumbing, not a number.  A WORLD_SIZE that disagrees with --gpus is an
error instead of a silent single-rank line (VERDICT r01 missing #3)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MBHIP_BENCH_STUB="1", OMP_NUM_THREADS="1", **kw)
    return env


def test_gpus_2_spawns_two_ranks_and_reports_them():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["data"] == "stub" and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["utterances"] == 2
    # whole-job aggregate: both ranks' samples over the max-over-ranks time
    assert abs(r["value"] - 2 * 3 * 4096 / (r["ms_per_step"] * 3 / 1000.0)) / r["value"] < 1e-6


def test_single_rank_needs_no_launcher_and_world_size_mismatch_is_fatal():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["ranks_seen"] == 1
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert q.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in q.stderr


def test_gpus_8_dry_run_maps_every_rank_to_its_own_device():
    """The driver's SCALE run, dry: `bench.py --gpus 8` (self-spawned, gloo, stubbed compute) -- eight ranks, one JSON line, whole-job
    value, and the per-rank compute / gather split with the device each rank selected (LOCAL_RANK r -> device r; on the GPU box
    torch.cuda.set_device(LOCAL_RANK) + the nccl (= RCCL) process group bound to that device).  VERDICT r03 item 9."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["ranks_seen"] == 8 and r["scaling"] == "weak" and r["config"]["utterances"] == 8
    pr = r["per_rank_ms"]
    assert len(pr["compute"]) == len(pr["gather"]) == 8 and all(c > 0 for c in pr["compute"]) and all(g >= 0 for g in pr["gather"])
    assert pr["device"] == list(range(8))
    assert abs(r["value"] - 8 * 2 * 4096 / (r["ms_per_step"] * 2 / 1000.0)) / r["value"] < 1e-6
