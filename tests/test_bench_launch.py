"""bench.py's launch contract on a box without GPUs: `--gpus N` either runs N ranks or fails loudly, never one rank quietly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_world_size_mismatch_is_refused():
    """a launcher that started another number of ranks than --gpus asks for: exit, do not measure the wrong job"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, cwd=ROOT, env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=600)
    assert out.returncode != 0
    assert "--gpus 8 but WORLD_SIZE=1" in out.stderr + out.stdout
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_gpus_flag_without_launcher_spawns_the_ranks():
    """no launcher environment: bench.py re-launches itself under torch.distributed.run with --nproc-per-node N (here the
    ranks then fail for want of a GPU -- the point is that N of them were started and that no JSON line came out)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, cwd=ROOT, env=_clean_env(), timeout=900)
    text = out.stderr + out.stdout
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert out.returncode == 0
        return
    assert out.returncode != 0
    assert "local_rank: 0" in text or "rank      : 0" in text or "rank 0" in text.lower()      # torch.distributed.run's failure report
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
