"""bench.py's launch contract (CPU side): `--gpus N` must be honoured or refused loudly, never ignored."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_mismatch_is_refused():
    """Launched under a 2-rank launcher with --gpus 1 (or the reverse): exit non-zero, say why."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "--gpus 1 but WORLD_SIZE=2" in p.stderr


def test_gpus_n_without_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher starts two rank processes itself (before touching HIP).  Without a
    GPU both ranks fail loudly and the launcher reports the failure - there is no CPU path to fall back to."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "0", "--steps", "1",
                        "--warmup", "0"], env=_env(HK_BENCH_ECHO_RANK="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""                              # no JSON line from a failed run
    assert "rank 0 of 2" in p.stderr and "rank 1 of 2" in p.stderr
