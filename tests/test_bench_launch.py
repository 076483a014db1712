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


def test_plan_classes_for_eight_gpus_matches_the_sharding_design():
    """`bench.py --gpus 8` on big-merkle (64 subcircuits per GPU, 512 in the job): contiguous shards (node.rs:471-493) and
    the proving-key classes each rank has to hold, by the reference's index -> class map (tree_hash_circuit.rs:192-216):
    rank 0 {0, 1}, ranks 1-3 {1} (leaves), ranks 4-6 {n-3} (parents), rank 7 {n-3, n-2, n-1} - DESIGN.md section 5."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    n = 512
    args = types.SimpleNamespace(config="big-merkle-512x64", subcircuits=64)
    sets, total = [], 0
    for rank in range(8):
        n_total, shard, class_of = bench.plan_classes(args, rank, 8, False)
        assert n_total == n and shard == list(range(64 * rank, 64 * rank + 64))
        total += len(shard)
        sets.append(sorted(set(class_of.values())))
    assert total == n
    assert sets == [[0, 1], [1], [1], [1], [n - 3], [n - 3], [n - 3], [n - 3, n - 2, n - 1]]
    # a single GPU holds all five
    _n, _shard, class_of = bench.plan_classes(types.SimpleNamespace(config="big-merkle-64x32", subcircuits=64), 0, 1, False)
    assert sorted(set(class_of.values())) == [0, 1, 61, 62, 63]
