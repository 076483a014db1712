set -o pipefail
true
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --config big-merkle-sha-64x32"
timeout -k 10 500 $B > gpurun_out/r02_sha_bench.json 2> gpurun_out/r02_sha_bench.err || { tail -20 gpurun_out/r02_sha_bench.err; exit 1; }
timeout -k 10 500 $B --witness-gen > gpurun_out/r02_sha_bench_wg.json 2> gpurun_out/r02_sha_bench_wg.err || { tail -20 gpurun_out/r02_sha_bench_wg.err; exit 1; }
for f in sha_bench sha_bench_wg; do python - <<PY
import json
d=json.loads(open("gpurun_out/r02_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],2), "proofs/s; n_c", d["config"]["n_constraints"], "domain", d["config"]["domain"], "classes", d["config"]["pk_classes_rank0"], d["timed_proofs_check"])
PY
done
grep "host setup\|resident" gpurun_out/r02_sha_bench.err | tail -4
