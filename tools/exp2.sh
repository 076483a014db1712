set -o pipefail
B="python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
for t in 8 12 16; do
  timeout -k 10 300 $B --threads $t > gpurun_out/r02_t$t.json 2> gpurun_out/r02_t$t.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r02_t$t.json").read().strip().splitlines()[-1])
print("threads $t:", round(d["value"],2), "proofs/s; total_ms/proof", round(d["phase_ms_per_proof"]["total_ms"],1), "finish", round(d["phase_ms_per_proof"]["finish_ms"],1))
PY
done
