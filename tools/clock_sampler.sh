#!/bin/bash
# Samples the GPU's shader clock and board power (rocm-smi, 2 Hz) while the default bench runs; writes
# gpurun_out/clock_samples.txt.  The bench process is a child of this script (no exec hop after HIP init).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/clock_samples.txt
: > $OUT
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 12 --warmup 2 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
t0=$(date +%s%N)
while kill -0 $BP 2>/dev/null; do
  t=$(( ($(date +%s%N) - t0) / 1000000 ))ms
  line=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power" | tr -s ' ' | tr '\n' ';')
  echo "$t $line" >> $OUT
  sleep 0.5
done
wait $BP; echo "bench rc=$?"
python - <<'PY'
import re
rows=[l for l in open("gpurun_out/clock_samples.txt")]
print(len(rows), "samples")
for l in rows[::6][:40]:
    print(l.strip()[:220])
PY
