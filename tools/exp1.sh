set -o pipefail
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gputests2.txt 2>&1; echo tests rc=$?; tail -3 gpurun_out/r02_gputests2.txt
B="python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 300 $B > gpurun_out/r02_e_default.json 2> gpurun_out/r02_e_default.err && echo default ok
timeout -k 10 300 $B --single-class > gpurun_out/r02_e_single.json 2> gpurun_out/r02_e_single.err && echo single ok
HK_LIB=$PWD/hekaton_system_amd/lib/libhekaton_w5.so timeout -k 10 300 $B > gpurun_out/r02_e_w5.json 2> gpurun_out/r02_e_w5.err && echo w5 ok
HK_LIB=$PWD/hekaton_system_amd/lib/libhekaton_w5.so timeout -k 10 300 $B --single-class > gpurun_out/r02_e_w5_single.json 2> gpurun_out/r02_e_w5_single.err && echo w5 single ok
for f in default single w5 w5_single; do python - <<PY
import json
d=json.loads(open("gpurun_out/r02_e_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],2), "proofs/s; H accum ms", round(d["roofline"]["h_query_launch"]["avg_ms"],3), "avg accum ms", round(d["roofline"]["avg_launch_ms"],3))
PY
done
