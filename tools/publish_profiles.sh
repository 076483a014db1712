#!/bin/bash
# Copies what `tools/collect_profiles.sh <round>` left under gpurun_out/prof_<round> into profiles/ under their tracked names.
set -e
cd "$(dirname "$0")/.."
R=${1:-r03}; O=gpurun_out/prof_$R; P=profiles
cp $O/bench.json $P/${R}_bench.json
cp $O/bench_noflags.json $P/${R}_bench_no_flags.json
cp $O/bench_wall.txt $P/${R}_bench_wall_time.txt
cp $O/stats_default/run_kernel_stats.csv $P/${R}_kernel_stats_default_bench.csv
cp $O/bench_under_rocprof_default.json $P/${R}_bench_under_rocprof_default.json
cp $O/stats_serial/run_kernel_stats.csv $P/${R}_kernel_stats_serial_streams.csv
cp $O/bench_under_rocprof_serial.json $P/${R}_bench_under_rocprof_serial_streams.json
for n in 64 1024; do [ -s $O/tipp_timeline_bn254_$n.txt ] && cp $O/tipp_timeline_bn254_$n.txt $P/${R}_tipp_prove_timeline_bn254_$n.txt; done
[ -s $O/setup_timeline_bn254_1024.txt ] && cp $O/setup_timeline_bn254_1024.txt $P/${R}_tipa_setup_timeline_bn254_1024.txt
[ -s $O/front_timeline_bn254_1024.txt ] && cp $O/front_timeline_bn254_1024.txt $P/${R}_agg_front_timeline_bn254_1024.txt
for a in bn254 bls12_381; do [ -s $O/agg_ops_$a.txt ] && cp $O/agg_ops_$a.txt $P/${R}_agg_ops_$a.txt; done
for c in big-merkle-512x64 vm-1024x1024 vkd-256 big-merkle-4x1 big-merkle-sha-64x32 big-merkle-sha-64x32_witness_gen; do
  [ -s $O/bench_$c.json ] && cp $O/bench_$c.json $P/${R}_bench_$c.json
done
python3 tools/pmc_traffic.py $O/pmc_fetch/run_counter_collection.csv $O/pmc_write/run_counter_collection.csv $P/${R}_pmc_accum0.json > /dev/null
python3 - <<PY
import json
d = json.load(open("$P/${R}_pmc_accum0.json"))
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh, tools/pmc_traffic.py); per-launch averages of k_msm_accum0<Fp<...FqP>>, raw counter values: for this pattern the counter reads 1.13 x the REQUESTED bytes on the H launch (2.57 GB against (m-1) x 16 rows x 64 B + entries = 2.28 GB), so the x2 correction of wide coalesced streams does not apply",
       "big-merkle-64x32/bn254": {"FETCH_SIZE_KiB_avg": d["FETCH_SIZE_KiB_avg"], "WRITE_SIZE_KiB_avg": d["WRITE_SIZE_KiB_avg"],
                                  "collected": "profiles/${R}_pmc_accum0.json"},
       "_collected": "round ${R}, build of commit $(git rev-parse --short HEAD)"}
json.dump(out, open("$P/pmc_accum0.json", "w"), indent=1)
print("traffic per launch: %.1f MB" % ((d["FETCH_SIZE_KiB_avg"] + d["WRITE_SIZE_KiB_avg"]) * 1024 / 1e6))
PY
