#!/bin/bash
# A round's measurement set (run on the GPU box from the repo root; `bash tools/collect_profiles.sh r03`): bench lines,
# rocprofv3 kernel summaries, PMC traffic.
# Every rocprofv3 run uses `-- python3 bench.py ...` directly (no env / shell hop) and keeps --pmc runs free of trace
# domains other than --kernel-trace.
set -o pipefail
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
R=${1:-r03}
PART=${2:-all}                      # a: bench lines + rocprofv3 / PMC passes;  b: other configs, SHA, aggregator;  all
O=gpurun_out/prof_$R; mkdir -p $O
B="--no-cpu-baseline --no-secondary --no-e2e --no-synthesis-leg"
if [ $PART != b ]; then
echo "== bench with no flags (wall time of the default invocation)"; t0=$(date +%s); timeout -k 10 600 python bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err || exit 1
echo "python bench.py (no flags) wall time: $(( $(date +%s) - t0 )) s" | tee $O/bench_wall.txt
echo "== bench (profile line: 6 timed steps)"; timeout -k 10 600 python bench.py --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err || exit 1
echo "== rocprof default bench"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o run -- python3 bench.py $B --steps 2 --warmup 1 > $O/bench_under_rocprof_default.json 2> $O/rocprof_default.err || exit 1
echo "== rocprof serial streams"
HK_SERIAL_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -o run -- python3 bench.py $B --single-class --steps 1 --warmup 1 --subcircuits 8 --threads 1 > $O/bench_under_rocprof_serial.json 2> $O/rocprof_serial.err || exit 1
echo "== pmc fetch"
HK_SERIAL_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py $B --single-class --steps 1 --warmup 0 --subcircuits 2 --threads 1 --no-verify > /dev/null 2> $O/pmc_fetch.err || exit 1
echo "== pmc write"
HK_SERIAL_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 bench.py $B --single-class --steps 1 --warmup 0 --subcircuits 2 --threads 1 --no-verify > /dev/null 2> $O/pmc_write.err || exit 1
fi
if [ $PART != a ]; then
echo "== other configs"
for c in big-merkle-512x64 vm-1024x1024 vkd-256 big-merkle-4x1; do
  timeout -k 10 500 python bench.py --no-cpu-baseline --no-secondary --no-synthesis-leg --config $c --steps 2 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err || echo "config $c failed"
done
echo "== real SHA-256 subcircuits"
timeout -k 10 500 python bench.py $B --config big-merkle-sha-64x32 --steps 3 --warmup 1 > $O/bench_big-merkle-sha-64x32.json 2> $O/bench_sha.err || echo "sha config failed"
timeout -k 10 500 python bench.py $B --config big-merkle-sha-64x32 --witness-gen --steps 3 --warmup 1 > $O/bench_big-merkle-sha-64x32_witness_gen.json 2> $O/bench_sha_wg.err || echo "sha witness-gen config failed"
echo "== aggregation primitives and the whole aggregator, both curves"
timeout -k 10 500 python tools/agg_ops_bench.py bn254 > $O/agg_ops_bn254.txt 2> $O/agg_ops_bn254.err || echo "agg ops bn254 failed"
timeout -k 10 500 python tools/agg_ops_bench.py bls12_381 > $O/agg_ops_bls12_381.txt 2> $O/agg_ops_bls12_381.err || echo "agg ops bls failed"
echo "== kernel timeline of one TIPP prove (64 and 1024 elements)"
for n in 64 1024; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tipp_tl_$n -o run -- python3 tools/tipp_timeline.py run bn254 $n > $O/tipp_tl_$n.txt 2>&1 \
    && python3 tools/tipp_timeline.py show $(find $O/tipp_tl_$n -name "run_kernel_trace.csv") > $O/tipp_timeline_bn254_$n.txt || echo "tipp timeline $n failed"
done
echo "== kernel timelines of TIPA setup and of the aggregator's front half (1024 proofs)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/setup_tl -o run -- python3 tools/tipp_timeline.py setup bn254 1024 > $O/setup_tl.txt 2>&1 \
  && python3 tools/tipp_timeline.py show $(find $O/setup_tl -name "run_kernel_trace.csv") > $O/setup_timeline_bn254_1024.txt || echo "setup timeline failed"
HK_AGG_FRONT_TRACE=1024 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/front_tl -o run -- python3 tools/agg_ops_bench.py bn254 > $O/front_tl.txt 2>&1 \
  && python3 tools/tipp_timeline.py show $(find $O/front_tl -name "run_kernel_trace.csv") > $O/front_timeline_bn254_1024.txt || echo "front timeline failed"
fi
find $O -name "*.csv" | head -30
echo done
