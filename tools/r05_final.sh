#!/bin/bash
# usage (inside gpurun): tools/r05_final.sh   -- the end-of-round evidence on the shipped library, one box, one call:
# bench line (with the CPU baseline), rocprofv3 kernel-trace stats of the same command, every configuration's launch time,
# counters for every configuration (tools/pmc_all.sh r05), a five-minute soak.
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r05/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/trace_bench -- python3 bench.py --no-cpu-baseline > gpurun_out/r05/trace_bench.log 2>&1; echo "trace rc=$?"
find gpurun_out/r05/trace_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/kernel_stats_bench_default.csv \;
rm -rf gpurun_out/r05/trace_bench
tools/all_configs.sh > gpurun_out/r05/all_configs.txt 2>&1; cat gpurun_out/r05/all_configs.txt
[ "$1" = pmc ] && tools/pmc_all.sh r05
[ "$1" = pmc ] || timeout -k 10 420 python3 tools/soak.py --minutes 5 > gpurun_out/r05/soak.txt 2>&1; echo "soak rc=$?"; tail -n 3 gpurun_out/r05/soak.txt
