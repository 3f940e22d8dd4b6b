#!/bin/bash
# usage (inside gpurun): tools/final_records.sh <tag> [pmc]   e.g.  tools/final_records.sh r05   then   tools/final_records.sh r05 pmc
# The end-of-round evidence on the shipped library, one box per call (a gpurun call is limited to 20 minutes):
#   without "pmc": the bench line (with the CPU baseline), rocprofv3 kernel-trace stats of the same command, every configuration's
#                  launch time (tools/all_configs.sh), a five-minute soak (tools/soak.py)
#   with "pmc":    counters for every configuration (tools/pmc_all.sh <tag>)
# Afterwards, in the build container: python3 tools/pmc_merge.py --tag <tag>; copy gpurun_out/<tag>/{bench_default.json,
# kernel_stats_bench_default.csv, all_configs.txt, soak.txt} to profiles/<tag>_*.
tag=${1:?tag}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
if [ "$2" = pmc ]; then tools/pmc_all.sh $tag; exit; fi
python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err; echo "bench rc=$?"; tail -c 600 gpurun_out/$tag/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace_bench -- python3 bench.py --no-cpu-baseline > gpurun_out/$tag/trace_bench.log 2>&1; echo "trace rc=$?"
find gpurun_out/$tag/trace_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats_bench_default.csv \;
rm -rf gpurun_out/$tag/trace_bench
tools/all_configs.sh > gpurun_out/$tag/all_configs.txt 2>&1; cat gpurun_out/$tag/all_configs.txt
timeout -k 10 420 python3 tools/soak.py --minutes 5 > gpurun_out/$tag/soak.txt 2>&1; echo "soak rc=$?"; tail -n 3 gpurun_out/$tag/soak.txt
