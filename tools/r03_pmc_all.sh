#!/bin/bash
# usage (inside gpurun): counter evidence for every BASELINE configuration on the current binary (one configuration after another)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for args in "--config 3" "--config 3 --complex" "--config 3 --exact-qp" "--config 2" "--config 4" "--config 5 --batch 131072"; do
  timeout -k 10 900 python3 tools/pmc_collect.py --tag r03 $args > gpurun_out/r03/pmc_collect_$(echo $args | tr -d ' -').log 2>&1; echo "pmc $args rc=$?"; tail -2 gpurun_out/r03/pmc_collect_$(echo $args | tr -d ' -').log
done
rm -rf gpurun_out/r03/pmc_tmp_* gpurun_out/r03/trace_tmp
