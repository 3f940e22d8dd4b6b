#!/bin/bash
# usage (inside gpurun): counter evidence for every BASELINE configuration on the current binary (one configuration after another)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for args in "--config 3" "--config 3 --complex" "--config 3 --exact-qp" "--config 2" "--config 4" "--config 5 --batch 131072"; do
  timeout -k 10 900 python3 tools/pmc_collect.py --tag r04 $args > gpurun_out/r04/pmc_collect_$(echo $args | tr -d ' -').log 2>&1; echo "pmc $args rc=$?"; tail -2 gpurun_out/r04/pmc_collect_$(echo $args | tr -d ' -').log
done
# the DPP sweeps on the headline configuration, for comparison with the tile sweep (M4Q_NO_TILE=1)
M4Q_NO_TILE=1 timeout -k 10 900 python3 tools/pmc_collect.py --tag r04 --config 3 > gpurun_out/r04/pmc_collect_config3_notile.log 2>&1; echo "pmc config 3 (DPP sweeps) rc=$?"
rm -rf gpurun_out/r04/pmc_tmp_* gpurun_out/r04/trace_tmp
