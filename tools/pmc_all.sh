#!/bin/bash
# usage (inside gpurun):  tools/pmc_all.sh <tag> [lib]     e.g. tools/pmc_all.sh r05
# Counter evidence for every BASELINE configuration on the current binary (or the variant library given), one configuration after
# another on ONE box: tools/pmc_collect.py per configuration (eight separate --pmc passes + one --kernel-trace --stats run each,
# never combined), records under gpurun_out/<tag>/; afterwards, in the build container:  python3 tools/pmc_merge.py --tag <tag>
# (-> profiles/<tag>_pmc.json, <tag>_pmc_<key>.txt, <tag>_kernel_stats_<key>.csv).  Replaces the per-round r02 / r03 / r04 copies.
tag=${1:?tag}; [ -n "$2" ] && export M4Q_LIB=$2
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for args in "--config 3" "--config 3 --complex" "--config 3 --exact-qp" "--config 2" "--config 4" "--config 5 --batch 131072"; do
  log=gpurun_out/$tag/pmc_collect_$(echo $args | tr -d ' -').log
  timeout -k 10 900 python3 tools/pmc_collect.py --tag $tag $args > $log 2>&1; echo "pmc $args rc=$?"; tail -n 2 $log
done
# the DPP sweeps on the headline configuration, beside the tile sweep (M4Q_NO_TILE=1)
M4Q_NO_TILE=1 timeout -k 10 900 python3 tools/pmc_collect.py --tag $tag --config 3 > gpurun_out/$tag/pmc_collect_config3_notile.log 2>&1; echo "pmc config 3 (DPP sweeps) rc=$?"
# config 4 on per-member models, beside its shared-generator kernel (M4Q_NO_SG=1)
M4Q_NO_SG=1 timeout -k 10 900 python3 tools/pmc_collect.py --tag $tag --config 4 > gpurun_out/$tag/pmc_collect_config4_nosg.log 2>&1; echo "pmc config 4 (per-member models) rc=$?"
rm -rf gpurun_out/$tag/pmc_tmp_* gpurun_out/$tag/trace_tmp
