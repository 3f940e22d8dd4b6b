#!/bin/bash
# usage (inside gpurun): tools/pmc_pass.sh <tag> <extra bench args...>  -- PMC passes (own runs, no tracing) over one launch
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VALU_FMA_F64" \
             "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print("%-28s %.6g" % (k, tot[k]))
PY
