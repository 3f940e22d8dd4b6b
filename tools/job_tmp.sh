mkdir -p gpurun_out/r04
timeout -k 10 120 ./tools/bin/ubench_tile_chain > gpurun_out/r04/tile_chain3.txt 2>&1
python -m pytest tests/test_gpu_parity.py -q -x -k "exact" > gpurun_out/r04/t_exact.log 2>&1; echo "exact tests rc=$?"; tail -3 gpurun_out/r04/t_exact.log
tools/ab_bench.sh "--steps 5 --warmup 1 --exact-qp" 2 base= XC=tools/bin/libXC.so > gpurun_out/r04/ab_exact1.txt 2>&1; cat gpurun_out/r04/ab_exact1.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/r04/pmc_lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --exact-qp > gpurun_out/r04/pmc_lds.log 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/r04/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpc_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print("%-28s %.6g" % (k, tot[k]))
PY
cat gpurun_out/r04/tile_chain3.txt
