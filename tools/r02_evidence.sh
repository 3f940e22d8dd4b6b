#!/bin/bash
# usage (inside gpurun): tools/r02_evidence.sh   -- tests, bench, counter evidence for every BASELINE config, calibration
mkdir -p gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/gpu_tests.log; tail -4 gpurun_out/r02/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; tail -c 600 gpurun_out/r02/bench_default.json; echo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist --no-cpu-baseline > gpurun_out/r02/bench_forcedist_nccl.json 2> gpurun_out/r02/bench_forcedist_nccl.err; echo "force-dist nccl rc=$?"; tail -c 300 gpurun_out/r02/bench_forcedist_nccl.json; echo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --batch 8192 --no-cpu-baseline > gpurun_out/r02/bench_gloo2.json 2> gpurun_out/r02/bench_gloo2.err; echo "gloo x2 rc=$?"; tail -c 300 gpurun_out/r02/bench_gloo2.json; echo
for args in "--config 3" "--config 3 --complex" "--config 3 --exact-qp" "--config 2" "--config 4" "--config 5 --batch 131072"; do
  timeout -k 10 900 python3 tools/pmc_collect.py --tag r02 $args > gpurun_out/r02/pmc_collect_$(echo $args | tr -d ' -').log 2>&1; echo "pmc $args rc=$?"; tail -3 gpurun_out/r02/pmc_collect_$(echo $args | tr -d ' -').log
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r02/fetchcal_$c -- ./tools/bin/ubench_fetch > gpurun_out/r02/fetchcal_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob
print("# tools/ubench_fetch.hip under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB as printed); 2 GiB streamed per kernel")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/r02/fetchcal_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("rd" in r["Kernel_Name"] or "wr" in r["Kernel_Name"]):
                kb = float(r["Counter_Value"])
                print("%-12s %-60s %14.0f KB   = %.3f of 2 GiB" % (c, r["Kernel_Name"][:60], kb, kb * 1024 / (2 << 30)))
PY
