#!/bin/bash
# usage (inside gpurun): tools/sustained_ab.sh "<bench args>" STEPS NAME1 NAME2 ...   -- like tools/abn.sh, but every run keeps the GPU
# busy for STEPS launches back to back (20 s and more), so that the clock the chip settles at under the load is part of the comparison.
args=$1; steps=$2; shift 2
for r in 1 2; do
  for v in "$@"; do
    if [ "$v" = product ]; then lib=mpc4quantum_amd/libm4q_hip.so; else lib=tools/bin/lib$v.so; fi
    M4Q_LIB=$lib timeout -k 10 600 python bench.py --steps $steps --warmup 20 --no-cpu-baseline $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['avg_launch_ms'],3), 'ms', '%.4g' % d['value'])" || echo "$v FAILED"
  done
done
