#!/bin/bash
# usage (inside gpurun): tools/abn.sh "<bench args>" NAME1 NAME2 ...   -- alternates tools/bin/lib<NAME>.so (NAME=product: the
# in-tree library) three rounds; prints the HIP-event launch time of each run.  Same box, same minute: box-to-box spread is 1.5 %.
args=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = product ]; then lib=mpc4quantum_amd/libm4q_hip.so; else lib=tools/bin/lib$v.so; fi
    M4Q_LIB=$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['avg_launch_ms'],3), 'ms', '%.4g' % d['value'])" || echo "$v FAILED"
  done
done
