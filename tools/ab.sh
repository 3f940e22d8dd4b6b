#!/bin/bash
# usage (inside gpurun): tools/ab.sh [bench args]  -- alternates tools/bin/libA.so and libB.so in the scratch copy, 3 rounds
for r in 1 2 3; do
  for v in A B; do
    cp tools/bin/lib$v.so mpc4quantum_amd/libm4q_hip.so
    timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['avg_launch_ms'],3))"
  done
done
