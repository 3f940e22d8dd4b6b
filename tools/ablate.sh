#!/bin/bash
# timing-only ablation of the fused kernel's phases (inside gpurun): rebuilds the library per variant
for e in 1 3 5 9 17 33 7; do
  python mpc4quantum_amd/csrc/build.py --define M4Q_EXP=$e > /dev/null 2>&1
  r=$(timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms' % d['ms_per_step'], d['config']['qp_solves_per_step'])")
  c=$(M4Q_FORCE_COMPLEX=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms' % d['ms_per_step'])")
  echo "EXP=$e real $r | complex $c"
done
