#!/bin/bash
# A/B of library variants on ONE box, alternated:  tools/ab_bench.sh "<bench.py args>" <rounds> <name>=<lib or ''> [ENV=VAL,...] ...
# e.g. tools/ab_bench.sh "--steps 10 --warmup 2" 2 base= tile=:M4Q_TILE=1 TM=tools/bin/libTM.so:M4Q_TILE=1
args=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    name=${v%%=*}; rest=${v#*=}; lib=${rest%%:*}; envs=""
    [[ "$rest" == *:* ]] && envs=${rest#*:}
    out=$( (export ${envs//,/ } >/dev/null 2>&1; [ -n "$lib" ] && export M4Q_LIB=$lib; python bench.py $args --no-cpu-baseline 2>/dev/null) | tail -1)
    echo "$name $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%.3f ms/launch  %.3f ms/step  %s' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], d['config']['workload'].split(';')[-1]))" "$out" 2>/dev/null || echo FAILED)"
  done
done
