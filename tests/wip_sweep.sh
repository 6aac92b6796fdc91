#!/bin/bash
# scratch: phase times over the batch size, clustered vs batch-partitioned kernels
out=$1; shift
for spec in "$@"; do
  cfg=${spec%%:*}; B=${spec##*:}
  for nc in "" "--no-cluster"; do
    timeout -k 10 120 python bench.py --config $cfg --batch $B --no-cpu-baseline --steps 3 --warmup 1 --allow-variant $nc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg B=$B $nc', 'step %.2f ms'%d['ms_per_step'], {k: round(v,3) for k,v in d['phase_ms'].items()}, d.get('forward_kernels'))
" >> $out || exit 1
  done
done
