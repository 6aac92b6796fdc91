#!/bin/bash
# scratch: clustered phase times over the batch size for library variants (PMG_LIB)
out=$1; shift
for lib in libpmg_pref0.so libpmg_pref1.so; do
for spec in "$@"; do
  cfg=${spec%%:*}; B=${spec##*:}
  PMG_LIB=$PWD/paccmann_generator_amd/lib/$lib timeout -k 10 120 python bench.py --config $cfg --batch $B --no-cpu-baseline --steps 3 --warmup 1 --allow-variant 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg B=$B', 'step %.2f ms'%d['ms_per_step'], {k: round(v,3) for k,v in d['phase_ms'].items()}, d.get('forward_kernels'))
" >> $out || exit 1
done
done
