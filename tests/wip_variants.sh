#!/bin/bash
# runs wip_phases.py once per library variant; stops at the first failure
set -e
for lib in "$@"; do
  PMG_LIB=$PWD/paccmann_generator_amd/lib/$lib timeout -k 10 120 python tests/wip_phases.py
  if [ -n "$PROF" ]; then PMG_LIB=$PWD/paccmann_generator_amd/lib/$lib WHICH=$PROF timeout -k 10 100 python tests/wip_prof.py | grep "wave [048] "; fi
done
