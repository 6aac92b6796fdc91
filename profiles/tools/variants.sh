#!/bin/bash
# runs phases.py once per library variant; stops at the first failure
set -e
for lib in "$@"; do
  PMG_LIB=$PWD/paccmann_generator_amd/lib/$lib timeout -k 10 120 python profiles/tools/phases.py
  if [ -n "$PROF" ]; then PMG_LIB=$PWD/paccmann_generator_amd/lib/$lib WHICH=$PROF timeout -k 10 100 python profiles/tools/prof.py | grep "wave [048] "; fi
done
