#!/bin/bash
# usage: bash profiles/profile_all.sh <tag> [bench.py arguments, e.g. --config cfg4 --batch 4096]
# (run on the GPU box from the repo root; the program comes directly after `--`, one pass per counter set)
set -e
R=$GRAFT_REPO_ROOT; T=$1; shift; A="$@"; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $A > $R/gpurun_out/${T}_kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $A > $R/gpurun_out/${T}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -o write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $A > $R/gpurun_out/${T}_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O -o mfma -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline $A > $R/gpurun_out/${T}_mfma.log 2>&1
find $O -name "*.csv" | head -20
