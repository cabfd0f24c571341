#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + separate PMC passes.
# Usage: bash tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}/...
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --alsz-log2n 0 --lut-log2b 0"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $ARGS --no-prof > $OUT/fetch.log 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $ARGS --no-prof > $OUT/write.log 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES -d $OUT/sq -o sq -- python3 $ARGS --no-prof > $OUT/sq.log 2>&1 || exit 4
ls -R $OUT | head -50
