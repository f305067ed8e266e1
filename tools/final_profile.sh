#!/bin/bash
# Round artefacts for the current build: default bench line, the one-frame line, and the rocprofv3 kernel-trace summary of
# the same bench command.  usage (on the GPU box, from the repo root): bash tools/final_profile.sh <tag>
set -e
TAG=${1:-final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --frames 1 --no-cpu-baseline > $OUT/${TAG}_bench_T1.json 2>> $OUT/${TAG}_bench.err
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o k -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_rocprof.json 2> $OUT/${TAG}_rocprof.err
cd $REPO
find $OUT/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -delete
head -c 600 $OUT/${TAG}_bench.json; echo; head -c 300 $OUT/${TAG}_bench_T1.json; echo; head -c 300 $OUT/${TAG}_bench_rocprof.json; echo; head -8 $OUT/${TAG}_kernel_stats.csv
