#!/bin/bash
# Round artefacts for the current build (GPU box, from the repo root):  bash tools/final_profile.sh <tag>
#   gpurun_out/<tag>_bench.json              default bench line (timed region: weight gradients on the second stream)
#   gpurun_out/<tag>_bench_T1.json           the one-frame (reference pre-training semantics) line
#   gpurun_out/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the same bench command
#   gpurun_out/<tag>_s0_kernel_stats.csv     the same with AVSIAM_WGRAD_STREAM=0: everything on ONE stream, so every kernel's
#                                            duration is its own (no waiting for CUs a concurrent kernel holds)
set -e
TAG=${1:-final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --frames 1 --no-cpu-baseline > $OUT/${TAG}_bench_T1.json 2>> $OUT/${TAG}_bench.err
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o k -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $OUT/${TAG}_bench_rocprof.json 2> $OUT/${TAG}_rocprof.err
export AVSIAM_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_s0 -o k -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $OUT/${TAG}_s0_bench_rocprof.json 2> $OUT/${TAG}_s0_rocprof.err
unset AVSIAM_WGRAD_STREAM
cd $REPO
find $OUT/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
find $OUT/${TAG}_prof_s0 -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_s0_kernel_stats.csv \;
find $OUT/${TAG}_prof $OUT/${TAG}_prof_s0 -name "*kernel_trace.csv" -delete
head -c 400 $OUT/${TAG}_bench.json; echo; head -c 300 $OUT/${TAG}_bench_T1.json; echo; head -c 300 $OUT/${TAG}_s0_bench_rocprof.json; echo; head -8 $OUT/${TAG}_s0_kernel_stats.csv
