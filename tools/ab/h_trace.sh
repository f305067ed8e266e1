#!/bin/bash
# kernel trace of the ViT-H/14 fp8 mode-3 step (batch 64 x 10 frames, one activation pool), one stream:  bash tools/ab/h_trace.sh
set -e
OUT=$PWD/gpurun_out/r06; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
AVSIAM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_h -o k -- python3 $REPO/bench.py --secondary-steps 0 --no-cpu-baseline --steps 3 --warmup 2 --model vit_huge14 --recompute auto --share-pass-buffers --fp8 --fp8-wgrad --roofline-steps 0 --no-kernel-events > $OUT/h_s0_bench_rocprof.json 2> $OUT/h_rocprof.err
cd $REPO
find $OUT/prof_h -name "*kernel_stats.csv" -exec cp {} $OUT/h_kernel_stats.csv \;
rm -rf $OUT/prof_h
head -30 $OUT/h_kernel_stats.csv
