#!/bin/bash
# ViT-H/14 fp8 mode 3 (batch 64 x 10 frames, one activation pool): fused backward for the <= 64-token sequences at head dim 80 on / off, one box, alternating
OUT=gpurun_out/ab_h_fused80.txt; : > $OUT
for rep in 1 2; do
  for f in 0 1; do
    AVSIAM_ATTN_FUSED=$f python bench.py --secondary-steps 0 --no-cpu-baseline --steps 4 --warmup 2 --model vit_huge14 --recompute auto --share-pass-buffers --fp8 --fp8-wgrad --roofline-steps 0 --no-kernel-events 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('attn_fused=$f rep $rep', round(d['value'],2), 'samples/s', round(d['ms_per_step'],1), 'ms')" >> $OUT
    tail -1 $OUT
  done
done
