#!/bin/bash
# EngineOptions.batch_reduce on / off (AVSIAM_BATCH_REDUCE), one box, alternating: the headline shape and the one-frame shape
OUT=gpurun_out/ab_batch_reduce.txt; : > $OUT
for rep in 1 2 3; do
  for f in 0 1; do
    AVSIAM_BATCH_REDUCE=$f python bench.py --secondary-steps 0 --no-cpu-baseline --steps 12 --warmup 3 --roofline-steps 0 --no-kernel-events 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('T=10 batch_reduce=$f rep $rep', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms')" >> $OUT
    tail -1 $OUT
    AVSIAM_BATCH_REDUCE=$f python bench.py --frames 1 --secondary-steps 0 --no-cpu-baseline --steps 30 --warmup 5 --roofline-steps 0 --no-kernel-events 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('T=1  batch_reduce=$f rep $rep', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms')" >> $OUT
    tail -1 $OUT
  done
done
