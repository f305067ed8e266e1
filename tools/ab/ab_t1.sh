run() { # name env...
  name=$1; shift
  env "$@" python bench.py --frames 1 --steps 30 --warmup 5 --no-cpu-baseline --secondary-steps 0 --roofline-steps 0 --no-kernel-events > gpurun_out/t1_$name.json 2> gpurun_out/t1_$name.err || { echo FAIL $name; tail -3 gpurun_out/t1_$name.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/t1_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"
}
for i in 1 2; do
run base_$i A=1
run ws1_$i AVSIAM_WGRAD_STREAM=1
run ws0_$i AVSIAM_WGRAD_STREAM=0
run tile64_$i AVSIAM_ATTN_TILE=64
run tile128_$i AVSIAM_ATTN_TILE=128
done
