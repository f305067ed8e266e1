run() { name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary-steps 0 --roofline-steps 0 --no-kernel-events > gpurun_out/qb_$name.json 2> gpurun_out/qb_$name.err || { echo FAIL $name; tail -3 gpurun_out/qb_$name.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/qb_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"
}
for i in 1 2; do
run fused_$i AVSIAM_QBIAS_FUSED=1
run colsum_$i AVSIAM_QBIAS_FUSED=0
done
