set -e
REPO=$PWD
OUT=$PWD/gpurun_out/attn_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 $REPO/tools/bench_attn.py --step --decoder-only > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/attn_pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'attn' in k:
        print(k, {c: f"{v:.3e}" for c, v in sorted(d.items())})
PY
