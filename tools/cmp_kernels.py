import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
print("A", a["value"], a["ms_per_step"]); print("B", b["value"], b["ms_per_step"])
ks=sorted(set(a["kernels"])|set(b["kernels"]))
n=a["steps"]
for k in ks:
    x=a["kernels"].get(k,{}); y=b["kernels"].get(k,{})
    print("%-18s A %5d %8.2f ms/step %8.1f us | B %5d %8.2f ms/step %8.1f us" % (k, x.get("launches",0), x.get("total_ms",0)/n, x.get("avg_us",0), y.get("launches",0), y.get("total_ms",0)/n, y.get("avg_us",0)))
