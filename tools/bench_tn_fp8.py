"""fp8 weight-gradient kernel (avs_gemm_tn_fp8_group3) against the bf16 8-phase kernel at the step's shapes: a block's fc2 | fc1 | proj
gradients in one launch, and the qkv gradient alone (us per launch, TFLOP/s).  python tools/bench_tn_fp8.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from avsiam_amd import ops as o
dev = "cuda"
def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
recs = o.Fp8Records(2, dev); recs.q[:, 0] = 1.0; recs.q[:, 1] = 1.0
for name, M, D, Hd in (("P1", 95630, 768, 3072), ("towers", 39552, 768, 3072), ("decoder", 158208, 512, 2048)):
    rp = o.pad_rows(M, 256)
    mk8 = lambda N: torch.randint(0, 120, (rp, N), device=dev, dtype=torch.uint8)
    mkb = lambda N: torch.randn(rp, N, device=dev).to(torch.bfloat16)
    shapes = [(D, Hd), (Hd, D), (D, D)]
    j8 = [(mk8(a), mk8(b), torch.zeros(a * b, device=dev), recs.rec(0), recs.rec(1)) for a, b in shapes]
    jb = [(mkb(a), mkb(b), torch.zeros(a * b, device=dev)) for a, b in shapes]
    q8 = [(mk8(3 * D), mk8(D), torch.zeros(3 * D * D, device=dev), recs.rec(0), recs.rec(1))]
    qb = (mkb(3 * D), mkb(D), torch.zeros(3 * D * D, device=dev))
    fl3 = sum(2.0 * M * a * b for a, b in shapes); flq = 2.0 * M * 3 * D * D
    u8 = t(lambda: o.gemm_tn_fp8_group(j8, M)); ub = t(lambda: o.gemm_tn_group(jb, M))
    v8 = t(lambda: o.gemm_tn_fp8_group(q8, M)); vb = t(lambda: o.gemm_tn(qb[0], qb[1], qb[2], M))
    print(f"{name:8s} M={M:6d}: group3 fp8 {u8:7.1f} us ({fl3/u8/1e6:6.0f} TF/s)  bf16 {ub:7.1f} us ({fl3/ub/1e6:6.0f} TF/s)   qkv fp8 {v8:7.1f} us ({flq/v8/1e6:6.0f})  bf16 {vb:7.1f} us ({flq/vb/1e6:6.0f})", flush=True)
