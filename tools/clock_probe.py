#!/usr/bin/env python3
"""Average shader clock while the step's GEMMs run: a one-wave probe kernel on a second stream reads the shader-clock counter
(s_memtime) against the 100 MHz reference (s_memrealtime) over a window in which the main stream runs GEMMs back to back
(persistent grid capped at 255 CUs so that the probe has somewhere to live).
    python tools/clock_probe.py [--power] [--all]     (--power: rocm-smi socket power beside each workload; --all: + attention, LayerNorm)
    AVSIAM_NT_GRID=128 python tools/clock_probe.py    (the persistent forward/dgrad GEMM on half of the CUs)
Builds tools/csrc/clock_probe.hip with hipcc on first use."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from avsiam_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def build():
    src, so = os.path.join(HERE, "csrc", "clock_probe.hip"), os.path.join(HERE, "csrc", "clock_probe.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_void_p]
    return lib


def smi_sampler(stop, samples):
    """rocm-smi's socket power / sclk of the visible GPU, sampled until `stop` is set (each call takes a few 100 ms)"""
    import re
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        except (OSError, subprocess.TimeoutExpired):
            return
        w = re.search(r"Power \(W\): ([0-9.]+)", txt)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
        samples.append((float(w.group(1)) if w else float("nan"), int(c.group(1)) if c else -1))


def main():
    power = "--power" in sys.argv
    lib = build()
    dev = "cuda"
    M, D, Hd = 95630, 768, 3072
    Mp = ops.pad_rows(M, 256)
    x = (torch.randn(Mp, D, device=dev) * 0.5).to(BF16)
    xh = (torch.randn(Mp, Hd, device=dev) * 0.5).to(BF16)
    w1 = (torch.randn(Hd, D, device=dev) * 0.03).to(BF16)
    w2 = (torch.randn(D, Hd, device=dev) * 0.03).to(BF16)
    oh = torch.empty(Mp, Hd, device=dev, dtype=BF16)
    od = torch.empty(Mp, D, device=dev, dtype=BF16)
    dw = torch.empty(Hd, D, device=dev, dtype=F32)
    side = torch.cuda.Stream()
    out = torch.zeros(2, dtype=torch.int64, device=dev)

    def measure(name, work, ms=40.0):
        # warm the clocks with the workload for a while, then probe for `ms` beside it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        for _ in range(30):
            work()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            work()
        with torch.cuda.stream(side):
            lib.clock_probe(out.data_ptr(), int(ms * 1e5), 1, side.cuda_stream)
        for _ in range(300):
            work()
            n += 1
        e1.record()
        torch.cuda.synchronize()
        cyc, ref = out.tolist()
        line = f"{name:34s} shader clock {cyc / ref * 100:7.1f} MHz over {ref / 1e5:6.1f} ms   ({e0.elapsed_time(e1) / 320 * 1e3:7.1f} us per call)"
        if power:
            # keep the same work running for ~4 s with rocm-smi sampling beside it
            import threading
            import time
            stop, samples = threading.Event(), []
            th = threading.Thread(target=smi_sampler, args=(stop, samples))
            th.start()
            t0 = time.time()
            while time.time() - t0 < 4.0:
                for _ in range(50):
                    work()
                torch.cuda.synchronize()
            stop.set(); th.join()
            mid = samples[1:-1] or samples
            line += "   rocm-smi: " + " ".join(f"{w:.0f}W/{c}MHz" for w, c in mid[:6])
        print(line, flush=True)

    measure("idle", lambda: None)
    g = os.environ.get("AVSIAM_NT_GRID", "all")
    measure(f"gemm_nt K=3072 N=768 grid {g}", lambda: ops.gemm_nt(xh, w2, od, M))
    measure(f"gemm_nt K=768 N=3072 grid {g}", lambda: ops.gemm_nt(x, w1, oh, M))
    measure("gemm_tn wgrad", lambda: ops.gemm_tn(oh, x, dw, M))
    if "--all" in sys.argv:
        # the other kernel families of the step: attention (encoder-like 12 x 64, decoder-like 16 x 32) and LayerNorm
        for H, hd, L, nseq in ((12, 64, 196, 320), (12, 64, 512, 64), (16, 32, 2472, 64)):
            Da, rows = H * hd, nseq * L
            rp = ops.pad_rows(rows)
            qkv = torch.zeros(rp, 3 * Da, device=dev, dtype=BF16)
            qkv[:rows] = torch.randn(rows, 3 * Da, device=dev).to(BF16)
            ao, do_, dqkv = torch.zeros(rp, Da, device=dev, dtype=BF16), torch.zeros(rp, Da, device=dev, dtype=BF16), torch.zeros_like(qkv)
            do_[:rows] = torch.randn(rows, Da, device=dev).to(BF16)
            lse, delta = torch.zeros(H, rp, device=dev), torch.zeros(H, rp, device=dev)
            tiles = ops.AttnTiles([L] * nseq, dev, tile_rows=128)
            measure(f"attn fwd H={H} hd={hd} L={L}", lambda: ops.attn_fwd(qkv, tiles, H, ao, lse))
            measure(f"attn bwd H={H} hd={hd} L={L}", lambda: ops.attn_bwd(qkv, tiles, H, ao, do_, lse, delta, dqkv))
        rows = 95630
        rp = ops.pad_rows(rows, 128)
        xf, dres, dx = (torch.randn(rp, D, device=dev) for _ in range(3))
        dy, y, dxb = torch.randn(rp, D, device=dev).to(BF16), torch.zeros(rp, D, device=dev, dtype=BF16), torch.zeros(rp, D, device=dev, dtype=BF16)
        g0, b0, g1, b1 = (torch.randn(D, device=dev) for _ in range(4))
        dg0, db0, dg1, db1, dcol = (torch.zeros(D, device=dev) for _ in range(5))
        mean, rstd = torch.zeros(rp, device=dev), torch.zeros(rp, device=dev)
        mod = (torch.arange(rows, device=dev) >= rows // 4).to(torch.uint8)
        ws = torch.zeros(ops.layernorm_ws(rows, D), device=dev)
        measure("layernorm fwd 95630 x 768", lambda: ops.layernorm_fwd(xf, g0, b0, y, mean, rstd, rows, 1e-5, g1, b1, mod))
        measure("layernorm bwd 95630 x 768", lambda: ops.layernorm_bwd(dy, xf, mean, rstd, g0, dx, dg0, db0, ws, rows, g1, dg1, db1, mod, None, dres, dxb, dcol))


if __name__ == "__main__":
    main()
