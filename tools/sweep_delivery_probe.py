#!/usr/bin/env python3
"""What bounds the operand delivery of the sweep kernel (round 6): the stamped diagnostic build of k_sweep_f16x3 at cfg2's
inner-fold shape (4 alphas x 480 validation rows, depth 1920, 80 000 voxels) with the kernel's experiment bits
(lc_gemm16_kernel.h): contiguous 16 KB fetches instead of two 8 KB hi planes 16 KB apart, delivery alone (no MFMAs), every
workgroup on tile (0, 0) (all L2 hits).  Prints ms per launch and cycles per 32 KB ring step.
    python tools/sweep_delivery_probe.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "debug_kernels"))
from litcoder_core_amd import ops  # noqa: E402
from litcoder_core_amd._lib import LC_MB, LC_NB  # noqa: E402
import build as debug_build  # noqa: E402

dev = ops.device()
dbg = debug_build.load()
A, n_v, n_i, V, T = 4, 480, 1920, 80000, 3000
M, N = ops.pad_to(n_v, LC_MB), ops.pad_to(n_i, LC_NB)
g = torch.Generator(device=dev); g.manual_seed(0)
H = torch.randn((A * M, N), generator=g, device=dev, dtype=torch.float32) * 0.02
Y = torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
tr = ops.idx_tensor(np.r_[0:1920], N, dev)
va = ops.idx_tensor(np.r_[1920:2400], M, dev)
ystat = torch.empty((3, V), dtype=torch.float32, device=dev)
yblk = torch.empty((M // LC_MB, V), dtype=torch.float32, device=dev)
part = torch.empty((A * M // LC_MB, 4, V), dtype=torch.float32, device=dev)
yv = torch.empty((M, V), dtype=torch.float32, device=dev)
ops.val_stats(Y, V, va, M, n_v, ystat, yblk, yv)
rows_pad = ops.pad_to(A * M, 256)
Ht = torch.empty(rows_pad * N * 2, dtype=torch.float16, device=dev)
rs_inv = torch.empty(rows_pad, dtype=torch.float32, device=dev)
Yt = torch.empty(ops.pad_to(V, 256) * N * 2, dtype=torch.float16, device=dev)
cs, _flag = ops.col_scales_f16(Y, T, V)
ops.split_rows_f16_alphas(H, 1, A, M, N, Ht, rs_inv)
ops.split_cols_f16(Y, V, tr, N, cs, Yt)
p_ = lambda t: ctypes.c_void_p(t.data_ptr())


def run(hi2, bits, reps=6):
    st = torch.zeros(32, dtype=torch.int64, device=dev)
    call = lambda: dbg.lc_debug_sweep16_stamps_exp(p_(Ht), p_(rs_inv), A, M, N, p_(Yt), p_(cs[V:]), p_(yv), ctypes.c_int64(V), n_v,
                                                   p_(ystat), p_(part), p_(st), int(hi2), int(bits),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(2):
        assert call() == 0
    torch.cuda.synchronize()
    st.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        assert call() == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    h = st.cpu().numpy().reshape(2, 16).astype(np.float64)
    steps = N // (32 if hi2 else 16)
    out = []
    for g_ in range(2):
        nw = max(h[g_, 5], 1) / steps
        ghz = h[g_, 0] / max(h[g_, 14], 1) * 0.1
        out.append(f"{h[g_, 0] / nw / steps:.0f} cyc/step @ {ghz:.2f} GHz, prologue {h[g_, 6] / nw:.0f}, epilogue {h[g_, 7] / nw:.0f}")
    gb = steps * 32768 * ((A * M + 255) // 256) * ((V + 255) // 256) / 1e9
    return ms, gb, out


def name(bits):
    parts = [n for b, n in ((1, "contiguous 16 KB"), (2, "delivery alone"), (4, "tile (0,0)"), (8, "no DMA"), (16, "K loops rotated"), (32, "dword DMA"), (64, "no reads"), (128, "no MFMA")) if bits & b]
    return ", ".join(parts) or "as shipped"


combos = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,4,6,38,36,32,34,68,132,196,100".split(","))]
for rnd in range(2):
    for hi2 in (1, 0):
        for bits in combos:
            if not hi2 and (bits & (1 | 32 | 64 | 128)):
                continue
            ms, gb, out = run(hi2, bits)
            print(f"round {rnd} {'HI2 ' if hi2 else '3-MFMA'} [{bits:2d}] {name(bits):44s}: {ms:.3f} ms, {gb:.2f} GB into LDS = {gb / ms:.2f} TB/s "
                  f"= {gb / ms * 1e3 / 256:.1f} GB/s per CU | waves 0-3: {out[0]} | waves 4-7: {out[1]}", flush=True)
