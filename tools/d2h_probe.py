"""How do strided (2-D) device-to-host copies behave beside compute?  Times a column-panel D2H of a (3072, 80000) f32
matrix into page-locked host memory as hipMemcpy2DAsync (what _range_finished issues) and as a linear copy of a
contiguous staging buffer, alone and beside a plain fp16x3 GEMM loop on another stream (does the copy slow the GEMM: a
shader blit would take CUs, the SDMA engines would not)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from litcoder_core_amd import ops  # noqa: E402

dev = ops.device(0)
p, V, w = 3072, 80000, 24576
W = torch.randn((p, V), device=dev)
host = torch.empty((p, V), dtype=torch.float32, pin_memory=True)
stage_d = torch.empty((p, w), device=dev)
stage_h = torch.empty((p, w), dtype=torch.float32, pin_memory=True)
dl = torch.cuda.Stream()
main = torch.cuda.current_stream()
# a GEMM workload like the refit's weight rows: (3072 x 2400) . (2400 x 24576)
K, rows, Vs = 2400, 3072, 24576
A = torch.randn((rows, K), device=dev)
B = torch.randn((K, Vs), device=dev)
At = torch.empty(ops.pad_to(rows, 256) * K * 2, dtype=torch.float16, device=dev)
rs = torch.empty(ops.pad_to(rows, 256), dtype=torch.float32, device=dev)
ops.split_rows_f16(A, rows, K, At, rs)
cs, _ = ops.col_scales_f16(B, K, Vs)
Bt = torch.empty(Vs * K * 2, dtype=torch.float16, device=dev)
ops.split_cols_f16(B, Vs, ops.idx_tensor(np.arange(K), K, dev), K, cs, Bt)
C = torch.empty((rows, Vs), device=dev)


def gemm(n=8):
    for _ in range(n):
        ops.gemm_grouped_f16x3(At, rs, rows, Bt, cs[Vs:], C, Vs, Vs, K, [0, Vs // 256])


def copy2d():
    ops.download_cols(W[:, 8192:8192 + w], host, 8192, w, dl)


def copylin():
    with torch.cuda.stream(dl):
        stage_h.copy_(stage_d, non_blocking=True)


def timed(fn_main, fn_side, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    if fn_side:
        fn_side()
    if fn_main:
        fn_main()
    e1.record()
    e1.synchronize()
    t_main = e0.elapsed_time(e1)
    torch.cuda.synchronize()
    t_all = 1e3 * (time.perf_counter() - t0)
    print(f"{label:50s} main stream {t_main:7.2f} ms   everything done after {t_all:7.2f} ms", flush=True)


for _ in range(2):
    gemm(2); copy2d(); copylin(); torch.cuda.synchronize()
mb = p * w * 4 / 1e6
print(f"panel = {mb:.0f} MB")
timed(None, copy2d, "2-D D2H alone")
timed(None, copylin, "linear D2H alone")
timed(gemm, None, "8 GEMMs alone")
timed(gemm, copy2d, "8 GEMMs beside a 2-D D2H")
timed(gemm, copylin, "8 GEMMs beside a linear D2H")
