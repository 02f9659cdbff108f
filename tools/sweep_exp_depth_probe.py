#!/usr/bin/env python3
"""The un-stamped experiment build of the screening sweep (tools/debug_kernels, LC_SWEEP_EXPERIMENTS: this translation unit's
copy of k_sweep_f16x3<score, HI2> with the experiment bits live) timed over contraction depths: slope = time per ring step
and tile, intercept = everything outside the main loop.  Bits: lc_gemm16_kernel.h.
    python tools/sweep_exp_depth_probe.py [bits,bits,...]
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
from litcoder_core_amd._lib import LC_MB  # noqa: E402
import build as debug_build  # noqa: E402

dev = ops.device()
CONST = os.environ.get("PROBE_CONST", "1") != "0"      # one library per variant, the bits a compile-time constant
libs = {}


def lib_of(bits):
    key = int(bits) if CONST else None
    if key not in libs:
        libs[key] = debug_build.load(key)
    return libs[key]


V, n_v, A = int(os.environ.get("PROBE_V", 80000)), 480, 4
M = ops.pad_to(n_v, LC_MB)
g = torch.Generator(device=dev); g.manual_seed(0)
p_ = lambda t: ctypes.c_void_p(t.data_ptr())
NAMES = ((1, "contiguous"), (2, "delivery alone"), (4, "tile (0,0)"), (8, "no DMA"), (16, "rotated"), (32, "dword DMA"),
         (64, "no reads"), (128, "no MFMA"), (256, "private L2 region"))


def name(bits):
    return ", ".join(n for b, n in NAMES if bits & b) or "as shipped"


def setup(N):
    T = N + n_v
    H = torch.randn((A * M, N), generator=g, device=dev, dtype=torch.float32) * 0.02
    Y = torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    tr = ops.idx_tensor(np.r_[0:N], N, dev)
    va = ops.idx_tensor(np.r_[N:T], M, dev)
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
    keep = (H, Y, tr, va, ystat, yblk, part, yv, Ht, rs_inv, Yt, cs)

    def call(bits):
        rc = lib_of(bits).lc_debug_sweep16_stamps_exp(p_(Ht), p_(rs_inv), A, M, N, p_(Yt), p_(cs[V:]), p_(yv), ctypes.c_int64(V), n_v,
                                             p_(ystat), p_(part), None, 1, int(bits),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, (rc, keep is None)
    return call


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


combos = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,8,2,64,128,72,1,256").split(",")]
depths = (640, 1280, 1920, 2560)
calls = {N: setup(N) for N in depths}
tiles = -(-A * 480 // 256) * -(-V // 256)
rounds = tiles / 256.0
for rnd in range(2):
    for bits in combos:
        ts = [timeit(lambda: calls[N](bits)) for N in depths]
        slope, icpt = np.polyfit(np.asarray(depths, dtype=np.float64), np.asarray(ts), 1)
        print(f"round {rnd} [{bits:3d}] {name(bits):34s}: " + ", ".join(f"N {n}: {t:.3f}" for n, t in zip(depths, ts)) +
              f" ms | {slope * 32 * 1e6 / rounds:.0f} ns per ring step and tile, {icpt / rounds * 1e3:.1f} us per tile outside the main loop",
              flush=True)
