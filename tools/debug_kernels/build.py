#!/usr/bin/env python3
"""Builds tools/bin/liblitcoder_debug.so (diagnostic kernels, gfx950) against the product library.
    python tools/debug_kernels/build.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def build(exp_bits=None):
    """``exp_bits``: a build whose sweep kernels carry these experiment bits as a compile-time constant
    (lc_gemm16_kernel.h, LC_SWEEP_EXP_BITS): tools/bin/liblitcoder_debug_exp<bits>.so."""
    from litcoder_core_amd import build as product
    lib = product.build()
    out_dir = os.path.join(ROOT, "tools", "bin")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liblitcoder_debug.so" if exp_bits is None else f"liblitcoder_debug_exp{int(exp_bits)}.so")
    src = os.path.join(HERE, "lc_debug_gemm16.hip")
    deps = [src, os.path.join(HERE, "lc_debug.h"), os.path.join(ROOT, "litcoder_core_amd", "csrc", "lc_gemm16_kernel.h"), lib]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    csrc = os.path.dirname(lib)
    extra = [] if exp_bits is None else [f"-DLC_SWEEP_EXP_BITS={int(exp_bits)}"]
    cmd = [product.HIPCC, *product.FLAGS, *extra, "-shared", src, "-o", out, f"-L{csrc}", "-llitcoder_hip", "-Wl,-rpath,$ORIGIN/../../litcoder_core_amd/csrc"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{r.stdout}")
    return out


def load(exp_bits=None):
    """ctypes handle of the debug library (built on first use); errors of its calls are in lc_last_error() of the product."""
    import ctypes
    from litcoder_core_amd import _lib
    _lib.load()                                           # the product library first: the debug one links against it
    return ctypes.CDLL(build(exp_bits))


if __name__ == "__main__":
    print(build())
    for b in sys.argv[1:]:
        print(build(int(b)))
