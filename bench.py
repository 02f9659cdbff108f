#!/usr/bin/env python3
"""Headline benchmark: voxels/sec of one full nested-CV ridge fit (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): synthetic T=3000, F=768 x 4 FIR delays
(p=3072), V=80000 voxels PER GPU, 20 alphas logspace(-1, 8), 5 outer x 5 inner contiguous K-folds,
per-voxel alpha, normalpha, correlation scoring.  One "step" = one complete fit (Gram, 25 inner
alpha sweeps, 5 refits, test scoring, host statistics) with the fp32 inputs already resident in HBM
and the weights left resident; `value` = voxels of all ranks x steps / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--voxels V] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  The `roofline` object is for the dominant kernel, the fp16x3 MFMA
contraction k_sweep_f16x3, in its fused-score launches (the alphas that go through the batched
Cholesky: all their predictions reduced to scores in the epilogue): algorithmic flops per launch x 3
MFMAs per product / mean HIP-event duration over the timed steps, against the 2.5 PFLOP/s dense fp16
MFMA peak.  The same kernel's plain launches (the shared series terms of the large alphas, the refit)
are summarised beside it.  `cpu_baseline` times the CPU oracle (the reference algorithm restated, SVD
route) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, F0, DELAYS, A, N_OUTER, N_INNER = 3000, 768, [1, 2, 3, 4], 20, 5, 5
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0         # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
FIT_KW = dict(folding_type="kfold", n_outer_folds=N_OUTER, n_inner_folds=N_INNER, chunk_length=20,
              single_alpha=False, normalpha=True, use_corr=True, normalize_features=False,
              normalize_targets=False)


def synth_inputs(V, rank, dev):
    """SURVEY.md 8d generator: X0 ~ N(0,1) (T, 768) -> FIR delays (HIP kernel) -> X (T, 3072);
    Y = X (0.02 N(0,1)) + N(0,1), made on the device in fp32 (data synthesis only)."""
    from litcoder_core_amd import ops
    rng = np.random.default_rng(0)
    X0 = rng.standard_normal((T, F0))
    Xd = ops.fir_delay(torch.from_numpy(X0).to(dev), DELAYS, False)          # (T, 3072) f64 on device
    p = Xd.shape[1]
    dX = torch.zeros((T, ops.pad_to(p, 32)), dtype=torch.float32, device=dev)
    dX[:, :p] = Xd.to(torch.float32)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    Vp = ops.pad_to(V, 128)
    dY = torch.zeros((T, Vp), dtype=torch.float32, device=dev)
    W = 0.02 * torch.randn((p, V), generator=g, device=dev, dtype=torch.float32)
    dY[:, :V] = dX[:, :p] @ W + torch.randn((T, V), generator=g, device=dev, dtype=torch.float32)
    del W
    return dX, dY, p


def cpu_baseline(dX, dY, p, V_full, alphas, v_sample=2000):
    """The oracle (reference algorithm, fp32 torch-CPU SVD route + scipy pearsonr loop) on ONE of
    the five outer folds and `v_sample` voxels; SVD time is V-independent, the rest is
    proportional to V, so  T_cpu(V) = 5 * (t_svd + t_rest * V / v_sample)."""
    import oracle.folds as ofolds
    import oracle.nested_cv as onc
    import oracle.ridge as oridge
    import oracle.stats as ostats
    X = dX[:, :p].cpu()
    Y = dY[:, :v_sample].cpu()
    svd_time = [0.0]
    raw_svd = oridge.thin_svd

    def timed_svd(M, cutoff):
        t = time.perf_counter()
        out = raw_svd(M, cutoff)
        svd_time[0] += time.perf_counter() - t
        return out

    oridge.thin_svd = timed_svd
    try:
        t0 = time.perf_counter()
        tr, te = ofolds.create_folds(T, "kfold", N_OUTER)[0]
        inner = ofolds.create_folds(len(tr), "kfold", N_INNER)
        chosen, _ = onc.select_alphas(X[tr], Y[tr], inner, alphas, False, True, True, 1e-10)
        W = oridge.ridge_weights(X[tr], Y[tr], chosen, normalpha=True, singcutoff=1e-10)
        ostats.pearson_per_voxel(Y[te].numpy(), (X[te] @ W).numpy())
        total = time.perf_counter() - t0
    finally:
        oridge.thin_svd = raw_svd
    t_svd, t_rest = svd_time[0], total - svd_time[0]
    t_full = N_OUTER * (t_svd + t_rest * V_full / v_sample)
    return {
        "value": V_full / t_full, "unit": "voxels/sec", "cores": int(torch.get_num_threads()), "kind": "port",
        "sample": (f"oracle (reference algorithm restated: fp32 torch-CPU SVD route, scipy pearsonr loop) on 1 of "
                   f"{N_OUTER} outer folds x {v_sample} of {V_full} voxels, full T={T} p={p} A={len(alphas)}: "
                   f"{total:.1f}s measured ({t_svd:.1f}s in 6 V-independent SVDs, {t_rest:.1f}s proportional to V); "
                   f"extrapolated to the whole job as {N_OUTER}*(t_svd + t_rest*V/{v_sample}) = {t_full:.0f}s"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--voxels", type=int, default=80000, help="voxels per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "f16x3"],
                    help="arithmetic of the alpha sweep (auto = f16x3 unless the targets' dynamic range forbids it)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))

    from litcoder_core_amd import NestedCVModel, ShardContext, ops
    dev = ops.device(local)
    V = args.voxels
    alphas = np.logspace(-1, 8, A)
    dX, dY, p = synth_inputs(V, rank, dev)
    shard = ShardContext(device=dev) if world > 1 else None
    model = NestedCVModel("ridge_regression", shard=shard, precision=args.precision)

    def step():
        # every rank passes its own V-voxel block; the gather at the end of the fit spans V*world voxels
        return model.fit_predict_device(dX, dY, p, V, n_voxels_total=None if world == 1 else V * world,
                                        alphas=alphas, **FIT_KW)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    metrics = None
    for _ in range(args.warmup):
        metrics, _, _ = step()
    ops.timing_enable(True)
    ops.timing_read()
    from litcoder_core_amd.nested_cv import LAST_SWEEP as _ls
    plain_flops0 = _ls["plain_flops"]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics, _, _ = step()
    fence()
    elapsed = time.perf_counter() - t0
    kern = ops.timing_read()
    ops.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_o = T - T // N_OUTER
        n_v = n_o // N_INNER
        n_i = n_o - n_v
        from litcoder_core_amd.nested_cv import LAST_SWEEP
        split = LAST_SWEEP["precision"] == "f16x3"
        A_fused = LAST_SWEEP.get("fused_alphas", A)           # alphas scored inside the fused launch
        flops_per_launch = 2.0 * A_fused * n_v * n_i * V      # algorithmic: those alphas of one inner fold
        ms, launches = kern.get("alpha_sweep_gemm", (0.0, 0))
        avg_ms = ms / max(launches, 1)
        alg_tflops = flops_per_launch / (avg_ms * 1e-3) / 1e12 if launches else None
        # f16x3: every algorithmic product is three fp16 MFMAs (hi*hi + hi*lo + lo*hi); the MFMA roofline
        # is priced on the MFMA flops the kernel executes, the algorithmic rate is reported beside it.
        mfma_per_product = 3 if split else 1
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        achieved = alg_tflops * mfma_per_product if alg_tflops else None
        plain_ms, plain_n = kern.get("grouped_gemm", (0.0, 0))
        plain = None
        if split and plain_n:
            pf = LAST_SWEEP["plain_flops"] - plain_flops0
            plain = {"launches": plain_n, "ms_per_step": plain_ms / args.steps,
                     "algorithmic_tflops": pf / (plain_ms * 1e-3) / 1e12,
                     "mfma_tflops": 3 * pf / (plain_ms * 1e-3) / 1e12,
                     "what": f"{LAST_SWEEP.get('series_terms', 0)} shared series terms x 25 inner folds + 5 refits "
                             "(weights and test predictions) per step; includes the f32 MFMA launches of that slot"}
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "alpha_sweep_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(LAST_SWEEP["precision"], tj).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the fit's main stream leaves 32 of the 256 CUs to the auxiliary (fp64) stream: the kernel's launches are
        # 256/224 longer than on the whole chip, `frac` stays priced against the whole chip
        from litcoder_core_amd.nested_cv import _main_stream
        cus_main = 224 if _main_stream() is not None else 256
        out = {
            "metric": "voxels/sec full nested-CV ridge fit (LeBel UTS03, GPT-2 768x4 delays, ~80k voxels)",
            "value": world * V * args.steps / elapsed, "unit": "voxels/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3+f32acc (fp16 hi+lo operands, fp32 accumulate; Gram/Cholesky in f64)" if split
                     else "f32 (Gram/Cholesky in f64)", "data": "synthetic",
            "config": {"workload": f"cfg2 synthetic T={T} F={F0}x{len(DELAYS)} delays (p={p}) V={V}/GPU "
                                   f"A={A} alphas {N_OUTER}x{N_INNER} kfold, per-voxel alpha, normalpha, corr",
                       "voxels_per_gpu": V, "inputs": "fp32 resident in HBM; weights left resident; "
                                                      "per-voxel scores/alphas/p-values on host",
                       "parallelism": f"voxel-shard x{world}", "median_score": metrics["median_score"]},
            "roofline": {"bound": "mfma",
                         "kernel": "k_sweep_f16x3 (fused alpha sweep, 3 fp16 MFMAs per product)" if split
                                   else "k_gemm_f32<score> (fused alpha sweep, f32-input MFMA)",
                         "note": "peak = dense fp16 MFMA datasheet figure at 2.4 GHz; under this kernel the chip holds "
                                 "1.4-1.8 GHz (in-kernel s_memtime/s_memrealtime, profiles/), where the same MFMA stream "
                                 "tops out at 1.5-1.8 PFLOP/s" if split else None,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": traffic,
                         "cus_of_256_the_kernel_runs_on": cus_main,
                         "frac_of_the_cus_it_runs_on": (achieved / (peak * cus_main / 256.0)) if achieved else None,
                         "algorithmic_tflops": alg_tflops, "mfma_per_product": mfma_per_product,
                         "flops_per_launch": flops_per_launch, "avg_launch_ms": avg_ms, "launches": launches,
                         "fused_alphas_per_launch": A_fused, "plain_launches_same_kernel": plain},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in sorted(kern.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dX, dY, p, V, alphas)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
