"""Is the fused alpha sweep (k_sweep_f16x3<score>) bit-reproducible while OTHER kernels run beside it on another stream?
The real operands of fold 0 of the cfg2 fit; `_sweeps` repeated, each time with a disturbance of the given kind queued on
a side stream, scores compared with the undisturbed first run.
    python tools/sweep_stress.py [kind ...]      kinds: none memset gather gemm chol copy"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from litcoder_core_amd import ops  # noqa: E402
import litcoder_core_amd.nested_cv as ncv  # noqa: E402
from litcoder_core_amd.folding import create_folds  # noqa: E402

kinds = sys.argv[1:] or ["none", "memset", "gather", "gemm", "chol"]
V = int(os.environ.get("STRESS_V", "80000"))
reps = int(os.environ.get("STRESS_REPS", "12"))
dev = ops.device(0)
dX, dY, p = bench.synth_inputs(V, 0, dev)
alphas = np.logspace(-1, 8, bench.A)
eng = ncv.RidgeCVEngine(ncv._DeviceShapes(dX, p), ncv._DeviceShapes(dY, V), alphas, True, True, False, False)
eng.begin_fit(1)
T = bench.T
splits = create_folds(T, "kfold", 5)
tr, te = splits[0]
inner = create_folds(len(tr), "kfold", 5)
base = eng.fold_prepare(tr, te, inner)
torch.cuda.synchronize()
side = torch.cuda.Stream()
N = 1920
Kn = torch.randn((4, N, N), device=dev)
Q = torch.randn((N, 4 * 512), device=dev)
big = torch.empty((20, 2048, 1920), device=dev)
tr_i = ops.idx_matrix([np.arange(N)] * 4, N, dev)
lm = torch.ones(4, dtype=torch.float64, device=dev)
aug_src = None


def disturb(kind):
    with torch.cuda.stream(side):
        for _ in range(6):
            if kind == "memset":
                ops.zeros((20, 2048, 1920), torch.float32, dev)
            elif kind == "copy":
                big.copy_(big.flip(0))
            elif kind == "gather":
                out = torch.empty((4, N, N), dtype=torch.float32, device=dev)
                ops.gather_sub_f32(eng.K, tr_i, tr_i, 4, N, N, lm, out)
            elif kind == "gemm":
                Qn = torch.empty_like(Q)
                ops.gemm_grouped(Kn, N, N * N, Q, 4 * 512, None, Qn, 4 * 512, N, 4 * 512, N, [0, 4, 8, 12, 16])
            elif kind == "chol":
                B, M = 8, 480
                aug = torch.randn((B, N + M, N), dtype=torch.float64, device=dev) * 0.01
                aug[:, :N] += torch.eye(N, dtype=torch.float64, device=dev) * 50.0
                H = torch.empty((B, M, N), dtype=torch.float32, device=dev)
                ops.batch_chol_solve(aug, B, N, M, H)


def sweep():
    st = dict(base)
    hat = dict(base["hat"])
    cs, split = eng._target_scales(eng.dY_full, eng.full)
    hat.update(cs=cs, split=split)
    return eng._sweeps(hat, eng.dY_full, None)


ref = sweep()
torch.cuda.synchronize()
ref = ref.clone()
for kind in kinds:
    bad = 0
    for it in range(reps):
        if kind != "none":
            disturb(kind)
        s = sweep()
        torch.cuda.synchronize()
        d = (s[:, :V] != ref[:, :V])
        if bool(d.any()):
            bad += 1
            rows = torch.nonzero(d.any(dim=1)).flatten().tolist()
            cols = torch.nonzero(d.any(dim=0)).flatten()
            print(f"  {kind} rep {it}: {int(d.sum())} entries differ, alpha rows {rows}, {cols.numel()} voxels, "
                  f"tiles {sorted(set((cols // 256).tolist()))[:10]}", flush=True)
    print(f"{kind}: {bad} of {reps} sweeps differ from the undisturbed run", flush=True)
