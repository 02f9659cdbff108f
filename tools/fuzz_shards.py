#!/usr/bin/env python3
"""Random configurations through the voxel-sharded fit with the REAL engine: 2 and 3 ranks on the one GPU of the box (gloo
exchanges, as tests/test_gpu_shards.py) against the unsharded fit of the same inputs -- every rank must return the metrics
of ALL voxels, the alphas and its own block of the weights BIT FOR BIT.  Shapes, fold types, normalisers, scoring, single /
per-voxel alpha, CV / train-test, precisions, host panels and outlier entries in the targets (the f32 side panel of one
rank's block) are drawn at random.  A bug hunt, not a test.
    python tools/fuzz_shards.py [n_cases [seed]]"""
import os
import pickle
import socket
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, pickle, random, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from litcoder_core_amd import NestedCVModel, ShardContext
backend, out_dir, n_cases, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
shard = None
if backend != "none":
    dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=world)
    shard = ShardContext(device=torch.device("cuda", 0))
rng = np.random.default_rng(seed)
out = {}
for case in range(n_cases):
    T = int(rng.integers(120, 420))
    p = int(rng.choice([2, 5, 8, 16, 17, 24, 40, 70, 130, 300]))
    V = int(rng.choice([64, 100, 129, 257, 300, 777, 1300]))
    fold = str(rng.choice(["kfold", "chunked", "kfold_trimmed", "chunked_trimmed", "timeseries", "group"]))
    use_corr = bool(rng.random() < 0.8)
    kw = dict(folding_type=fold, n_outer_folds=int(rng.integers(2, 4)), n_inner_folds=int(rng.integers(2, 4)),
              alphas=np.logspace(rng.uniform(-2, 0), rng.uniform(1, 5), int(rng.integers(1, 9))),
              normalpha=bool(rng.random() < 0.7), use_corr=use_corr, single_alpha=bool(rng.random() < 0.3),
              normalize_features=bool(rng.random() < 0.2), normalize_targets=bool(rng.random() < 0.2))
    if "chunked" in fold:
        kw["chunk_length"] = int(rng.integers(5, 30))
    tt = int(rng.integers(30, 90)) if rng.random() < 0.3 else 0
    if fold == "group":
        kw["groups"] = rng.integers(0, 8, size=T - tt)
    signal = 1.0 if not use_corr else float(rng.choice([0.3, 1.0]))
    X = rng.standard_normal((T, p)) * rng.uniform(0.5, 2.0, p)
    Y = X @ (rng.standard_normal((p, V)) * (signal / np.sqrt(p))) + rng.standard_normal((T, V)) + rng.uniform(-3, 3)
    spiked = []
    if rng.random() < 0.35:
        for c in rng.choice(V, size=int(rng.integers(1, 4)), replace=False):
            Y[int(rng.integers(0, T - tt)), int(c)] = float(rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(4, 6))
            spiked.append(int(c))
    if rng.random() < 0.2:
        Y[:, int(rng.integers(0, V))] = 2.5                  # a constant voxel
    precision = str(rng.choice(["auto", "auto", "f32"]))
    panel_cols = int(rng.choice([0, 0, 256]))
    local = bool(rng.random() < 0.25)
    args = (X[:T - tt], Y[:T - tt])
    extra = dict(X_test=X[T - tt:], y_test=Y[T - tt:]) if tt else {}
    kw_run = {k: v for k, v in kw.items() if not (tt and k == "n_outer_folds")}
    tag = f"case {case}: T{T} p{p} V{V} {fold} tt{tt} {precision} panels{panel_cols} local{int(local)} " + " ".join(
        f"{k}={v}" for k, v in kw.items() if k not in ("alphas", "groups", "folding_type")) + f" A={len(kw['alphas'])}"
    lo, hi = shard.bounds(V) if shard else (0, V)
    random.seed(case); np.random.seed(case)
    try:
        model = NestedCVModel("r", shard=shard, precision=precision, panel_cols=panel_cols,
                              local_targets=bool(local and shard is not None))
        a_ = args
        e_ = dict(extra)
        if local and shard is not None:
            a_ = (args[0], args[1][:, lo:hi])
            if tt:
                e_["y_test"] = extra["y_test"][:, lo:hi]
        m, W, a = model.fit_predict(*a_, **e_, **kw_run)
        out[case] = dict(tag=tag + f" spiked{spiked}", m=m, W=np.asarray(W), a=np.asarray(a), lo=lo, hi=hi, form=model.last_form,
                         prec=model.last_fit.get("precision"), side=model.last_fit.get("side_panel_cols"))
        if shard is None and V >= 600:
            # the unsharded fit once more with the OTHER panel plan of its host targets (none / 256-column panels): per-voxel
            # results must not depend on the plan, bit for bit
            random.seed(case); np.random.seed(case)
            m2, W2, a2 = NestedCVModel("r", precision=precision, panel_cols=0 if panel_cols else 256).fit_predict(*args, **extra, **kw_run)
            bad = [k for k, v in m.items() if not (np.array_equal(np.asarray(m2[k]), np.asarray(v), equal_nan=True)
                                                  if isinstance(v, list) else (m2[k] == v or (m2[k] != m2[k] and v != v)))]
            if bad or not np.array_equal(np.asarray(W2), np.asarray(W), equal_nan=True) or not np.array_equal(a2, a):
                out[case]["plan"] = f"panel plans differ: metrics {bad[:4]}, weights {not np.array_equal(np.asarray(W2), np.asarray(W), equal_nan=True)}"
    except ValueError as e:
        out[case] = dict(tag=tag, err="ValueError: " + str(e)[:120], lo=lo, hi=hi)
name = "single" if shard is None else f"w{world}_r{shard.rank}"
pickle.dump(out, open(os.path.join(out_dir, name + ".pkl"), "wb"))
if shard is not None:
    dist.destroy_process_group()
'''


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tmp = tempfile.mkdtemp(prefix="fuzz_shards_")
    script = os.path.join(tmp, "worker.py")
    open(script, "w").write(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, script, ROOT, "none", tmp, str(n_cases), str(seed)], check=True, env=env)
    ref = pickle.load(open(os.path.join(tmp, "single.pkl"), "rb"))
    fails = 0
    for world in (2, 3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), script, ROOT, "gloo", tmp, str(n_cases),
                        str(seed)], check=True, env=env)
        ranks = [pickle.load(open(os.path.join(tmp, f"w{world}_r{r}.pkl"), "rb")) for r in range(world)]
        for case in range(n_cases):
            want = ref[case]
            why = []
            for r, got_all in enumerate(ranks):
                got = got_all[case]
                if "err" in want or "err" in got:
                    if ("err" in want) != ("err" in got):
                        why.append(f"rank {r}: {got.get('err', 'fit went through')} vs {want.get('err', 'fit went through')}")
                    continue
                lo, hi = got["lo"], got["hi"]
                if sorted(got["m"]) != sorted(want["m"]):
                    why.append(f"rank {r}: metrics keys differ")
                    continue
                for k, v in want["m"].items():
                    g = got["m"][k]
                    same = (np.array_equal(np.asarray(g), np.asarray(v), equal_nan=True) if isinstance(v, (list, np.ndarray))
                            else (g == v or (g != g and v != v)))
                    if not same:
                        if isinstance(v, list) and len(v) == len(g):
                            gv, vv = np.asarray(g, dtype=np.float64), np.asarray(v, dtype=np.float64)
                            d = np.nonzero(~((gv == vv) | ((gv != gv) & (vv != vv))))[0]
                            why.append(f"rank {r}: metrics[{k}] differs at voxels {d[:6].tolist()} ({d.size} in all), "
                                       f"max |d| {np.nanmax(np.abs(gv[d] - vv[d])):.3g}")
                        else:
                            why.append(f"rank {r}: metrics[{k}] differs: {g!r} vs {v!r}")
                if not np.array_equal(got["a"], want["a"]):
                    why.append(f"rank {r}: alphas differ at {int((got['a'] != want['a']).sum())} voxels")
                if got["W"].shape != want["W"][:, lo:hi].shape or not np.array_equal(got["W"], want["W"][:, lo:hi], equal_nan=True):
                    dw = np.nonzero((got["W"] != want["W"][:, lo:hi]).any(axis=0))[0] + lo
                    why.append(f"rank {r}: weights differ in columns {dw[:6].tolist()} ({dw.size} in all)")
                if got["form"] != want["form"] or got["prec"] != want["prec"]:
                    why.append(f"rank {r}: {got['form']} / {got['prec']} vs {want['form']} / {want['prec']}")
            if world == 2 and want.get("plan"):
                why.append(want["plan"])
            if why:
                fails += 1
                print(f"FAIL world {world} {want['tag']}\n      " + "\n      ".join(why[:8]), flush=True)
            elif world == 2:
                print("ok  ", want["tag"], "->", want.get("form", want.get("err")), want.get("prec"), "side", want.get("side"),
                      flush=True)
    print(f"{n_cases} configurations x (2, 3) ranks, seed {seed}: {fails} (case, world) pairs differ from the unsharded fit")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
