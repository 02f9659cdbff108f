"""The voxel-sharded orchestration (gather of per-voxel vectors, single-alpha all-reduce) on CPU:
two processes, gloo backend.  The device engine is replaced by an oracle-backed stand-in so that
only the host-side sharding logic of litcoder_core_amd.nested_cv is under test; the result must
equal the unsharded run on every rank."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, pickle, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import litcoder_core_amd.nested_cv as ncv
from litcoder_core_amd.dist import ShardContext
import oracle.ridge as oridge

class OracleEngine:
    """Same interface as RidgeCVEngine.run_fold/weights, arithmetic by the CPU oracle."""
    def __init__(self, X, Y, alphas, normalpha, use_corr, nf, nt, shard, precision="auto", singcutoff=0.0):
        self.X, self.Y = torch.tensor(np.asarray(X), dtype=torch.float32), torch.tensor(np.asarray(Y), dtype=torch.float32)
        self.alphas, self.normalpha, self.use_corr, self.shard = list(alphas), normalpha, use_corr, shard
        self.V = self.Y.shape[1]
        self.W = np.zeros((self.X.shape[1], self.V), dtype=np.float32)
    def run_fold(self, tr, te, inner, single_alpha, scale):
        tr, te = np.asarray(tr), np.asarray(te)
        Xtr, Ytr = self.X[tr], self.Y[tr]
        tot = 0
        for a, b in inner:
            tot = tot + oridge.alpha_sweep_scores(Xtr[a], Xtr[b], Ytr[a], Ytr[b], self.alphas, 1e-10, self.use_corr, self.normalpha)
        if single_alpha:
            total = self.shard.allreduce_sum(tot.double().sum(dim=1).numpy())
            idx = np.full(self.V, int(np.argmax(total)), dtype=np.int32)
        else:
            idx = tot.argmax(dim=0).numpy().astype(np.int32)
        W = oridge.ridge_weights(Xtr, Ytr, torch.tensor([self.alphas[i] for i in idx], dtype=torch.float32), 1e-10, self.normalpha)
        pred = (self.X[te] @ W).numpy().astype(np.float64); yt = self.Y[te].numpy().astype(np.float64)
        pc, yc = pred - pred.mean(0), yt - yt.mean(0)
        with np.errstate(all="ignore"):
            r = (pc * yc).sum(0) / np.sqrt((pc ** 2).sum(0) * (yc ** 2).sum(0))
        self.W += scale * W.numpy()
        from litcoder_core_amd import stats
        return ncv._FoldResult(r, stats.pearson_pvalues(r.astype(np.float32), len(te)), idx, len(te))
    def begin_fit(self):
        pass
    def precompute_lmax(self, outer):
        return [None] * len(outer)
    def fold_prepare(self, tr, te, inner, lmax_pre=None, chol_after=None):
        return (tr, te, inner)
    def fold_begin(self, tr, te, inner, prepared=None):
        return (tr, te, inner)
    def fold_refit(self, st, single_alpha, scale):
        return self.run_fold(*st, single_alpha, scale)
    def fold_select(self, st, single_alpha):
        return (st, single_alpha)
    def fold_finish(self, sel, scale):
        return self.run_fold(*sel[0], sel[1], scale)
    def fold_collect(self, pend):
        return pend
    def weights(self):
        return self.W

ncv.RidgeCVEngine = OracleEngine
mode = sys.argv[2]
rng = np.random.default_rng(0)
X = rng.standard_normal((120, 12)); Y = X @ rng.standard_normal((12, 37)) * 0.4 + rng.standard_normal((120, 37))
Y[:, 11] = 2.0
kw = dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=2, alphas=np.logspace(-1, 3, 5), single_alpha=(mode == "single"))
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    dist.init_process_group("gloo")
    shard = ShardContext()
    assert shard.world == 2
    m, W, a = ncv.NestedCVModel("r", shard=shard).fit_predict(X, Y, **kw)
    lo, hi = shard.bounds(37)
    assert W.shape == (12, hi - lo)
    # train/test mode through the same sharded path
    m2, W2, a2 = ncv.NestedCVModel("r", shard=shard).fit_predict(X[:90], Y[:90], X_test=X[90:], y_test=Y[90:], **kw)
    out = dict(m=m, W=W, a=a, lo=lo, hi=hi, m2=m2, a2=a2)
    pickle.dump(out, open(os.path.join(sys.argv[3], f"rank{shard.rank}_{mode}.pkl"), "wb"))
    dist.destroy_process_group()
else:
    m, W, a = ncv.NestedCVModel("r").fit_predict(X, Y, **kw)
    m2, W2, a2 = ncv.NestedCVModel("r").fit_predict(X[:90], Y[:90], X_test=X[90:], y_test=Y[90:], **kw)
    pickle.dump(dict(m=m, W=W, a=a, m2=m2, a2=a2), open(os.path.join(sys.argv[3], f"single_{mode}.pkl"), "wb"))
'''


@pytest.mark.parametrize("mode", ["pervoxel", "single"])
def test_two_rank_shard_equals_unsharded(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, str(script), ROOT, mode, str(tmp_path)], check=True, env=env, timeout=300)
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "single" else 1)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, mode, str(tmp_path)],
                   check=True, env=env, timeout=600)
    ref = pickle.load(open(tmp_path / f"single_{mode}.pkl", "rb"))
    ranks = [pickle.load(open(tmp_path / f"rank{r}_{mode}.pkl", "rb")) for r in range(2)]
    for out in ranks:
        for key in ("m", "m2"):
            assert sorted(out[key]) == sorted(ref[key])
            for k, v in ref[key].items():
                got = out[key][k]
                if isinstance(v, list):
                    assert np.array_equal(np.asarray(got), np.asarray(v)), (key, k)
                else:
                    assert got == v, (key, k)
        assert np.array_equal(out["a"], ref["a"]) and np.array_equal(out["a2"], ref["a2"])
        assert np.array_equal(out["W"], ref["W"][:, out["lo"]:out["hi"]])
    assert ranks[0]["hi"] == ranks[1]["lo"] and ranks[1]["hi"] == 37
