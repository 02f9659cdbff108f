"""The voxel-sharded path on CPU: two processes, gloo backend.

Three layers (the real engine needs a GPU: its two-rank run is tests/test_gpu_shards.py, ``-m gpu``):
  * ShardContext's collectives themselves (all_gather / all_reduce_ / allgather_cols / bounds / job_share) on host
    tensors;
  * the driver loop of litcoder_core_amd.nested_cv (fold phases, the order of the collectives, the host tail) with
    the device engine replaced by an oracle-backed stand-in that implements the engine's phase interface -- the
    result must equal the unsharded run on every rank, per-voxel alpha and single_alpha, CV and train/test.
"""
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
from litcoder_core_amd import stats
from litcoder_core_amd.dist import ShardContext, job_share, shard_bounds
import oracle.ridge as oridge

class OracleEngine:
    """The phase interface of RidgeCVEngine (what NestedCVModel's driver loop calls), arithmetic by the CPU oracle,
    exchanges through the same ShardContext."""
    primal = False
    def __init__(self, X, Y, alphas, normalpha, use_corr, nf, nt, shard, precision="auto", singcutoff=0.0, V_total=None,
                 min_train_rows=None, form="dual", panels=None, options=None, down_panels=None):
        Y = np.concatenate([b for _, b in Y.blocks], axis=0)              # ops.HostRows: the targets' row blocks
        self.X, self.Y = torch.tensor(np.asarray(X), dtype=torch.float32), torch.tensor(np.asarray(Y), dtype=torch.float32)
        self.alphas, self.normalpha, self.use_corr, self.shard = list(alphas), normalpha, use_corr, shard
        self.V = self.Y.shape[1]
        self.V_total = V_total or self.V
        self.W = np.zeros((self.X.shape[1], self.V), dtype=np.float32)
        self.W_acc = self.W
        self.alpha_fdr = 0.05
        self.p_rows = []
    def begin_fit(self, n_folds=1):
        self.p_rows = []
    def precompute_lmax(self, outer):
        return [None] * len(outer)
    def prepare_folds(self, folds, lmax_pre, chol_after=None):
        return [dict(tr=np.asarray(tr), te=np.asarray(te), inner=inner) for tr, te, inner in folds]
    def plan_steps(self, n_folds, single_alpha=False, ahead=False):
        return [(f, (0, self.V)) for f in range(n_folds)]
    def fold_begin(self, tr, te, inner, prepared=None, lmax_pre=None, step=None):
        return prepared
    def chain_gate(self):
        return None
    def fold_choose(self, st, single_alpha):
        Xtr, Ytr = self.X[st["tr"]], self.Y[st["tr"]]
        tot = 0
        for a, b in st["inner"]:
            tot = tot + oridge.alpha_sweep_scores(Xtr[a], Xtr[b], Ytr[a], Ytr[b], self.alphas, 1e-10, self.use_corr, self.normalpha)
        if single_alpha:
            total = self.shard.all_reduce_(tot.double().sum(dim=1), "sum")          # host tensor under gloo
            st["idx"] = np.full(self.V, int(torch.argmax(total)), dtype=np.int32)
        else:
            st["idx"] = tot.argmax(dim=0).numpy().astype(np.int32)
        return st
    def fold_select(self, st, single_alpha):
        count = torch.from_numpy(np.bincount(st["idx"], minlength=len(self.alphas)).astype(np.int32))
        st["used"] = [a for a in range(len(self.alphas)) if count[a] > 0]
        st["used_all"] = [a for a, c in enumerate(self.shard.all_reduce_(count.clone(), "sum").tolist()) if c > 0]
        assert set(st["used"]) <= set(st["used_all"])
        return st
    def fold_speculate(self, st, alphas_idx, early=False):
        pass
    def refit_ahead(self, states):
        pass
    def fold_finish(self, st, scale):
        tr, te, idx = st["tr"], st["te"], st["idx"]
        Xtr, Ytr = self.X[tr], self.Y[tr]
        W = oridge.ridge_weights(Xtr, Ytr, torch.tensor([self.alphas[i] for i in idx], dtype=torch.float32), 1e-10, self.normalpha)
        pred = (self.X[te] @ W).numpy().astype(np.float64); yt = self.Y[te].numpy().astype(np.float64)
        pc, yc = pred - pred.mean(0), yt - yt.mean(0)
        with np.errstate(all="ignore"):
            r = (pc * yc).sum(0) / np.sqrt((pc ** 2).sum(0) * (yc ** 2).sum(0))
        self.W += scale * W.numpy()
        p = stats.pearson_pvalues(r.astype(np.float32), len(te))
        rp = self.shard.allgather_cols(np.stack([r, p]), self.V_total)
        gi = self.shard.allgather_cols(idx[None, :], self.V_total)[0]
        p_clean = np.where(np.isnan(rp[0].astype(np.float32)), 1.0, rp[1])
        self.p_rows.append(p_clean)
        return ncv._FoldResult(rp[0], rp[1], gi, len(te), stats.fdrcorrection(p_clean, alpha=self.alpha_fdr))
    def fold_collect(self, pend):
        return pend
    def combined_significance(self):
        pc = stats.fisher_combine(np.stack(self.p_rows))
        sig, padj = stats.fdrcorrection(pc, alpha=self.alpha_fdr)
        return pc, sig, padj
    def reserve_host_weights(self):
        pass
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
    assert shard.world == 2 and shard.backend == "gloo"
    # ---- the collectives themselves, on host tensors
    r = shard.rank
    g = shard.all_gather(torch.full((2, 3), float(r)))
    assert g.shape == (2, 2, 3) and torch.equal(g[0], torch.zeros(2, 3)) and torch.equal(g[1], torch.ones(2, 3))
    t = torch.tensor([1.0 + r, 10.0 * (r + 1)], dtype=torch.float64)
    assert shard.all_reduce_(t.clone(), "sum").tolist() == [3.0, 30.0] and shard.all_reduce_(t.clone(), "max").tolist() == [2.0, 20.0]
    lo, hi = shard.bounds(37)
    cols = shard.allgather_cols(np.arange(lo, hi, dtype=np.float64)[None, :] * 2.0, 37)
    assert np.array_equal(cols[0], 2.0 * np.arange(37)) and list(shard.all_bounds(37)) == [0, 19, 37]
    assert shard.allreduce_sum(np.array([r + 1.0, 0.5])).tolist() == [3.0, 1.0]
    # local_targets: the job's total from the ranks' own block widths -- and a block that is not ShardContext.bounds' block is
    # refused on EVERY rank alike, from the one all-reduced vector (ADVICE r5: it used to raise on one rank and hang the others)
    assert shard.total_of_local_blocks(hi - lo) == 37
    try:
        shard.total_of_local_blocks((hi - lo) + (5 if shard.rank == 0 else 0))
        raise AssertionError("uneven caller-supplied blocks must be refused")
    except ValueError as e:
        assert "local_targets" in str(e)
    # ---- the driver loop
    m, W, a = ncv.NestedCVModel("r", shard=shard).fit_predict(X, Y, **kw)
    assert W.shape == (12, hi - lo)
    # train/test mode through the same sharded path
    m2, W2, a2 = ncv.NestedCVModel("r", shard=shard).fit_predict(X[:90], Y[:90], X_test=X[90:], y_test=Y[90:], **kw)
    # local lists: the V-long containers cover the rank's own voxels only, the scalars stay global
    shard_l = ShardContext(global_lists=False)
    m3, W3, a3 = ncv.NestedCVModel("r", shard=shard_l).fit_predict(X, Y, **kw)
    m4, W4, a4 = ncv.NestedCVModel("r", shard=shard_l).fit_predict(X[:90], Y[:90], X_test=X[90:], y_test=Y[90:], **kw)
    # ---- the story pipeline's share of a sharded job (harness.StoryPipeline, round 5): which columns of every story's brain
    # array a rank stages -- views of its block [lo, hi) when it is handed all voxels; with local_targets the array it is
    # handed IS its block and the job's voxel count comes from one all-reduce (trainer.py:235-257 z-scores per voxel)
    from litcoder_core_amd.harness import StoryPipeline
    rs = np.random.default_rng(5)
    brain = {s: rs.standard_normal((n, 37)) for s, n in (("s1", 40), ("s2", 31), ("s3", 25))}
    trim = dict(train_targets_start=3, train_targets_end=-2, test_targets_start=1, test_targets_end=None)
    names = list(brain)
    pipe = StoryPipeline([1, 2], trim, model=ncv.NestedCVModel("r", shard=shard))
    assert pipe._voxel_block(brain, names) == (lo, hi, 37)
    tg = pipe._targets(brain, names, [35, 26, 24], lo, hi)
    assert tg.zscore and tg.shape == (85, hi - lo) and [r0 for r0, _ in tg.blocks] == [0, 35, 61]
    assert all(np.shares_memory(b, brain[s]) for (_, b), s in zip(tg.blocks, names))            # views, no copies
    assert np.array_equal(tg.blocks[1][1], brain["s2"][3:-2, lo:hi]) and np.array_equal(tg.blocks[2][1], brain["s3"][1:, lo:hi])
    mine = {s: np.ascontiguousarray(b[:, lo:hi]) for s, b in brain.items()}
    pipe_l = StoryPipeline([1, 2], trim, model=ncv.NestedCVModel("r", shard=shard, local_targets=True))
    assert pipe_l._voxel_block(mine, names) == (0, hi - lo, 37)
    try:
        pipe._targets(brain, names, [35, 27, 24], lo, hi)
        raise AssertionError("a row count that differs from the features' must be refused")
    except RuntimeError:
        pass
    out = dict(m=m, W=W, a=a, lo=lo, hi=hi, m2=m2, a2=a2, m3=m3, a3=a3, m4=m4, a4=a4)
    pickle.dump(out, open(os.path.join(sys.argv[3], f"rank{shard.rank}_{mode}.pkl"), "wb"))
    dist.destroy_process_group()
    # a second world in the same process (notebooks, repeated jobs): the bulk lanes of the destroyed one must not be
    # reused (ADVICE r3) -- a gather on a lane and a whole sharded fit again
    # (a rendezvous of its own: a file store in the test's directory -- re-using the launcher's TCP store for a second
    # world in the same job proved flaky)
    dist.init_process_group("gloo", init_method="file://" + os.path.join(sys.argv[3], f"rendezvous2_{mode}"),
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    shard2 = ShardContext()
    g = shard2.all_gather(torch.full((3,), float(shard2.rank)), lane="hat")
    assert g[:, 0].tolist() == [0.0, 1.0]
    m5, W5, a5 = ncv.NestedCVModel("r", shard=shard2).fit_predict(X, Y, **kw)
    assert np.array_equal(W5, W) and np.array_equal(a5, a) and m5["median_score"] == m["median_score"]
    dist.destroy_process_group()
else:
    m, W, a = ncv.NestedCVModel("r").fit_predict(X, Y, **kw)
    m2, W2, a2 = ncv.NestedCVModel("r").fit_predict(X[:90], Y[:90], X_test=X[90:], y_test=Y[90:], **kw)
    pickle.dump(dict(m=m, W=W, a=a, m2=m2, a2=a2), open(os.path.join(sys.argv[3], f"single_{mode}.pkl"), "wb"))
'''


def test_job_share_and_bounds_partition():
    """Every job / voxel belongs to exactly one rank, slots of the gathered buffer are the job numbers."""
    from litcoder_core_amd.dist import ShardContext, job_share, shard_bounds
    for n in (0, 1, 4, 7, 20, 24, 100):
        for world in (1, 2, 3, 8):
            n_per = job_share(n, world, 0)[0]
            jobs = [j for r in range(world) for j in job_share(n, world, r)[1]]
            assert jobs == list(range(n)) and world * n_per >= n
            assert all(len(job_share(n, world, r)[1]) <= n_per for r in range(world))
            for r in range(world):
                assert all(j // n_per == r for j in job_share(n, world, r)[1])      # slot j = job j
            assert [shard_bounds(n, world, r)[1] for r in range(world)][-1] == n
    sim = ShardContext.simulated(8, 3)
    assert (sim.world, sim.rank, sim.simulate) == (8, 3, True) and sim.bounds(80000) == (30000, 40000)
    import torch
    g = sim.all_gather(torch.arange(4.0))
    assert g.shape == (8, 4) and torch.equal(g[5], torch.arange(4.0))                 # local copies, timing studies only


@pytest.mark.parametrize("mode", ["pervoxel", "single"])
def test_two_rank_shard_equals_unsharded(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    subprocess.run([sys.executable, str(script), ROOT, mode, str(tmp_path)], check=True, env=env, timeout=300)
    for attempt in range(3):                   # (a free port asked of the kernel; once more should the agent still lose it)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        done = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, mode,
                               str(tmp_path)], env=env, timeout=600)
        if done.returncode == 0:
            break
        for f in tmp_path.glob("rendezvous2_*"):
            f.unlink()                          # the second world's file store must start empty
    assert done.returncode == 0
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
        lo, hi = out["lo"], out["hi"]
        for key, rkey, akey in (("m3", "m", "a3"), ("m4", "m2", "a4")):          # ShardContext(global_lists=False)
            for k, v in ref[rkey].items():
                got = out[key][k]
                if isinstance(v, list):
                    assert np.array_equal(np.asarray(got), np.asarray(v)[lo:hi]), (key, k)
                else:
                    assert got == v, (key, k)
            assert np.array_equal(out[akey], ref["a" if key == "m3" else "a2"][lo:hi])
    assert ranks[0]["hi"] == ranks[1]["lo"] and ranks[1]["hi"] == 37
