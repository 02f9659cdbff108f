"""The sharded fit with the REAL engine (``-m gpu``).

* Two ranks on the one GPU of the box: two fresh processes (torch.distributed.run, gloo backend for the exchanges --
  RCCL refuses two ranks on one device), each with its own RidgeCVEngine on cuda:0 fitting its block of voxel
  columns, the (fold, alpha) Cholesky systems dealt out over the two (``RidgeCVEngine._sharded_solve``) and
  all-gathered.  Every rank must return the unsharded fit BIT FOR BIT: metrics of all voxels, chosen alphas, and its
  own block of the weights -- per-voxel alpha and single_alpha, full CV and train/test, the moments path and the
  per-alpha hat-matrix paths (R2 scoring, raw alphas, f32 sweep), and the primal form of a tall design.
* One rank through RCCL: a one-rank "nccl" group with ``always_collective`` runs the same code with every collective
  issued as a real RCCL call on device tensors (all_gather_into_tensor of f32 / f64 blocks, all_reduce of f64 and int32
  vectors, on the engine's auxiliary / communication streams): the calls the 8-GPU run makes, checked for API /
  dtype / stream-ordering mistakes on the hardware we have; results identical to the plain fit.
"""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, pickle, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from litcoder_core_amd import NestedCVModel, ShardContext
from litcoder_core_amd.engine.common import FitOptions
backend, out_dir = sys.argv[2], sys.argv[3]
torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
rng = np.random.default_rng(3)
T, p, V = 420, 96, 777
X = rng.standard_normal((T, p))
Y = X @ (rng.standard_normal((p, V)) * (0.3 / np.sqrt(p)) * rng.uniform(0.0, 3.0, V)) + rng.standard_normal((T, V))
Y[:, 5] = 1.5                                     # constant voxel: NaN r -> (0, p = 1)
cases = {
    "pervoxel": dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 5, 8)),
    "single": dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 5, 8), single_alpha=True),
    "rawalpha_r2": dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=np.logspace(0, 3, 5), normalpha=False,
                        use_corr=False),
    "one_alpha": dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=[0.5]),   # one refit system: row slices
    "normalised": dict(folding_type="chunked_contiguous", n_outer_folds=3, n_inner_folds=2, chunk_length=10,
                       alphas=np.logspace(-1, 4, 6), normalize_features=True, normalize_targets=True),
    "tall": dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=2, alphas=np.logspace(-1, 3, 5), _p=8),   # primal form
    # a column too wide for the fp16 split in the LAST rank's block only: precision "auto" must take the same arithmetic
    # (hence the same collectives) on every rank -- the flag is all-reduced before anybody acts on it (ADVICE r2); since
    # round 5 that column alone is recomputed in f32 by the rank that holds it
    "outlier": dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 5, 8), _spike=(7, 700)),
    # factorised alphas on BOTH sides of FitOptions.refit_inverse_min_alpha (0.1): a sharded fit solves the refit systems of
    # all of them ahead, an unsharded one those in use -- each alpha must take the route (explicit inverse / augmented
    # solves) IT qualifies for, whatever its companions (round 5's shard fuzzing: the route was decided per list)
    "straddle": dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=[0.02, 0.06, 0.3, 2.0, 300.0]),
    # ... and a grid where NO factorised alpha reaches FitOptions.refit_ahead_min_alpha (0.15) while one (0.12) is above
    # refit_inverse_min_alpha: the sharded fit must leave 0.12 to the route the unsharded fit takes (ADVICE r5)
    "straddle_low": dict(folding_type="kfold", n_outer_folds=2, n_inner_folds=2, alphas=[0.05, 0.12, 10.0]),
    # the two-precision inner CV (round 6; on by default in every case above) with a refinement panel too small for the
    # undecided voxels of a wide block and a generous gap: a block of > 256 voxels overflows the panel and is scored again,
    # a block of <= 256 never does -- the ranks of a sharded fit take that decision TOGETHER (MAX all-reduce of the flag:
    # the choice that follows all-reduces its histogram), and the fit is the same whichever way its blocks went
    "screen_overflow": dict(folding_type="kfold", n_outer_folds=3, n_inner_folds=3, alphas=np.logspace(-1, 5, 8),
                            _opts=dict(screen_tau=0.2, screen_panel_cols=256)),
}
shard = None
if backend != "none":
    dist.init_process_group(backend, rank=int(os.environ.get("RANK", "0")), world_size=world,
                            **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
    shard = ShardContext(device=torch.device("cuda", 0), always_collective=(world == 1))
    assert shard.active
lo, hi = (shard.bounds(V) if shard else (0, V))
out = {"lo": lo, "hi": hi, "direct_rccl": bool(shard is not None and shard._comms),
       "rccl_lanes": bool(shard is not None and len(shard._comms) > 1)}
for name, kw in cases.items():
    kw = dict(kw)
    Xc = X[:, : kw.pop("_p", p)]
    spike = kw.pop("_spike", None)
    opts = kw.pop("_opts", None)
    Yc = Y
    if spike is not None:
        Yc = Y.copy()
        Yc[spike] = 1e6
    for prec in (("auto", "f32") if name == "pervoxel" else ("auto",)):
        model = NestedCVModel("r", shard=shard, precision=prec, options=FitOptions(**opts) if opts else None)
        out[name, prec, "cv"] = model.fit_predict(Xc, Yc, **kw)
        assert model.last_form == ("primal" if name == "tall" else "dual")
        if name == "pervoxel" and prec == "auto":
            assert model.last_fit["screen_terms"] == 1 and model.last_fit.get("screen_overflows", 0) == 0, model.last_fit
        if opts:
            out["screen_overflows"] = model.last_fit.get("screen_overflows", 0)
        if spike is not None:
            # (round 5: the one wide column goes through the f32 side path of the rank that holds it; every rank keeps the
            # fp16x3 arithmetic -- and agrees on that together, the decision is all-reduced)
            assert model.last_fit["precision"] == "f16x3", "every rank takes the same arithmetic"
            assert model.last_fit["side_panel_cols"] == int(lo <= spike[1] < hi)
        kw_tt = {k: v for k, v in kw.items() if k != "n_outer_folds"}
        out[name, prec, "tt"] = model.fit_predict(Xc[:330], Yc[:330], X_test=Xc[330:], y_test=Yc[330:], **kw_tt)
# ---- the story pipeline (harness.StoryPipeline.fit_words: BASELINE configs[2]'s route) under the same shard context: word
# features -> Lanczos -> FIR -> per-story zs -> train/test fit in the primal form with shared series terms (p_pad = 256,
# 2 p <= the inner training sets), the brain data of every rank's voxel block z-scored in ITS upload threads, panel by
# panel (panel_cols = 256: 3-4 upload panels per rank); single_alpha (the per-alpha sums of all panels and all ranks
# added on the device, the early panels refitted with the alpha they choose) and per-voxel alpha; ``local_targets``:
# every rank is handed its own block of the brain data only
from litcoder_core_amd import StoryPipeline
rs = np.random.default_rng(11)
Vs_, D = 2600, 64
n_trs = [150, 170, 160, 180, 165, 175, 155, 120]
Wt = rs.standard_normal((4 * D, Vs_)) * 0.05 * rs.uniform(0.0, 2.0, Vs_)
words, wtimes, trtimes, brain = {}, {}, {}, {}
for i, n_tr in enumerate(n_trs):
    nm = "s%d" % i
    nw = int(6.5 * n_tr)
    wtimes[nm] = np.sort(rs.uniform(0, 2.0 * (n_tr + 15), nw))
    emb = rs.standard_normal((nw, D)).astype(np.float32)
    emb[1:] = 0.6 * emb[:-1] + 0.8 * emb[1:]
    words[nm] = emb
    trtimes[nm] = 1.0 + 2.0 * np.arange(n_tr + 15)
    brain[nm] = 3.0 * (rs.standard_normal((n_tr, 4 * D)) @ Wt + rs.standard_normal((n_tr, Vs_))) + 100.0
brain["s2"][:, 9] = 7.0                            # a voxel that is constant within one story
TRIM = {"train_features_start": 10, "train_features_end": -5, "train_targets_start": 0, "train_targets_end": None,
        "test_features_start": 10, "test_features_end": -5, "test_targets_start": 0, "test_targets_end": None}
slo, shi = (shard.bounds(Vs_) if shard else (0, Vs_))
out["story_lo"], out["story_hi"] = slo, shi
for name, kw, local in (("story_single", dict(single_alpha=True), False), ("story_pervoxel", dict(single_alpha=False), False),
                        ("story_single_local", dict(single_alpha=True), True)):
    model = NestedCVModel("r", shard=shard, panel_cols=256, local_targets=local)
    pipe = StoryPipeline([1, 2, 3, 4], TRIM, model=model)
    data = {k: v[:, slo:shi] for k, v in brain.items()} if (local and shard is not None) else brain
    out["story", name] = pipe.fit_words(words, wtimes, trtimes, data, folding_type="kfold", n_inner_folds=5,
                                        alphas=np.logspace(-1, 8, 10), **kw)
    assert model.last_form == "primal" and model.last_fit["series_terms"] == 4, model.last_fit
    assert len(model.last_fit["panels"]) >= 3, model.last_fit["panels"]
    if kw["single_alpha"]:
        assert model.last_fit.get("single_alpha_guess") in ("held", "not decisive"), model.last_fit
    out["story_info", name] = {k: model.last_fit.get(k) for k in ("single_alpha_guess", "single_alpha_lead", "panels")}
tag = "single" if backend == "none" else f"{backend}{world}_rank{shard.rank}" + ("_direct" if out["direct_rccl"] else "")
pickle.dump(out, open(os.path.join(out_dir, tag + ".pkl"), "wb"))
if shard is not None:
    shard.close()
    dist.destroy_process_group()
'''


def _same(got, want, lo, hi, what):
    (m, W, al), (m0, W0, al0) = got, want
    assert sorted(m) == sorted(m0), what
    for k, v in m0.items():
        if isinstance(v, list):
            assert np.array_equal(np.asarray(m[k]), np.asarray(v), equal_nan=True), (what, k)
        else:
            assert m[k] == v or (v != v and m[k] != m[k]), (what, k)
    assert np.array_equal(al, al0), what
    assert W.shape == (W0.shape[0], hi - lo) and np.array_equal(W, W0[:, lo:hi]), what


def _free_port():
    """A TCP port the kernel just handed out (derived-from-the-pid ports collide now and then on a shared host)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    d = tmp_path_factory.mktemp("shards")
    script = d / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500), OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, str(script), ROOT, "none", str(d)], check=True, env=env, timeout=900)
    return d, script, env, pickle.load(open(d / "single.pkl", "rb"))


def test_two_real_engine_ranks_on_one_gpu_equal_the_unsharded_fit(runs):
    d, script, env, ref = runs
    port = _free_port()
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                    "127.0.0.1", "--master-port", str(port), str(script), ROOT, "gloo", str(d)], check=True, env=env,
                   timeout=1500)
    ranks = [pickle.load(open(d / f"gloo2_rank{r}.pkl", "rb")) for r in range(2)]
    assert (ranks[0]["lo"], ranks[0]["hi"], ranks[1]["lo"], ranks[1]["hi"]) == (0, 389, 389, 777)
    n = 0
    for key, want in ref.items():
        if not isinstance(key, tuple) or key[0] in ("story", "story_info"):
            continue
        for r, out in enumerate(ranks):
            _same(out[key], want, out["lo"], out["hi"], (key, r))
            n += 1
    assert n == 2 * 2 * 11
    assert ref["screen_overflows"] >= 1, "the unsharded fit must have gone through the overflow path"
    _stories_same(ranks, ref, [(0, 1300), (1300, 2600)])


def test_three_ranks_uneven_shares(runs):
    """Three ranks: 777 voxels split 259 / 259 / 259, job counts that do not divide by three (ranks with one job fewer
    or none at all, row-sliced refit systems with an idle rank) -- still the unsharded fit bit for bit on every rank."""
    d, script, env, ref = runs
    port = _free_port()
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
                    "127.0.0.1", "--master-port", str(port), str(script), ROOT, "gloo", str(d)], check=True, env=env,
                   timeout=1500)
    ranks = [pickle.load(open(d / f"gloo3_rank{r}.pkl", "rb")) for r in range(3)]
    assert [(o["lo"], o["hi"]) for o in ranks] == [(0, 259), (259, 518), (518, 777)]
    for key, want in ref.items():
        if isinstance(key, tuple) and key[0] not in ("story", "story_info"):
            for r, out in enumerate(ranks):
                _same(out[key], want, out["lo"], out["hi"], (key, r))
    _stories_same(ranks, ref, [(0, 867), (867, 1734), (1734, 2600)])


def _stories_same(ranks, ref, blocks):
    """harness.StoryPipeline.fit_words on every rank (its voxel block of every story, z-scored in its own upload threads,
    panel by panel) equals the one-GPU pipeline bit for bit: metrics of all voxels, alphas, its block of the weights --
    single_alpha (panelled joint choice, all-reduced sums) and per-voxel alpha, whole brain arrays or local blocks."""
    assert [(o["story_lo"], o["story_hi"]) for o in ranks] == blocks
    n = 0
    for key, want in ref.items():
        if isinstance(key, tuple) and key[0] == "story":
            base = ref["story", key[1].replace("_local", "")]
            _same(want, base, 0, blocks[-1][1], (key, "one GPU, local == global"))
            for r, out in enumerate(ranks):
                _same(out[key], want, out["story_lo"], out["story_hi"], (key, r))
                n += 1
    assert n == 3 * len(ranks)
    # every rank took the same decision about the early panels' alpha (it is taken from all-reduced sums)
    for name in ("story_single", "story_single_local"):
        tags = {o["story_info", name]["single_alpha_guess"] for o in ranks}
        assert len(tags) == 1, tags


def test_one_rank_through_rccl_collectives(runs):
    """torch.distributed's own RCCL calls (LITCODER_AMD_RCCL_DIRECT=0: the fallback transport since round 6)."""
    d, script, env, ref = runs
    env = dict(env, RANK="0", WORLD_SIZE="1", MASTER_PORT=str(29950 + os.getpid() % 40), LITCODER_AMD_RCCL_DIRECT="0")
    subprocess.run([sys.executable, str(script), ROOT, "nccl", str(d)], check=True, env=env, timeout=900)
    out = pickle.load(open(d / "nccl1_rank0.pkl", "rb"))
    for key, want in ref.items():
        if isinstance(key, tuple) and key[0] != "story_info":
            _same(out[key], want, 0, out["story_hi"] if key[0] == "story" else out["hi"], key)


def test_one_rank_through_the_librarys_own_rccl_wrappers(runs):
    """The same one-rank run on the DEFAULT transport under "nccl" (round 6; LITCODER_AMD_RCCL_DIRECT unset), then once more
    with LITCODER_AMD_RCCL_LANES=0 (every collective on one communicator): every device-tensor collective of the sharded fit is the
    library's own RCCL call (lc_allgather_f32 / lc_allgather_bytes / lc_allreduce on communicators made by lc_comm_create from
    a unique id that torch.distributed only hands round; include/litcoder_hip.h) instead of torch.distributed's --
    the results are the plain fit's, bit for bit."""
    d, script, env, ref = runs
    for lanes in ("1", "0"):
        env_ = {k: v for k, v in env.items() if k != "LITCODER_AMD_RCCL_DIRECT"}
        env_.update(RANK="0", WORLD_SIZE="1", MASTER_PORT=str(_free_port()), LITCODER_AMD_RCCL_LANES=lanes)
        subprocess.run([sys.executable, str(script), ROOT, "nccl", str(d)], check=True, env=env_, timeout=900)
        out = pickle.load(open(d / "nccl1_rank0_direct.pkl", "rb"))
        assert out["direct_rccl"] and out["rccl_lanes"] == (lanes == "1")
        for key, want in ref.items():
            if isinstance(key, tuple) and key[0] != "story_info":
                _same(out[key], want, 0, out["story_hi"] if key[0] == "story" else out["hi"], key)
        os.remove(d / "nccl1_rank0_direct.pkl")


def test_bench_runs_on_two_ranks_of_one_gpu(tmp_path):
    """The N > 1 path of bench.py itself (what the driver launches on an 8-GPU node: one process per GPU under
    torch.distributed.run, barrier + MAX-over-ranks timing, per-rank inputs, the other scaling mode beside the headline)
    on the hardware there is: two ranks on the one GPU, LITCODER_BENCH_ONE_GPU=1 (every rank on device 0, gloo for the
    exchanges -- RCCL refuses two ranks on one device).  The numbers mean nothing; the line must be complete, the job's
    voxel count right in both scaling modes, the scores sane -- so that this path cannot rot unmeasured."""
    import json
    env = dict(os.environ, LITCODER_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    for k, scaling in enumerate(("weak", "strong")):
        port = _free_port()
        # the default at N > 1 IS the strong-scaled job (BASELINE's configs fix the voxel total): no flag in that run, which
        # also carries the cfg3 story pipeline sharded over the two ranks
        flags = ["--scaling", "weak", "--no-cfg3"] if scaling == "weak" else []
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                            "--warmup", "1", "--voxels", "6144", "--no-cpu-baseline"] + flags,
                           env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["unit"] == "voxels/sec" and out["value"] > 0
        total = 2 * 6144 if scaling == "weak" else 6144
        assert out["config"]["voxels_total"] == total and abs(out["value"] - total / (1e-3 * out["ms_per_step"])) < 1e-6 * out["value"]
        other = out["other_scaling"]
        assert other["scaling"] == ("strong" if scaling == "weak" else "weak") and other["voxels_total"] == (6144 if scaling == "weak" else 2 * 6144)
        assert 0.2 < out["config"]["median_score"] < 0.6 and 0.2 < other["median_score"] < 0.6
        assert out["roofline"]["launches"] > 0 and "cpu_baseline" not in out
        dist_ = out["ms_per_step_distribution"]
        assert dist_["steps"] == 1 and dist_["min"] <= dist_["median"] <= dist_["max"]
        if scaling == "weak":
            assert "cfg3_pipeline" not in out
        else:
            c3 = out["cfg3_pipeline"]
            assert c3["n_gpus"] == 2 and c3["voxels_total"] == 6144 and c3["voxels_rank0"] == 3072 and c3["form"] == "primal"
            assert abs(c3["value"] - 6144 / (1e-3 * c3["ms_per_step"])) < 1e-6 * c3["value"] and 0.0 < c3["median_score"] < 0.9
            assert "in total" in out["config"]["workload"] and "split over 2 GPUs" in out["config"]["workload"]


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 --steps 1 --warmup 0` with NO launcher around it (the form of the driver's bench command;
    VERDICT r5 #2): the process starts its two ranks itself, relays rank 0's ONE JSON line and returns 0.  Two ranks on the
    one GPU (LITCODER_BENCH_ONE_GPU=1, gloo exchanges): the numbers mean nothing, the line and the return code do."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(LITCODER_BENCH_ONE_GPU="1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--voxels", "6144", "--no-cfg3", "--no-extra-legs"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo" and out["value"] > 0
    assert len(out["ms_per_step_by_rank"]) == 2 and max(out["ms_per_step_by_rank"]) == pytest.approx(out["ms_per_step"], rel=1e-9)
    assert out["scaling"] == "strong" and out["config"]["voxels_total"] == 6144
