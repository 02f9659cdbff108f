"""Reference-generated config fixtures (tests/golden/configs.npz, made by tests/golden/make_golden_configs.py from the
reference itself) in the shape tests/_oracle_check.assert_matches_oracle takes an oracle fit in."""
import json
import os

import numpy as np

import _config_problems as cp
import oracle.folds as ofolds


def load(golden_dir):
    return np.load(os.path.join(golden_dir, "configs.npz")), json.load(open(os.path.join(golden_dir, "configs.json")))["fits"]


def reference_fit(g, tag, n_rows=None, n_outer=5):
    """((metrics-like, W[:, :32], alphas), detail) of the REFERENCE's fit ``tag`` (cfg2 | cfg4 | cfg5 | cfg3s | cfg3v).
    ``n_rows``: rows of a cross-validated fit (its outer K-folds are rebuilt: oracle.folds == the reference's, pinned by
    folds.json); None: a train/test fit."""
    m_o = {"correlations": g[f"{tag}__correlations"]}
    oracle = (m_o, g[f"{tag}__W"], g[f"{tag}__alphas"])
    if n_rows is None:
        detail = {"mean_scores": g[f"{tag}__fold_tables"][0], "test_scores": g[f"{tag}__fold_r"][0]}
    else:
        detail = {"fold_alphas": g[f"{tag}__fold_alphas"], "fold_mean_scores": g[f"{tag}__fold_tables"],
                  "fold_scores": g[f"{tag}__fold_r"], "outer": ofolds.create_folds(n_rows, "kfold", n_outer)}
    return oracle, detail


def check_inputs(g, key, *arrays):
    """The inputs rebuilt from seeds on this box are the ones the reference saw (float64 fingerprints, 1e-9)."""
    np.testing.assert_allclose(cp.checks(*arrays), g[key], rtol=1e-9, atol=1e-6, err_msg=f"{key}: rebuilt inputs differ")
