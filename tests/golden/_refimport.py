"""Container-only helper: import the reference's hot-path modules from
/root/reference without touching it.  Used ONLY by make_golden.py (and by the
optional cross-check test that skips when /root/reference is absent).  Nothing
here travels to the GPU box as reference code: it only seeds ``sys.modules``
with empty stand-ins for third-party packages the reference's package
``__init__`` files import eagerly but the hot path never uses.
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "encoding"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install(fdr):
    """``fdr(pvals, alpha) -> (reject, p_adj)`` stands in for the absent
    statsmodels ``fdrcorrection`` (unpinned boundary)."""
    for name in ("transformer_lens", "gensim", "gensim.models", "h5py", "wandb", "nibabel", "nilearn",
                 "nilearn.plotting", "nilearn.plotting.cm", "nilearn.datasets", "nilearn.surface", "seaborn",
                 "statsmodels", "statsmodels.stats"):
        _stub(name)
    sys.modules["transformer_lens"].HookedTransformer = type("HookedTransformer", (), {})
    sys.modules["gensim.models"].KeyedVectors = type("KeyedVectors", (), {})
    sys.modules["nilearn.plotting.cm"].cold_hot = None
    _stub("statsmodels.stats.multitest", fdrcorrection=lambda pvals, alpha=0.05, **kw: fdr(pvals, alpha))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def load(fdr):
    """Returns a namespace with the reference's hot-path callables."""
    install(fdr)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        from encoding.models import nested_cv, ridge_regression, ridge_utils, folding
        from encoding.features import FIR_expander
        from encoding.downsample import downsampling, interpdata
        from encoding import utils as ref_utils
        from encoding import trainer as ref_trainer
    ns = types.SimpleNamespace(nested_cv=nested_cv, ridge_regression=ridge_regression, ridge_utils=ridge_utils,
                               folding=folding, FIR_expander=FIR_expander, downsampling=downsampling,
                               interpdata=interpdata, utils=ref_utils, trainer=ref_trainer)
    return ns
