"""CPU oracle for the LITcoder nested-CV ridge hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``litcoder_core_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / the timed CPU
baseline -- never as the thing shipped.

It is a restatement, in this repo's own words, of the reference algorithm
(``/root/reference/encoding/...``; every function cites the file:line it
follows).  The reference is pure Python on torch-CPU/numpy/scipy/sklearn, so
the oracle is too (fp32 torch-CPU where the reference computes in torch,
float64 numpy where it computes in numpy): there is no C/C++ reference source
to compile, hence no ``oracle/_ref`` build.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(reference imported unmodified with third-party import stubs) and committed as
``tests/golden/*.npz``.  One boundary stays unpinned and says so:
``statsmodels.stats.multitest.fdrcorrection`` is not installed anywhere, so the
Benjamini-Hochberg step (``stats.bh_fdr``) is pinned only by hand-checkable
known-answer vectors ("parity unpinned" for that step).
"""

from . import fir, folds, lanczos, ridge, stats, nested_cv, harness, banded  # noqa: F401
