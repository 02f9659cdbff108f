"""The parts of RidgeCVEngine (assembled in litcoder_core_amd/nested_cv.py)."""
from .common import FitOptions, check_penalties  # noqa: F401
