"""litcoder_core_amd -- MI355X-native implementation of LITcoder's nested-CV ridge fit path.

Public surface (names and signatures of the reference, ``/root/reference/encoding``):

    NestedCVModel, fit_nested_cv      encoding/models/nested_cv.py, README.md:212-226
    FIR                               encoding/features/FIR_expander.py
    Downsampler                       encoding/downsample/downsampling.py
    create_folds                      encoding/models/folding.py
    ModelSaver                        encoding/utils.py (on-disk result format)
    BandedNestedCVModel               (extension: per-band penalty scale, not in the reference)

All numerical work on the fit path is done by hand-written gfx950 HIP kernels in
``csrc/`` reached through the C ABI of ``include/litcoder_hip.h``; importing this package
does not need a GPU, calling it does (there is no CPU fallback).
"""
from .downsample import Downsampler
from .fir import FIR
from .folding import create_folds
from .nested_cv import BasePredictivityModel, NestedCVModel, fit_nested_cv
from .dist import ShardContext, shard_bounds
from .harness import StoryPipeline
from .saver import ModelSaver
from .banded import BandedNestedCVModel

__all__ = ["NestedCVModel", "fit_nested_cv", "FIR", "Downsampler", "create_folds", "BasePredictivityModel",
           "ShardContext", "shard_bounds", "StoryPipeline", "ModelSaver", "BandedNestedCVModel"]
