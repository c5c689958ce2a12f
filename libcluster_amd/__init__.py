"""libcluster_amd: MI355X (gfx950) implementation of libcluster's variational
E-step / sufficient-statistic hot path behind libcluster's own API.

The product is the C-ABI shared library (include/libcluster_hip.h, built from
libcluster_amd/csrc by `python -m libcluster_amd.build`).  This package is the
Python-side plumbing over it: `capi` (ctypes), `api` (learnVDP / learnBGMM /
learnGMC with the reference Python binding's return shape) and `dist`
(row-sharded multi-GPU driver over torch.distributed).  None of them contains
a CPU implementation of the data path.
"""
from . import capi  # noqa: F401
from .api import (learnBEMM, learnBGMM, learnDGMC, learnDGMM, learnEGMC, learnGMC, learnMCM,  # noqa: F401
                  learnSCM, learnSGMC, learnVDP)
