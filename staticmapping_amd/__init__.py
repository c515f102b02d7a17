"""staticmapping_amd -- MI355X (gfx950) scan-matching backend for StaticMapping's
per-frame registration hot path (registrators::IcpFast / Ndt).

The product is the C-ABI shared library `staticmapping_amd/lib/libsmhip.so`
(include/smhip.h) plus the C++ `registrator::Interface` mirror in
include/smhip/registrator.h.  This Python package is plumbing around it: a ctypes
view for tests / bench, the synthetic Velodyne-64 workload generator, and the
scan-pair sharding helper built on torch.distributed (RCCL).
"""
from . import synth  # noqa: F401
from .matcher import (IcpFastHip, IcpPointMatcherHip, NdtGicpHip, NdtHip, SmhipError, se3_error, calculate_normals,  # noqa: F401
                      NN_BRUTE, NN_GRID, NN_NABO)
from .mrvm import MultiResolutionVoxelMapHip  # noqa: F401
