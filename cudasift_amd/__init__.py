"""cudasift_amd — MI355X-native SIFT extraction + matching behind the CudaSift API.

The product is the C-ABI library `libmisift.so` (hand-written gfx950 HIP kernels,
include/misift.h) plus the C++ drop-in shim `libcudasift.so` (include/cudaSift.h,
include/cudaImage.h).  This Python package is only the ctypes binding used by the
tests and bench.py: `capi` (raw C-ABI over ctypes; `capi.HostComm` drives the library's own gather code over a
caller's transport, which is how the gloo tests run it without a GPU).
"""
from . import capi  # noqa: F401
