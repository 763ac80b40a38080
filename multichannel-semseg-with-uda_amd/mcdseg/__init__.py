"""mcdseg -- Python binding of libmcdseg.so (hand-written HIP kernels for gfx950 / MI355X).

The binding is ctypes over the C ABI declared in ``include/mcdseg.h``; tensors are handed over as
raw device pointers plus the current HIP stream.  There is no CPU or eager-PyTorch fallback: every
op raises if the shared library is missing or a tensor is not on the GPU.
"""
from ._lib import LIB_PATH, build, get_option, lib, option_default, option_names, options, set_option  # noqa: F401
