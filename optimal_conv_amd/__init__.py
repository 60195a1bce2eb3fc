"""optimal_conv_amd — MI355X (gfx950) engine for the homomorphic-convolution hot path of dwkim606/optimal_conv.

The product is libhconv.so (HIP kernels + C ABI, include/hconv.h) and the C++ host side under host/.
This Python package only binds the C ABI for tests and bench.py; it has no CPU fallback.
"""
from .abi import Context, DevBuf, HconvError, SYMBOLS, load  # noqa: F401
