"""cuopt_amd: MI355X-native PDLP linear-programming solver behind the libcuopt C API.

The product is the shared library cuopt_amd/lib/libcuopt.so (C API + C++ host driver + HIP/gfx950
kernels).  This Python package only holds the ctypes mirror of its C headers (`capi`) and the
synthetic LP generator used by the benchmark and the tests (`synthetic`)."""
__version__ = "0.1.0"
