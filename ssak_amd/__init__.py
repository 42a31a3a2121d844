"""ssak_amd -- MI355X-native acoustic-model hot path for SSAK (linto-ai/ssak).

The package is a thin Python host layer (the reference is Python) over ``libssak_hip.so``:
hand-written HIP kernels for gfx950 behind the C ABI of ``include/ssak_hip.h``.  There is no
CPU fallback: importing :mod:`ssak_amd.hip` without the built library raises.
"""
__version__ = "0.1.0"
