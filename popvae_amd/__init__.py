"""popvae_amd -- MI355X-native engine for popVAE's VAE training hot path.

Python host code (this package) calls hand-written HIP kernels through a C ABI (include/pvae.h,
include/pvae_prep.h) loaded with ctypes.  There is no CPU fallback: if ``libpvae.so`` is missing or no
GPU is visible the engine raises.
"""
__version__ = "0.1.0"
