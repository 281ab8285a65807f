"""ktransformers_amd — MI355X-native (gfx950) quantized-MoE + MLA hot path behind KTransformers' operator surface.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels + the C ABI (include/*.h)
  _native.py ctypes binding of libktx_hip.so (fails loudly when the library is missing)
  operators/ host-side mirror of the reference's injection surface (KExperts*, KTransformersExperts, ...)
"""
__version__ = "0.1.0"
