"""Drop-in for the reference's live Python package `kt_kernel` (kt-kernel/python/__init__.py): the expert wrapper SGLang
drives, backed by the HBM-resident HIP experts of this library instead of the CPU worker pool.

    from ktransformers_amd.kt_kernel import KTMoEWrapper        # was: from kt_kernel import KTMoEWrapper
"""
from .experts import INFERENCE_METHODS, SUPPORTED_METHODS, KTMoEWrapper
from .experts_base import BaseMoEWrapper, KExpertsDeviceBuffer, generate_gpu_experts_masks

__all__ = ["KTMoEWrapper", "BaseMoEWrapper", "KExpertsDeviceBuffer", "generate_gpu_experts_masks", "INFERENCE_METHODS", "SUPPORTED_METHODS"]
