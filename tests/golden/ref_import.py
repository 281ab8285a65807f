"""Import the reference's pure-torch model code (/root/reference/archive/ktransformers/models) in the build container.
transformers here is newer than the reference expects; two symbols it imports only for feature probing are stubbed."""
import sys


def reference_models():
    sys.path.insert(0, "/root/reference/archive")
    import transformers.utils as tu
    import transformers.utils.import_utils as iu

    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    if not hasattr(tu, "is_flash_attn_greater_or_equal_2_10"):
        tu.is_flash_attn_greater_or_equal_2_10 = lambda *a, **k: False
    from ktransformers.models import modeling_deepseek_v3 as v3
    from ktransformers.models.configuration_deepseek_v3 import DeepseekV3Config
    return v3, DeepseekV3Config
