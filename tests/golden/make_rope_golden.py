"""Generates tests/golden/rope_golden.npz: cos / sin of the REFERENCE's own rotary modules
(archive/ktransformers/models/modeling_deepseek_v3.py DeepseekV3YarnRotaryEmbedding :270-342 and DeepseekV3RotaryEmbedding
:199-239, imported here) for the DeepSeek-V3 and V2-Lite YaRN settings and a plain-RoPE setting.  Build container only."""
import os

import numpy as np
import torch

from ref_import import reference_models

v3, _ = reference_models()
out = {}
pos = torch.tensor([0, 1, 2, 63, 64, 1000, 4095, 4096, 40000, 163839])
cases = {
    "v3": dict(dim=64, base=10000.0, factor=40, original_max_position_embeddings=4096, beta_fast=32, beta_slow=1, mscale=1.0, mscale_all_dim=1.0),
    "v2lite": dict(dim=64, base=10000.0, factor=40, original_max_position_embeddings=4096, beta_fast=32, beta_slow=1, mscale=0.707, mscale_all_dim=0.707),
    "odd": dict(dim=64, base=50000.0, factor=8, original_max_position_embeddings=2048, beta_fast=16, beta_slow=2, mscale=1.0, mscale_all_dim=0.0),
}
for name, c in cases.items():
    m = v3.DeepseekV3YarnRotaryEmbedding(c["dim"], max_position_embeddings=163840, base=c["base"], scaling_factor=c["factor"],
                                         original_max_position_embeddings=c["original_max_position_embeddings"], beta_fast=c["beta_fast"],
                                         beta_slow=c["beta_slow"], mscale=c["mscale"], mscale_all_dim=c["mscale_all_dim"])
    cos, sin = m(torch.zeros(1, dtype=torch.float32), seq_len=163840)
    out[f"{name}_cos"], out[f"{name}_sin"] = cos[pos].float().numpy(), sin[pos].float().numpy()
    out[f"{name}_cfg"] = np.array([c[k] for k in ("dim", "base", "factor", "original_max_position_embeddings", "beta_fast", "beta_slow", "mscale", "mscale_all_dim")], np.float64)
m = v3.DeepseekV3RotaryEmbedding(64, max_position_embeddings=8192, base=10000.0)
cos, sin = m(torch.zeros(1, dtype=torch.float32), seq_len=8192)
p2 = pos[pos < 8192]
out["plain_cos"], out["plain_sin"], out["plain_pos"] = cos[p2].float().numpy(), sin[p2].float().numpy(), p2.numpy()
out["pos"] = pos.numpy()
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rope_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
