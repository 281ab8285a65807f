"""Toy DeepSeek-shaped modules for the injection tests (HF-style names, no weights needed)."""
import torch
from torch import nn


class ToyConfig:
    def __init__(self, **kw):
        self.hidden_size = 256
        self.moe_intermediate_size = 128
        self.intermediate_size = 128
        self.n_routed_experts = 8
        self.num_experts_per_tok = 2
        self.n_shared_experts = None
        self.n_group = 2
        self.topk_group = 1
        self.scoring_func = "sigmoid"
        self.topk_method = "noaux_tc"
        self.norm_topk_prob = True
        self.routed_scaling_factor = 2.5
        self.hidden_act = "silu"
        self.__dict__.update(kw)


class ToyMLP(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.gate_proj = nn.Linear(h, i, bias=False)
        self.up_proj = nn.Linear(h, i, bias=False)
        self.down_proj = nn.Linear(i, h, bias=False)


class ToyGate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cfg.n_routed_experts, cfg.hidden_size))
        self.e_score_correction_bias = nn.Parameter(torch.empty(cfg.n_routed_experts))
        self.top_k = cfg.num_experts_per_tok


class ToyMoE(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.experts = nn.ModuleList([ToyMLP(cfg.hidden_size, cfg.moe_intermediate_size) for _ in range(cfg.n_routed_experts)])
        self.gate = ToyGate(cfg)


class ToyLayer(nn.Module):
    def __init__(self, cfg, moe):
        super().__init__()
        self.mlp = ToyMoE(cfg) if moe else ToyMLP(cfg.hidden_size, cfg.intermediate_size)
        self.tag = "layer"


class ToyInner(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([ToyLayer(cfg, moe=i > 0) for i in range(3)])


class ToyModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.model = ToyInner(cfg)
        self.lm_head = nn.Linear(cfg.hidden_size, 32, bias=False)
