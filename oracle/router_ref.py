"""TEST INFRASTRUCTURE ONLY — torch (CPU, fp32) restatement of the reference router, line for line:
  V3 / K2 : archive/ktransformers/models/modeling_deepseek_v3.py:430-481  (MoEGate.forward, sigmoid + noaux_tc)
  V2      : archive/ktransformers/models/modeling_deepseek.py:413-455     (softmax + greedy / group_limited_greedy)
Returned index ORDER is torch.topk(sorted=False)'s, i.e. implementation-defined: compare as sets."""
import torch
import torch.nn.functional as F


def moe_gate_ref(x, weight, bias, *, top_k, n_group, topk_group, scoring_func, topk_method, norm_topk_prob,
                 routed_scaling_factor):
    n = x.shape[0]
    logits = F.linear(x.type(torch.float32), weight.type(torch.float32), None)
    if scoring_func == "sigmoid":
        scores = logits.sigmoid()
    elif scoring_func == "softmax":
        scores = logits.softmax(dim=-1, dtype=torch.float32)
    else:
        raise NotImplementedError(scoring_func)
    E = scores.shape[-1]
    if topk_method == "noaux_tc":
        scores_for_choice = scores.view(n, -1) + bias.unsqueeze(0)
        group_scores = scores_for_choice.view(n, n_group, -1).topk(2, dim=-1)[0].sum(dim=-1)
        group_idx = torch.topk(group_scores, k=topk_group, dim=-1, sorted=False)[1]
        group_mask = torch.zeros_like(group_scores)
        group_mask.scatter_(1, group_idx, 1)
        score_mask = group_mask.unsqueeze(-1).expand(n, n_group, E // n_group).reshape(n, -1)
        tmp_scores = scores_for_choice.masked_fill(~score_mask.bool(), float("-inf"))
        _, topk_idx = torch.topk(tmp_scores, k=top_k, dim=-1, sorted=False)
        topk_weight = scores.gather(1, topk_idx)
        if top_k > 1 and norm_topk_prob:
            topk_weight = topk_weight / (topk_weight.sum(dim=-1, keepdim=True) + 1e-20)
        topk_weight = topk_weight * routed_scaling_factor
        return topk_idx, topk_weight
    if topk_method == "greedy":
        topk_weight, topk_idx = torch.topk(scores, k=top_k, dim=-1, sorted=False)
    elif topk_method == "group_limited_greedy":
        group_scores = scores.view(n, n_group, -1).max(dim=-1).values
        group_idx = torch.topk(group_scores, k=topk_group, dim=-1, sorted=False)[1]
        group_mask = torch.zeros_like(group_scores)
        group_mask.scatter_(1, group_idx, 1)
        score_mask = group_mask.unsqueeze(-1).expand(n, n_group, E // n_group).reshape(n, -1)
        tmp_scores = scores.masked_fill(~score_mask.bool(), 0.0)
        topk_weight, topk_idx = torch.topk(tmp_scores, k=top_k, dim=-1, sorted=False)
    else:
        raise NotImplementedError(topk_method)
    if top_k > 1 and norm_topk_prob:
        topk_weight = topk_weight / (topk_weight.sum(dim=-1, keepdim=True) + 1e-20)
    else:
        topk_weight = topk_weight * routed_scaling_factor
    return topk_idx, topk_weight
