"""TEST INFRASTRUCTURE ONLY — fp32 torch restatement of the reference's MLA oracle attention_ref_torch
(archive/ktransformers/operators/flashinfer_wrapper.py:30-76), applied per request to a paged latent cache.
PINNED (round 6): tests/golden/triton_golden.npz holds outputs of the reference's Triton MLA decode (decode_attention_fwd_grouped,
operators/triton_attention.py:358-385: 4 KV splits + log-sum-exp merge over a permuted page table) run on the CPU by Triton's
interpreter (tests/golden/make_triton_golden.py); tests/test_triton_pin_cpu.py holds this file to 2e-6 of them (fp32 operands)."""
import math

import torch


def attention_ref_torch(batch_size, q, k, v, causal, sm_scale):
    qo_len = q.shape[0] // batch_size
    kv_len = k.shape[0] // batch_size
    num_qo_heads, head_dim_qk, head_dim_vo = q.shape[1], q.shape[2], v.shape[2]
    logits = torch.einsum("bmhd,bnhd->bhmn", q.view(batch_size, qo_len, num_qo_heads, head_dim_qk).float(),
                          k.view(batch_size, kv_len, num_qo_heads, head_dim_qk).float()) * sm_scale
    if causal:
        mask = torch.arange(kv_len - qo_len, kv_len).unsqueeze(1) >= torch.arange(0, kv_len).unsqueeze(0)
    else:
        mask = torch.ones(qo_len, kv_len)
    logits = logits.masked_fill(mask.unsqueeze(0).unsqueeze(0) == 0, float("-inf"))
    lse_ref = torch.logsumexp(logits, -1).transpose(-1, -2)
    p = torch.softmax(logits, dim=-1)
    o_ref = torch.einsum("bhmn,bnhd->bmhd", p, v.view(batch_size, kv_len, num_qo_heads, head_dim_vo).float())
    return o_ref.contiguous().view(batch_size * qo_len, num_qo_heads, head_dim_vo), lse_ref * math.log2(math.e)


def mla_paged_ref(q_nope, q_pe, kv_buf, qo_indptr, kv_indptr, kv_indices, kv_len_arr, sm_scale):
    """q_nope [T,Hq,512], q_pe [T,Hq,64] bf16; kv_buf [pages, page, 576] bf16 -> fp32 out [T,Hq,512], lse [T,Hq]."""
    T, Hq, _ = q_nope.shape
    page = kv_buf.shape[1]
    out = torch.zeros((T, Hq, 512), dtype=torch.float32)
    lse = torch.zeros((T, Hq), dtype=torch.float32)
    q = torch.cat([q_nope, q_pe], dim=-1)
    for b in range(len(kv_len_arr)):
        q0, q1 = int(qo_indptr[b]), int(qo_indptr[b + 1])
        n = int(kv_len_arr[b])
        pages = kv_indices[int(kv_indptr[b]):int(kv_indptr[b + 1])].long()
        lat = kv_buf[pages].reshape(-1, 576)[:n]
        k = lat.view(n, 1, 576).repeat_interleave(Hq, dim=1)
        v = lat[:, :512].reshape(n, 1, 512).repeat_interleave(Hq, dim=1)
        o, l = attention_ref_torch(1, q[q0:q1], k, v, True, sm_scale)
        out[q0:q1] = o
        lse[q0:q1] = l.reshape(q1 - q0, Hq)
    return out, lse
