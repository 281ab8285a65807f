import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ktransformers_amd import _native as n
from oracle.linear_ref import linear_w4_ref, quantize_weights_ref
torch.manual_seed(0)
K, N, G = 256, 64, 64
w = (torch.randn(N, K) / 10).to(torch.bfloat16)
q, s = quantize_weights_ref(w.T.contiguous(), G)
h = n.LinearHandle(K, N, "W4", G, 256)
h.load_bf16(w.cuda())
for mode in ("onehot", "ones", "rand"):
    for T in (1, 5):
        x = torch.zeros(T, K)
        if mode == "onehot": x[:, 3] = 1.0
        elif mode == "ones": x[:] = 1.0
        else: x = torch.randn(T, K) / 10
        x = x.to(torch.bfloat16)
        y = h.forward(x.cuda()).float().cpu()
        r = linear_w4_ref(x, q, s, G).float()
        print(mode, T, "maxdiff", float((y - r).abs().max()), "y", y[0, :6].tolist(), "ref", r[0, :6].tolist())
