"""Generate tests/golden/moe_amx_golden.npz from the REFERENCE's own kernels (oracle/_ref, built from /root/reference).

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
The fixture stores the inputs (bf16 bits) and the reference outputs, so it travels to machines without the reference.
Shapes are the smallest the reference accepts (K % 128 == 0, N % 32 == 0) to keep the file small.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import bf16_to_f32, fp8_block_quant, fp8_perchannel_quant, make_case, rawint4_quantize  # noqa: E402
from oracle.oracle import FMT_AMXINT4, FMT_AMXINT8, FMT_BF16, FMT_FP8, FMT_FP8_PERCHANNEL, FMT_RAWINT4, Reference  # noqa: E402

E, k, H, I = 4, 2, 128, 128
ref = Reference(threads=2)
base = make_case(20260921, E, k, H, I, 1)
out = dict(E=E, k=k, H=H, I=I, gate=base["gate"], up=base["up"], down=base["down"])
cases = [("t1", 1, False), ("t7_invalid", 7, True), ("t33_prefill", 33, False)]
for fmt, fname in ((FMT_AMXINT4, "int4"), (FMT_AMXINT8, "int8")):
    moe = ref.make_moe(fmt, base["gate"], base["up"], base["down"], k=k, max_len=64)
    for name, T, inv in cases:
        c = make_case(1000 + T, E, k, H, I, T, invalid_ids=inv)
        y = ref.moe_forward(moe, c["ids"], c["w"], c["x"])
        y2 = ref.moe_forward(moe, c["ids"], c["w"], c["x"], y_prev=y)
        out[f"{fname}_{name}_x"] = c["x"]
        out[f"{fname}_{name}_ids"] = c["ids"]
        out[f"{fname}_{name}_w"] = c["w"]
        out[f"{fname}_{name}_y"] = y
        out[f"{fname}_{name}_yinc"] = y2
    ref.free_moe(moe)
# FP8 (block scales) and BF16 experts: same weights; fp8 codes/scales are stored so the fixture is self-contained
gq, gs = fp8_block_quant(bf16_to_f32(base["gate"]))
uq, us = fp8_block_quant(bf16_to_f32(base["up"]))
dq, ds = fp8_block_quant(bf16_to_f32(base["down"]))
out.update(fp8_gate=gq, fp8_up=uq, fp8_down=dq, fp8_gate_s=gs, fp8_up_s=us, fp8_down_s=ds)
moe8 = ref.make_moe_quant(FMT_FP8, E, H, I, k, gq, uq, dq, gs, us, ds, max_len=64, group_size=128)
moeb = ref.make_moe(FMT_BF16, base["gate"], base["up"], base["down"], k=k, max_len=64)
for fname, moe in (("fp8", moe8), ("bf16", moeb)):
    for name, T, inv in cases:
        c = make_case(1000 + T, E, k, H, I, T, invalid_ids=inv)
        y = ref.moe_forward(moe, c["ids"], c["w"], c["x"])
        out[f"{fname}_{name}_y"] = y
        out[f"{fname}_{name}_yinc"] = ref.moe_forward(moe, c["ids"], c["w"], c["x"], y_prev=y)

# FP8_PERCHANNEL (one fp32 scale per output row): the reference's own TP_MOE<AMX_FP8_PERCHANNEL_MOE_TP<GemmKernel224FP8PerChannel>>
# (kt-kernel/operators/amx/fp8-perchannel-moe.hpp) on the same weights, same seeded inputs
pq = [fp8_perchannel_quant(bf16_to_f32(base[n])) for n in ("gate", "up", "down")]
out.update(fp8pc_gate=pq[0][0], fp8pc_up=pq[1][0], fp8pc_down=pq[2][0], fp8pc_gate_s=pq[0][1], fp8pc_up_s=pq[1][1],
           fp8pc_down_s=pq[2][1])
moepc = ref.make_moe_quant(FMT_FP8_PERCHANNEL, E, H, I, k, pq[0][0], pq[1][0], pq[2][0], pq[0][1], pq[1][1], pq[2][1], max_len=64)
for name, T, inv in cases:
    c = make_case(1000 + T, E, k, H, I, T, invalid_ids=inv)
    y = ref.moe_forward(moepc, c["ids"], c["w"], c["x"])
    out[f"fp8pc_{name}_y"] = y
    out[f"fp8pc_{name}_yinc"] = ref.moe_forward(moepc, c["ids"], c["w"], c["x"], y_prev=y)

# RAWINT4 (Kimi-K2 native int4, group 32): the reference's own TP_MOE<AMX_K2_MOE_TP<GemmKernel224Int4SmallKGroup>>
# (kt-kernel/operators/amx/k2-moe.hpp:124-191) on weights quantised by the reference test's own rawint4_quantize
# (kt-kernel/test/per_commit/test_moe_rawint4_accuracy.py:69-94, lifted from its source: the module imports the compiled
# extension).  K % 512 == 0 is the smallest shape the K2 kernels take here; both the vec_mul (qlen <= 4E/k) and the
# mat_mul path (t33) are covered.
import ast  # noqa: E402

import torch  # noqa: E402

SRC = "/root/reference/kt-kernel/test/per_commit/test_moe_rawint4_accuracy.py"
ns = {"torch": torch, "group_size": 32}
for node in ast.parse(open(SRC).read()).body:
    if isinstance(node, ast.FunctionDef) and node.name == "rawint4_quantize":
        exec(compile(ast.Module([node], []), SRC, "exec"), ns)
Ek, kk, Hk, Ik = 4, 2, 512, 512
bk = make_case(20260922, Ek, kk, Hk, Ik, 1)
packs = {}
for nm in ("gate", "up", "down"):
    w = torch.from_numpy(bf16_to_f32(bk[nm]))
    # the reference's python quantiser on the first expert (slow scalar loops) pins the vectorised helper, which then
    # quantises all experts
    q0, s0 = ns["rawint4_quantize"](w[0].to(torch.bfloat16))
    pq, ps = rawint4_quantize(bf16_to_f32(bk[nm]))
    assert np.array_equal(q0.numpy(), pq[0]) and np.array_equal(s0.view(torch.int16).numpy().view(np.uint16), ps[0]), nm
    packs[nm] = (pq, ps)
    out[f"k2_{nm}_p"], out[f"k2_{nm}_s"] = pq, ps
out.update(k2_E=Ek, k2_k=kk, k2_H=Hk, k2_I=Ik)
moek = ref.make_moe_quant(FMT_RAWINT4, Ek, Hk, Ik, kk, packs["gate"][0], packs["up"][0], packs["down"][0], packs["gate"][1],
                          packs["up"][1], packs["down"][1], max_len=64, group_size=32)
for name, T, inv in cases:
    c = make_case(2000 + T, Ek, kk, Hk, Ik, T, invalid_ids=inv)
    y = ref.moe_forward(moek, c["ids"], c["w"], c["x"])
    out[f"k2_{name}_x"], out[f"k2_{name}_ids"], out[f"k2_{name}_w"] = c["x"], c["ids"], c["w"]
    out[f"k2_{name}_y"] = y
    out[f"k2_{name}_yinc"] = ref.moe_forward(moek, c["ids"], c["w"], c["x"], y_prev=y)

# the reference quantiser's own dequantised weights + scales for one matrix (BufferBInt4Impl::from_mat -> to_mat)
deq, d = ref.quant_roundtrip_int4(base["gate"][0])
out["int4_gate0_dequant"] = deq
out["int4_gate0_scale"] = d
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "moe_amx_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
