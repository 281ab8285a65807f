"""Generate tests/golden/kt_tp_golden.npz: a NUMA-sharded AMXINT4 checkpoint (two parts, down sharded over K with one scale per
(row, part) — kt-kernel/python/utils/loader.py:179-290, operators/amx/moe.hpp:103-147) together with what the REFERENCE computes
from the same bf16 weights with two sub-pools: TP_MOE<AMX_MOE_TP<GemmKernel224Int4>> in oracle/_ref/libkt_ref.so, tp_count = 2,
i.e. two MoEs of width I/2 whose fp32 outputs merge_results adds (operators/amx/moe_base.hpp:749-791), plain and incremental.
The packed tensors are written by the reference's own packer (ktref_pack_b on each part's slice).

    python tests/golden/make_kt_tp_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import make_case  # noqa: E402
from test_amx_packed_cpu import pack_with_reference  # noqa: E402

from oracle.oracle import FMT_AMXINT4, Reference  # noqa: E402

E, k, H, I, T, P = 4, 2, 256, 512, 9, 2
c = make_case(41, E, k, H, I, T)
Ip = I // P
out = dict(E=np.int64(E), k=np.int64(k), H=np.int64(H), I=np.int64(I), P=np.int64(P), x=c["x"], ids=c["ids"], w=c["w"])
for fam, (n, kk) in (("gate", (Ip, H)), ("up", (Ip, H)), ("down", (H, Ip))):
    for e in range(E):
        for p in range(P):
            sl = c[fam][e][:, p * Ip:(p + 1) * Ip] if fam == "down" else c[fam][e][p * Ip:(p + 1) * Ip]
            packed, scale = pack_with_reference(0, np.ascontiguousarray(sl), n, kk)
            out[f"blk.3.ffn_{fam}_exps.{e}.numa.{p}.weight"] = packed.view(np.int8)
            out[f"blk.3.ffn_{fam}_exps.{e}.numa.{p}.scale"] = scale
ref = Reference(threads=2, subpools=P)
moe = ref.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"], k=k, max_len=32)
out["y"] = ref.moe_forward(moe, c["ids"], c["w"], c["x"])
rng = np.random.default_rng(5)
y_prev = (rng.standard_normal((T, H)).astype(np.float32) * 0.05).view(np.uint32) >> 16
out["y_prev"] = y_prev.astype(np.uint16)
out["y_inc"] = ref.moe_forward(moe, c["ids"], c["w"], c["x"], y_prev=out["y_prev"])
one = Reference(threads=2, subpools=1)
y1 = one.moe_forward(one.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"], k=k, max_len=32), c["ids"], c["w"], c["x"])
out["differs_from_one_part"] = np.int64((y1 != out["y"]).sum())
np.savez_compressed(os.path.join(HERE, "kt_tp_golden.npz"), **out)
print("outputs that differ between tp_count 2 and 1:", int(out["differs_from_one_part"]), "of", y1.size)
