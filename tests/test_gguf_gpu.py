"""GPU: GGUF k-quant experts (KTX_FMT_GGUF, csrc/ktx_moe_gguf.inc) against oracle/ktx_oracle_gguf.c through the C ABI.

Every integer stage (Q8_K codes, sub-block dot products, scale/min products) is exact on both sides and the fp32 combine is
the same FMA chain, so gate/up/down outputs agree to the last bit EXCEPT where the device expf and glibc expf differ in
the final ulp of SiLU: that perturbs single intermediates by 1 ulp, which can flip a Q8_K code by one step.  Bound used:
|y - oracle| <= 2^-7 |oracle| + 2e-3 max|oracle|, and >= 99 % of the output elements bit-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import bf16_to_f32  # noqa: E402
from oracle.gguf_ref import GGML_TYPE_IQ1_S, GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, QUANT, GgufOracle  # noqa: E402
from oracle.oracle import f32_to_bf16  # noqa: E402

Q4, Q6, IQ1 = GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_IQ1_S


def make(E, H, I, types, seed):
    rng = np.random.default_rng(seed)
    gate = QUANT[types[0]]((rng.standard_normal((E, I, H)) / 10).astype(np.float32))
    up = QUANT[types[1]]((rng.standard_normal((E, I, H)) / 10).astype(np.float32))
    down = QUANT[types[2]]((rng.standard_normal((E, H, I)) / 10).astype(np.float32))
    return gate, up, down


def random_iq1s(E, N, K, rng):
    """Random but valid IQ1_S blocks (any byte pattern is a legal block; d kept small): the encoder in oracle/gguf_ref.py is a
    brute-force nearest-grid search, far too slow for V3-sized matrices."""
    b = rng.integers(0, 256, (E, N, K // 256, 50), dtype=np.uint8)
    d = (rng.random((E, N, K // 256)).astype(np.float16) * np.float16(0.004) + np.float16(0.001))
    b[..., 0:2] = d.view(np.uint8).reshape(E, N, K // 256, 2)
    return b.reshape(E, N, -1)


Q2, Q3, Q5, IQ4 = 10, 11, 13, 23          # round 3: native kernels for Q2_K, Q3_K, Q5_K, IQ4_XS as well
Q40, Q50, Q80 = 2, 6, 8                   # round 5: the legacy 32-element types, Q8_0 activations (csrc/ktx_moe_legacy.inc)
BLOCK_BYTES = {Q2: 84, Q3: 110, Q4: 144, Q5: 176, Q6: 210, IQ1: 50, IQ4: 136, Q40: 18, Q50: 22, Q80: 34}
BLOCK_ELEMS = {Q40: 32, Q50: 32, Q80: 32}
F16_FIELDS = {Q2: (80, 82), Q3: (108,), Q4: (0, 2), Q5: (0, 2), Q6: (208,), IQ1: (0,), IQ4: (0,), Q40: (0,), Q50: (0,), Q80: (0,)}     # byte offsets of the fp16 scales
F16_RANGE = {Q2: (1e-3, 4e-3), Q3: (1e-4, 4e-4), Q4: (1e-4, 2e-4), Q5: (1e-4, 2e-4), Q6: (1e-5, 2e-5), IQ1: (1e-3, 4e-3), IQ4: (2e-5, 4e-5),
             Q40: (1e-3, 4e-3), Q50: (5e-4, 2e-3), Q80: (6e-5, 2.4e-4)}


def random_kquant(t, E, N, K, rng):
    """Random but valid blocks of any native type (every byte pattern is a legal block; the fp16 super-scales are kept small and
    positive) for shapes where quantising real matrices in numpy would take minutes, and for the types the oracle has no
    quantiser for (Q2_K, Q3_K, IQ4_XS)."""
    nb = K // BLOCK_ELEMS.get(t, 256)
    b = rng.integers(0, 256, (E, N, nb, BLOCK_BYTES[t]), dtype=np.uint8)
    lo, span = F16_RANGE[t]
    for c in F16_FIELDS[t]:
        d = (rng.random((E, N, nb)).astype(np.float16) * np.float16(span) + np.float16(lo))
        b[..., c:c + 2] = d.view(np.uint8).reshape(E, N, nb, 2)
    return b.reshape(E, N, -1)


def run_case(E, k, H, I, T, types, seed=0, invalid=False, max_len=None, random_blocks=False):
    from ktransformers_amd import _native as n
    o = GgufOracle()
    if random_blocks or any(t not in QUANT for t in types):
        r0 = np.random.default_rng(seed)
        gate, up, down = random_kquant(types[0], E, I, H, r0), random_kquant(types[1], E, I, H, r0), random_kquant(types[2], E, H, I, r0)
    else:
        gate, up, down = make(E, H, I, types, seed)
    rng = np.random.default_rng(seed + 1)
    x = f32_to_bf16(rng.standard_normal((T, H)).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:k] for _ in range(T)]).astype(np.int64)
    if invalid:
        ids[0, 0] = -1
        ids[T - 1, k - 1] = E + 3
    w = rng.random((T, k)).astype(np.float32)
    ref = bf16_to_f32(o.moe_forward(gate, up, down, types, E, H, I, ids, w, x))
    h = n.MoEHandle(E, k, H, I, max_len or max(T, 16), "GGUF", 0)
    h.load_gguf(torch.from_numpy(gate).cuda(), torch.from_numpy(up).cuda(), torch.from_numpy(down).cuda(), *types)
    y = h.forward(torch.from_numpy(x.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(ids).cuda(),
                  torch.from_numpy(w).cuda())
    y = y.float().cpu().numpy()
    tol = 2.0 ** -7 * np.abs(ref) + 2e-3 * np.abs(ref).max()
    assert (np.abs(y - ref) <= tol).all(), f"max diff {np.abs(y - ref).max()} (ref max {np.abs(ref).max()})"
    same = float((y == ref).mean())
    assert same >= 0.99, same
    if T * k <= 64:     # the two-launch decode kernels ran above: the grouped path must give the same bits (one device expf)
        n.force_generic_path(True)
        try:
            yg = h.forward(torch.from_numpy(x.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(ids).cuda(),
                           torch.from_numpy(w).cuda()).float().cpu().numpy()
        finally:
            n.force_generic_path(False)
        # the decode gate/up kernel adds its two k-slices' fp32 sums at the end (round 3: two wavefronts per SIMD), the grouped
        # path folds the 256-blocks in one chain: a handful of outputs may land on the other side of a bf16 rounding ...
        assert float((y == yg).mean()) >= 0.99 and (np.abs(y - yg) <= 2.0 ** -7 * np.abs(yg) + 1e-3 * np.abs(yg).max()).all(), \
            f"decode and grouped paths differ in {int((y != yg).sum())} of {y.size} outputs"
        n.lib.ktx_debug_set(20, 1)      # ... and with the k-slices off the two paths are the same chain: the same bits
        try:
            y1 = h.forward(torch.from_numpy(x.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(ids).cuda(),
                           torch.from_numpy(w).cuda()).float().cpu().numpy()
        finally:
            n.lib.ktx_debug_set(20, 0)
        assert np.array_equal(y1, yg), f"decode (one k-slice) and grouped paths differ in {int((y1 != yg).sum())} of {y.size} outputs"
    return h


@pytest.mark.parametrize("types", [(Q4, Q4, Q6), (Q4, Q4, Q4), (Q6, Q6, Q6), (Q6, Q6, Q4), (IQ1, IQ1, IQ1), (IQ1, IQ1, Q4)])
@pytest.mark.parametrize("T", [1, 3, 19])
def test_small(types, T):
    run_case(4, 2, 256, 512, T, types, seed=T)


@pytest.mark.parametrize("types", [(Q5, Q5, Q6), (Q5, Q5, Q5), (Q2, Q2, Q3), (Q3, Q3, Q2), (Q2, Q2, Q2), (Q3, Q3, Q3), (IQ4, IQ4, IQ4),
                                   (IQ4, IQ4, Q5), (Q4, Q4, Q2)])
@pytest.mark.parametrize("T", [1, 3, 19])
def test_small_round3_types(types, T):
    """Q5_K (real quantiser + the q5_k_m mix), Q2_K / Q3_K / IQ4_XS (random valid blocks) against oracle/ktx_oracle_gguf.c's
    restatements (each pinned against the reference's iqk kernels in tests/test_gguf_ref_pin_cpu.py): decode kernels (T = 1, 3),
    grouped GEMM tiles (T = 19), every type on both sides of the activation re-quantisation."""
    run_case(4, 2, 256, 512, T, types, seed=40 + T)


@pytest.mark.parametrize("types", [(Q40, Q40, Q40), (Q50, Q50, Q50), (Q80, Q80, Q80), (Q40, Q40, Q80), (Q50, Q50, Q40)])
@pytest.mark.parametrize("T", [1, 3, 19, 130])
def test_legacy_types(types, T):
    """Q4_0 / Q5_0 / Q8_0 experts with llamafile's arithmetic: Q8_0 activations (ggml's partner format of these
    types), one exact int8 MFMA per 32-block, one fma per block in k order — the oracle's restatement (pinned to the reference's
    compiled iqk kernels for Q4_0 / Q5_0 by tests/test_gguf_ref_pin_cpu.py), so the outputs agree to the last bit except where the
    device expf and glibc's differ in SiLU's final ulp; held to the k-quants' bound.  Ragged expert tiles at T = 19 / 130."""
    h = run_case(6, 3, 512, 256, T, types, seed=5 + T, invalid=(T == 19))
    assert h.weight_bytes == 6 * (2 * 256 * 512 // 32 * BLOCK_BYTES[types[0]] + 512 * 256 // 32 * BLOCK_BYTES[types[2]])   # the file's bytes, re-tiled


@pytest.mark.parametrize("types", [(Q40, Q40, Q40), (Q80, Q80, Q80)])
def test_legacy_types_v3_expert_shape(types):
    run_case(8, 8, 7168, 2048, 2, types, seed=9)


def test_legacy_and_kquant_families_do_not_mix():
    from ktransformers_amd import _native as n
    h = n.MoEHandle(2, 1, 256, 256, 4, "GGUF", 0)
    g = torch.zeros((2, 256, 256 // 32 * 18), dtype=torch.uint8, device="cuda")
    d = torch.zeros((2, 256, 144), dtype=torch.uint8, device="cuda")
    with pytest.raises(n.KtxError, match="all be k- / i-quants or all be legacy"):
        h.load_gguf(g, g, d, Q40, Q40, Q4)


@pytest.mark.parametrize("types", [(Q5, Q5, Q6), (Q2, Q2, Q3), (IQ4, IQ4, IQ4)])
def test_round3_types_v3_expert_shape(types):
    """The DeepSeek-V3 / R1 expert shape (7168 x 2048, the decode kernels' 2-k-slice variants and deep rings) for the type mixes
    of the q5_k_m, q2_k and iq4_xs GGUF files; random valid blocks."""
    run_case(4, 2, 7168, 2048, 1, types, seed=61, random_blocks=True)
    run_case(4, 2, 7168, 2048, 5, types, seed=62, random_blocks=True)


def test_invalid_ids_and_ragged_tiles():
    run_case(8, 3, 512, 256, 37, (Q4, Q4, Q6), seed=5, invalid=True)


@pytest.mark.parametrize("T", [1, 130])
def test_mixtral_like_shape(T):
    """q4_k_m mix at a Mixtral-like aspect (H % 256 == 0, I % 256 == 0), prefill tile sizes MT = 1 / 4."""
    run_case(8, 2, 1024, 3584, T, (Q4, Q4, Q6), seed=9)


@pytest.mark.parametrize("types", [(Q4, Q4, Q6), (Q6, Q6, Q4), (IQ1, IQ1, IQ1), (IQ1, IQ1, Q4)])
@pytest.mark.parametrize("T", [3, 19, 70, 300])
def test_folded_prompt_kernels_give_the_unfolded_bits(types, T):
    """Round 6: the Q4_K / Q6_K grouped GEMM folds the sub-block scales into the int8 MFMA operand (csrc/ktx_moe_gguf.inc,
    gg_fold) where gg_block multiplied every 32-wide partial product on the vector ALU; the IQ1_S one runs ONE int8 chain per block
    on the operand (s'(8 g - delta) - 1) / 2 plus the block's sum of codes.  Every integer is the same and so is the
    fp32 chain, hence the SAME BITS as the unfolded kernel (dev knob 21 = 1) — at every tile height (MT = 1, 2, 4), on ragged
    tiles, with ids out of range, on random valid blocks (all 6-bit scales / mins and all int8 Q6_K scales occur)."""
    from ktransformers_amd import _native as n
    E, k, H, I = 6, 2, 512, 768
    r0 = np.random.default_rng(100 + T)
    gate, up, down = random_kquant(types[0], E, I, H, r0), random_kquant(types[1], E, I, H, r0), random_kquant(types[2], E, H, I, r0)
    x = torch.from_numpy(f32_to_bf16(r0.standard_normal((T, H)).astype(np.float32)).view(np.int16)).view(torch.bfloat16).cuda()
    ids = np.stack([r0.permutation(E)[:k] for _ in range(T)]).astype(np.int64)
    ids[0, 0] = -1
    ids[T - 1, k - 1] = E + 1
    ids, w = torch.from_numpy(ids).cuda(), torch.from_numpy(r0.random((T, k)).astype(np.float32)).cuda()
    h = n.MoEHandle(E, k, H, I, max(T, 16), "GGUF", 0)
    h.load_gguf(torch.from_numpy(gate).cuda(), torch.from_numpy(up).cuda(), torch.from_numpy(down).cuda(), *types)
    n.force_generic_path(True)          # (T = 3: the grouped kernels, not the two decode launches)
    try:
        y_fold = h.forward(x, ids, w).view(torch.int16).cpu().numpy()
        n.lib.ktx_debug_set(21, 1)
        y_ref = h.forward(x, ids, w).view(torch.int16).cpu().numpy()
    finally:
        n.lib.ktx_debug_set(21, 0)
        n.force_generic_path(False)
    assert np.array_equal(y_fold, y_ref), f"{int((y_fold != y_ref).sum())} of {y_ref.size} outputs differ"
    assert np.abs(y_ref.view(np.uint16).astype(np.int32)).max() > 0


@pytest.mark.parametrize("T", [1, 4])
def test_mixtral_8x7b_real_dims(T):
    """BASELINE.json configs[0] (Mixtral-8x7B q4_k_m, kt-kernel/bench/bench_moe.py:166-170): 8 experts, top-2, hidden 4096,
    intermediate 14336, Q4_K gate/up + Q6_K down, decode batches."""
    run_case(8, 2, 4096, 14336, T, (Q4, Q4, Q6), seed=21 + T, random_blocks=True)


def test_iq1s_v3_expert_shape():
    """BASELINE.json configs[4] (DeepSeek-R1, IQ1_S experts): the V3 expert shape, a few experts, decode and a small batch."""
    run_case(4, 2, 7168, 2048, 1, (IQ1, IQ1, IQ1), seed=11, random_blocks=True)
    run_case(4, 2, 7168, 2048, 9, (IQ1, IQ1, IQ1), seed=12, random_blocks=True)


def test_errors():
    from ktransformers_amd import _native as n
    with pytest.raises(n.KtxError):
        n.MoEHandle(4, 2, 384, 512, 16, "GGUF", 0)                       # H % 256
    h = n.MoEHandle(4, 2, 256, 512, 16, "GGUF", 0)
    x = torch.zeros(1, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(n.KtxError):
        h.forward(x, torch.zeros(1, 2, dtype=torch.long, device="cuda"), torch.zeros(1, 2, device="cuda"))   # not loaded
    g = torch.zeros(4, 512, 144, dtype=torch.uint8, device="cuda")
    d = torch.zeros(4, 256, 2 * 210, dtype=torch.uint8, device="cuda")
    with pytest.raises(n.KtxError):
        h.load_gguf(g, g, d, 12, 12, 8)                                   # Q8_0 down behind k-quant gate / up: the families do not mix


def test_llamafile_backend_through_the_operator_and_gguf_file(tmp_path):
    """GGUF file -> GGUFLoader -> KTransformersExperts(backend="llamafile") -> HIP, against the oracle on the same blocks."""
    from test_gguf_loader_cpu import build_file
    from toy_model import ToyConfig
    from ktransformers_amd.operators.experts import KTransformersExperts
    from ktransformers_amd.util.gguf_loader import GGUFLoader
    from ktransformers_amd.util.utils import InferenceState

    src = build_file(str(tmp_path / "toy.gguf"))
    E, H, I, k, T = src["E"], src["H"], src["I"], 2, 5
    cfg = ToyConfig(hidden_size=H, moe_intermediate_size=I, intermediate_size=I, n_routed_experts=E, num_experts_per_tok=k)
    ld = GGUFLoader(str(tmp_path))
    orig = torch.nn.ModuleList([torch.nn.Identity() for _ in range(E)])
    ex = KTransformersExperts("model.layers.1.mlp.experts", ld, cfg, orig, prefill_device="cuda", prefill_op="KExpertsTorch",
                              generate_device="cpu", generate_op="KExpertsCPU", out_device="cuda", backend="llamafile",
                              max_len=16)
    ex.load(mode=InferenceState.GENERATE)
    assert ex.generate_experts.method == "GGUF"
    rng = np.random.default_rng(3)
    x = f32_to_bf16(rng.standard_normal((T, H)).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:k] for _ in range(T)]).astype(np.int64)
    w = rng.random((T, k)).astype(np.float32)
    y = ex.forward(torch.from_numpy(x.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(ids).cuda(),
                   torch.from_numpy(w).cuda()).float().cpu().numpy()
    ref = bf16_to_f32(GgufOracle().moe_forward(src["gate"], src["up"], src["down"], (Q4, Q4, Q6), E, H, I, ids, w, x))
    assert (np.abs(y - ref) <= 2.0 ** -7 * np.abs(ref) + 2e-3 * np.abs(ref).max()).all()


def test_expert_types_without_a_native_kernel_are_served_as_dequantised_bf16_experts(tmp_path):
    """A GGUF file whose experts are Q5_K gate / up with a Q8_0 down (a mix of the two activation families the expert kernels do not
    serve — Q8_K and Q8_0 inputs in one expert): the operator says so (RuntimeWarning),
    de-quantises the blocks with the loader's reference-pinned codecs and serves BF16 experts.  Checked against fp64 math on
    the same de-quantised weights (bf16 kernels: 2^-7 relative + a small absolute term), not against llamafile arithmetic."""
    from helpers import write_gguf
    from toy_model import ToyConfig
    from ktransformers_amd.operators.experts import KTransformersExperts
    from ktransformers_amd.util.gguf_loader import GGUFLoader, _dequant
    from ktransformers_amd.util.utils import InferenceState

    rng = np.random.default_rng(11)
    E, H, I, k, T = 4, 256, 512, 2, 3

    def blocks(n, nbytes, f16_cols):       # any bit pattern is a valid block; keep the fp16 scales small and positive
        b = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
        for c in f16_cols:
            b[:, c:c + 2] = (rng.random(n).astype(np.float16) * np.float16(0.004) + np.float16(0.001)).view(np.uint8).reshape(n, 2)
        return b

    gate, up = blocks(E * I * H // 256, 176, (0, 2)), blocks(E * I * H // 256, 176, (0, 2))        # Q5_K
    down = blocks(E * H * I // 32, 34, (0,))                                                         # Q8_0
    write_gguf(str(tmp_path / "toy.gguf"), {"blk.1.ffn_gate_exps.weight": (13, [H, I, E], gate.tobytes()),
                                            "blk.1.ffn_up_exps.weight": (13, [H, I, E], up.tobytes()),
                                            "blk.1.ffn_down_exps.weight": (8, [I, H, E], down.tobytes())},
               {"deepseek2.expert_count": E})
    cfg = ToyConfig(hidden_size=H, moe_intermediate_size=I, intermediate_size=I, n_routed_experts=E, num_experts_per_tok=k)
    orig = torch.nn.ModuleList([torch.nn.Identity() for _ in range(E)])
    ex = KTransformersExperts("model.layers.1.mlp.experts", GGUFLoader(str(tmp_path)), cfg, orig, prefill_device="cuda",
                              prefill_op="KExpertsTorch", generate_device="cpu", generate_op="KExpertsCPU", out_device="cuda",
                              backend="llamafile", max_len=16)
    with pytest.warns(RuntimeWarning, match="no native expert kernel"):
        ex.load(mode=InferenceState.GENERATE)
    ge = ex.generate_experts
    assert ge.loaded_method == "BF16" and ge.method == "GGUF"       # the configured method survives the fallback ...
    ge.unload()
    with pytest.warns(RuntimeWarning, match="no native expert kernel"):
        ge.load()                                                    # ... so UNLOAD -> load takes the GGUF branch again
    assert ge.loaded_method == "BF16"
    # a hybrid (fp8 + GGUF) checkpoint hands the raw blocks over as DEVICE tensors (util/loader.py load_experts(device=...))
    from ktransformers_amd.util.gguf_loader import dequantize_expert_blocks
    blk = torch.from_numpy(down.reshape(-1)).cuda()
    assert torch.equal(dequantize_expert_blocks(blk, 8, E, H, I), dequantize_expert_blocks(blk.cpu(), 8, E, H, I))
    x = f32_to_bf16((rng.standard_normal((T, H)) / 4).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:k] for _ in range(T)]).astype(np.int64)
    w = rng.random((T, k)).astype(np.float32)
    y = ex.forward(torch.from_numpy(x.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(ids).cuda(),
                   torch.from_numpy(w).cuda()).float().cpu().numpy()
    rb = lambda a: bf16_to_f32(f32_to_bf16(a)).astype(np.float64)      # noqa: E731  (the weights the handle holds)
    G, U = rb(_dequant(13, gate.reshape(-1)).reshape(E, I, H)), rb(_dequant(13, up.reshape(-1)).reshape(E, I, H))
    D = rb(_dequant(8, down.reshape(-1)).reshape(E, H, I))
    xf = bf16_to_f32(x).astype(np.float64)
    ref = np.zeros((T, H))
    for t in range(T):
        for j in range(k):
            e = ids[t, j]
            g, u = G[e] @ xf[t], U[e] @ xf[t]
            ref[t] += w[t, j] * (D[e] @ (g / (1 + np.exp(-g)) * u))
    assert (np.abs(y - ref) <= 2.0 ** -6 * np.abs(ref) + 1e-2 * np.abs(ref).max()).all()


def test_legacy_type_experts_through_the_operator_and_gguf_file(tmp_path, monkeypatch):
    """A GGUF file whose routed experts are Q4_0 (gate / up) and Q8_0 (down): KTransformersExperts(backend="llamafile") serves them
    NATIVELY (no warning, the file's bytes re-tiled, llamafile's arithmetic with Q8_0 activations) — against the oracle on the same
    blocks; KTX_GGUF_BF16_FALLBACK=1 brings back the de-quantised BF16 experts of rounds 2-4 (opt-in, with the warning)."""
    import warnings
    from helpers import write_gguf
    from toy_model import ToyConfig
    from ktransformers_amd.operators.experts import KTransformersExperts
    from ktransformers_amd.util.gguf_loader import GGUFLoader
    from ktransformers_amd.util.utils import InferenceState

    rng = np.random.default_rng(21)
    E, H, I, k, T = 4, 256, 512, 2, 5
    gate, up, down = random_kquant(Q40, E, I, H, rng), random_kquant(Q40, E, I, H, rng), random_kquant(Q80, E, H, I, rng)
    write_gguf(str(tmp_path / "toy.gguf"), {"blk.1.ffn_gate_exps.weight": (Q40, [H, I, E], gate.tobytes()),
                                            "blk.1.ffn_up_exps.weight": (Q40, [H, I, E], up.tobytes()),
                                            "blk.1.ffn_down_exps.weight": (Q80, [I, H, E], down.tobytes())},
               {"deepseek2.expert_count": E})
    cfg = ToyConfig(hidden_size=H, moe_intermediate_size=I, intermediate_size=I, n_routed_experts=E, num_experts_per_tok=k)
    orig = torch.nn.ModuleList([torch.nn.Identity() for _ in range(E)])

    def build():
        return KTransformersExperts("model.layers.1.mlp.experts", GGUFLoader(str(tmp_path)), cfg, orig, prefill_device="cuda",
                                    prefill_op="KExpertsTorch", generate_device="cpu", generate_op="KExpertsCPU", out_device="cuda",
                                    backend="llamafile", max_len=16)
    ex = build()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        ex.load(mode=InferenceState.GENERATE)
    assert ex.generate_experts.loaded_method == "GGUF"
    x = f32_to_bf16((rng.standard_normal((T, H)) / 4).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:k] for _ in range(T)]).astype(np.int64)
    w = rng.random((T, k)).astype(np.float32)
    args = (torch.from_numpy(x.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(w).cuda())
    y = ex.forward(*args).float().cpu().numpy()
    ref = bf16_to_f32(GgufOracle().moe_forward(gate, up, down, (Q40, Q40, Q80), E, H, I, ids, w, x))
    assert (np.abs(y - ref) <= 2.0 ** -7 * np.abs(ref) + 2e-3 * np.abs(ref).max()).all() and float((y == ref).mean()) >= 0.99
    monkeypatch.setenv("KTX_GGUF_BF16_FALLBACK", "1")
    ex2 = build()
    with pytest.warns(RuntimeWarning, match="no native expert kernel"):
        ex2.load(mode=InferenceState.GENERATE)
    assert ex2.generate_experts.loaded_method == "BF16"
    y2 = ex2.forward(*args).float().cpu().numpy()
    assert (np.abs(y2 - ref) <= 2.0 ** -5 * np.abs(ref) + 2e-2 * np.abs(ref).max()).all()      # same weights, un-quantised activations
