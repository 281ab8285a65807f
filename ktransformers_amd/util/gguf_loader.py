"""GGUF weight ingest for the hot path — mirror of the reader half of archive/ktransformers/util/custom_loader.py:278-526
(GGUFLoader) and custom_gguf.py:177-217,665-760 (read_value, translate_name_to_gguf).

Same public surface the operators use: has_tensor / get_mmap_tensor / get_ggml_type / tensor_info / tensor_device_map /
gguf_file_meta / load_gguf_tensor (de-quantise to a torch tensor) plus load_experts(key) which hands KExpertsHIP the RAW
k-quant blocks and their ggml types (the reference's KExpertsBase.load_weights contract, operators/experts.py:89-135).
Files are memory-mapped; nothing here touches the GPU except the final `.to(device)`.

De-quantisation on load covers the types the hot path's non-expert tensors use (F32, F16, BF16, Q8_0, Q4_K, Q6_K); the
expert tensors are not de-quantised at all — they are re-tiled on the GPU (ktx_moe_load_gguf)."""
from __future__ import annotations

import math
import os
import re
import struct

import numpy as np
import torch

GGML_TYPES = {"F32": 0, "F16": 1, "Q4_0": 2, "Q5_0": 6, "Q8_0": 8, "Q2_K": 10, "Q3_K": 11, "Q4_K": 12, "Q5_K": 13, "Q6_K": 14,
              "IQ4_XS": 23, "BF16": 30}
GGML_NAMES = {v: k for k, v in GGML_TYPES.items()}
# (elements per block, bytes per block) — custom_gguf.py:72-100
GGML_QUANT_SIZES = {0: (1, 4), 1: (1, 2), 2: (32, 18), 3: (32, 20), 6: (32, 22), 7: (32, 24), 8: (32, 34), 9: (32, 40),
                    10: (256, 84), 11: (256, 110), 12: (256, 144), 13: (256, 176), 14: (256, 210), 15: (256, 292),
                    16: (256, 66), 17: (256, 74), 18: (256, 98), 19: (256, 50), 20: (32, 18), 21: (256, 110), 22: (256, 82),
                    23: (256, 136), 24: (1, 1), 25: (1, 2), 26: (1, 4), 27: (1, 8), 28: (1, 8), 29: (256, 56), 30: (1, 2)}
_T = {"uint8": 0, "int8": 1, "uint16": 2, "int16": 3, "uint32": 4, "int32": 5, "float32": 6, "bool": 7, "string": 8,
      "array": 9, "uint64": 10, "int64": 11, "float64": 12}
_FMT = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


def read_value(f, data_type):
    if data_type == _T["string"]:
        n = struct.unpack("<Q", f.read(8))[0]
        return f.read(n).decode("utf-8")
    if data_type == _T["array"]:
        elem, count = struct.unpack("<IQ", f.read(12))
        return [read_value(f, elem) for _ in range(count)]
    fmt = _FMT[data_type]
    v = struct.unpack(fmt, f.read(struct.calcsize(fmt)))[0]
    return bool(v) if data_type == _T["bool"] else v


def translate_name_to_gguf(name: str) -> str:
    """HF parameter name -> GGUF tensor name (custom_gguf.py:665-760, the DeepSeek / Mixtral / Qwen-MoE rules)."""
    m = re.match(r"model\.layers\.(\d+)\.block_sparse_moe\.experts\.(\d+)\.(w\d)\.weight", name)
    if m:
        return f"blk.{m.group(1)}.{ {'w1': 'ffn_gate', 'w2': 'ffn_down', 'w3': 'ffn_up'}[m.group(3)] }.{m.group(2)}.weight"
    for a in ("gate", "up", "down"):
        name = name.replace(f".ffn_{a}_exp.", f".ffn_{a}_exps.")
    m = re.match(r"(?:model\.layers|blk)\.(\d+)\.mlp\.experts\.(\d+)\.(gate_proj|up_proj|down_proj)", name)
    if m:
        return f"blk.{m.group(1)}.{m.group(2)}.ffn_{m.group(3)[:-5]}_exps"
    for a, b in (("lm_head.", "output."), ("model.embed_tokens.", "token_embd."), ("model.norm.", "output_norm."),
                 ("model.layers.", "blk."), (".input_layernorm", ".attn_norm"), (".mlp.down_proj", ".ffn_down"),
                 (".mlp.gate_proj", ".ffn_gate"), (".mlp.up_proj", ".ffn_up"), (".post_attention_layernorm", ".ffn_norm"),
                 (".self_attn.q_proj", ".attn_q"), (".self_attn.k_proj", ".attn_k"), (".self_attn.v_proj", ".attn_v"),
                 (".self_attn.o_proj", ".attn_output"), (".self_attn.qkv_proj", ".attn_qkv"),
                 (".self_attn.kv_a_proj_with_mqa", ".attn_kv_a_mqa"), (".self_attn.kv_a_layernorm", ".attn_kv_a_norm"),
                 (".self_attn.kv_b_proj", ".attn_kv_b"), (".self_attn.q_a_proj", ".attn_q_a"),
                 (".self_attn.q_a_layernorm", ".attn_q_a_norm"), (".self_attn.q_b_proj", ".attn_q_b"),
                 (".self_attn.q_norm", ".attn_q_norm"), (".self_attn.k_norm", ".attn_k_norm"),
                 (".shared_expert.", ".shared_experts."), (".shared_expert_", ".shared_experts_"),
                 (".mlp.shared_experts.down_proj", ".ffn_down_shexp"), (".mlp.gate.e_score_correction_bias", ".exp_probs_b.bias"),
                 (".mlp.gate", ".ffn_gate_inp"), (".mlp.shared_experts.gate_proj", ".ffn_gate_shexp"),
                 (".mlp.shared_experts.up_proj", ".ffn_up_shexp"), (".mlp.shared_experts_gate", ".ffn_gate_inp_shexp"),
                 (".mlp.experts", ""), (".block_sparse_moe.gate.", ".ffn_gate_inp."), (".block_sparse_moe.experts", "")):
        name = name.replace(a, b)
    return name


def _dequant(ggml_type: int, raw: np.ndarray) -> np.ndarray:
    """raw uint8 bytes of whole blocks -> float32 values (flat)."""
    if ggml_type == 0:
        return raw.view(np.float32)
    if ggml_type == 1:
        return raw.view(np.float16).astype(np.float32)
    if ggml_type == 30:
        return (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    if ggml_type == 8:                                       # Q8_0: fp16 d + 32 int8 (custom_gguf.py dequantize_q8_0)
        b = raw.reshape(-1, 34)
        d = b[:, :2].copy().view(np.float16).astype(np.float32)
        return (d * b[:, 2:].view(np.int8).astype(np.float32)).reshape(-1)
    b = raw
    if ggml_type == 12:                                      # Q4_K (custom_gguf.py:326-343)
        b = b.reshape(-1, 144)
        nb = b.shape[0]
        d = b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
        dmin = b[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
        s1, qs = b[:, 4:16].reshape(nb, 12, 1), b[:, 16:].reshape(nb, 4, 32)
        fac = d * np.concatenate([s1[:, 0:4] & 63, (s1[:, 8:] & 15) | ((s1[:, 0:4] >> 6) << 4)], axis=1)
        off = dmin * np.concatenate([s1[:, 4:8] & 63, (s1[:, 8:] >> 4) | ((s1[:, 4:8] >> 6) << 4)], axis=1)
        q = np.stack([qs & 0xF, qs >> 4], axis=2).reshape(nb, 8, 32)
        return (fac * q - off).astype(np.float32).reshape(-1)
    if ggml_type == 14:                                      # Q6_K
        b = b.reshape(-1, 210)
        nb = b.shape[0]
        ql, qh = b[:, :128].reshape(nb, 2, 64).astype(np.int16), b[:, 128:192].reshape(nb, 2, 32).astype(np.int16)
        sc = b[:, 192:208].copy().view(np.int8).astype(np.float32).reshape(nb, 2, 8)
        d = b[:, 208:210].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
        q = np.stack([((ql[:, :, :32] & 0xF) | (((qh >> 0) & 3) << 4)) - 32, ((ql[:, :, 32:] & 0xF) | (((qh >> 2) & 3) << 4)) - 32,
                      ((ql[:, :, :32] >> 4) | (((qh >> 4) & 3) << 4)) - 32, ((ql[:, :, 32:] >> 4) | (((qh >> 6) & 3) << 4)) - 32],
                     axis=2).reshape(nb, 2, 128).astype(np.float32)
        return (d * np.repeat(sc, 16, axis=2) * q).astype(np.float32).reshape(-1)
    f16 = lambda cols: np.ascontiguousarray(cols).view(np.float16).astype(np.float32)  # noqa: E731
    if ggml_type == 2:                                       # Q4_0: fp16 d | 16 bytes, low nibbles then high nibbles, offset 8
        b = b.reshape(-1, 18)
        d, qs = f16(b[:, 0:2]), b[:, 2:]
        q = np.concatenate([qs & 0xF, qs >> 4], axis=1).astype(np.float32) - 8.0
        return (d * q).reshape(-1)
    if ggml_type == 6:                                       # Q5_0: fp16 d | 32 high bits | 16 bytes of nibbles, offset 16
        b = b.reshape(-1, 22)
        d, qs = f16(b[:, 0:2]), b[:, 6:]
        hi = np.unpackbits(b[:, 2:6], axis=1, bitorder="little")          # bit j of the little-endian word -> value j
        q = (np.concatenate([qs & 0xF, qs >> 4], axis=1) | (hi << 4)).astype(np.float32) - 16.0
        return (d * q).reshape(-1)
    if ggml_type == 10:                                      # Q2_K: 16 x (4-bit scale | 4-bit min) | 64 bytes of 2-bit q | d | dmin
        b = b.reshape(-1, 84)
        nb = b.shape[0]
        sc = b[:, :16]
        d, dmin = f16(b[:, 80:82]), f16(b[:, 82:84])
        qs = b[:, 16:80].reshape(nb, 2, 1, 32)                               # two halves of 32 bytes, four 2-bit planes each
        q = ((qs >> np.array([0, 2, 4, 6], np.uint8).reshape(1, 1, 4, 1)) & 3).reshape(nb, 16, 16).astype(np.float32)
        dl = (d * (sc & 0xF).astype(np.float32)).reshape(nb, 16, 1)
        ml = (dmin * (sc >> 4).astype(np.float32)).reshape(nb, 16, 1)
        return (dl * q - ml).reshape(-1)
    if ggml_type == 11:                                      # Q3_K: 32 bytes of high bits | 64 bytes of 2-bit q | 12 bytes = 16 six-bit scales | d
        b = b.reshape(-1, 110)
        nb = b.shape[0]
        d = f16(b[:, 108:110]).reshape(nb, 1, 1)
        hbit = np.unpackbits(b[:, :32].reshape(nb, 32, 1), axis=2, bitorder="little")      # [nb, l, plane]
        hbit = hbit.transpose(0, 2, 1).reshape(nb, 2, 4, 32)                                # plane = half * 4 + shift
        qs = b[:, 32:96].reshape(nb, 2, 1, 32)
        q2 = (qs >> np.array([0, 2, 4, 6], np.uint8).reshape(1, 1, 4, 1)) & 3
        q = (q2.astype(np.int16) - 4 * (1 - hbit.astype(np.int16))).reshape(nb, 16, 16).astype(np.float32)
        lo, hi2 = b[:, 96:104], b[:, 104:108]                                               # 4-bit parts, 2-bit parts
        s6 = np.concatenate([(lo & 0xF) | ((np.tile(hi2, 2) >> np.repeat(np.array([0, 2], np.uint8), 4)) & 3) << 4,
                             (lo >> 4) | ((np.tile(hi2, 2) >> np.repeat(np.array([4, 6], np.uint8), 4)) & 3) << 4], axis=1)
        dl = d * (s6.astype(np.float32) - 32.0).reshape(nb, 16, 1)
        return (dl * q).reshape(-1)
    if ggml_type == 13:                                      # Q5_K: d | dmin | 12 bytes of 6-bit scales / mins | 32 bytes of high bits | 128 bytes of nibbles
        b = b.reshape(-1, 176)
        nb = b.shape[0]
        d, dmin = f16(b[:, 0:2]).reshape(nb, 1, 1), f16(b[:, 2:4]).reshape(nb, 1, 1)
        s1, qs = b[:, 4:16].reshape(nb, 12, 1), b[:, 48:].reshape(nb, 4, 32)
        fac = d * np.concatenate([s1[:, 0:4] & 63, (s1[:, 8:] & 15) | ((s1[:, 0:4] >> 6) << 4)], axis=1).astype(np.float32)
        off = dmin * np.concatenate([s1[:, 4:8] & 63, (s1[:, 8:] >> 4) | ((s1[:, 4:8] >> 6) << 4)], axis=1).astype(np.float32)
        hbit = np.unpackbits(b[:, 16:48].reshape(nb, 32, 1), axis=2, bitorder="little").transpose(0, 2, 1)   # [nb, plane, l]
        q = (np.stack([qs & 0xF, qs >> 4], axis=2).reshape(nb, 8, 32) + (hbit << 4)).astype(np.float32)
        return (fac * q - off).reshape(-1)
    if ggml_type == 23:                                      # IQ4_XS: d | 16 bits of scale high parts | 4 bytes of low parts | 128 bytes of codebook indices
        b = b.reshape(-1, 136)
        nb = b.shape[0]
        d = f16(b[:, 0:2])
        sh = np.ascontiguousarray(b[:, 2:4]).view(np.uint16).astype(np.uint32)              # [nb, 1]
        ib = np.arange(8, dtype=np.uint32)
        lo = (np.repeat(b[:, 4:8], 2, axis=1).astype(np.uint32) >> (4 * (ib % 2))) & 0xF
        ls = (lo | (((sh >> (2 * ib)) & 3) << 4)).astype(np.float32) - 32.0
        dl = (d * ls).reshape(nb, 8, 1)
        qs = b[:, 8:].reshape(nb, 8, 16)
        kv = np.array([-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113], np.float32)   # the IQ4_NL codebook
        return (dl * kv[np.concatenate([qs & 0xF, qs >> 4], axis=2)]).reshape(-1)
    raise NotImplementedError(f"ggml_type {ggml_type} ({GGML_NAMES.get(ggml_type, '?')}) is not de-quantised here")


def dequantize_expert_blocks(raw, ggml_type: int, num_experts: int, rows: int, cols: int) -> torch.Tensor:
    """Raw ggml blocks of one stacked expert tensor ([E, rows, cols] elements, any layout of whole blocks) -> bf16
    [E, rows, cols], one expert at a time (the fp32 intermediate of a whole DeepSeek-sized tensor would not fit in host
    memory).  For expert types the HIP expert kernels do not read natively."""
    n_el, n_by = GGML_QUANT_SIZES[ggml_type]
    flat = (raw.detach().cpu().numpy() if isinstance(raw, torch.Tensor) else np.asarray(raw)).reshape(-1).view(np.uint8)   # (hybrid checkpoints hand over device tensors)
    per = rows * cols // n_el * n_by
    if cols % n_el or flat.size != num_experts * per:
        raise ValueError(f"expert blocks: {flat.size} bytes for {num_experts} x {rows} x {cols} of ggml type {ggml_type}")
    out = torch.empty((num_experts, rows, cols), dtype=torch.bfloat16)
    for e in range(num_experts):
        out[e] = torch.from_numpy(_dequant(ggml_type, flat[e * per:(e + 1) * per]).reshape(rows, cols)).to(torch.bfloat16)
    return out


class GGUFLoader:
    def __init__(self, gguf_path: str):
        if not os.path.exists(gguf_path):
            raise FileNotFoundError(f"GGUF dir not found: {gguf_path}")
        if os.path.isfile(gguf_path):
            gguf_path = os.path.dirname(gguf_path)
        self.gguf_path = gguf_path
        self.tensor_info, self.tensor_file_map, self.file_data_map = {}, {}, {}
        self.gguf_file_meta, self.tensor_device_map = {}, {}
        found = False
        for root, _, files in os.walk(gguf_path):
            for fn in sorted(files):
                if fn.endswith(".gguf"):
                    found = True
                    path = os.path.join(root, fn)
                    with open(path, "rb") as f:
                        self._load_header(f)
                    self.file_data_map[path] = np.memmap(path, mode="r")
        if not found:
            raise FileNotFoundError(f"Cannot find any .gguf files in: {gguf_path}")

    def _load_header(self, f):
        if f.read(4) != b"GGUF":
            raise ValueError(f"{f.name}: not a GGUF file")
        _version, n_tensors, n_kv = struct.unpack("<IQQ", f.read(20))
        info = {}
        for _ in range(n_kv):
            name = read_value(f, _T["string"])
            info[name] = read_value(f, struct.unpack("<I", f.read(4))[0])
        tinfo = {}
        for _ in range(n_tensors):
            name = read_value(f, _T["string"])
            ndim = read_value(f, _T["uint32"])
            shape = [read_value(f, _T["uint64"]) for _ in range(ndim)]
            ggml_type = read_value(f, _T["uint32"])
            bad_offset = read_value(f, _T["uint64"])
            blk, tsz = GGML_QUANT_SIZES[ggml_type]
            tinfo[name] = {"ggml_type": ggml_type, "shape": shape, "bad_offset": bad_offset,
                           "n_bytes": int(math.prod(shape)) * tsz // blk}
        start = f.tell()
        align = info.get("general.alignment", 32)             # custom_loader.py:389-399
        for t in tinfo.values():
            off = start + t["bad_offset"]
            t["offset"] = off + (align - off % align) % align
        for name in tinfo:
            self.tensor_file_map[name] = f.name
        self.tensor_info.update(tinfo)
        self.gguf_file_meta.update(info)

    # ---- the reference loader's query surface -------------------------------------------------------------------
    def has_tensor(self, name: str) -> bool:
        return translate_name_to_gguf(name) in self.tensor_info

    def get_ggml_type(self, name: str) -> int:
        name = translate_name_to_gguf(name)
        if name not in self.tensor_info:
            raise KeyError(f"Key {name} not found in GGUF files")
        return self.tensor_info[name]["ggml_type"]

    def get_mmap_tensor(self, name: str) -> np.ndarray:
        name = translate_name_to_gguf(name)
        t = self.tensor_info[name]
        return self.file_data_map[self.tensor_file_map[name]][t["offset"]: t["offset"] + t["n_bytes"]]

    def load_gguf_tensor(self, name: str, device: str = "cpu", target_dtype=None) -> torch.Tensor:
        name = translate_name_to_gguf(name)
        t = self.tensor_info[name]
        vals = _dequant(t["ggml_type"], np.ascontiguousarray(self.get_mmap_tensor(name)))
        out = torch.from_numpy(np.array(vals, copy=True)).view(t["shape"][::-1])
        # llama-architecture files (Mixtral's too) hold attn_q / attn_k with each head's rows interleaved for ggml's pairwise
        # RoPE; the reference undoes that so the HF rotate-half modules see their own row order (custom_loader.py:507-517).
        if self.gguf_file_meta.get("general.architecture") == "llama" and ("attn_q" in name or "attn_k" in name):
            n_head = int(self.gguf_file_meta["llama.attention.head_count" if "attn_q" in name
                                             else "llama.attention.head_count_kv"])
            out = out.reshape(n_head, out.shape[0] // n_head // 2, 2, *out.shape[1:]).swapaxes(1, 2).reshape(out.shape)
        if target_dtype is None:
            target_dtype = torch.get_default_dtype()
        return out.to(device=device, dtype=target_dtype)

    load_tensor = load_gguf_tensor                             # DictLoader / SafeTensorLoader spelling

    def get_expert_count(self, key: str) -> int:
        return int(self.tensor_info[translate_name_to_gguf(key) + ".ffn_gate_exps.weight"]["shape"][-1])

    def load_experts(self, key: str, device: str = "cpu") -> dict:
        """Raw k-quant blocks of blk.N.ffn_{gate,up,down}_exps.weight + ggml types (experts.py:89-135); `device` is
        ignored — the operator uploads the blocks itself."""
        base = translate_name_to_gguf(key)
        res = {}
        for w in ("gate", "up", "down"):
            name = f"{base}.ffn_{w}_exps.weight"
            if name not in self.tensor_info:
                raise ValueError(f"Experts {key} not found in gguf_loader")
            res[w] = torch.from_numpy(np.array(self.get_mmap_tensor(name), copy=True))
            res[f"{w}_type"] = self.tensor_info[name]["ggml_type"]
        return res
