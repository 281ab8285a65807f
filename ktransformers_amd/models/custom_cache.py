"""Paged latent KV cache for MLA — mirror of the MLA branch of archive/ktransformers/models/custom_cache.py:26-215
(StaticCache): one `[max_pages, page_size, 1, kv_lora_rank + qk_rope_head_dim]` bf16 tensor per layer, pages of 64
tokens, an identity page table, `update()` scatters the new `[ckv | k_pe]` rows at `cache_position`.

The scatter itself is `mla_cache_append_kernel` (csrc/ktx_mla.hip); the decode path of KDeepseekV2Attention does not even
call update(): the MLA kernel stores the new row while it reads it (ktx_mla_decode_append)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch


class StaticCache:
    def __init__(self, config, max_batch_size: int, max_cache_len: Optional[int], device, dtype=None) -> None:
        self._max_batch_size = max_batch_size
        self._max_cache_len = config.max_position_embeddings if max_cache_len is None else max_cache_len
        self.dtype = dtype if dtype is not None else torch.bfloat16
        if self.dtype != torch.bfloat16:
            raise ValueError("the HIP MLA kernels read a bf16 latent cache")
        self.page_size = 64                                                      # custom_cache.py:81
        self.max_pages = (self._max_cache_len + self.page_size - 1) // self.page_size
        self.kv_lora_rank = config.kv_lora_rank
        self.qk_rope_head_dim = config.qk_rope_head_dim
        self.num_hidden_layers = config.num_hidden_layers
        self.is_MLA, self.is_page = True, True
        self.identity_page_table = True      # page_table_list[i] = arange (custom_cache.py:99-104): pages are used in order
        latent_shape = (self.max_pages, self.page_size, 1, self.kv_lora_rank + self.qk_rope_head_dim)
        self.key_cache, self.value_cache, self.page_table_list, self.past_tokens = [], [], [], []
        self.page_table_map: Dict[Any, torch.Tensor] = {}
        for idx in range(self.num_hidden_layers):
            dev = device[f"blk.{idx}.self_attn"]["generate_device"] if isinstance(device, dict) else device
            if dev not in self.page_table_map:                                   # custom_cache.py:99-104
                pt = torch.arange(max_batch_size * self.max_pages, dtype=torch.int32, device=dev)
                self.page_table_map[dev] = pt.view(max_batch_size, self.max_pages)
            self.page_table_list.append(self.page_table_map[dev])
            self.key_cache.append(torch.zeros(latent_shape, dtype=self.dtype, device=dev))
            self.value_cache.append(None)
            self.past_tokens.append(0)

    @property
    def max_batch_size(self):
        return self._max_batch_size

    @property
    def max_cache_len(self):
        return self._max_cache_len

    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int,
               cache_kwargs: Optional[Dict[str, Any]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """key_states = compressed_kv [.., T, 1, kv_lora], value_states = k_pe [.., T, 1, rope] (custom_cache.py:147-199)."""
        from ktransformers_amd._native import mla_cache_append

        cache_position = cache_kwargs.get("cache_position")
        k_out = self.key_cache[layer_idx]
        if self.past_tokens[layer_idx] + cache_position.size(0) > self.max_pages * self.page_size:
            # the reference's indexed assignment raises on an out-of-range position (custom_cache.py:189-195)
            raise IndexError(f"StaticCache.update: {self.past_tokens[layer_idx]} + {cache_position.size(0)} tokens exceed the "
                             f"cache ({self.max_pages * self.page_size} tokens)")
        self.past_tokens[layer_idx] += cache_position.size(0)
        page_idx = cache_position // self.page_size
        page_offset = cache_position % self.page_size
        mla_cache_append(k_out, key_states.reshape(-1, self.kv_lora_rank), value_states.reshape(-1, self.qk_rope_head_dim),
                         page_idx, page_offset)
        return k_out, self.page_table_list[layer_idx]

    def note_appended(self, layer_idx: int, n: int) -> None:
        """Bookkeeping for rows the attention kernel appended itself."""
        self.past_tokens[layer_idx] += n

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        return self.past_tokens[layer_idx]

    def get_usable_length(self, kv_seq_len: int, layer_idx: Optional[int] = 0) -> int:
        return 0                                                                 # custom_cache.py:225-226 (callers add it to q_len)

    def change_seq_length(self, bias: Optional[int] = 0) -> None:
        for i in range(self.num_hidden_layers):
            self.past_tokens[i] += bias

    def get_max_length(self) -> Optional[int]:
        return self._max_cache_len

    def reset(self):
        for i in range(self.num_hidden_layers):
            self.key_cache[i].zero_()
            self.past_tokens[i] = 0

    def remove_suffix(self, start_pos: int) -> None:
        """Forget everything from token `start_pos` on (prefix reuse between requests, custom_cache.py:240-250)."""
        for i in range(self.num_hidden_layers):
            k = self.key_cache[i]
            k.view(-1, k.shape[-1])[start_pos:].zero_()
            self.past_tokens[i] = start_pos

    def get_max_cache_shape(self):
        return self.max_cache_len                                                # custom_cache.py:252-254


class KDeepSeekV3Cache:
    """The batched-serving latent cache (archive/ktransformers/models/custom_cache.py:387-466): per layer one
    [pages, page_size, 1, kv_lora_rank + rope] bf16 tensor, pages handed out by the scheduler.  The reference takes the tensors
    from its kvc2 `InferenceContext`; here `allocate(num_pages)` creates them (the storage engine is out of scope) and `load`
    accepts any object with the same `k_cache[0][layer]` shape of attribute."""

    def __init__(self, config, page_size: int = 256, dtype=torch.bfloat16, device="cuda:0"):
        if dtype != torch.bfloat16:
            raise ValueError("the HIP MLA kernels read a bf16 latent cache")
        self.config, self.dtype, self.device = config, dtype, torch.device(device)
        self.kv_lora_rank, self.qk_rope_head_dim, self.page_size = config.kv_lora_rank, config.qk_rope_head_dim, page_size
        self.k_caches, self.v_caches = [], []
        self.max_cache_len = 0

    def allocate(self, num_pages: int) -> None:
        shape = (num_pages, self.page_size, 1, self.kv_lora_rank + self.qk_rope_head_dim)
        self.k_caches = [torch.zeros(shape, dtype=self.dtype, device=self.device) for _ in range(self.config.num_hidden_layers)]
        self.max_cache_len = num_pages * self.page_size

    def load(self, inference_context) -> None:
        self.k_caches = [inference_context.k_cache[0][i] for i in range(self.config.num_hidden_layers)]
        self.max_cache_len = self.k_caches[0].shape[0] * self.k_caches[0].shape[1]

    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int, page_idx: torch.Tensor,
               page_offset: torch.Tensor, cache_kwargs: Optional[Dict[str, Any]] = None) -> torch.Tensor:
        """Scatter the new latent rows (key_states = compressed_kv, value_states = k_pe) to [page_idx, page_offset]; returns the
        layer's whole cache tensor (custom_cache.py:413-443)."""
        from ktransformers_amd._native import mla_cache_append

        k_out = self.k_caches[layer_idx]
        mla_cache_append(k_out, key_states.reshape(-1, self.kv_lora_rank), value_states.reshape(-1, self.qk_rope_head_dim),
                         page_idx, page_offset)
        return k_out

    def get_page_table(self, cache_position: torch.Tensor, q_indptr: torch.Tensor, kv_indptr: torch.Tensor, kv_indices: torch.Tensor,
                       bsz_tensors: torch.Tensor):
        """(page_idx, page_offset) of every scheduled token (custom_cache.py:446-463): token i of request r at position p lives
        in that request's p // page_size-th page, kv_indices[kv_indptr[r] + p // page_size]; tokens beyond bsz_tensors[0], and
        positions beyond the request's allotted pages, get page 0 like the reference's zero initialisation.  Vectorised: the
        reference loops over tokens on the host."""
        page_offset = cache_position % self.page_size
        local = cache_position // self.page_size
        n = cache_position.numel()
        counts = (q_indptr[1:] - q_indptr[:-1]).to(torch.long)
        query_ids = torch.repeat_interleave(torch.arange(counts.numel(), device=cache_position.device), counts, output_size=int(q_indptr[-1]))
        if query_ids.numel() < n:   # tokens past the last request keep request 0, as torch.zeros_like does in the reference
            query_ids = torch.cat([query_ids, query_ids.new_zeros(n - query_ids.numel())])
        query_ids = query_ids[:n]
        start = kv_indptr[query_ids].to(torch.long)
        have = (kv_indptr[query_ids + 1] - kv_indptr[query_ids]).to(torch.long)
        ok = (local < have) & (torch.arange(n, device=cache_position.device) < int(bsz_tensors[0]))
        idx = (start + torch.where(ok, local.to(torch.long), torch.zeros_like(start))).clamp_(max=max(kv_indices.numel() - 1, 0))
        page_idx = torch.where(ok, kv_indices[idx].to(local.dtype), torch.zeros_like(local))
        return page_idx, page_offset

