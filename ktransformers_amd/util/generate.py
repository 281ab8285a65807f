"""Decode-loop glue — mirror of archive/ktransformers/util/utils.py:356-540 (prefill_and_generate:
chunked prefill, then one token per step through a captured graph, greedy sampling) and util/cuda_graph_runner.py:19-100
(CUDAGraphRunner: static input buffers `cur_token`, `position_ids`, `cache_position`, one captured decode step, logits out).

Everything between the embedding and the logits is HIP kernels from this package enqueued by the injected operators; the
graph is a HIP graph (torch.cuda.CUDAGraph on ROCm).  The kv length the MLA kernel uses is derived ON THE DEVICE from
position_ids, so the captured graph stays valid as the sequence grows."""
from __future__ import annotations

import torch

from ktransformers_amd.util.utils import InferenceState


def check_handoffs(full: bool = True) -> None:
    """Bounded in-launch hand-offs must not fail silently.  The one-launch attention step (include/ktx_attn.h) writes its status word
    into pinned host memory when a poll gives up: reading it is a host load, so it is checked after EVERY token, for every device
    (`full=False`: only that).  The expert-parallel peer-write transport (parallel.py) keeps its
    status word on the device (a small copy): checked with `full=True` — every 64 tokens and at the end of a generation."""
    from ktransformers_amd import _native, parallel
    dev, st = _native.attn_status_any()
    if st != 0:
        raise RuntimeError(f"one-launch attention step on cuda:{dev}: a hand-off inside the launch timed out (status {st:#x}); the "
                           "outputs since the previous token are not valid.  Another kernel held CUs for longer than the poll bound, or "
                           "two persistent launches overlapped; ktransformers_amd._native.attn_reset(device) re-arms the workspace")
    if not full:
        return
    if parallel.EP_STATE.get("exchange") is not None:
        parallel.check_exchange_status()


_check_ep = check_handoffs      # (the name rounds 2-4 used)


def set_inference_mode(model: torch.nn.Module, mode: InferenceState) -> None:
    for m in model.modules():
        if hasattr(m, "set_inference_mode") and not isinstance(getattr(type(m), "set_inference_mode", None), property):
            try:
                m.set_inference_mode(mode)
            except NotImplementedError:
                pass


class CUDAGraphRunner:
    """util/cuda_graph_runner.py:19-100."""

    def __init__(self):
        self.graph = None
        self.input_buffers = {}
        self.output_buffers = {}

    def capture(self, model, cur_token, position_ids, cache_position, past_key_values, main_device="cuda:0", trace=None,
                **kwargs):
        """`trace` (a list, measurement only): receives the (name, args) of every library call issued during the capture
        (ktransformers_amd._native.TRACE); their pointers stay valid as long as this runner's graph lives."""
        assert self.graph is None
        self.model = model
        dev = torch.device(main_device)
        self.input_buffers = {"cur_token": cur_token.clone(), "position_ids": position_ids.clone(),
                              "cache_position": cache_position.clone()}
        ib = self.input_buffers
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            for _ in range(2):   # warm-up outside the capture: lazy handle creation, workspace growth, attribute setting
                logits = model(ib["cur_token"], ib["position_ids"], past_key_values, ib["cache_position"])
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        from ktransformers_amd import _native
        _native.TRACE = trace
        try:
            with torch.cuda.graph(self.graph, stream=stream):
                logits = model(ib["cur_token"], ib["position_ids"], past_key_values, ib["cache_position"])
        finally:
            _native.TRACE = None
        torch.cuda.synchronize(dev)
        self.output_buffers = {"logits": logits}

    def forward(self, cur_token, position_ids, cache_position):
        ib = self.input_buffers
        ib["cur_token"].copy_(cur_token)
        ib["position_ids"].copy_(position_ids)
        ib["cache_position"].copy_(cache_position)
        self.graph.replay()
        return self.output_buffers["logits"]

    __call__ = forward


def warp_logits(scores: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
                min_tokens_to_keep: int = 1) -> torch.Tensor:
    """The reference's `tf_logits_warper` chain for the settings its chat front end exposes (utils.py:356-396 builds HF's
    TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper in this order): scores [..., vocab] -> scores with the
    excluded tokens at -inf.  Same arithmetic as the HF classes (tests/test_generate_cpu.py compares with them)."""
    if temperature is not None and temperature != 1.0:
        scores = scores / temperature
    if top_k is not None and top_k != 0:
        k = min(max(int(top_k), min_tokens_to_keep), scores.size(-1))
        scores = scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], -float("inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(scores, descending=False)
        cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cumulative <= (1 - top_p)
        remove[..., -min_tokens_to_keep:] = 0
        scores = scores.masked_fill(remove.scatter(-1, sorted_indices, remove), -float("inf"))
    return scores


def sample_next_token(scores: torch.Tensor, do_sample: bool = False, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
                      generator: torch.Generator | None = None) -> torch.Tensor:
    """decode_one_tokens' tail (utils.py:486-491): warp, then multinomial over the softmax when sampling, argmax otherwise."""
    scores = warp_logits(scores.float(), temperature, top_k, top_p) if do_sample else scores
    if not do_sample:
        return scores.argmax(dim=-1)
    return torch.multinomial(torch.softmax(scores, dim=-1), num_samples=1, generator=generator).squeeze(-1)


@torch.no_grad()
def prefill_and_generate(model, input_ids: torch.Tensor, past_key_values, max_new_tokens: int = 16, use_cuda_graph: bool = True,
                         chunk_size: int = 8192, return_logits: bool = False, do_sample: bool = False, temperature: float = 1.0,
                         top_k: int = 0, top_p: float = 1.0, generator: torch.Generator | None = None, eos_token_id=None):
    """Generation loop (utils.py:356-540): chunked prefill, then one token per step through a captured HIP graph; greedy by
    default, temperature / top-k / top-p sampling with do_sample.  Returns the generated token ids [<= max_new_tokens]
    (generation stops after `eos_token_id`), and the fp32 logits of every generated position when return_logits."""
    pick = lambda lg: sample_next_token(lg[0, -1], do_sample, temperature, top_k, top_p, generator)
    dev = input_ids.device
    T = input_ids.shape[1]
    cap = getattr(past_key_values, "max_cache_len", None)
    if hasattr(past_key_values, "max_pages"):
        cap = past_key_values.max_pages * past_key_values.page_size
    if cap is not None and T + max_new_tokens > cap:
        raise ValueError(f"prefill_and_generate: {T} prompt + {max_new_tokens} new tokens exceed the cache ({cap} tokens)")
    set_inference_mode(model, InferenceState.PREFILL)
    logits = None
    for s in range(0, T, chunk_size):                                    # chunk_prefill (utils.py:496-511)
        e = min(T, s + chunk_size)
        pos = torch.arange(s, e, device=dev).unsqueeze(0)
        logits = model(input_ids[:, s:e], pos, past_key_values, pos[0], last_token_only=True)
    set_inference_mode(model, InferenceState.GENERATE)
    tokens, all_logits = [], []
    nxt = pick(logits)
    runner = None
    for i in range(max_new_tokens):
        tokens.append(nxt.clone())
        if return_logits:
            all_logits.append(logits[0, -1].clone())
        if i == max_new_tokens - 1 or (eos_token_id is not None and int(nxt) == eos_token_id):
            break
        cur = nxt.view(1, 1)
        pos = torch.tensor([[T + i]], device=dev, dtype=torch.long)
        if use_cuda_graph:
            if runner is None:
                runner = CUDAGraphRunner()
                runner.capture(model, cur, pos, pos[0], past_key_values, main_device=str(dev))
            logits = runner(cur, pos, pos[0])
        else:
            logits = model(cur, pos, past_key_values, pos[0])
        nxt = pick(logits)
        check_handoffs(full=(i & 63) == 63)
    check_handoffs()
    out = torch.stack(tokens)
    return (out, torch.stack(all_logits)) if return_logits else out
