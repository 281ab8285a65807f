"""CPU: the sampling tail of the generation loop against the Hugging Face logits warpers the reference instantiates
(archive/ktransformers/util/utils.py:356-396 `tf_logits_warper`, :486-491), and the loop itself on a stand-in model."""
import pytest
import torch
from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

from ktransformers_amd.util.generate import prefill_and_generate, sample_next_token, warp_logits


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.7, 0, 1.0), (1.0, 5, 1.0), (1.0, 0, 0.9), (0.6, 50, 0.95), (1.3, 3, 0.5),
                                                     (1.0, 10_000, 0.999)])
def test_warp_logits_matches_hf_warpers(temperature, top_k, top_p):
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(4, 257, generator=g) * 3
    want = scores
    if temperature != 1.0:
        want = TemperatureLogitsWarper(temperature)(None, want)
    if top_k:
        want = TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(None, want)
    if top_p < 1.0:
        want = TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(None, want)
    got = warp_logits(scores, temperature, top_k, top_p)
    assert torch.equal(got, want)


def test_sample_next_token():
    scores = torch.tensor([0.1, 3.0, 0.2, 2.9])
    assert int(sample_next_token(scores)) == 1
    assert int(sample_next_token(scores, do_sample=True, top_k=1)) == 1
    g = torch.Generator().manual_seed(1)
    draws = torch.stack([sample_next_token(scores, do_sample=True, temperature=0.5, top_k=2, generator=g) for _ in range(200)])
    assert set(draws.tolist()) == {1, 3}
    a = sample_next_token(scores, do_sample=True, generator=torch.Generator().manual_seed(7))
    b = sample_next_token(scores, do_sample=True, generator=torch.Generator().manual_seed(7))
    assert int(a) == int(b)


class Counter(torch.nn.Module):
    """next token = (last token + 1) % vocab; records the (start, end) of every call."""

    def __init__(self, vocab=11):
        super().__init__()
        self.vocab, self.calls = vocab, []

    def forward(self, ids, pos, cache, cache_position, last_token_only=False):
        self.calls.append((int(pos[0, 0]), int(pos[0, -1]) + 1))
        return torch.nn.functional.one_hot((ids[:, -1:] + 1) % self.vocab, self.vocab).float() * 10


def test_generation_loop_chunked_prefill_eos_and_sampling():
    m = Counter()
    ids = torch.arange(7).view(1, 7) % 11
    out = prefill_and_generate(m, ids, None, max_new_tokens=5, use_cuda_graph=False, chunk_size=3)
    assert out.tolist() == [7, 8, 9, 10, 0]
    assert m.calls[:3] == [(0, 3), (3, 6), (6, 7)] and m.calls[3:] == [(7, 8), (8, 9), (9, 10), (10, 11)]
    out = prefill_and_generate(Counter(), ids, None, max_new_tokens=9, use_cuda_graph=False, eos_token_id=9)
    assert out.tolist() == [7, 8, 9]
    out, lg = prefill_and_generate(Counter(), ids, None, max_new_tokens=3, use_cuda_graph=False, return_logits=True, do_sample=True,
                                   top_k=1, generator=torch.Generator().manual_seed(0))
    assert out.tolist() == [7, 8, 9] and lg.shape == (3, 11)
