"""GPU: the expert-parallel wrappers on a single-rank RCCL group (the 8-GPU run belongs to the driver): prefill
(all-to-all-v dispatch + ktx_moe_combine) must reproduce MoEHandle.forward bit for bit; decode (all-gather + fp32
partial + reduce-scatter) to 1 bf16 ulp."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from helpers import make_case  # noqa: E402


@pytest.fixture(scope="module")
def group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield None
    dist.destroy_process_group()


def test_prefill_and_decode_wrappers(group):
    from ktransformers_amd import _native as n
    from ktransformers_amd.parallel import ExpertParallelMoE
    E, k, H, I, T = 8, 2, 256, 128, 37
    c = make_case(5, E, k, H, I, T, invalid_ids=True)
    h = n.MoEHandle(E, k, H, I, T * k, "AMXINT4", 0)
    tt = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).cuda()
    h.load_bf16(tt(c["gate"]), tt(c["up"]), tt(c["down"]))
    x, ids, w = tt(c["x"]), torch.from_numpy(c["ids"]).cuda(), torch.from_numpy(c["w"]).cuda()
    ref = h.forward(x, ids, w)
    ep = ExpertParallelMoE(h)
    y = ep.forward_prefill(x, ids, w)
    assert torch.equal(y, ref)
    yd = ep.forward(x[:4], ids[:4], w[:4]).float()
    r4 = ref[:4].float()
    assert torch.all((yd - r4).abs() <= r4.abs() * 2.0 ** -7 + 1e-5 * r4.abs().max())
