"""GPU: a decode step of the block-fp8 model family (BASELINE.json configs[4]: IQ1_S routed experts + KLinearFP8 linears, the
reference's DeepSeek-V3-Chat-fp8-linear-ggml-experts.yaml combination) through the injected operators — the router rides in the
launch of the shared experts' fp8 [gate ; up] GEMV (lin_dec_gate_kernel<FP8>, round 5) and the step's logits are those of the two
separate launches (KTX_MOE_SEPARATE_ROUTER=1) up to the fp32 summation order of the GEMV's k-slices."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fp8_decode_step_router_rides_with_the_shared_gate_up(monkeypatch):
    import bench
    from ktransformers_amd import _native as n
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS["r1-iq1s"]
    L = wl["dense"] + 2
    torch.manual_seed(0)
    mr = bench.ModelDecodeRunner(wl, L, dev, 64, 8, seed=0, use_graph=False)

    def logits():
        with torch.no_grad():
            out = mr.model(mr.cur.clone(), mr.pos.clone(), mr.cache, mr.pos[0].clone())[0, -1].float()
        torch.cuda.synchronize()
        return out

    try:
        monkeypatch.setenv("KTX_MOE_SEPARATE_ROUTER", "1")
        want = logits()
        monkeypatch.delenv("KTX_MOE_SEPARATE_ROUTER")
        n.timing_collect()
        n.timing_enable(2)
        try:
            got = logits()
            labels = [lab for lab, _, _ in n.timing_collect()]
        finally:
            n.timing_enable(0)
        assert sum("lin_dec_gate_kernel<FP8>" in lab for lab in labels) == 2, labels
        assert n.attn_status_any()[1] == 0
        assert torch.isfinite(got).all()
        assert (got - want).abs().max() <= 2e-2 * want.abs().max(), float((got - want).abs().max() / want.abs().max())
    finally:
        mr.close()
