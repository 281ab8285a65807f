#!/usr/bin/env python3
"""bench.py — decode / prefill throughput of the MI355X-native quantized-MoE + MLA hot path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched under torch.distributed.run)
prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], the largest configuration that fits one GPU with parity pinned): DeepSeek-V2-Lite
(27 layers, H=2048, 16 heads, MLA kv_lora 512 + rope 64, 64 routed experts top-6 + 2 shared, I=1408, vocab 102400) with
AMXINT4 ("int4") routed experts and W4-g64 (Marlin semantics) linears, all resident in HBM; synthetic seeded weights.

One "step" (default --hot-path model) = one greedy decode token (batch 1) through the WHOLE YAML-injected decoder stack —
every §8(a) row: embedding, RMSNorm, MLA attention operator (projections, YaRN RoPE, absorb, paged MQA over --ctx cached
tokens, cache append), router, routed + shared experts, lm_head, argmax — replayed as one HIP graph; the sampled token is
fed back, so routing follows the model.  value = tokens/s of the whole job.
N>1: every rank decodes its own token stream (weak scaling); attention / dense parts are replicated, the routed experts are
sharded E/N per rank (expert parallel): per MoE layer all-gather [x, ids, w] -> local experts -> reduce-scatter (RCCL).
Extra fields at N=1: `mla_router_experts_only` (the MLA kernel + router + routed experts of every layer, nothing else),
`moe_only`, `prefill` (2048-token chunk through the routed experts), `roofline` (dominant decode kernel: algorithmic bytes
/ launch time measured live with HIP events, + PMC traffic), `cpu_baseline` (the reference's own kernels on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (H, I, E, k, L_moe, description)
    "v2lite-int4": dict(H=2048, I=1408, E=64, k=6, L=26, method="AMXINT4", heads=16, attn_layers=27,
                        gate=dict(n_group=1, topk_group=1, scoring_func="softmax", topk_method="greedy",
                                  norm_topk_prob=False, routed_scaling_factor=1.0, bias=False),
                        desc="DeepSeek-V2-Lite 16B routed experts AMXINT4, 26 MoE layers, decode bs=1"),
    "v3-int4-layers": dict(H=7168, I=2048, E=256, k=8, L=8, method="AMXINT4", heads=128, attn_layers=8,
                           gate=dict(n_group=8, topk_group=4, scoring_func="sigmoid", topk_method="noaux_tc",
                                     norm_topk_prob=True, routed_scaling_factor=2.5, bias=True),
                           desc="DeepSeek-V3 routed experts AMXINT4, layer subset (8 distinct resident layers), decode bs=1"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_layers(wl, dev, max_len, expert_begin=0, expert_num=None, seed=0):
    from ktransformers_amd._native import MoEHandle

    H, I, E, k, L = wl["H"], wl["I"], wl["E"], wl["k"], wl["L"]
    e_local = expert_num or E
    layers = []
    g = torch.Generator(device=dev)
    for li in range(L):
        g.manual_seed(seed * 1000 + li)
        h = MoEHandle(e_local, k, H, I, max_len=max_len, method=wl["method"], device=dev.index,
                      expert_begin=expert_begin, global_expert_num=E)
        # randn/10 bf16 weights (reference tests: test_moe_rawint4_accuracy.py:175-183), quantised by the GPU restatement
        # of the reference quantiser.  All E experts are generated so every EP rank sees the same global weights.
        gate = (torch.randn((E, I, H), generator=g, device=dev, dtype=torch.float32) / 10).to(torch.bfloat16)
        up = (torch.randn((E, I, H), generator=g, device=dev, dtype=torch.float32) / 10).to(torch.bfloat16)
        down = (torch.randn((E, H, I), generator=g, device=dev, dtype=torch.float32) / 10).to(torch.bfloat16)
        sl = slice(expert_begin, expert_begin + e_local)
        h.load_bf16(gate[sl].contiguous(), up[sl].contiguous(), down[sl].contiguous())
        del gate, up, down
        layers.append(h)
    torch.cuda.synchronize(dev)
    return layers


def make_routing(wl, T, nsets, dev, seed):
    """nsets x L distinct routings: randperm(E)[:k] per token, rand weights (reference kernel tests' convention)."""
    E, k, L = wl["E"], wl["k"], wl["L"]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    scores = torch.rand((nsets, L, T, E), generator=g, device=dev)
    ids = scores.topk(k, dim=-1).indices.to(torch.int64).contiguous()
    w = torch.rand((nsets, L, T, k), generator=g, device=dev, dtype=torch.float32).contiguous()
    return ids, w


class DecodeRunner:
    """One token (or T tokens) through L MoE layers; static buffers so the whole step is one HIP graph."""

    def __init__(self, wl, layers, T, dev, nsets=16, seed=1, ep_group=None):
        self.wl, self.layers, self.T, self.dev = wl, layers, T, dev
        self.ids_all, self.w_all = make_routing(wl, T, nsets, dev, seed)
        self.nsets = nsets
        self.ids = self.ids_all[0].clone()
        self.w = self.w_all[0].clone()
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 77)
        self.x = (torch.randn((T, wl["H"]), generator=g, device=dev) / 100).to(torch.bfloat16)
        self.y = [torch.empty_like(self.x) for _ in range(2)]
        self.graph = None
        self.ep_group = ep_group

    def set_step(self, i):
        s = i % self.nsets
        self.ids.copy_(self.ids_all[s])
        self.w.copy_(self.w_all[s])

    def step_eager(self):
        # residual-free chain: layer l consumes the (bf16) output of layer l-1 re-scaled into activation range by
        # feeding the original hidden state; the experts' arithmetic does not depend on what produced x.
        for li, h in enumerate(self.layers):
            h.forward(self.x, self.ids[li], self.w[li], out=self.y[li & 1])

    def capture(self):
        self.step_eager()
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step_eager()
        torch.cuda.synchronize(self.dev)

    def step(self, i):
        self.set_step(i)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.step_eager()


class FullDecodeRunner:
    """One token through the WHOLE hot path of every layer, one HIP graph:
         MLA: latent-cache append + absorbed paged attention over `ctx` cached tokens        (a14/a15, every attention layer)
         MoE: router (logits + group-limited top-k) -> routed experts                        (a1, a5-a12, every MoE layer)
    The hidden state fed to router and experts changes every step (nsets pre-generated rows), so routing changes too."""

    def __init__(self, wl, layers, dev, ctx=4096, nsets=16, seed=3, ep=False):
        from ktransformers_amd._native import GateHandle, MLAWrapper
        from ktransformers_amd.parallel import ExpertParallelMoE

        self.wl, self.layers, self.dev, self.nsets = wl, layers, dev, nsets
        # N > 1: every rank decodes its own token (attention + router replicated), the routed experts are sharded
        # expert-parallel: all-gather (x, ids, w) -> local experts -> reduce-scatter (ktransformers_amd/parallel.py)
        self.ep = [ExpertParallelMoE(h) for h in layers] if ep else None
        H, E, k, L = wl["H"], wl["E"], wl["k"], wl["L"]
        gc = wl["gate"]
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        self.x_all = (torch.randn((nsets, 1, H), generator=g, device=dev) / 100).to(torch.bfloat16)
        self.x = self.x_all[0].clone()
        self.y = [torch.empty_like(self.x) for _ in range(2)]
        self.gate = GateHandle(E, H, k, gc["n_group"], gc["topk_group"], gc["scoring_func"], gc["topk_method"],
                               gc["norm_topk_prob"], gc["routed_scaling_factor"])
        self.gate_w = [(torch.randn((E, H), generator=g, device=dev) * H ** -0.5).to(torch.bfloat16) for _ in range(L)]
        self.gate_b = [(torch.randn((E,), generator=g, device=dev) * 0.1) if gc["bias"] else None for _ in range(L)]
        # MLA state: page 64 single-request cache like StaticCache (custom_cache.py:81)
        self.heads, self.ctx, self.nattn = wl["heads"], ctx, wl["attn_layers"]
        pages = (ctx + 1 + 63) // 64
        self.kv = [torch.randn((pages, 64, 576), generator=g, device=dev).to(torch.bfloat16) for _ in range(self.nattn)]
        self.qn = (torch.randn((1, self.heads, 512), generator=g, device=dev)).to(torch.bfloat16)
        self.qp = (torch.randn((1, self.heads, 64), generator=g, device=dev)).to(torch.bfloat16)
        self.new_ckv = torch.randn((1, 512), generator=g, device=dev).to(torch.bfloat16)
        self.new_kpe = torch.randn((1, 64), generator=g, device=dev).to(torch.bfloat16)
        self.page_idx = torch.tensor([ctx // 64], dtype=torch.int32, device=dev)
        self.page_off = torch.tensor([ctx % 64], dtype=torch.int32, device=dev)
        self.mla = MLAWrapper(1, pages, device=dev, max_q_tokens=1, max_splits=int(os.environ.get('KTX_MLA_SPLITS', '256')))
        self.kv_len = torch.tensor([ctx + 1], dtype=torch.int32, device=dev)
        self.mla.plan(None, None, None, self.kv_len, None, self.heads, 512, 64, 64, 192 ** -0.5, max_kv_len=ctx + 1)
        self.graph = None

    def step_eager(self):
        for a in range(self.nattn):
            ckv, k_pe = torch.split(self.kv[a], [512, 64], dim=-1)
            # latent-cache append of the new token is fused into the attention launch
            self.attn_out = self.mla.run(self.qn, self.qp, ckv, k_pe, new_ckv=self.new_ckv, new_kpe=self.new_kpe)
            li = a - (self.nattn - len(self.layers))
            if li >= 0:
                ids, w = self.gate.forward(self.x, self.gate_w[li], self.gate_b[li])
                if self.ep is not None:
                    self.y[li & 1] = self.ep[li].forward(self.x, ids, w)
                else:
                    self.layers[li].forward(self.x, ids, w, out=self.y[li & 1])

    def capture(self):
        self.step_eager()
        torch.cuda.synchronize(self.dev)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step_eager()
            torch.cuda.synchronize(self.dev)
            self.graph = g
        except Exception as e:   # collectives not capturable on this stack: stay eager, say so
            if self.ep is None:
                raise
            log(f"[bench] EP graph capture failed ({type(e).__name__}: {e}); running eagerly")
            self.graph = None

    def step(self, i):
        self.x.copy_(self.x_all[i % self.nsets])
        if self.graph is not None:
            self.graph.replay()
        else:
            self.step_eager()


class RandomLoader:
    """Weight source for the whole-model run: tensors are generated on the device on demand (seeded by their name) with
    the shapes of the meta-device skeleton — there is no network for checkpoints.  Same protocol as util/loader.py."""

    def __init__(self, shapes, dev):
        self.shapes, self.dev, self.tensor_device_map = shapes, dev, {}

    def has_tensor(self, name):
        return name in self.shapes

    def _gen(self, name, shape, scale, mean=0.0):
        import zlib
        g = torch.Generator(device=self.dev)
        g.manual_seed(zlib.crc32(name.encode()))
        return (torch.randn(tuple(shape), generator=g, device=self.dev) * scale + mean).to(torch.bfloat16)

    def load_tensor(self, name, device="cpu"):
        shape = self.shapes[name]
        if "layernorm" in name or name.endswith("norm.weight"):
            return self._gen(name, shape, 0.1, 1.0)
        if "embed_tokens" in name:
            return self._gen(name, shape, 1.0)
        if name.endswith("e_score_correction_bias"):
            return self._gen(name, shape, 0.1).float()
        return self._gen(name, shape, shape[-1] ** -0.5)

    def get_expert_count(self, key):
        n = 0
        while f"{key}.{n}.gate_proj.weight" in self.shapes:
            n += 1
        return n

    def load_experts(self, key, device="cpu"):
        n = self.get_expert_count(key)
        out = {}
        for proj in ("gate", "up", "down"):
            shape = (n,) + tuple(self.shapes[f"{key}.0.{proj}_proj.weight"])
            out[proj] = self._gen(f"{key}.{proj}", shape, 0.1)          # randn/10 like the reference's MoE tests
        return out


class GreedyStep(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, cur_token, position_ids, past_key_values, cache_position):
        logits = self.model(cur_token, position_ids, past_key_values, cache_position)
        return logits[0, -1].argmax(dim=-1).view(1, 1)                   # greedy sampling (utils.py:485-494, do_sample=False)


class ModelDecodeRunner:
    """Whole-model greedy decode of DeepSeek-V2-Lite through the YAML-injected operators, one HIP graph per token:
    embedding -> 27 x [RMSNorm, MLA attention operator (W4 q/kv_a/o projections, RoPE, absorb, paged MQA over `ctx` cached
    tokens, cache append), RMSNorm, dense MLP (layer 0) | router + 6-of-64 int4 routed experts + shared experts] -> RMSNorm ->
    lm_head (W4) -> argmax.  The sampled token is fed back, so routing follows the model."""

    def __init__(self, dev, ctx, max_new, seed=0, use_graph=True):
        from ktransformers_amd.models.custom_cache import StaticCache
        from ktransformers_amd.models.modeling_deepseek import DeepseekForCausalLM, make_config
        from ktransformers_amd.optimize.optimize import optimize_and_load
        from ktransformers_amd.util.generate import CUDAGraphRunner, set_inference_mode
        from ktransformers_amd.util.utils import InferenceState

        cfg = make_config(max_position_embeddings=max(4096, ctx + max_new + 64),
                          rope_scaling={"type": "yarn", "factor": 40, "mscale": 0.707, "mscale_all_dim": 0.707,
                                        "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1})
        self.cfg, self.dev = cfg, dev
        torch.set_default_dtype(torch.bfloat16)
        try:
            with torch.device("meta"):
                model = DeepseekForCausalLM(cfg)
            shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
            rules = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ktransformers_amd", "optimize", "optimize_rules",
                                 "DeepSeek-V2-Lite-Chat.yaml")
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):                   # "Injecting ..." lines
                optimize_and_load(model, rules, RandomLoader(shapes, dev), cfg, default_device=str(dev))
        finally:
            torch.set_default_dtype(torch.float32)
        set_inference_mode(model, InferenceState.GENERATE)
        self.model = model
        self.cache = StaticCache(cfg, 1, ctx + max_new + 64, str(dev), torch.bfloat16)
        for kc in self.cache.key_cache:
            kc.normal_()
        self.cache.past_tokens = [ctx] * cfg.num_hidden_layers           # the prompt the cache pretends to hold
        self.step_mod = GreedyStep(model)
        self.pos = torch.tensor([[ctx]], device=dev, dtype=torch.long)
        self.cur = torch.tensor([[1 + 17 * seed]], device=dev, dtype=torch.long)
        self.runner, self.graph_ok = None, False
        if use_graph:
            try:
                r = CUDAGraphRunner()
                with torch.no_grad():
                    r.capture(self.step_mod, self.cur, self.pos, self.pos[0], self.cache, main_device=str(dev))
                self.runner, self.graph_ok = r, True
            except Exception as e:   # e.g. collectives that cannot be captured on this stack: stay eager, say so
                log(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize(dev)

    def moe_handles(self):
        return [l.mlp.experts.generate_experts.handle for l in self.model.model.layers if hasattr(l.mlp, "experts")]

    def linear_bytes(self):
        tot = 0
        for m in self.model.modules():
            h = getattr(m, "_h", None)
            if h is not None and hasattr(h, "weight_bytes"):
                tot += h.weight_bytes()
        return tot

    def rewind(self, n):
        """Forget n generated positions (pre-warm steps): the timed run starts at the configured context length again."""
        self.pos -= n

    @torch.no_grad()
    def step(self, i):
        nxt = self.runner(self.cur, self.pos, self.pos[0]) if self.runner is not None else \
            self.step_mod(self.cur, self.pos, self.cache, self.pos[0])
        self.cur.copy_(nxt)
        self.pos += 1


def timed(fn, steps, warmup, dev, dist_on):
    import torch.distributed as dist

    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(wl, budget_s=20.0):
    """The reference's own AVX512 MoE kernels (oracle/_ref) on this box's host cores, same layer shape, bs=1 decode.
    Bounded sample: 3 distinct layers' weights (> the host's L3), forward rotating over them for ~budget_s."""
    import numpy as np

    try:
        from oracle.oracle import FMT_AMXINT4, Reference, f32_to_bf16, reference_available
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference", "sample": f"unavailable: {e}"}
    if not reference_available():
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/libkt_ref.so not present or host lacks AVX512-VNNI/BF16"}
    H, I, E, k, L = wl["H"], wl["I"], wl["E"], wl["k"], wl["L"]
    ncpu = os.cpu_count() or 8
    threads = max(1, min(64, ncpu // 2))  # physical cores of one socket-ish; reference guidance: physical cores only
    ref = Reference(threads=threads, subpools=1)
    rng = np.random.default_rng(0)
    nlayers = 3
    moes = []
    t_load = time.perf_counter()
    # one block of randn/10 bf16 values, re-used with cheap permutations so that every matrix of every layer is
    # distinct in memory (what matters for a bandwidth-bound baseline) without minutes of single-threaded numpy RNG
    base = f32_to_bf16((rng.standard_normal((E, I, H), dtype=np.float32) / 10))
    for li in range(nlayers):
        gate = np.roll(base, li + 1, axis=0)
        up = np.ascontiguousarray(base[::-1]) if li % 2 == 0 else np.roll(base, -(li + 2), axis=0)
        down = np.roll(base, li + 3, axis=0).reshape(E, H, I)
        moes.append(ref.make_moe(FMT_AMXINT4, gate, up, down, k=k, max_len=32))
    t_load = time.perf_counter() - t_load
    x = f32_to_bf16((rng.standard_normal((1, H), dtype=np.float32) / 100))
    sets = [(np.stack([rng.permutation(E)[:k]]).astype(np.int64), rng.random((1, k), dtype=np.float32)) for _ in range(64)]
    for i in range(30):
        ref.moe_forward(moes[i % nlayers], sets[i % 64][0], sets[i % 64][1], x)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 20000:
        for _ in range(50):
            ref.moe_forward(moes[n % nlayers], sets[n % 64][0], sets[n % 64][1], x)
            n += 1
    dt = time.perf_counter() - t0
    t_layer = dt / n
    return {"value": round(1.0 / (L * t_layer), 3), "unit": "tok/s", "cores": threads, "kind": "reference",
            "us_per_layer": round(t_layer * 1e6, 1),
            "sample": f"{n} bs=1 forwards of TP_MOE<AMX_MOE_TP<GemmKernel224Int4>> (AVX512-VNNI path, no AMX on this host) "
                      f"rotating over {nlayers} distinct {wl['desc'].split(',')[0]} layers, {threads} threads, 1 subpool; "
                      f"tok/s = 1/({L} layers x t_layer); weight quant took {t_load:.1f}s (untimed)"}


def cpu_baseline_subprocess(workload, timeout_s=240):
    """Run the CPU leg in a child process: the reference kernels abort() on assertion failures and hold ~GBs of
    host memory; neither may take the bench line down with it."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload],
                           capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference",
                "sample": f"child failed rc={r.returncode}: {r.stderr.strip()[-300:]}"}
    except Exception as e:
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference", "sample": f"child error: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="v2lite-int4", choices=sorted(WORKLOADS))
    ap.add_argument("--prefill-tokens", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--hot-path", default="model", choices=["model", "full", "moe"],
                    help="model (default): the whole injected DeepSeek-V2-Lite decoder stack, greedy decode (every §8a row: "
                         "linears, norms, RoPE, MLA attention operator, router, routed + shared experts, lm_head); "
                         "full: only the MLA kernel + router + routed experts of every layer; moe: routed experts only")
    ap.add_argument("--ctx", type=int, default=4096, help="cached tokens the MLA decode attends over")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(WORKLOADS[args.workload])), flush=True)
        return

    wl = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ktransformers_amd import _native
    from ktransformers_amd.parallel import ExpertParallelMoE

    H, I, E, k, L = wl["H"], wl["I"], wl["E"], wl["k"], wl["L"]
    assert E % world == 0
    e_local = E // world
    max_len = max(args.prefill_tokens if not args.no_prefill else 1, world, 1)
    whole = args.hot_path == "model" and args.workload == "v2lite-int4"
    layers = None
    if not (whole and dist_on):   # stand-alone expert layers: the kernel-level measurements (and the non-model step types)
        t0 = time.perf_counter()
        layers = build_layers(wl, dev, max_len=max_len, expert_begin=rank * e_local, expert_num=e_local)
        if rank == 0:
            log(f"[bench] {L} layers x {e_local} experts resident: {sum(h.weight_bytes for h in layers) / 2**30:.2f} GiB packed, "
                f"built in {time.perf_counter() - t0:.1f}s")

    # ---------------- decode ----------------
    if whole:
        if dist_on:   # experts sharded over the ranks, attention / dense parts replicated, one token stream per rank
            from ktransformers_amd.parallel import enable_expert_parallel
            enable_expert_parallel()
        t0 = time.perf_counter()
        mr = ModelDecodeRunner(dev, args.ctx, args.steps + args.warmup + 464, seed=rank, use_graph=not args.no_graph)
        if rank == 0:
            log(f"[bench] whole-model skeleton injected and loaded in {time.perf_counter() - t0:.1f}s"
                + ("" if mr.graph_ok else " (graph capture failed: eager launches)"))
        step = mr.step
        # bring the GPU to its sustained clocks before the driver's W warm-up + K timed steps: a fresh process that has only
        # loaded weights starts the first few hundred graph replays at idle clocks (measured: 390 vs 465 tok/s)
        t_pre = time.perf_counter()
        n_pre = 0
        # (N > 1: every step issues collectives, so all ranks must run the SAME number of steps — a fixed count, not a time box)
        while n_pre < 200 if dist_on else (time.perf_counter() - t_pre < 0.7 and n_pre < 400):
            for _ in range(20):
                mr.step(n_pre)
                n_pre += 1
            torch.cuda.synchronize(dev)
        mr.rewind(n_pre)
    elif dist_on and args.hot_path == "moe":
        runner = ExpertParallelMoE.bench_runner(wl, layers, dev, world, rank, use_graph=not args.no_graph)
        step = runner.step
    else:
        r = FullDecodeRunner(wl, layers, dev, ctx=args.ctx, seed=3 + rank, ep=dist_on) if args.hot_path == "full" \
            else DecodeRunner(wl, layers, T=1, dev=dev)
        if not args.no_graph:
            r.capture()
        step = r.step
    tokens_per_step = world  # one token per rank per step (weak scaling)
    dt = timed(step, args.steps, args.warmup, dev, dist_on)
    ms_per_step = dt / args.steps * 1e3
    decode_tps = tokens_per_step * args.steps / dt

    out = {
        "metric": "decode tokens/s (DeepSeek-V2-Lite, int4 experts + W4 linears + MLA resident in HBM, whole-model greedy decode)"
                  if whole else "decode tokens/s (MoE + MLA hot path, int4 experts resident in HBM)",
        "value": round(decode_tps, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8xint4->int32 (bf16 io)", "data": "synthetic",
        "config": {"workload": wl["desc"], "hidden": H, "intermediate": I, "experts": E, "top_k": k, "moe_layers": L,
                   "batch_per_gpu": 1, "parallelism": f"ep{world}" if dist_on else "single",
                   "hip_graph": not args.no_graph,
                   "step": ("one greedy token through the YAML-injected DeepSeek-V2-Lite: embedding, 27 x [RMSNorm, MLA attention "
                            "operator (W4-g64 q/kv_a/o projections, YaRN RoPE, absorb, paged MQA over ctx %d, cache append), "
                            "RMSNorm, dense MLP | router + 6-of-64 AMXINT4 routed experts + W4 shared experts], RMSNorm, W4 "
                            "lm_head, argmax; the sampled token is fed back; random weights" % args.ctx)
                   if whole else
                           ("MLA cache-append + absorbed paged attention (ctx %d, %d heads, %d layers) + router + routed "
                            "experts (%d layers)" % (args.ctx, wl["heads"], wl["attn_layers"], L))
                   + ("; routed experts sharded expert-parallel over %d ranks (all-gather + reduce-scatter per layer), "
                      "attention and router replicated, one token per rank" % world if dist_on else "")
                   if args.hot_path in ("full", "model") else "router-less routed experts only"},
    }

    if not dist_on:
        # ---------------- roofline of the dominant kernel: HIP events on the launch stream ----------------------------
        # The dominant kernel (decode gate/up) is launched alone, once per layer, from a HIP graph (library test hook
        # ktx_debug_set(2, 1) = "gate/up only"); two events on the replay stream bracket R replays, so the average
        # covers exactly L*R back-to-back launches of that kernel — the same quantity rocprofv3 --kernel-trace reports.
        def kernel_only_us(which):
            _native.lib.ktx_debug_set(2, which)
            rk = DecodeRunner(wl, layers, T=1, dev=dev)
            rk.capture()
            for i in range(5):
                rk.step(i)
            torch.cuda.synchronize(dev)
            R = max(20, min(args.steps, 200))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for i in range(R):
                rk.set_step(i)
                e0.record()
                rk.graph.replay()
                e1.record()
                e1.synchronize()
                tot += e0.elapsed_time(e1)
            _native.lib.ktx_debug_set(2, 0)
            return tot / (R * L) * 1e3

        if whole:
            out["weight_bytes_streamed_per_token"] = int(mr.linear_bytes() + L * (k * 3 * H * I * 0.5))
            del mr
            torch.cuda.empty_cache()
            # the MLA kernel + router + routed experts alone (the step this bench timed in its first profiles)
            rf = FullDecodeRunner(wl, layers, dev, ctx=args.ctx)
            rf.capture()
            dtf = timed(rf.step, max(50, args.steps // 2), 10, dev, False)
            out["mla_router_experts_only"] = {"value": round(max(50, args.steps // 2) / dtf, 2), "unit": "tok/s",
                                              "ms_per_step": round(dtf / max(50, args.steps // 2) * 1e3, 4)}
            del rf
        # routed-experts-only decode rate, for continuity with earlier profiles
        rm = DecodeRunner(wl, layers, T=1, dev=dev)
        rm.capture()
        dtm = timed(rm.step, max(50, args.steps // 2), 10, dev, False)
        out["moe_only"] = {"value": round(max(50, args.steps // 2) / dtm, 2), "unit": "tok/s",
                           "ms_per_step": round(dtm / max(50, args.steps // 2) * 1e3, 4)}
        gu_us = kernel_only_us(1)
        dn_us = kernel_only_us(2)
        gu_bytes = k * 2 * I * H * 0.5 + k * 2 * I * 4 + H * 2  # packed gate+up of k experts + fp32 row scales + bf16 x row
        dn_bytes = k * H * I * 0.5 + k * H * 4 + k * I * 2 + H * 2
        ach = gu_bytes / (gu_us * 1e-6) / 1e9
        # HBM bytes per launch from the PMC passes (collected offline with rocprofv3 --pmc, see scripts/pmc_summary.py;
        # counters cannot be read from inside this process)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_decode.json")))
            for name, e in pmc["kernels"].items():
                if name.startswith("moe_dec_gateup_kernel") and args.workload == "v2lite-int4":
                    traffic = e["hbm_bytes"]
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "hbm",
                           "kernel": "moe_dec_gateup_kernel (x-quant + gate/up W4A8 MFMA GEMV + SiLU*up, decode path)",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_source": "profiles/r01_pmc_decode.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if traffic else None,
                           "algorithmic_bytes_per_launch": int(gu_bytes), "avg_launch_us": round(gu_us, 3)}
        out["kernels_us"] = {"gate_up": round(gu_us, 3), "down_combine": round(dn_us, 3)}
        out["down_kernel_GBs"] = round(dn_bytes / (dn_us * 1e-6) / 1e9, 1)
        layer_bytes = gu_bytes + dn_bytes
        out["step_GBs"] = round(L * layer_bytes / (ms_per_step * 1e-3) / 1e9, 1)

        # ---------------- prefill: one prompt chunk through the same layers -------------------------------------------
        if not args.no_prefill:
            Tp = args.prefill_tokens
            rp = DecodeRunner(wl, layers, T=Tp, dev=dev, nsets=2, seed=5)
            psteps = max(3, min(10, args.steps // 20))
            dtp = timed(lambda i: rp.step(i), psteps, 2, dev, False)
            out["prefill"] = {"value": round(Tp * psteps / dtp, 1), "unit": "tok/s", "tokens": Tp,
                              "ms_per_chunk": round(dtp / psteps * 1e3, 3),
                              "tflops": round(2 * 3 * H * I * k * Tp * L / (dtp / psteps) / 1e12, 1)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args.workload)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
