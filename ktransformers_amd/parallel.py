"""Expert parallelism for the routed-expert path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no GPU expert parallelism of its own for this path: its analogue is the NUMA tensor-parallel split
of TP_MOE_Common::forward + merge_results (kt-kernel/operators/moe-tp.hpp:201-246, operators/amx/moe_base.hpp:749-791),
where every part sees all tokens, computes an fp32 partial [T,H], and the partials are summed in fp32 before the single
bf16 rounding.  Sharding by EXPERT instead of by intermediate column keeps exactly that reduce shape:

  prefill / large T: all-to-all-v dispatch + combine (ep_prefill_forward) — bit-identical to the single-GPU forward.
  decode / small T ("replicate + reduce"): all-gather the ranks' token rows (x, ids, w — a few KiB),
  every rank runs the experts it owns on all gathered tokens (ids outside its range are skipped inside the kernel),
  reduce-scatter the fp32 partials so each rank receives the sum for its own tokens, round to bf16 once.

The fp32 summation order across ranks differs from the single-GPU slot order j=0..k-1, so EP outputs match the
single-GPU ones to fp32 rounding (<=1 bf16 ulp), not bit-for-bit; tests state that tolerance.

Decode has two transports behind the same ep_decode_forward signature:
  collectives (default, and the only one on CPU / gloo): all_gather_into_tensor + reduce_scatter_tensor;
  peer writes (enable_peer_exchange, GPU only; include/ktx_ep.h): two launches per layer that write tagged 8-byte
  granules straight into the peers' buffers over xGMI and add the partials in RANK order — the fixed part order of the
  reference's merge_results — so the result is bit-identical to sum_r partial_r evaluated left to right.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


# Process-wide expert-parallel setting consulted by KExpertsHIP at load(): rule files cannot carry a process group.
EP_STATE = {"enabled": False, "group": None, "exchange": None, "replicated_input": False}


def set_replicated_input(on: bool = True) -> None:
    """Strong scaling (ONE token stream over all ranks): every rank holds the SAME decode rows (attention and router are
    replicated, so they are bit-identical), so the gather half of the decode exchange is dropped — each rank runs its own
    experts on its own copy of the rows and only the fp32 partials travel: one all-reduce (collectives) or the reduce launch
    of the peer exchange, parts added in rank order on every rank alike, so every rank ends up with the same bits and the
    streams cannot drift apart."""
    EP_STATE["replicated_input"] = bool(on)


def enable_expert_parallel(group=None, enabled: bool = True) -> None:
    """After torch.distributed.init_process_group: every KExpertsHIP loaded from now on owns experts
    expert_range(E, world, rank) and runs its forward through the EP choreography below."""
    EP_STATE["enabled"] = bool(enabled)
    EP_STATE["group"] = group


def _all_ranks_ok(ok: bool, what: str, group) -> None:
    """Collective: raise the same error on EVERY rank if any rank failed, so that no rank goes on alone."""
    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, None if ok is True else str(ok), group=group)
    bad = [(r, f) for r, f in enumerate(flags) if f is not None]
    if bad:
        raise RuntimeError(f"peer exchange: {what} failed on rank(s) " + "; ".join(f"{r}: {f}" for r, f in bad))


def enable_peer_exchange(hidden: int, topk: int, max_tokens: int, device, group=None, memory: str | None = None,
                         verify: bool = True):
    """Collective call (every rank of `group`, after init_process_group): allocate this rank's symmetric buffer, swap the
    inter-process handles through the process group, map the peers, and — verify=True — push one known pattern through
    both kernels on the real fabric before anything depends on it.  From then on ExpertParallelMoE.forward (decode) takes
    the peer-write transport for T <= max_tokens.  Returns the EpExchange; raises ON EVERY RANK if any rank could not set
    the transport up or did not get the pattern back (the caller decides whether to go on with the collectives — never
    silently, and never with the ranks disagreeing about the transport)."""
    import os

    from ktransformers_amd._native import EpExchange
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ex, handle, ok = None, None, True
    try:
        ex = EpExchange(world, rank, max_tokens, hidden, topk, device, memory or os.environ.get("KTX_EP_MEMORY", "uncached"))
        handle = ex.export_handle()
    except Exception as e:
        ok = f"{type(e).__name__}: {e}"
    handles = [None] * world
    dist.all_gather_object(handles, handle, group=group)
    if ok is True and all(h is not None for h in handles):
        try:
            for r in range(world):
                if r != rank:
                    ex.import_handle(r, handles[r])
        except Exception as e:
            ok = f"{type(e).__name__}: {e}"
    elif ok is True:
        ok = "a peer has no buffer to map"
    try:
        _all_ranks_ok(ok, "mapping the peers' buffers", group)      # also: every buffer mapped everywhere before the first put
        if verify:
            try:
                verify_peer_exchange(ex)
            except Exception as e:
                ok = f"{type(e).__name__}: {e}"
            _all_ranks_ok(ok, "the transport self-check", group)
    except Exception:
        if ex is not None:
            ex.close()
        raise
    EP_STATE["exchange"] = ex
    return ex


def check_exchange_status(group=None) -> None:
    """Raise — on EVERY rank — if a bounded poll of the peer-write exchange has given up since the last check (a dead or
    very slow peer: the kernels then finished on stale granules and every token since is suspect).  The collectives path
    would have raised by itself; the peer path only sets a status word, so the decode loop asks for it at a cheap cadence
    (util/generate.py: every 64 tokens and at the end; bench.py: after the timed region).  One tiny all-reduce."""
    ex = EP_STATE.get("exchange")
    if ex is None or not (dist.is_available() and dist.is_initialized()):
        return
    st = torch.tensor([int(ex.status())], device=ex.device, dtype=torch.int32)
    dist.all_reduce(st, op=dist.ReduceOp.MAX, group=group if group is not None else EP_STATE.get("group"))
    if int(st.item()) != 0:
        raise RuntimeError(f"expert-parallel peer exchange: a poll gave up waiting for a peer (status {int(st.item())}); "
                           "outputs since the previous check are not valid")


def verify_peer_exchange(ex, T: int = 1) -> None:
    """One gather + reduce of rank-dependent patterns; raises unless every row and every sum arrives exactly."""
    dev, R, H, k = ex.device, ex.world, ex.H, ex.k
    col = torch.arange(H, device=dev, dtype=torch.float32)

    def xrow(r):       # exactly representable in bf16
        return ((col % 61) - 30 + r).to(torch.bfloat16).expand(T, H).contiguous()

    ids = (torch.arange(T * k, device=dev, dtype=torch.int64).view(T, k) * 7 + ex.rank)
    w = (torch.arange(T * k, device=dev, dtype=torch.float32).view(T, k) + 0.5 * ex.rank)
    xg, idsg, wg = ex.gather(xrow(ex.rank), ids, w)
    # part[row of rank r] = what THIS rank contributes to rank r's tokens: small integers, so every order of adding is exact
    part = torch.stack([(col % 17) * (ex.rank + 1) + r for r in range(R)]).repeat_interleave(T, dim=0).contiguous()
    out = ex.reduce(part)
    st = ex.status()
    if st != 0:
        raise RuntimeError(f"peer exchange self-check: a poll gave up waiting for a peer (status {st})")
    for r in range(R):
        sl = slice(r * T, (r + 1) * T)
        ok = torch.equal(xg[sl], xrow(r)) and torch.equal(idsg[sl], ids - ex.rank + r) and torch.equal(wg[sl], w + 0.5 * (r - ex.rank))
        if not ok:
            raise RuntimeError(f"peer exchange self-check: rank {r}'s token row arrived damaged at rank {ex.rank}")
    want = ((col % 17) * (R * (R + 1) // 2) + R * ex.rank).to(torch.bfloat16).expand(T, H)
    if not torch.equal(out, want):
        raise RuntimeError(f"peer exchange self-check: reduced partials differ at rank {ex.rank}")


def ep_decode_forward(local_partial: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                      x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor, group=None, exchange=None) -> torch.Tensor:
    """x bf16 [T,H], ids int64 [T,k], w fp32 [T,k] (this rank's tokens) -> bf16 [T,H].

    `local_partial(xg, idsg, wg) -> fp32 [world*T, H]` computes this rank's experts' contribution for all gathered
    tokens (MoEHandle.forward_partial on GPU; the oracle in the gloo tests).  `exchange` (an EpExchange whose peers are
    mapped) selects the peer-write transport: two launches, partials added in rank order."""
    if EP_STATE["replicated_input"]:
        part = local_partial(x.contiguous(), ids.contiguous(), w.contiguous())       # fp32 [T, H]: this rank's experts only
        if exchange is not None:
            # the reduce launch hands row block r of `part` to rank r: every rank's block is the same T rows here.  No gather runs
            # in this mode, so the launch must own its call tag (ktx_ep_reduce_only): with the plain reduce every call would
            # re-use the tag the set-up check left behind and could add a previous call's granules
            return exchange.reduce(part.repeat(exchange.world, 1), reduce_only=True)
        # rank-ordered sum on every rank (all_reduce's ring order differs between ranks at fp32 rounding level, which would let
        # the replicas drift apart): gather the R partials and add them left to right
        world = dist.get_world_size(group)
        Tn = part.shape[0]
        parts = torch.empty((world * Tn, part.shape[1]), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(parts, part.contiguous(), group=group)
        out = parts[:Tn]
        for r in range(1, world):
            out = out + parts[r * Tn:(r + 1) * Tn]
        return out.to(torch.bfloat16)
    if exchange is not None:
        xg, idsg, wg = exchange.gather(x.contiguous(), ids.contiguous(), w.contiguous())
        return exchange.reduce(local_partial(xg, idsg, wg))
    world = dist.get_world_size(group)
    T, H = x.shape
    k = ids.shape[1]
    # one all-gather per layer instead of three: the token's activation row, expert ids and routing weights travel as one
    # byte row (decode collectives are latency-bound: ~20 us each on xGMI, 26 MoE layers per token)
    nx, ni = H * x.element_size(), k * ids.element_size()
    row = torch.cat([x.contiguous().view(torch.uint8).view(T, nx), ids.contiguous().view(torch.uint8).view(T, ni),
                     w.contiguous().view(torch.uint8).view(T, k * w.element_size())], dim=1)
    g = torch.empty((world * T, row.shape[1]), dtype=torch.uint8, device=x.device)
    dist.all_gather_into_tensor(g, row, group=group)
    xg = g[:, :nx].contiguous().view(x.dtype).view(world * T, H)
    idsg = g[:, nx:nx + ni].contiguous().view(ids.dtype).view(world * T, k)
    wg = g[:, nx + ni:].contiguous().view(w.dtype).view(world * T, k)
    part = local_partial(xg, idsg, wg)
    out = torch.empty((T, H), dtype=torch.float32, device=x.device)
    dist.reduce_scatter_tensor(out, part, op=dist.ReduceOp.SUM, group=group)
    return out.to(torch.bfloat16)


def ep_prefill_forward(local_rows: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                       combine: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                       x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor, E: int, group=None,
                       stats: dict | None = None) -> torch.Tensor:
    """Prefill / large-T expert parallelism: all-to-all-v dispatch of token rows to the ranks that own their
    slots' experts, local experts, all-to-all-v back of the per-pair outputs, slot-ordered combine at the token's home rank.

    x bf16 [T,H], ids int64 [T,k] (global expert ids), w fp32 [T,k] — this rank's tokens; returns bf16 [T,H].
    `local_rows(rows bf16 [n,H], expert_ids int64 [n]) -> bf16 [n,H]` = Expert_id(row) (MoEHandle.forward with k=1 and
    weight 1.0, which returns the expert's bf16 output unchanged); `combine(rows, row_of_pair int32 [T,k], w) -> bf16 [T,H]`
    is the single-GPU combine (ktransformers_amd._native.moe_combine).  Because the per-pair expert outputs and the
    combine are the single-GPU ones, the result is bit-identical to the single-GPU forward (unlike the decode path,
    whose cross-rank fp32 reduction reorders the sum).

    Dispatch is de-duplicated per destination (round 6): a token whose k slots name several experts of ONE rank crosses the
    fabric once — the row goes with the (token, destination) pair, the slots travel as 16 bytes of metadata (expert id, index
    of the row inside the sender's block) and the destination re-expands rows to pairs from its own HBM.  With k = 8 slots dealt
    uniformly over 8 ranks a token reaches 5.25 distinct ranks on average: 34 % fewer dispatch bytes.  Volume per rank:
    (#distinct (token, destination) pairs leaving) * H * 2 B out, (#pairs) * H * 2 B back (the outputs differ per pair).
    The split sizes are read on the host (one small sync): prefill is not graph-captured in the reference either.
    `stats` (optional dict) receives rows_out / pairs_out / rows_in / pairs_in of this call."""
    world = dist.get_world_size(group)
    T, H = x.shape
    k = ids.shape[1]
    if E % world != 0:
        raise ValueError(f"expert count {E} not divisible by world size {world}")
    per = E // world
    dev = x.device
    flat = ids.reshape(-1)
    valid = (flat >= 0) & (flat < E)
    dest = torch.where(valid, torch.div(flat, per, rounding_mode="floor"), torch.full_like(flat, world))
    tok = torch.div(torch.arange(T * k, device=dev), k, rounding_mode="floor")
    key = dest * T + tok                                            # destination-major, token-minor; stable: slot order kept
    order = torch.argsort(key, stable=True)
    pair_counts_t = torch.bincount(dest, minlength=world + 1)[:world]
    n_send = int(pair_counts_t.sum())
    sel = order[:n_send]                                            # the valid pairs, grouped by destination then token
    sk = key[sel]
    first = torch.ones(n_send, dtype=torch.bool, device=dev)
    first[1:] = sk[1:] != sk[:-1]
    urow = torch.cumsum(first.to(torch.int64), 0) - 1               # pair -> its row among the distinct (destination, token) rows
    ukey = sk[first]
    u_dest, u_tok = torch.div(ukey, T, rounding_mode="floor"), ukey % T
    row_counts_t = torch.bincount(u_dest, minlength=world)[:world] if n_send else torch.zeros(world, dtype=torch.int64, device=dev)
    row_off = torch.cumsum(row_counts_t, 0) - row_counts_t
    send_counts_t = torch.stack([row_counts_t, pair_counts_t], dim=1).contiguous()      # [world, 2]
    recv_counts_t = torch.empty_like(send_counts_t)
    dist.all_to_all_single(recv_counts_t, send_counts_t, group=group)
    send_rows, send_pairs = send_counts_t[:, 0].tolist(), send_counts_t[:, 1].tolist()
    recv_rows, recv_pairs = recv_counts_t[:, 0].tolist(), recv_counts_t[:, 1].tolist()
    n_rows_out, n_rows_in, n_recv = sum(send_rows), sum(recv_rows), sum(recv_pairs)
    if stats is not None:
        stats.update(rows_out=n_rows_out, pairs_out=n_send, rows_in=n_rows_in, pairs_in=n_recv)
    send_x = x[u_tok].contiguous()
    meta = torch.stack([flat[sel], urow - row_off[dest[sel]]], dim=1).contiguous()      # (expert id, row inside this rank's block)
    recv_x = torch.empty((n_rows_in, H), dtype=x.dtype, device=dev)
    recv_meta = torch.empty((n_recv, 2), dtype=meta.dtype, device=dev)
    dist.all_to_all_single(recv_x, send_x, recv_rows, send_rows, group=group)
    dist.all_to_all_single(recv_meta, meta, recv_pairs, send_pairs, group=group)
    if n_recv:
        rr = recv_counts_t[:, 0]
        src = torch.repeat_interleave(torch.arange(world, device=dev), recv_counts_t[:, 1])
        rows_of_pairs = recv_x[recv_meta[:, 1] + (torch.cumsum(rr, 0) - rr)[src]]
        out_rows = local_rows(rows_of_pairs, recv_meta[:, 0].contiguous())
    else:
        out_rows = recv_x[:0]
    back = torch.empty((n_send, H), dtype=x.dtype, device=dev)
    dist.all_to_all_single(back, out_rows.contiguous(), send_pairs, recv_pairs, group=group)
    row_of_pair = torch.full((T * k,), -1, dtype=torch.int32, device=dev)
    row_of_pair[sel] = torch.arange(n_send, dtype=torch.int32, device=dev)
    if n_send == 0:
        back = torch.zeros((1, H), dtype=x.dtype, device=dev)
    return combine(back, row_of_pair.view(T, k), w)


def expert_range(E: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous expert shard [begin, begin+count) of rank `rank`."""
    if E % world != 0:
        raise ValueError(f"expert count {E} not divisible by world size {world}")
    n = E // world
    return rank * n, n


class ExpertParallelMoE:
    """One MoE layer sharded over the ranks of `group`; wraps a local MoEHandle created with
    expert_begin/expert_num = expert_range(E, world, rank)."""

    def __init__(self, handle, group=None):
        self.handle = handle
        self.group = group

    def forward(self, x, ids, w):
        ex = EP_STATE["exchange"]
        if ex is not None and (x.shape[0] > ex.max_tokens or x.shape[1] != ex.H or ids.shape[1] != ex.k):
            ex = None
        return ep_decode_forward(self.handle.forward_partial, x, ids, w, self.group, exchange=ex)

    def forward_prefill(self, x, ids, w):
        """All-to-all-v dispatch / combine; bit-identical to the single-GPU forward.  The local handle must have been
        created with max_len >= the rows this rank can receive (world * T * k in the worst case)."""
        from ktransformers_amd._native import moe_combine

        E = self.handle.E * dist.get_world_size(self.group)

        def local_rows(rows, eids):
            ones = torch.ones((rows.shape[0], 1), dtype=torch.float32, device=rows.device)
            return self.handle.forward(rows, eids.view(-1, 1), ones)

        return ep_prefill_forward(local_rows, moe_combine, x, ids, w, E, self.group)

    # ---- bench support ------------------------------------------------------------------------------------------
    @staticmethod
    def bench_runner(wl, layers, dev, world, rank, use_graph=True):
        return _EPBenchRunner(wl, layers, dev, world, rank, use_graph)


class _EPBenchRunner:
    """Each rank decodes its own token stream (weak scaling); per layer all-gather + local experts + reduce-scatter."""

    def __init__(self, wl, layers, dev, world, rank, use_graph, nsets=16):
        E, k, L, H = wl["E"], wl["k"], wl["L"], wl["H"]
        self.layers, self.dev, self.nsets = [ExpertParallelMoE(h) for h in layers], dev, nsets
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        scores = torch.rand((nsets, L, 1, E), generator=g, device=dev)
        self.ids_all = scores.topk(k, dim=-1).indices.to(torch.int64).contiguous()
        self.w_all = torch.rand((nsets, L, 1, k), generator=g, device=dev, dtype=torch.float32).contiguous()
        self.ids, self.w = self.ids_all[0].clone(), self.w_all[0].clone()
        self.x = (torch.randn((1, H), generator=g, device=dev) / 100).to(torch.bfloat16)
        self.graph = None
        self._eager()
        torch.cuda.synchronize(dev)
        if use_graph:
            try:
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph):
                    self._eager()
                torch.cuda.synchronize(dev)
                self.graph = gph
            except Exception as e:  # collectives not capturable on this stack: stay eager, say so
                import sys
                print(f"[bench] EP graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                self.graph = None

    def _eager(self):
        for li, m in enumerate(self.layers):
            self.out = m.forward(self.x, self.ids[li], self.w[li])

    def step(self, i):
        s = i % self.nsets
        self.ids.copy_(self.ids_all[s])
        self.w.copy_(self.w_all[s])
        if self.graph is not None:
            self.graph.replay()
        else:
            self._eager()
