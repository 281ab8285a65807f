mkdir -p gpurun_out/r6C
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from ktransformers_amd import _native as n
dev = torch.device("cuda", 0)
E_, k, H, I, T = 64, 8, 7168, 2048, 700
g = torch.Generator(device=dev); g.manual_seed(0)
h = n.MoEHandle(E_, k, H, I, max_len=T, method="AMXINT4", device=0)
mk = lambda *s: (torch.randn(s, generator=g, device=dev, dtype=torch.bfloat16) * 0.1)
h.load_bf16(mk(E_, I, H), mk(E_, I, H), mk(E_, H, I))
x = (torch.randn((T, H), generator=g, device=dev) * 0.5).to(torch.bfloat16)
ids = torch.multinomial(torch.ones(T, E_), k).to(torch.int64).to(dev)
w = torch.rand((T, k), generator=g, device=dev)
outs = []
for kn in (0, 1):
    n.lib.ktx_debug_set(25, kn)
    outs.append(h.forward(x, ids, w).clone())
torch.cuda.synchronize()
print("NWV=4 output identical to NWV=8:", bool(torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))))
PY
for T in 2048 4096 8192; do for kn in 0 1 0 1; do
python scripts/fmt_prompt_bench.py --fmt AMXINT4 --T $T --knob 25=$kn 2>&1 | tail -1
done; done | tee gpurun_out/r6C/stream_nwv4.txt
