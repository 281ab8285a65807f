import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle.oracle import *
from helpers import *
from ktransformers_amd import _native
from ktransformers_amd._native import MoEHandle
o = Oracle()
dev = torch.device('cuda', 0)
ok = True
for fmt, method in ((FMT_AMXINT4, 'AMXINT4'), (FMT_AMXINT8, 'AMXINT8')):
  for (E,k,H,I,T) in [(8,2,512,256,1),(8,2,512,256,5),(8,6,2048,1408,1),(16,8,7168,2048,2),(8,2,256,512,40),(16,4,2048,1408,3),(8,2,256,512,300),(4,2,256,256,700)]:
    c = make_case(1, E,k,H,I,T, invalid_ids=(T>=5))
    mo = o.make_moe(fmt, c['gate'], c['up'], c['down'])
    yo = o.moe_forward(mo, c['ids'], c['w'], c['x'])
    h = MoEHandle(E,k,H,I,max_len=max(T,8),method=method,device=0)
    h.load_bf16(torch_bf16(c['gate'],dev), torch_bf16(c['up'],dev), torch_bf16(c['down'],dev))
    for force in (False, True):
        _native.force_generic_path(force)
        y = h.forward(torch_bf16(c['x'],dev), torch.from_numpy(c['ids']).to(dev), torch.from_numpy(c['w']).to(dev))
        torch.cuda.synchronize()
        yg = numpy_u16(y)
        nd = int((yg != yo).sum())
        y2 = h.forward(torch_bf16(c['x'],dev), torch.from_numpy(c['ids']).to(dev), torch.from_numpy(c['w']).to(dev), out=y.clone(), incremental=True)
        yo2 = o.moe_forward(mo, c['ids'], c['w'], c['x'], y_prev=yo)
        nd2 = int((numpy_u16(y2)!=yo2).sum())
        print(method,(E,k,H,I,T),'generic' if force else 'auto','mismatch',nd,'of',yo.size, 'incr', nd2, flush=True)
        if nd or nd2:
            ok = False
            a=bf16_to_f32(yg); b=bf16_to_f32(yo); print('   max abs', np.abs(a-b).max(), 'ref absmean', np.abs(b).mean())
    _native.force_generic_path(False)
    h.close()
print('ALL OK' if ok else 'FAILED')
