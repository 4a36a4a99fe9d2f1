import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hifihr_amd.network import ResEncoder
torch.manual_seed(5)
ref = ResEncoder(pretrain="res50", conv_impl="aten").train()
hip = ResEncoder(pretrain="res50", conv_impl="mfma").cuda().train()
hip.load_state_dict(ref.state_dict())
B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 128
x = torch.rand(B, 3, H, H)
low_r, f_r = ref(x)
low_h, f_h = hip(x.cuda())
print("fwd err", float((f_h.cpu() - f_r).abs().max()) / float(f_r.abs().max()), float((low_h.cpu() - low_r).abs().max()) / float(low_r.abs().max()))
gen = torch.Generator().manual_seed(1)
wf, wl = torch.randn(f_r.shape, generator=gen), torch.randn(low_r.shape, generator=gen)
((f_r * wf).sum() + 0.1 * (low_r * wl).sum()).backward()
((f_h * wf.cuda()).sum() + 0.1 * (low_h * wl.cuda()).sum()).backward()
rows = []
for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
    if q.grad is None: continue
    scale = float(q.grad.abs().max())
    err = float((p.grad.cpu() - q.grad).abs().max())
    rows.append((err / max(scale, 1e-12), n, scale, tuple(q.shape)))
rows.sort(reverse=True)
for r in rows[:12]: print("%.3e  %-45s scale %.3e %s" % r)
# a second CPU run with perturbed input bits: the noise floor of the aten flavour itself
g1 = {n: q.grad.clone() for n, q in ref.named_parameters() if q.grad is not None}
ref.zero_grad()
low2, f2 = ref(x + 1e-7 * torch.randn_like(x))
((f2 * wf).sum() + 0.1 * (low2 * wl).sum()).backward()
print("aten flavour alone, input perturbed by 1e-7: its own gradients move by")
rows = sorted(((float((q.grad - g1[n]).abs().max()) / max(float(g1[n].abs().max()), 1e-12), n) for n, q in ref.named_parameters() if q.grad is not None), reverse=True)
for r in rows[:5]: print("%.3e  %s" % r)
