"""Per-tile phase durations of render_bwd_kernel (diagnostic build, tools/build_render_probe.sh)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import kernel_cases as kc
from hifihr_amd._lib import HifihrLib
from hifihr_amd.mano_tables import synthetic_mano_tables
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_render_stamp.so"))
t = synthetic_mano_tables(0); B, H, aa, V, F = 32, 224, 3, 778, 1538
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, V, image_size=H, aa=aa)
ws = torch.zeros(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
g = torch.randn_like(rgba); gv = torch.empty_like(verts); gc = torch.empty_like(verts); glc = torch.empty(B, 3, device="cuda"); gld = torch.empty(B, 3, device="cuda")
off = B * V * (4 * 16 + 12 * 4) + B * 196 * 4
ws[off:off + B * 196 * 16].zero_()
lib.render_bwd(h, verts, cam, lc, ld, fid, g, gv, gc, glc, gld, ws)
torch.cuda.synchronize()
d = ws[off:off + B * 196 * 16].view(torch.int32).cpu().numpy().reshape(B, 14, 14, 4)
busy = d[..., 3] == 1
us = d[..., :3] * 64 / 2400.0
tot = us.sum(-1)
print("busy tiles:", int(busy.sum()))
print("mean us: zero %.1f  main %.1f  flush+lights %.1f  total %.1f" % (us[busy][:, 0].mean(), us[busy][:, 1].mean(), us[busy][:, 2].mean(), tot[busy].mean()))
print("max  us: zero %.1f  main %.1f  flush+lights %.1f  total %.1f" % (us[busy][:, 0].max(), us[busy][:, 1].max(), us[busy][:, 2].max(), tot[busy].max()))
print("p90 total %.1f" % np.percentile(tot[busy], 90))
