#!/usr/bin/env python3
"""Where does render_fwd time go?  (GPU box)  binning-only (mesh behind the camera) vs full."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
import kernel_cases as kc
from time_kernels import timeit

lib = get_lib(); t = synthetic_mano_tables(0)
B, H, aa = 32, 224, 3
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, 778, image_size=H, aa=aa)
ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
def run(v): return timeit(lambda: lib.render_fwd(h, v, vcol, cam, lc, ld, rgba, fid, ws))
print("full            ", run(verts), "us  coverage", float((fid >= 0).float().mean()))
vb = verts.clone(); vb[..., 2] -= 5.0
print("behind camera   ", run(vb), "us (binning + rejection only)")
vf = verts.clone(); vf[..., 0] += 10.0
print("off-screen      ", run(vf), "us (binning, bbox never overlaps)")
vs = verts.clone(); vs[..., 2] *= 3.0; vs[..., 0] *= 3.0; vs[..., 1] *= 3.0
print("same pose, 3x further (small on screen)", run(vs.contiguous()), "us coverage", float((fid >= 0).float().mean()))
vn = verts.clone(); c = vn.mean(1, keepdim=True); vn = (vn - c) * 2.0 + c
print("2x larger hand  ", run(vn.contiguous()), "us coverage", float((fid >= 0).float().mean()))
