#!/usr/bin/env python3
"""B=32 render forward+backward, 5 launches (for rocprofv3 --pmc passes)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
import kernel_cases as kc
lib = get_lib(); t = synthetic_mano_tables(0)
B, H, aa = 32, 224, 3
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, 778, image_size=H, aa=aa)
ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
g = torch.randn_like(rgba); gv = torch.empty(B, 778, 3, device="cuda"); gc = torch.empty_like(gv); glc = torch.empty(B, 3, device="cuda"); gld = torch.empty(B, 3, device="cuda")
for _ in range(5):
    lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
    lib.render_bwd(h, verts, cam, lc, ld, fid, g, gv, gc, glc, gld, ws)
torch.cuda.synchronize()
