"""render_fwd alone (5 launches) for PMC passes: tools/pmc_probe.sh <tag> "<counters>" render_only2.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import kernel_cases as kc
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
lib = get_lib()
t = synthetic_mano_tables(0)
B, H, aa = 32, 224, 3
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, 778, image_size=H, aa=aa)
ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
for _ in range(5):
    lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
torch.cuda.synchronize()
