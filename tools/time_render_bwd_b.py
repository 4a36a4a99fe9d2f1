import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import kernel_cases as kc
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
lib = get_lib(); t = synthetic_mano_tables(0); H, aa = 224, 3
h = lib.renderer_create(t.faces, 778, image_size=H, aa=aa)
for B in (1, 4, 16, 32, 64, 128):
    verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
    ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
    rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
    lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
    g = torch.randn_like(rgba); gv = torch.empty_like(verts); gc = torch.empty_like(verts); glc = torch.empty(B, 3, device="cuda"); gld = torch.empty(B, 3, device="cuda")
    fn = lambda: lib.render_bwd(h, verts, cam, lc, ld, fid, g, gv, gc, glc, gld, ws)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"B={B:4d}: {us:8.1f} us  {us / B:6.2f} us/image")
