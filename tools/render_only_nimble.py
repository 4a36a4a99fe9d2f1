#!/usr/bin/env python3
"""B = 48 render forward + backward of the NIMBLE-SHAPED mesh (5 990 vertices / 11 976 faces, synthetic tables) with TexturesUV (64 x 64
texture image per sample), 5 launches -- for rocprofv3 kernel traces and --pmc passes (BASELINE configs[2] in its real shape)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
from hifihr_amd.nimble_tables import add_synthetic_uv, synthetic_nimble_tables
import kernel_cases as kc
lib = get_lib(); mt = synthetic_mano_tables(0)
nt = add_synthetic_uv(synthetic_nimble_tables(0))
B, H, aa = int(os.environ.get("B", "48")), 224, 3
V = int(nt.v_template.shape[0]); faces = np.asarray(nt.faces)
verts_m, _, cam, lc, ld = kc.make_render_inputs(mt, B, 7, H)
mv = torch.as_tensor(np.asarray(nt.v_template), dtype=torch.float32)[None]
mv = mv - mv.mean(1, keepdim=True)
hand = verts_m - verts_m.mean(1, keepdim=True)
mv = mv / mv.abs().max() * hand.abs().max() + verts_m.mean(1, keepdim=True)           # the skin at the MANO hands' placements and size
verts = mv.contiguous().cuda(); cam, lc, ld = cam.cuda(), lc.cuda(), ld.cuda()
TH, TW = nt.tex_hw
maps = torch.rand(B, TH, TW, 3, device="cuda")
h = lib.renderer_create(faces, V, image_size=H, aa=aa)
lib.renderer_set_uv(h, nt.faces_uvs, nt.verts_uvs)
ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
g = torch.randn_like(rgba); gv = torch.empty(B, V, 3, device="cuda"); gm = torch.zeros(B, TH, TW, 3, device="cuda")
glc = torch.empty(B, 3, device="cuda"); gld = torch.empty(B, 3, device="cuda")
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
f = lambda: lib.render_fwd_uv(h, verts, maps, cam, lc, ld, rgba, fid, None, ws)
b = lambda: lib.render_bwd_uv(h, verts, maps, cam, lc, ld, fid, g, None, None, gv, gm, glc, gld, ws)
print(f"NIMBLE-shaped mesh, TexturesUV, B = {B}: render_fwd_uv {timeit(f):.1f} us, render_bwd_uv {timeit(b):.1f} us, coverage {float((fid >= 0).float().mean()):.3f}")
