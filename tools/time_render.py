"""render_fwd / render_bwd back-to-back on a synthetic BASELINE batch (B = 32, 224, aa 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import kernel_cases as kc
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
lib = get_lib() if not os.environ.get("RENDER_LIB") else __import__("hifihr_amd._lib", fromlist=["HifihrLib"]).HifihrLib(os.environ["RENDER_LIB"])
t = synthetic_mano_tables(0)
B, H, aa = 32, 224, 3
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, 778, image_size=H, aa=aa)
ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("render_fwd (vertex + bin + tiles) us:", timeit(lambda: lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)))
print("coverage:", float((fid >= 0).float().mean()))
g = torch.randn_like(rgba); gv = torch.empty_like(verts); gc = torch.empty_like(verts); glc = torch.empty(B, 3, device="cuda"); gld = torch.empty(B, 3, device="cuda")
print("render_bwd us:", timeit(lambda: lib.render_bwd(h, verts, cam, lc, ld, fid, g, gv, gc, glc, gld, ws)))
# ablations by input: everything behind the camera (all tiles empty), and a tiny far-away hand (few tiles, few samples)
vb = verts.clone(); vb[..., 2] = -vb[..., 2]
print("all rejected us:", timeit(lambda: lib.render_fwd(h, vb.contiguous(), vcol, cam, lc, ld, rgba, fid, ws)), "coverage", float((fid >= 0).float().mean()))
vf = verts.clone(); vf[..., 2] += 3.0
print("far hand us:", timeit(lambda: lib.render_fwd(h, vf.contiguous(), vcol, cam, lc, ld, rgba, fid, ws)), "coverage", float((fid >= 0).float().mean()))
vn = verts.clone(); vn[..., 2] *= 0.5; vn[..., :2] *= 0.5
print("same projection, half depth us:", timeit(lambda: lib.render_fwd(h, vn.contiguous(), vcol, cam, lc, ld, rgba, fid, ws)), "coverage", float((fid >= 0).float().mean()))
vc = verts.clone(); vc[..., 2] *= 0.6
print("closer (bigger) hand us:", timeit(lambda: lib.render_fwd(h, vc.contiguous(), vcol, cam, lc, ld, rgba, fid, ws)), "coverage", float((fid >= 0).float().mean()))
