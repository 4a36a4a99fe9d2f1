"""Per-tile duration of render_fwd_kernel (diagnostic build, tools/build_render_probe.sh): which tiles are the critical path?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import kernel_cases as kc
from hifihr_amd._lib import HifihrLib
from hifihr_amd.mano_tables import synthetic_mano_tables
lib = HifihrLib(os.path.join(R, "tools", "_probe", "libhifihr_render_stamp.so"))
t = synthetic_mano_tables(0); B, H, aa, V = 32, 224, 3, 778
verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
h = lib.renderer_create(t.faces, V, image_size=H, aa=aa)
ws = torch.zeros(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
for _ in range(2):
    lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws)
torch.cuda.synchronize()
off = B * V * (4 * 16 + 12 * 4)
cnt = ws[off:off + B * 196 * 4].view(torch.int32).cpu().numpy().reshape(B, 14, 14)
faces, cyc = cnt >> 20, (cnt & 0xfffff) * 64
busy = faces > 0
print("busy tiles:", int(busy.sum()), "of", busy.size, "| faces per busy tile: mean %.0f max %d" % (faces[busy].mean(), faces.max()))
us = cyc / 2400.0          # ~2.4 GHz shader clock
print("busy tile duration us: mean %.1f  p50 %.1f  p90 %.1f  max %.1f" % (us[busy].mean(), np.percentile(us[busy], 50), np.percentile(us[busy], 90), us[busy].max()))
order = np.argsort(-us.reshape(-1))[:8]
for o in order:
    b, ty, tx = np.unravel_index(o, us.shape)
    print(f"  image {b:2d} tile ({ty:2d},{tx:2d}): {us[b, ty, tx]:7.1f} us, {faces[b, ty, tx]:5d} faces")
F = 1538
lst = ws[off + B * 196 * 4: off + B * 196 * 4 + B * 196 * F * 4].view(torch.int32).cpu().numpy().reshape(B, 14, 14, F)
stage_us, raster_us = lst[..., 0] * 64 / 2400.0, lst[..., 1] * 64 / 2400.0
for o in order:
    b, ty, tx = np.unravel_index(o, us.shape)
    print(f"  image {b:2d} tile ({ty:2d},{tx:2d}): total {us[b, ty, tx]:7.1f} us = stage {stage_us[b, ty, tx]:6.1f} + raster {raster_us[b, ty, tx]:6.1f} + rest (init, shade) {us[b, ty, tx] - stage_us[b, ty, tx] - raster_us[b, ty, tx]:6.1f}")
for o in order[:5]:
    b, ty, tx = np.unravel_index(o, us.shape)
    d = lst[b, ty, tx]
    print(f"  image {b:2d} tile ({ty:2d},{tx:2d}): rect {d[2] * 64 / 2400:6.1f} us, scan {d[3] * 64 / 2400:6.1f} us, candidate loop {d[4] * 64 / 2400:6.1f} us, candidates {d[5]}")
print("busy tiles mean: stage %.1f raster %.1f total %.1f" % (stage_us[busy].mean(), raster_us[busy].mean(), us[busy].mean()))
cov = (fid >= 0).float().mean(dim=(1, 2)).cpu().numpy()
print("coverage per image: min %.3f max %.3f" % (cov.min(), cov.max()))
