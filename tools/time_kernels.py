#!/usr/bin/env python3
"""Quick per-kernel timing on the GPU box (HIP events on torch's current stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hifihr_amd._lib import get_lib
from hifihr_amd.mano_tables import synthetic_mano_tables
import kernel_cases as kc


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    lib = get_lib()
    t = synthetic_mano_tables(0)
    for B in (32, 256):
        H, aa = 224, 3
        verts, vcol, cam, lc, ld = (x.cuda().contiguous() for x in kc.make_render_inputs(t, B, 7, H))
        h = lib.renderer_create(t.faces, 778, image_size=H, aa=aa)
        ws = torch.empty(lib.render_workspace_bytes(h, B), dtype=torch.uint8, device="cuda")
        rgba = torch.empty(B, 4, H, H, device="cuda"); fid = torch.empty(B, H * aa, H * aa, dtype=torch.int32, device="cuda")
        g = torch.randn_like(rgba)
        gv = torch.empty(B, 778, 3, device="cuda"); gc = torch.empty_like(gv); glc = torch.empty(B, 3, device="cuda"); gld = torch.empty(B, 3, device="cuda")
        tf = timeit(lambda: lib.render_fwd(h, verts, vcol, cam, lc, ld, rgba, fid, ws))
        tb = timeit(lambda: lib.render_bwd(h, verts, cam, lc, ld, fid, g, gv, gc, glc, gld, ws))
        cov = float((fid >= 0).float().mean())
        print(f"render B={B}: fwd {tf:.1f} us ({tf / B:.2f} us/frame)  bwd {tb:.1f} us ({tb / B:.2f} us/frame)  coverage {cov:.3f}")
        hm = lib.mano_create(t)
        pose = 0.5 * torch.randn(B, 48, device="cuda"); beta = 0.5 * torch.randn(B, 10, device="cuda")
        v = torch.empty(B, 778, 3, device="cuda"); j = torch.empty(B, 21, 3, device="cuda"); sv = torch.empty_like(v)
        gp = torch.empty(B, 48, device="cuda"); gb = torch.empty(B, 10, device="cuda")
        gvv = torch.randn_like(v); gj = torch.randn_like(j)
        t1 = timeit(lambda: lib.mano_lbs_fwd(hm, pose, beta, v, j, sv))
        t2 = timeit(lambda: lib.mano_lbs_bwd(hm, pose, beta, sv, gvv, gj, gp, gb))
        jr = torch.empty(B, 21, 3, device="cuda"); vr = torch.empty_like(v); rt = torch.empty(B, 3, device="cuda")
        t3 = timeit(lambda: lib.mano_joints_fwd(hm, v, 9, jr, vr, rt))
        t4 = timeit(lambda: lib.mano_joints_bwd(hm, gj, gvv, rt, 9, vr))
        print(f"mano   B={B}: lbs fwd {t1:.1f} us  lbs bwd {t2:.1f} us  joints fwd {t3:.1f} us  joints bwd {t4:.1f} us")


if __name__ == "__main__":
    main()
