"""The Phong shader's rule set against hand-derived known answers (tests/golden/shade_known.json: derived by tools/make_shade_known.py in float64
from the STATEMENT of SURVEY.md section 8 A13 and the exact rational barycentrics of tools/make_raster_known.py -- independently of the code under
test): area-weighted vertex normals, perspective-corrected interpolation of position / normal / vertex colour, the diffuse cosine, the
reflection vector and the masked specular power, ambient + material constants, a light behind the surface, a point light, the white
background with alpha 0.  aa = 1, so a rendered pixel IS one sample.  CPU: oracle/render_oracle.render; GPU (-m gpu): the HIP renderer."""
import json
import os

import numpy as np
import pytest
import torch


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "shade_known.json")))["cases"]


def _check(c, rgba, tol):
    worst = 0.0
    for s in c["samples"]:
        got = rgba[:, s["yi"], s["xi"]]
        want = np.asarray(s["rgba"], dtype=np.float64)
        err = float(np.abs(got - want).max())
        assert err <= tol, (c["name"], s["yi"], s["xi"], got.tolist(), s["rgba"])
        worst = max(worst, err)
    return worst


def test_oracle_shader_matches_known_answers(golden_dir):
    from oracle import render_oracle as ro
    for c in _cases(golden_dir):
        v = torch.tensor(c["verts_cam"], dtype=torch.float32).unsqueeze(0)
        col = torch.tensor(c["colours"], dtype=torch.float32).unsqueeze(0)
        consts = ro.ShadeConsts(ambient=tuple(c["ambient"]), mat_diffuse=tuple(c["mat_diffuse"]), specular=tuple(c["specular"]),
                                shininess=c["shininess"], background=(1.0, 1.0, 1.0))
        rgba, p2f = ro.render(v, col, torch.tensor([[1.0, 1.0, 0.0, 0.0]]), torch.tensor([c["light_colour"]], dtype=torch.float32),
                              torch.tensor([c["light"]], dtype=torch.float32), torch.tensor(c["faces"]), image_size=c["image_size"], aa=1,
                              consts=consts, point_lights=c["point_light"])
        for s in c["samples"]:
            assert int(p2f[0, s["yi"], s["xi"]]) == s["face"], (c["name"], s)
        worst = _check(c, rgba[0].double().numpy(), 3e-6)
        print(f"{c['name']}: worst |pixel - known answer| = {worst:.2e} (bound 3e-6)")


@pytest.mark.gpu
def test_hip_shader_matches_known_answers(golden_dir):
    """The HIP renderer against the same file through the C ABI (observed 1.5e-7; the bound of 5e-6 is 20 x under north_star's 1e-4 pixel
    tolerance and leaves room for the shader's approximate reciprocals / square roots)."""
    from hifihr_amd._lib import get_lib
    lib = get_lib()
    for c in _cases(golden_dir):
        verts = torch.tensor(c["verts_cam"], dtype=torch.float32)
        V, H = verts.shape[0], c["image_size"]
        faces = np.asarray(c["faces"], dtype=np.int32)
        h = lib.renderer_create(faces, V, image_size=H, aa=1, ambient=tuple(c["ambient"]), mat_diffuse=tuple(c["mat_diffuse"]),
                                specular=tuple(c["specular"]), shininess=c["shininess"], background=(1.0, 1.0, 1.0))
        try:
            if c["point_light"]:
                lib.renderer_set_light_mode(h, True)
            v = verts.unsqueeze(0).cuda().contiguous()
            col = torch.tensor(c["colours"], dtype=torch.float32).cuda().contiguous()
            cam = torch.tensor([[1.0, 1.0, 0.0, 0.0]], device="cuda")
            lc = torch.tensor([c["light_colour"]], dtype=torch.float32, device="cuda"); ld = torch.tensor([c["light"]], dtype=torch.float32, device="cuda")
            rgba = torch.empty(1, 4, H, H, device="cuda"); fid = torch.full((1, H, H), -7, dtype=torch.int32, device="cuda")
            ws = torch.empty(lib.render_workspace_bytes(h, 1), dtype=torch.uint8, device="cuda")
            lib.render_fwd(h, v, col, cam, lc, ld, rgba, fid, ws)
            torch.cuda.synchronize()
            for s in c["samples"]:
                assert int(fid[0, s["yi"], s["xi"]]) == s["face"], (c["name"], s)
            worst = _check(c, rgba[0].double().cpu().numpy(), 5e-6)
            print(f"{c['name']}: worst |pixel - known answer| = {worst:.2e} (bound 5e-6)")
        finally:
            lib.renderer_destroy(h)
