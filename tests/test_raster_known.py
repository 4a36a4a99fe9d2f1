"""The rasteriser's rule set against hand-derived known answers (tests/golden/raster_known.json, derived in exact rational arithmetic
from the statement of SURVEY.md section 8 A12 by tools/make_raster_known.py -- independently of the code under test):
coverage of a triangle on the aa-3 sample grid, perspective-correct vs affine barycentrics, a sample exactly on an edge (strict >),
two faces at equal depth (lower index wins), a nearer face listed later, a zero-area face, faces with one / two vertices behind the
camera (the zmax >= 1e-8 rule).  CPU: the C oracle; GPU (-m gpu): the HIP renderer's face ids through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "raster_known.json")))["cases"]


def test_oracle_rasteriser_matches_known_answers(golden_dir):
    from oracle import render_oracle as ro
    for c in _cases(golden_dir):
        v = torch.tensor(c["verts_cam"], dtype=torch.float32).unsqueeze(0)
        ndc = ro.project_ndc(v, torch.tensor([[1.0, 1.0, 0.0, 0.0]]))
        S = c["image_size"] * c["aa"]
        p2f, zbuf, bary = ro.rasterize(ndc, torch.tensor(c["faces"]), S)
        np.testing.assert_array_equal(p2f[0], np.asarray(c["pix_to_face"], dtype=np.int32), err_msg=c["name"])
        for s in c["samples"]:
            yi, xi = s["yi"], s["xi"]
            assert p2f[0, yi, xi] == s["face"], (c["name"], s)
            if s["face"] >= 0:
                np.testing.assert_allclose(zbuf[0, yi, xi], s["zbuf"], rtol=2e-6, err_msg=c["name"])
                np.testing.assert_allclose(bary[0, yi, xi], s["bary"], rtol=2e-6, atol=2e-6 * max(1.0, max(abs(b) for b in s["bary"])), err_msg=c["name"])
        if c["name"] == "single_triangle_aa3":        # the implementation applies the perspective correction (it is not the affine result)
            s = c["samples"][0]
            assert abs(bary[0, s["yi"], s["xi"], 0] - s["bary_affine"][0]) > 0.05


@pytest.mark.gpu
def test_hip_rasteriser_matches_known_answers(golden_dir):
    from hifihr_amd._lib import get_lib
    lib = get_lib()
    for c in _cases(golden_dir):
        verts = torch.tensor(c["verts_cam"], dtype=torch.float32)
        V, H, aa = verts.shape[0], c["image_size"], c["aa"]
        faces = np.asarray(c["faces"], dtype=np.int32)
        h = lib.renderer_create(faces, V, image_size=H, aa=aa, ambient=(0.5,) * 3, mat_diffuse=(0.8,) * 3, specular=(0.04,) * 3,
                                shininess=30.0, background=(1.0,) * 3)
        try:
            S = H * aa
            v = verts.unsqueeze(0).cuda().contiguous()
            col = torch.ones(V, 3, device="cuda")
            cam = torch.tensor([[1.0, 1.0, 0.0, 0.0]], device="cuda")
            lc = torch.zeros(1, 3, device="cuda"); ld = torch.tensor([[0.0, 0.0, -1.0]], device="cuda")
            rgba = torch.empty(1, 4, H, H, device="cuda"); fid = torch.full((1, S, S), -7, dtype=torch.int32, device="cuda")
            ws = torch.empty(lib.render_workspace_bytes(h, 1), dtype=torch.uint8, device="cuda")
            lib.render_fwd(h, v, col, cam, lc, ld, rgba, fid, ws)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(fid[0].cpu().numpy(), np.asarray(c["pix_to_face"], dtype=np.int32), err_msg=c["name"])
            # alpha of the resolved image = fraction of covered samples per pixel
            want = (np.asarray(c["pix_to_face"]) >= 0).reshape(H, aa, H, aa).mean(axis=(1, 3))
            np.testing.assert_allclose(rgba[0, 3].cpu().numpy(), want, atol=1e-6, err_msg=c["name"])
        finally:
            lib.renderer_destroy(h)
