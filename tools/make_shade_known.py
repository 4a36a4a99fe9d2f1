#!/usr/bin/env python3
"""Hand-derived known answers for the Phong shader behind the hard rasteriser: tests/golden/shade_known.json.

Derived from the STATEMENT of the rule in SURVEY.md section 8 A13 (HardPhongShader with DirectionalLights / PointLights, Materials, TexturesVertex,
hard_rgb_blend on a white background [recalled: PyTorch3D is not in the reference tree]) in float64, from the exact rational barycentrics of
tools/make_raster_known.rasterise -- NOT from oracle/render_oracle.py and not from the HIP kernel, both of which are checked against this file
(tests/test_shade_known.py):

  * vertex normal n_v = normalise( sum over incident faces of cross(v2 - v1, v0 - v1) )   (area-weighted, F.normalize eps 1e-6);
  * per covered sample with perspective-corrected barycentrics b: P = sum b_i V_i (camera space), N = sum b_i n_i, T = sum b_i colour_i;
  * l = normalise(direction)  (directional light; point light: normalise(location - P));  n = normalise(N);  c = n . l;
  * diffuse = light_colour * max(c, 0);  v = normalise(-P) (camera centre at the origin);  r = -l + 2 c n;
    specular = spec * max(v . r, 0)^shininess * [c > 0];
  * colour = (ambient + mat_diffuse * diffuse) * T + specular;   a miss is the background (1, 1, 1) with alpha 0, a hit has alpha 1.

aa = 1: a pixel is ONE sample, so the rendered pixel is the sample's colour.  Every case keeps its samples far from edges (the rasteriser's
decisions are covered by raster_known.json)."""
import json
import math
import os
import sys
from fractions import Fraction as Fr

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_raster_known as mrk  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "shade_known.json")
AMB, MD, SP, SHIN = (0.5, 0.5, 0.5), (0.8, 0.8, 0.8), (0.04, 0.04, 0.04), 30.0


def sub(a, b): return [a[i] - b[i] for i in range(3)]
def cross(a, b): return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
def dot(a, b): return sum(a[i] * b[i] for i in range(3))
def norm(a):
    n = max(math.sqrt(dot(a, a)), 1e-6)
    return [x / n for x in a]


def vertex_normals(verts, faces):
    acc = [[0.0, 0.0, 0.0] for _ in verts]
    for (i0, i1, i2) in faces:
        fn = cross(sub(verts[i2], verts[i1]), sub(verts[i0], verts[i1]))
        for i in (i0, i1, i2):
            acc[i] = [acc[i][k] + fn[k] for k in range(3)]
    return [norm(a) for a in acc]


def shade(P, N, T, light_colour, light_vec, point_light):
    l = norm(sub(light_vec, P)) if point_light else norm(light_vec)
    n = norm(N)
    c = dot(n, l)
    diffuse = [lc * max(c, 0.0) for lc in light_colour]
    v = norm([-p for p in P])
    r = [-l[k] + 2.0 * c * n[k] for k in range(3)]
    alpha = max(dot(v, r), 0.0) * (1.0 if c > 0 else 0.0)
    spec = [s * alpha ** SHIN for s in SP]
    return [(AMB[k] + MD[k] * diffuse[k]) * T[k] + spec[k] for k in range(3)]


def case(name, why, verts, faces, colours, light_colour, light_vec, point_light, picks, H=8):
    vf = [[Fr(c).limit_denominator(1 << 20) for c in v] for v in verts]
    p2f, bary, zbuf, margin = mrk.rasterise(vf, faces, H)
    assert margin is None or margin > Fr(1, 100), (name, float(margin))             # every decision far from an edge
    V = [[float(c) for c in v] for v in vf]
    nrm = vertex_normals(V, faces)
    samples = []
    for (yi, xi) in picks:
        f = p2f[yi][xi]
        if f < 0:
            samples.append({"yi": yi, "xi": xi, "face": -1, "rgba": [1.0, 1.0, 1.0, 0.0]})
            continue
        b = [float(x) for x in bary[yi][xi][0]]
        ids = faces[f]
        P = [sum(b[i] * V[ids[i]][k] for i in range(3)) for k in range(3)]
        N = [sum(b[i] * nrm[ids[i]][k] for i in range(3)) for k in range(3)]
        T = [sum(b[i] * colours[ids[i]][k] for i in range(3)) for k in range(3)]
        samples.append({"yi": yi, "xi": xi, "face": f, "rgba": shade(P, N, T, light_colour, light_vec, point_light) + [1.0]})
    return {"name": name, "why": why, "image_size": H, "verts_cam": V, "faces": [list(f) for f in faces], "colours": colours,
            "light_colour": list(light_colour), "light": list(light_vec), "point_light": bool(point_light), "samples": samples,
            "ambient": list(AMB), "mat_diffuse": list(MD), "specular": list(SP), "shininess": SHIN}


def main():
    cases = []
    # a triangle in the plane z = 1 whose normal cross(v2 - v1, v0 - v1) points at the camera (-z); NDC = camera-space x, y at z = 1
    tri = [(-0.75, -0.75, 1.0), (0.0, 0.75, 1.0), (0.75, -0.75, 1.0)]
    n = vertex_normals([list(v) for v in tri], [(0, 1, 2)])[0]
    assert n[2] < -0.99, n
    white = [[1.0, 1.0, 1.0]] * 3
    picks = [(4, 4), (5, 3), (3, 4), (6, 5), (0, 0)]
    cases.append(case("facing_the_light", "normal and light direction coincide: full diffuse term, specular = (v . l)^30 falling off from the image centre",
                      tri, [(0, 1, 2)], white, (0.6, 0.7, 0.8), (0.0, 0.0, -1.0), False, picks))
    cases.append(case("oblique_light", "light 60 degrees off the normal: diffuse scales with the cosine, the reflection vector leaves the view direction",
                      tri, [(0, 1, 2)], white, (0.9, 0.5, 0.3), (math.sin(math.pi / 3), 0.0, -math.cos(math.pi / 3)), False, picks))
    cases.append(case("light_behind_the_surface", "n . l < 0: no diffuse term and the specular term is masked: ambient * texel only",
                      tri, [(0, 1, 2)], white, (0.9, 0.9, 0.9), (0.2, 0.1, 1.0), False, picks))
    cols = [[0.9, 0.1, 0.2], [0.2, 0.8, 0.3], [0.1, 0.3, 0.95]]
    tilt = [(-0.75 * 1.0, -0.75 * 1.0, 1.0), (0.0, 0.75 * 2.0, 2.0), (0.75 * 1.5, -0.75 * 1.5, 1.5)]      # same NDC triangle, three depths
    cases.append(case("vertex_colours_at_three_depths", "texel, position and normal interpolated with PERSPECTIVE-CORRECTED barycentrics on a tilted face",
                      tilt, [(0, 1, 2)], cols, (0.7, 0.7, 0.7), (0.3, -0.2, -1.0), False, picks))
    # a tent: two faces sharing the ridge (vertices 1, 3); the ridge is nearer -> vertex normals differ from the face normals and vary across a face
    tent_v = [(-0.8, -0.7, 1.2), (0.0, -0.7 * (1.0 / 1.2), 1.0), (0.8, -0.7, 1.2), (0.0, 0.8 * (1.0 / 1.2), 1.0)]
    tent_v = [(x, y, z) for (x, y, z) in tent_v]
    tent_f = [(0, 3, 1), (1, 3, 2)]
    nn = vertex_normals([list(v) for v in tent_v], tent_f)
    assert all(q[2] < 0 for q in nn), nn
    cases.append(case("tent_interpolated_normals", "two faces meet at a ridge: area-weighted vertex normals, interpolated and re-normalised per sample",
                      tent_v, tent_f, [[0.8, 0.6, 0.5]] * 4, (0.8, 0.8, 0.7), (-0.4, 0.3, -1.0), False, [(4, 3), (4, 5), (5, 2), (3, 6), (5, 4)]))
    cases.append(case("point_light", "PointLights: the light vector is location - P, normalised per sample",
                      tri, [(0, 1, 2)], cols, (0.3, 0.3, 0.3), (0.0, 1.0, 0.0), True, picks))
    with open(OUT, "w") as fh:
        json.dump({"rule": "SURVEY.md section 8 A13", "cases": cases}, fh)
    for c in cases:
        print(c["name"], [(s["yi"], s["xi"], s["face"], [round(v, 4) for v in s["rgba"][:3]]) for s in c["samples"][:3]])


if __name__ == "__main__":
    main()
