#!/usr/bin/env python3
"""Hand-derived known answers for the hard rasteriser: tests/golden/raster_known.json.

Derived from the STATEMENT of the rule in SURVEY.md section 8 A12, in exact rational arithmetic (fractions.Fraction) -- not from
oracle/raster_oracle.c and not from the HIP kernel, both of which are checked against this file:

  * sample (xi, yi) of an S x S grid has its centre at NDC (c(xi), c(yi)), c(i) = -1 + (2 (S - 1 - i) + 1) / S  (index 0 is +1:
    +X is left, +Y is up);
  * a face is a candidate for a sample when the sample lies inside the face's bounding box (closed), |signed area| > 1e-8 and
    zmax >= 1e-8;
  * w_i = edge_i(sample) / (area + 1e-8) must be > 0 STRICTLY for i = 0, 1, 2 (a sample on an edge is not covered);
  * perspective-correct barycentrics b_i = w_i prod_{j != i} z_j / max(sum, 1e-8); depth pz = sum b_i z_i must be >= 0;
  * the face with the smallest pz wins; on equal pz the LOWER face index stays.

Every case uses dyadic coordinates, so the float32 arithmetic of the implementations is exact or far from every decision boundary
(the script asserts a margin), except where a case is built ON a boundary (edge, tie).  Vertices are given as camera-space points for
a camera with NDC = (X / Z, Y / Z) (cam = (1, 1, 0, 0)), so the same file drives the HIP renderer (which projects) and the C oracle
(which takes NDC).
"""
import json
import os
from fractions import Fraction as Fr

EPS = Fr(1, 10 ** 8)
ALTERNATIVE_RULE = False       # True: the inside test on the UN-corrected barycentrics (rounds 1-2); only cases 6 and 7 depend on it
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "raster_known.json")


def centre(i, S):
    return Fr(-1) + Fr(2 * (S - 1 - i) + 1, S)


def edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def rasterise(verts_cam, faces, S, margin_check=True):
    """verts_cam: [(X, Y, Z)] Fractions; -> (pix_to_face [S][S], bary, zbuf, min margin)."""
    ndc = [((x / z, y / z, z) if z != 0 else None) for x, y, z in verts_cam]
    p2f = [[-1] * S for _ in range(S)]
    bary = [[None] * S for _ in range(S)]
    zbuf = [[None] * S for _ in range(S)]
    margin = None
    for yi in range(S):
        for xi in range(S):
            px, py = centre(xi, S), centre(yi, S)
            best = None
            for f, (a, b, c) in enumerate(faces):
                (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = ndc[a], ndc[b], ndc[c]
                if px > max(x0, x1, x2) or px < min(x0, x1, x2) or py > max(y0, y1, y2) or py < min(y0, y1, y2):
                    continue
                area = edge(x0, y0, x1, y1, x2, y2)
                if -EPS <= area <= EPS:
                    continue
                if max(z0, z1, z2) < EPS:
                    continue
                den = edge(x2, y2, x0, y0, x1, y1) + EPS
                w = [edge(px, py, x1, y1, x2, y2) / den, edge(px, py, x2, y2, x0, y0) / den, edge(px, py, x0, y0, x1, y1) / den]
                for wi in w:                       # distance of the decision from its boundary (0 = a case built on the boundary)
                    m = abs(wi)
                    if m > 0 and (margin is None or m < margin):
                        margin = m
                t = [w[0] * z1 * z2, z0 * w[1] * z2, z0 * z1 * w[2]]
                d = max(t[0] + t[1] + t[2], EPS)
                bb = [ti / d for ti in t]
                pz = bb[0] * z0 + bb[1] * z1 + bb[2] * z2
                if pz < 0:
                    continue
                # INSIDE TEST ON THE PERSPECTIVE-CORRECTED BARYCENTRICS (round 3).  PyTorch3D's CheckPixelInsideFace /
                # RasterizeMeshesNaiveCpu [recalled]: bary0 = BarycentricCoordsForward, bary = perspective_correct ?
                # BarycentricPerspectiveCorrectionForward(bary0, z) : bary0, pz from bary, `if (pz < 0) continue`, then
                # `inside = bary.x > 0 && bary.y > 0 && bary.z > 0` -- on `bary`, not `bary0`.  With every z > 0 (a hand in front of the
                # camera) the two have the same signs and nothing changes; with a vertex behind the camera they do not (cases 6, 7).
                # Rounds 1-2 tested the un-corrected w (`ALTERNATIVE_RULE` below keeps that reading for the record).
                if ALTERNATIVE_RULE:
                    if not (w[0] > 0 and w[1] > 0 and w[2] > 0):
                        continue
                elif not (bb[0] > 0 and bb[1] > 0 and bb[2] > 0):
                    continue
                if best is None or pz < best[1]:
                    best = (f, pz, bb, w)
            if best is not None:
                p2f[yi][xi], zbuf[yi][xi], bary[yi][xi] = best[0], best[1], (best[2], best[3])
    return p2f, bary, zbuf, margin


def fl(x):
    return float(x)


def case(name, why, verts, faces, image_size, aa, picks=()):
    S = image_size * aa
    p2f, bary, zbuf, margin = rasterise(verts, faces, S)
    assert margin is None or margin > Fr(1, 10 ** 4), (name, float(margin))     # no undesigned near-boundary decision
    samples = []
    for (yi, xi) in picks:
        if p2f[yi][xi] >= 0:
            persp, affine = bary[yi][xi]
            samples.append({"yi": yi, "xi": xi, "face": p2f[yi][xi], "zbuf": fl(zbuf[yi][xi]), "bary": [fl(v) for v in persp],
                            "bary_affine": [fl(v) for v in affine]})
        else:
            samples.append({"yi": yi, "xi": xi, "face": -1})
    return {"name": name, "why": why, "image_size": image_size, "aa": aa, "verts_cam": [[fl(c) for c in v] for v in verts],
            "faces": [list(f) for f in faces], "pix_to_face": p2f, "covered": sum(v >= 0 for row in p2f for v in row), "samples": samples}


def main():
    F = Fr
    cases = []
    # 1. one triangle at three different depths, 8 x 8 pixels x aa 3 (24 x 24 samples): coverage, perspective-correct vs affine
    #    barycentrics, depth.  NDC vertices (-1/2, -1/2), (1/2, -1/2), (0, 3/4) at z = 1, 2, 4 (camera-space X = x z, Y = y z).
    tri = [(F(-1, 2) * 1, F(-1, 2) * 1, F(1)), (F(1, 2) * 2, F(-1, 2) * 2, F(2)), (F(0), F(3, 4) * 4, F(4))]
    c1 = case("single_triangle_aa3", "coverage of one triangle on the 24 x 24 sample grid; perspective-corrected barycentrics differ from the affine ones",
              tri, [(0, 1, 2)], 8, 3, picks=[(12, 12), (15, 10), (9, 12), (17, 16), (0, 0), (5, 12)])
    assert any(abs(s["bary"][0] - s["bary_affine"][0]) > 0.05 for s in c1["samples"] if s["face"] >= 0)
    cases.append(c1)
    # 2. a sample exactly ON an edge is not covered (strict w > 0): 8 x 8 samples (aa 1) have centres at odd multiples of 1/8, exact in
    #    float32.  Vertical edge x = 1/8 through the column of samples xi = 3 (c(3) = -1 + 9/8 = 1/8); the triangle extends to -x
    #    (x = -7/8 .. 1/8): columns with centre < 1/8 inside are covered, the column ON the edge is not.
    e = [(F(1, 8), F(-7, 8), F(1)), (F(1, 8), F(7, 8), F(1)), (F(-7, 8), F(0), F(1))]
    c2 = case("sample_on_edge", "samples whose centre lies exactly on an edge (w = 0) are NOT covered: strict > 0", e, [(0, 1, 2)], 8, 1,
              picks=[(3, 3), (4, 3), (3, 4), (4, 4)])
    assert all(row[3] == -1 for row in c2["pix_to_face"]) and c2["pix_to_face"][3][4] == 0 and c2["pix_to_face"][4][4] == 0
    cases.append(c2)
    # 3. equal depth: two coincident faces (same three vertices listed twice) -> the LOWER face index wins everywhere it is covered;
    #    and a nearer face listed later does win.
    q = [(F(-3, 4), F(-3, 4), F(1)), (F(3, 4), F(-3, 4), F(1)), (F(0), F(3, 4), F(1)),
         (F(-3, 4) * F(1, 2), F(-3, 4) * F(1, 2), F(1, 2)), (F(3, 4) * F(1, 2), F(-3, 4) * F(1, 2), F(1, 2)), (F(0), F(3, 4) * F(1, 2), F(1, 2))]
    c3 = case("equal_depth_lower_index_wins", "faces 0 and 1 are the same triangle at z = 1: the earlier index stays on ties", q[:3],
              [(0, 1, 2), (0, 1, 2)], 8, 3, picks=[(12, 12), (16, 8)])
    assert set(v for row in c3["pix_to_face"] for v in row) == {-1, 0}
    cases.append(c3)
    c3b = case("nearer_face_listed_later_wins", "face 1 covers the same NDC triangle at z = 1/2 (nearer): it replaces face 0 (strictly smaller depth)", q,
               [(0, 1, 2), (3, 4, 5)], 8, 3, picks=[(12, 12)])
    assert set(v for row in c3b["pix_to_face"] for v in row) == {-1, 1}
    cases.append(c3b)
    # 4. zero-area faces are skipped: face 0 = three collinear points in front, face 1 = a proper triangle behind them.
    z = [(F(-1, 2), F(-1, 2), F(1)), (F(0), F(0), F(1)), (F(1, 2), F(1, 2), F(1)),
         (F(-3, 4) * 2, F(-3, 4) * 2, F(2)), (F(3, 4) * 2, F(-3, 4) * 2, F(2)), (F(0), F(3, 4) * 2, F(2))]
    c4 = case("zero_area_face_skipped", "face 0 is degenerate (collinear, area 0 <= 1e-8) and nearer; only face 1 may appear", z,
              [(0, 1, 2), (3, 4, 5)], 8, 3, picks=[(12, 12)])
    assert set(v for row in c4["pix_to_face"] for v in row) == {-1, 1}
    cases.append(c4)
    # 5. one vertex behind the camera: zmax >= 1e-8 keeps the face, every sample then fails pz >= 0 (pz = z0 z1 z2 / denom < 0).
    b1 = [(F(-3, 4) * -1, F(-3, 4) * -1, F(-1)), (F(3, 4), F(-3, 4), F(1)), (F(0), F(3, 4), F(1))]
    c5 = case("one_vertex_behind_camera", "z = (-1, 1, 1): the face is a candidate (zmax >= 1e-8) but every covered sample has pz < 0: nothing is drawn",
              b1, [(0, 1, 2)], 8, 3, picks=[(12, 12)])
    assert c5["covered"] == 0
    cases.append(c5)
    # 6. two vertices behind the camera: this is where "zmax >= 1e-8" (built) and "zmin >= 1e-8" (round 1) differ.
    #    z = (-1, -1, 1): t = (-w0, -w1, w2), denom = max(w2 - w0 - w1, 1e-8), pz = 1 / denom > 0: the whole interior is a hit, with depth
    #    1 / (2 w2 - 1) where w2 > 1/2 and 1e8 elsewhere.
    b2 = [(F(-3, 4) * -1, F(-3, 4) * -1, F(-1)), (F(3, 4) * -1, F(-3, 4) * -1, F(-1)), (F(0), F(3, 4), F(1))]
    #    The inside test decides: on the perspective-CORRECTED barycentrics (PyTorch3D, built since round 3) two of the three are
    #    negative over the triangle's interior (b0 = -w0 / denom, b1 = -w1 / denom) -> nothing is drawn; on the un-corrected ones
    #    (rounds 1-2, ALTERNATIVE_RULE) the interior was a hit.
    c6 = case("two_vertices_behind_camera", "z = (-1, -1, 1): kept by the zmax rule, but over the triangle's interior two corrected barycentrics are "
              "negative: nothing is drawn (the un-corrected inside test of rounds 1-2 drew the interior at depth 1 / max(2 w2 - 1, 1e-8))",
              b2, [(0, 1, 2)], 8, 3, picks=[(6, 12), (15, 12)])
    assert c6["covered"] == (0 if not ALTERNATIVE_RULE else c6["covered"]) and (ALTERNATIVE_RULE or c6["covered"] == 0)
    cases.append(c6)
    # 7. the other side of the same rule: a thin triangle whose only front vertex v2 lies strictly INSIDE its own bounding box.  In the
    #    wedge opposite the triangle at v2 (w0 < 0, w1 < 0, w2 > 1) all three corrected barycentrics are positive (t = (-w0, -w1, w2) with
    #    z = (-1, -1, 1)) and pz = 1 / (w2 - w0 - w1) > 0: PyTorch3D draws the face THERE, outside its 2-D triangle (the artefact its
    #    z-clipping option exists for), and nowhere inside it.  NDC triangle (-7/8, -3/4), (7/8, 3/4), apex (0, 1/4) between them.
    b3 = [(F(-7, 8) * -1, F(-3, 4) * -1, F(-1)), (F(7, 8) * -1, F(3, 4) * -1, F(-1)), (F(0), F(1, 4), F(1))]
    c7 = case("straddling_face_opposite_wedge", "z = (-1, -1, 1), front vertex inside the bounding box: covered samples lie in the wedge beyond "
              "that vertex (all corrected barycentrics > 0), none inside the 2-D triangle", b3, [(0, 1, 2)], 8, 3, picks=[(4, 11), (12, 12)])
    if not ALTERNATIVE_RULE:
        assert c7["covered"] > 0
    cases.append(c7)
    with open(OUT, "w") as fh:
        json.dump({"rule": "SURVEY.md section 8 A12", "cases": cases}, fh)
    for c in cases:
        print(c["name"], "covered", c["covered"], "of", (c["image_size"] * c["aa"]) ** 2)


if __name__ == "__main__":
    main()
