/* ORACLE (test infrastructure, not product code): naive hard rasteriser on the CPU.
 *
 * Restates the per-sample loop of PyTorch3D's RasterizeMeshesNaiveCpu / CheckPixelInsideFace as the
 * reference configures it (reference models_res_nimble.py:74-78,89-91: image_size = 224*3, blur_radius = 0,
 * faces_per_pixel = 1; perspective-correct barycentrics because the cameras are PerspectiveCameras
 * (:184-186); no back-face culling, no z-clipping, clip_barycentric_coords = False).
 * PyTorch3D is a pip dependency of the reference (README.md:70-71, unpinned git HEAD, a 0.7.x snapshot);
 * its source is not under /root/reference, so this file restates its published algorithm from memory:
 * PARITY UNPINNED at the PyTorch3D boundary.  The rule set itself (sample centres, bounding-box test,
 * strict barycentric > 0 on the perspective-corrected weights, perspective correction with the 1e-8 clamp, nearest depth with the lower face index on ties, zero-area and
 * behind-the-plane faces) is pinned by hand-derived known answers: tests/golden/raster_known.json, derived in exact rational
 * arithmetic from the statement of SURVEY.md A12 by tools/make_raster_known.py (not from this file).  It IS the bit-exact target for the
 * HIP rasteriser's face indices: both are built with -ffp-contract=off and evaluate the same fp32
 * expressions in the same order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define K_EPS 1e-8f

static float pix_to_ndc(int i, int S) {
  const float range = 2.0f;
  const float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S;
}

static float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

/* verts_ndc [B][V][3] = (x_ndc, y_ndc, z_view); faces [F][3] (shared topology for the batch);
 * outputs per sample: pix_to_face [B][S][S] (local face index, -1 = miss), zbuf [B][S][S] (-1 on miss),
 * bary [B][S][S][3] (-1 on miss).  Sample (yi, xi): row 0 is the top of the image (+Y up => NDC y of row
 * yi uses index S-1-yi), column 0 the left (+X left => index S-1-xi). */
void raster_oracle(const float* verts_ndc, const int32_t* faces, int B, int V, int F, int S,
                   int32_t* pix_to_face, float* zbuf, float* bary) {
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
  for (int b = 0; b < B; ++b) {
    for (int yi = 0; yi < S; ++yi) {
      const float* vb = verts_ndc + (size_t)b * V * 3;
      const float yf = pix_to_ndc(S - 1 - yi, S);
      for (int xi = 0; xi < S; ++xi) {
        const float xf = pix_to_ndc(S - 1 - xi, S);
        int best_f = -1;
        float best_z = 0.f, bb0 = -1.f, bb1 = -1.f, bb2 = -1.f;
        for (int f = 0; f < F; ++f) {
          const float* v0 = vb + 3 * faces[3 * f + 0];
          const float* v1 = vb + 3 * faces[3 * f + 1];
          const float* v2 = vb + 3 * faces[3 * f + 2];
          const float x0 = v0[0], y0 = v0[1], z0 = v0[2];
          const float x1 = v1[0], y1 = v1[1], z1 = v1[2];
          const float x2 = v2[0], y2 = v2[1], z2 = v2[2];
          const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
          const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
          /* sample outside the face bounding box (blur_radius = 0) */
          if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
          /* faces wholly at / behind the image plane (zmax < eps), or of ~zero area, are skipped; a face that
           * straddles the plane goes on to the per-sample pz >= 0 test below */
          const float zmax = fmaxf(z0, fmaxf(z1, z2));
          if (zmax < K_EPS) continue;
          const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
          if (face_area <= K_EPS && face_area >= -K_EPS) continue;
          /* BarycentricCoordsForward */
          const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
          const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
          const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
          const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;
          /* BarycentricPerspectiveCorrectionForward */
          const float t0 = w0 * z1 * z2;
          const float t1 = z0 * w1 * z2;
          const float t2 = z0 * z1 * w2;
          const float denom = fmaxf(t0 + t1 + t2, K_EPS);
          const float b0 = t0 / denom, b1 = t1 / denom, b2 = t2 / denom;
          const float pz = b0 * z0 + b1 * z1 + b2 * z2;
          if (pz < 0.0f) continue;
          /* inside test, strict (blur_radius = 0), on the perspective-CORRECTED barycentrics: PyTorch3D's CheckPixelInsideFace tests
           * `bary` (= the corrected ones under perspective_correct), not `bary0` [recalled; decided in round 3, DESIGN.md section 5].
           * Same decisions as the un-corrected test of rounds 1-2 whenever z0, z1, z2 > 0. */
          if (!(b0 > 0.0f && b1 > 0.0f && b2 > 0.0f)) continue;
          /* faces_per_pixel = 1: keep the nearest; on equal depth the earlier face stays */
          if (best_f < 0 || pz < best_z) {
            best_f = f; best_z = pz; bb0 = b0; bb1 = b1; bb2 = b2;
          }
        }
        const size_t o = ((size_t)b * S + yi) * S + xi;
        pix_to_face[o] = best_f;
        zbuf[o] = (best_f >= 0) ? best_z : -1.f;
        bary[3 * o + 0] = bb0; bary[3 * o + 1] = bb1; bary[3 * o + 2] = bb2;
      }
    }
  }
}
