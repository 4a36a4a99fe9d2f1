"""Tables of a NIMBLE-shaped hand layer.

The reference's `hand_model: "nimble"` configurations call MyNIMBLELayer (reference models_res_nimble.py:56-57,133-142), an
un-vendored submodule whose source and assets are absent from the reference checkout.  What the reference's
own files fix about it is its SHAPE: 20 shape / 30 pose / 10 texture components (models_res_nimble.py:56), 25 bone joints with root id
11 (:34,:135,:171), a 5 990-vertex skin mesh (:136), a 778-vertex MANO-topology regression of it (:139) and 21 MANO-ordered joints that
`Mano2Frei` re-orders (:157).  `NimbleTables` holds tables of that shape; `synthetic_nimble_tables` builds seeded stand-ins (a closed
genus-0 blob with five lobes, F = 2 V - 4 = 11 976 faces) so that the layer, its kernels (csrc/lbs.hip, csrc/texpca.hip) and the model
branch run at the real sizes.  Real NIMBLE assets, if a user has them, load through `NimbleTables(...)` with the same fields; nothing
here is pinned to NIMBLE's numbers ("parity unpinned").
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

NV, NF, NJ, NS, NP, NT = 5990, 11976, 25, 20, 30, 10
# 25 bones: a carpal root, a 4-joint thumb and four 5-joint fingers (metacarpal .. tip)
PARENTS = [-1] + [0, 1, 2, 3] + sum(([0, b, b + 1, b + 2, b + 3] for b in (5, 10, 15, 20)), [])
# 21 MANO-ordered joints (wrist, index, middle, pinky, ring, thumb -- the order Mano2Frei expects, fh_utils.py:547-552) picked
# from the 25: fingers drop their metacarpal joint
_FINGER = {"thumb": [1, 2, 3, 4], "index": [6, 7, 8, 9], "middle": [11, 12, 13, 14], "ring": [16, 17, 18, 19], "pinky": [21, 22, 23, 24]}
JOINT21 = [0] + _FINGER["index"] + _FINGER["middle"] + _FINGER["pinky"] + _FINGER["ring"] + _FINGER["thumb"]


@dataclass
class NimbleTables:
    v_template: np.ndarray      # [V,3]
    shapedirs: np.ndarray       # [V,3,S]
    J_regressor: np.ndarray     # [J,V]
    weights: np.ndarray         # [V,J]   <= 8 non-zeros per row, rows sum to 1
    parents: np.ndarray         # [J]     int32
    faces: np.ndarray           # [F,3]   int32
    pose_basis: np.ndarray      # [P, J*3]   theta = pose_mean + pose_params . pose_basis
    pose_mean: np.ndarray       # [J*3]
    tex_basis: np.ndarray       # [T, V*3]   per-vertex colours = tex_mean + texture_params . tex_basis
    tex_mean: np.ndarray        # [V*3]
    mano_vreg_fidx: np.ndarray  # [778]    skin face each MANO-topology vertex lies on
    mano_vreg_bc: np.ndarray    # [778,3]  barycentric coordinates on that face
    joint21: np.ndarray         # [21]     the 21 MANO-ordered joints among the J bones
    source: str = "user"
    # optional TexturesUV data (NIMBLE textures are images: reference models_res_nimble.py:203-208 hands the layer's texture image to the
    # renderer with per-face UV indices): uv per face corner, and a texture-IMAGE PCA instead of the per-vertex one
    faces_uvs: np.ndarray | None = None       # [F,3]  int32 indices into verts_uvs
    verts_uvs: np.ndarray | None = None       # [n,2]  (u, v) in [0, 1], v up (PyTorch3D's convention)
    tex_img_basis: np.ndarray | None = None   # [T, TH*TW*3]   texture image = tex_img_mean + texture_params . tex_img_basis
    tex_img_mean: np.ndarray | None = None    # [TH*TW*3]
    tex_hw: tuple | None = None               # (TH, TW)

    def astype32(self):
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(self, k, np.ascontiguousarray(v, dtype=np.int32 if v.dtype.kind in "iu" else np.float32))
        return self

    def check(self):
        V, J = self.v_template.shape[0], self.weights.shape[1]
        assert self.v_template.shape == (V, 3) and self.shapedirs.shape[:2] == (V, 3)
        assert self.J_regressor.shape == (J, V) and self.parents.shape == (J,) and self.parents[0] == -1
        assert all(0 <= self.parents[j] < j for j in range(1, J))
        assert (self.weights != 0).sum(1).max() <= 8 and np.allclose(self.weights.sum(1), 1.0, atol=1e-5)
        assert self.pose_basis.shape[1] == J * 3 and self.pose_mean.shape == (J * 3,)
        assert self.tex_basis.shape[1] == V * 3 and self.tex_mean.shape == (V * 3,)
        assert self.faces.min() >= 0 and self.faces.max() < V
        assert self.mano_vreg_fidx.shape == (778,) and self.mano_vreg_bc.shape == (778, 3)
        assert self.joint21.shape == (21,) and self.joint21.max() < J


def _lobed_sphere(V):
    """Closed genus-0 mesh with V = 2 + rings * 12 vertices: poles on +-z, `rings` latitude rings of 12."""
    rings, segs = (V - 2) // 12, 12
    assert rings * segs + 2 == V
    th = (np.arange(rings) + 1.0) / (rings + 1) * np.pi
    ph = np.arange(segs) / segs * 2 * np.pi
    T, P = np.meshgrid(th, ph, indexing="ij")
    unit = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1).reshape(-1, 3)
    unit = np.concatenate([[[0.0, 0.0, 1.0]], unit, [[0.0, 0.0, -1.0]]], 0)
    faces = []
    last = 1 + (rings - 1) * segs
    for j in range(segs):
        faces.append((0, 1 + j, 1 + (j + 1) % segs))
        faces.append((V - 1, last + (j + 1) % segs, last + j))
    for i in range(rings - 1):
        for j in range(segs):
            a, b = 1 + i * segs + j, 1 + i * segs + (j + 1) % segs
            faces.append((a, a + segs, b))
            faces.append((b, a + segs, b + segs))
    return unit, np.asarray(faces, dtype=np.int32)


def synthetic_nimble_tables(seed: int = 0) -> NimbleTables:
    """Deterministic NIMBLE-shaped tables (numpy RandomState => identical on every machine)."""
    rng = np.random.RandomState(seed)
    unit, faces = _lobed_sphere(NV)
    assert faces.shape[0] == NF
    # the blob: the z axis runs wrist (-z) to finger tips (+z); five lobes of the upper half stand in for the fingers
    z = unit[:, 2]
    az = np.arctan2(unit[:, 1], unit[:, 0])
    lobes = np.clip(np.cos(2.5 * az), 0.0, None) ** 2 * np.clip(z + 0.2, 0.0, None)
    radius = 1.0 + 0.35 * lobes
    verts = unit * radius[:, None] * np.array([0.045, 0.018, 0.09])

    # joints: the root near the wrist end, chains running up the blob at five azimuths
    joints = np.zeros((NJ, 3))
    joints[0] = (0.0, 0.0, -0.07)
    chains = [(1, 4, 0.9)] + [(b, 5, a) for b, a in zip((5, 10, 15, 20), (0.45, 0.15, -0.15, -0.45))]
    for base, n, x_frac in chains:
        for l in range(n):
            t = (l + 1.0) / n
            joints[base + l] = (0.04 * x_frac * (0.4 + 0.6 * t), 0.002 * ((base + l) % 3 - 1), -0.06 + 0.14 * t)

    d2 = ((verts[:, None, :] - joints[None, :, :]) ** 2).sum(-1)              # [V,J]
    w = np.exp(-d2 / (2 * 0.02 ** 2)) + 1e-12
    keep = np.argsort(-w, axis=1)[:, :4]                                       # <= 4 non-zeros per vertex, rows sum to 1
    wm = np.zeros_like(w)
    np.put_along_axis(wm, keep, np.take_along_axis(w, keep, axis=1), axis=1)
    wm[wm < 1e-3 * wm.max(axis=1, keepdims=True)] = 0.0
    weights = wm / wm.sum(axis=1, keepdims=True)

    jreg = np.zeros((NJ, NV))
    for j in range(NJ):
        idx = np.argsort(d2[:, j])[:96]
        ww = np.exp(-d2[idx, j] / (2 * 0.02 ** 2)) * (0.5 + rng.rand(idx.size))
        jreg[j, idx] = ww / ww.sum()

    shapedirs = np.zeros((NV, 3, NS))
    for k in range(NS):
        freq, phase = rng.uniform(8.0, 40.0, size=3), rng.uniform(0, 2 * np.pi, size=3)
        shapedirs[:, :, k] = np.sin(verts * freq + phase) * (rng.randn(3) * 0.003) + rng.randn(NV, 3) * 1.5e-4

    q, _ = np.linalg.qr(rng.randn(NJ * 3, NJ * 3))
    pose_basis = q[:NP] * 0.6                                                  # orthogonal rows, about 0.6 rad per unit coefficient
    pose_mean = 0.1 * rng.randn(NJ * 3)

    tone = np.array([0.78, 0.60, 0.50])
    tex_mean = np.tile(tone, NV) + 0.02 * np.sin(40.0 * verts).reshape(-1)
    tex_basis = np.zeros((NT, NV * 3))
    for k in range(NT):
        freq, phase = rng.uniform(20.0, 120.0, size=3), rng.uniform(0, 2 * np.pi, size=3)
        tex_basis[k] = (np.sin(verts * freq + phase).sum(1, keepdims=True) * (0.04 * rng.randn(3))).reshape(-1)

    fidx = rng.permutation(NF)[:778]
    bc = rng.dirichlet((2.0, 2.0, 2.0), size=778)

    t = NimbleTables(verts, shapedirs, jreg, weights, np.asarray(PARENTS), faces, pose_basis, pose_mean, tex_basis, tex_mean,
                     fidx, bc, np.asarray(JOINT21), source=f"synthetic(seed={seed})").astype32()
    t.check()
    return t


def add_synthetic_uv(t: NimbleTables, tex_hw=(64, 64), seed: int = 0) -> NimbleTables:
    """Seeded TexturesUV data for any tables: a cylindrical unwrap of the template (u = azimuth, v = height), one uv per face corner so that
    the faces crossing the azimuth seam stay un-stretched, and a texture-image PCA of smooth patterns around a skin tone."""
    rng = np.random.RandomState(1000 + seed)
    v = t.v_template - t.v_template.mean(0)
    az = np.arctan2(v[:, 1] / (np.abs(v[:, 1]).max() + 1e-9), v[:, 0] / (np.abs(v[:, 0]).max() + 1e-9)) / (2 * np.pi) + 0.5      # [0, 1)
    hz = (v[:, 2] - v[:, 2].min()) / (v[:, 2].max() - v[:, 2].min() + 1e-9)
    F_ = t.faces.shape[0]
    cu, cv = az[t.faces].copy(), hz[t.faces]                                            # [F,3]
    wrap = (cu.max(1) - cu.min(1)) > 0.5
    cu[wrap] = np.where(cu[wrap] < 0.5, cu[wrap] + 1.0, cu[wrap])                       # faces across the seam: unwrap ...
    uv = np.stack([cu / 1.25 * 0.9 + 0.05, cv * 0.9 + 0.05], -1).reshape(-1, 2)         # ... and keep everything inside [0.05, 0.95]
    TH, TW = tex_hw
    yy, xx = np.mgrid[:TH, :TW]
    tone = np.array([0.78, 0.60, 0.50])
    mean = tone[None, None, :] + 0.05 * np.sin(xx / TW * 14.0)[..., None] * np.cos(yy / TH * 9.0)[..., None]
    NT_ = t.tex_basis.shape[0]
    basis = np.zeros((NT_, TH, TW, 3))
    for k in range(NT_):
        fx, fy, ph = rng.uniform(2.0, 12.0), rng.uniform(2.0, 12.0), rng.uniform(0, 2 * np.pi)
        basis[k] = (np.sin(xx / TW * fx * 2 * np.pi + ph) * np.cos(yy / TH * fy * 2 * np.pi))[..., None] * (0.04 * rng.randn(3))
    t.faces_uvs = np.arange(3 * F_, dtype=np.int32).reshape(F_, 3)
    t.verts_uvs = uv.astype(np.float32)
    t.tex_img_mean = mean.reshape(-1).astype(np.float32)
    t.tex_img_basis = basis.reshape(NT_, -1).astype(np.float32)
    t.tex_hw = (int(TH), int(TW))
    return t
