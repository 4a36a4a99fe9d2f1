"""MANO model tables: container, loader for a user-supplied MANO_RIGHT.pkl, and a seeded
synthetic MANO-shaped generator.

The MANO licence forbids redistribution, so the repository ships NO MANO data.  Perf runs, the
GPU parity tests and `bench.py` use `synthetic_mano_tables()` (same shapes, sparsity pattern and
topology class as MANO: 778 verts, 1538 faces = a triangulated disc with a 16-edge boundary,
16 joints, parents [-,0,1,2,0,4,5,0,7,8,0,10,11,0,13,14]).  Users who own the MANO file load it with
`load_mano_pkl()`; the table layout follows what the reference's `ManoLayer.__init__` registers
(reference utils/my_mano.py:283-313).
"""
from __future__ import annotations

import pickle
import sys
import types
from dataclasses import dataclass

import numpy as np

NV = 778      # vertices
NF = 1538     # faces
NJ = 16       # joints (root + 5 fingers x 3)
NB = 10       # shape components
NP = 135      # pose-corrective components = 15 joints x 9
NPCA = 45     # articulated pose dims

KINTREE_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
# reference utils/my_mano.py:457 (ManoLayer tips) and Freihand_trainer_mano_fullsup.py:177-183
TIPS_MANOLAYER = (745, 317, 444, 556, 673)
TIPS_XYZ_FROM_VERTICE = (744, 320, 443, 555, 672)


@dataclass
class ManoTables:
    v_template: np.ndarray        # [778,3]  f32
    shapedirs: np.ndarray         # [778,3,10]
    posedirs: np.ndarray          # [778,3,135]
    J_regressor: np.ndarray       # [16,778] dense
    weights: np.ndarray           # [778,16]
    hands_components: np.ndarray  # [45,45]  rows = PCA components
    hands_mean: np.ndarray        # [45]
    faces: np.ndarray             # [1538,3] int32
    source: str = "synthetic"

    def astype32(self) -> "ManoTables":
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        return ManoTables(f(self.v_template), f(self.shapedirs), f(self.posedirs), f(self.J_regressor),
                          f(self.weights), f(self.hands_components), f(self.hands_mean),
                          np.ascontiguousarray(self.faces, dtype=np.int32), self.source)

    def check(self) -> None:
        assert self.v_template.shape == (NV, 3)
        assert self.shapedirs.shape == (NV, 3, NB)
        assert self.posedirs.shape == (NV, 3, NP)
        assert self.J_regressor.shape == (NJ, NV)
        assert self.weights.shape == (NV, NJ)
        assert self.hands_components.shape == (NPCA, NPCA)
        assert self.hands_mean.shape == (NPCA,)
        assert self.faces.shape == (NF, 3) and self.faces.min() >= 0 and self.faces.max() < NV


# ----------------------------------------------------------------------------------------------
# synthetic, MANO-shaped
# ----------------------------------------------------------------------------------------------
def _capsule_mesh():
    """778 verts / 1538 faces: 48 rings x 16 verts (open at ring 0 = the "wrist"), closed at the far
    end by a cap of an 8-ring and two centre verts.  Disc topology, 16 boundary edges (like MANO)."""
    nring, nseg = 48, 16
    verts = []
    length = 0.19
    for r in range(nring):
        t = r / (nring - 1)
        y = length * t
        # flattened, tapering cross-section: palm-like near the wrist, finger-like at the far end
        rx = 0.042 * (1.0 - 0.55 * t)
        rz = 0.016 * (1.0 - 0.35 * t)
        for s in range(nseg):
            a = 2.0 * np.pi * (s + 0.5 * (r & 1)) / nseg
            verts.append((rx * np.cos(a), y, rz * np.sin(a)))
    base = nring * nseg
    t_end = length
    for s in range(8):                      # inner 8-ring of the cap
        a = 2.0 * np.pi * (s + 0.25) / 8
        verts.append((0.010 * np.cos(a), t_end + 0.004, 0.005 * np.sin(a)))
    verts.append((-0.003, t_end + 0.006, 0.0))   # c0
    verts.append((0.003, t_end + 0.006, 0.0))    # c1
    verts = np.asarray(verts, dtype=np.float64)
    assert verts.shape[0] == NV

    faces = []
    for r in range(nring - 1):
        for s in range(nseg):
            a0 = r * nseg + s
            a1 = r * nseg + (s + 1) % nseg
            b0 = (r + 1) * nseg + s
            b1 = (r + 1) * nseg + (s + 1) % nseg
            if r & 1:
                faces.append((a0, a1, b1)); faces.append((a0, b1, b0))
            else:
                faces.append((a0, a1, b0)); faces.append((a1, b1, b0))
    last = (nring - 1) * nseg
    for s in range(8):                      # 16-ring -> 8-ring: 24 triangles
        o0 = last + 2 * s
        o1 = last + (2 * s + 1) % nseg
        o2 = last + (2 * s + 2) % nseg
        i0 = base + s
        i1 = base + (s + 1) % 8
        faces.append((o0, o1, i0)); faces.append((o1, o2, i0)); faces.append((o2, i1, i0))
    a = [base + s for s in range(8)]
    c0, c1 = base + 8, base + 9
    for s in range(4):
        faces.append((a[s], a[s + 1], c0))
    for s in range(4, 8):
        faces.append((a[s], a[(s + 1) % 8], c1))
    faces.append((a[0], c0, c1)); faces.append((c0, a[4], c1))
    faces = np.asarray(faces, dtype=np.int32)
    assert faces.shape[0] == NF, faces.shape
    return verts, faces


def synthetic_mano_tables(seed: int = 0) -> ManoTables:
    """Deterministic MANO-shaped tables (numpy RandomState ⇒ identical on every machine)."""
    rng = np.random.RandomState(seed)
    verts, faces = _capsule_mesh()

    # joints: root at the wrist opening, 5 "fingers" fanned across x, 3 joints each along y
    joints = np.zeros((NJ, 3))
    joints[0] = (0.0, 0.005, 0.0)
    for f in range(5):
        x = (f - 2) * 0.012
        for l in range(3):
            joints[1 + 3 * f + l] = (x * (1.0 - 0.25 * l), 0.05 + 0.045 * l, 0.002 * ((f + l) % 3 - 1))

    d2 = ((verts[:, None, :] - joints[None, :, :]) ** 2).sum(-1)          # [V,J]
    # skinning weights: <=4 non-zeros per vertex, rows sum to 1
    w = np.exp(-d2 / (2 * 0.018 ** 2)) + 1e-12
    keep = np.argsort(-w, axis=1)[:, :4]
    wm = np.zeros_like(w)
    np.put_along_axis(wm, keep, np.take_along_axis(w, keep, axis=1), axis=1)
    wm[wm < 1e-3 * wm.max(axis=1, keepdims=True)] = 0.0
    weights = wm / wm.sum(axis=1, keepdims=True)

    # joint regressor: each joint = convex combination of its ~118 nearest verts (MANO nnz = 1896)
    jreg = np.zeros((NJ, NV))
    for j in range(NJ):
        idx = np.argsort(d2[:, j])[:118]
        ww = np.exp(-d2[idx, j] / (2 * 0.02 ** 2)) * (0.5 + rng.rand(idx.size))
        jreg[j, idx] = ww / ww.sum()

    # smooth blend-shape fields: a few low-frequency modes of the rest position
    def smooth_field(ncomp, scale):
        out = np.zeros((NV, 3, ncomp))
        for k in range(ncomp):
            freq = rng.uniform(8.0, 40.0, size=3)
            phase = rng.uniform(0, 2 * np.pi, size=3)
            amp = rng.randn(3) * scale
            s = np.sin(verts @ np.diag(freq) + phase)                       # [V,3]
            out[:, :, k] = s * amp + rng.randn(NV, 3) * (0.05 * scale)
        return out

    shapedirs = smooth_field(NB, 0.004)
    posedirs = smooth_field(NP, 0.0015)

    q, _ = np.linalg.qr(rng.randn(NPCA, NPCA))
    hands_components = q                                                     # orthonormal rows
    hands_mean = 0.15 * rng.randn(NPCA)

    t = ManoTables(verts, shapedirs, posedirs, jreg, weights, hands_components, hands_mean, faces,
                   source=f"synthetic(seed={seed})").astype32()
    t.check()
    return t


# ----------------------------------------------------------------------------------------------
# user-supplied MANO_RIGHT.pkl (chumpy pickle) -- no chumpy needed
# ----------------------------------------------------------------------------------------------
class _ChStub:
    """Absorbs the state of a pickled chumpy object; only raw ndarray attributes are read."""

    def __setstate__(self, state):
        self.__dict__.update(state)


def _install_chumpy_stubs():
    added = []
    for name in ("chumpy", "chumpy.ch", "chumpy.reordering"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            added.append(name)
    ch, chch, reo = sys.modules["chumpy"], sys.modules["chumpy.ch"], sys.modules["chumpy.reordering"]
    for mod, names in ((ch, ("Ch",)), (chch, ("Ch",)), (reo, ("Select",))):
        for n in names:
            if not hasattr(mod, n):
                setattr(mod, n, type(n, (_ChStub,), {}))
    if not hasattr(ch, "ch"):
        ch.ch = chch
    if not hasattr(ch, "reordering"):
        ch.reordering = reo
    return added


def _as_array(obj) -> np.ndarray:
    if isinstance(obj, np.ndarray):
        return obj
    if hasattr(obj, "toarray"):                       # scipy sparse (J_regressor)
        return np.asarray(obj.toarray())
    d = getattr(obj, "__dict__", {})
    if "idxs" in d and "a" in d:                      # chumpy.reordering.Select (shapedirs)
        flat = _as_array(d["a"]).ravel()[np.asarray(d["idxs"])]
        return flat.reshape(d["preferred_shape"])
    if "x" in d:                                      # chumpy.Ch leaf
        return np.asarray(d["x"])
    raise TypeError(f"cannot recover an array from pickled {type(obj)}")


def load_mano_pkl(path: str) -> ManoTables:
    """Read the arrays `ManoLayer.__init__` uses (reference utils/my_mano.py:277-313) from a
    MANO_RIGHT.pkl, without chumpy."""
    added = _install_chumpy_stubs()
    try:
        with open(path, "rb") as fh:
            dd = pickle.load(fh, encoding="latin1")
    finally:
        for name in added:
            sys.modules.pop(name, None)
    t = ManoTables(
        v_template=_as_array(dd["v_template"]),
        shapedirs=_as_array(dd["shapedirs"]),
        posedirs=_as_array(dd["posedirs"]),
        J_regressor=_as_array(dd["J_regressor"]),
        weights=_as_array(dd["weights"]),
        hands_components=_as_array(dd["hands_components"]),
        hands_mean=_as_array(dd["hands_mean"]),
        faces=_as_array(dd["f"]).astype(np.int32),
        source=path,
    ).astype32()
    t.check()
    return t
