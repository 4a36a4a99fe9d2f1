"""Kernel-level parity cases shared by the hostsim (CPU, `not gpu`) and HIP (`gpu`) test modules.

Each case drives the C ABI through hifihr_amd._lib.HifihrLib with tensors on `device` and compares with the
oracle.  `lib` is either the product library (device='cuda') or tests/hostsim's emulator build (device='cpu').
"""
import os
import subprocess

import numpy as np
import torch

from hifihr_amd._lib import HifihrLib
from oracle import mano_oracle as mo

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM_DIR = os.path.join(REPO, "tests", "hostsim")
HOSTSIM_LIB = os.path.join(HOSTSIM_DIR, "libhifihr_hostsim.so")


def build_hostsim() -> HifihrLib:
    subprocess.run(["make", "-s", "-C", HOSTSIM_DIR, "-j8"], check=True)
    return HifihrLib(HOSTSIM_LIB)


def _dev(x, device):
    return torch.as_tensor(x).to(device).contiguous()


def mano_fwd_bwd_case(lib, tables, g, device, atol_v=5e-6, gtol=3e-4):
    """g: golden dict with pose, beta, wv, wj, verts, jtr, gpose, gbeta (reference ManoLayer outputs)."""
    h = lib.mano_create(tables)
    try:
        pose, beta = _dev(g["pose"], device), _dev(g["beta"], device)
        B = pose.shape[0]
        verts = torch.empty(B, 778, 3, device=device)
        jtr = torch.empty(B, 21, 3, device=device)
        saved = torch.empty(B, 778, 3, device=device)
        lib.mano_lbs_fwd(h, pose, beta, verts, jtr, saved)
        np.testing.assert_allclose(verts.cpu().numpy(), g["verts"], atol=atol_v, rtol=0)
        np.testing.assert_allclose(jtr.cpu().numpy(), g["jtr"], atol=atol_v, rtol=0)
        gpose = torch.empty(B, 48, device=device)
        gbeta = torch.empty(B, 10, device=device)
        lib.mano_lbs_bwd(h, pose, beta, saved, _dev(g["wv"], device), _dev(g["wj"], device), gpose, gbeta)
        scale_p = np.abs(g["gpose"]).max()
        scale_b = np.abs(g["gbeta"]).max()
        np.testing.assert_allclose(gpose.cpu().numpy(), g["gpose"], atol=gtol * scale_p, rtol=1e-4)
        np.testing.assert_allclose(gbeta.cpu().numpy(), g["gbeta"], atol=gtol * scale_b, rtol=1e-4)
    finally:
        lib.mano_destroy(h)


def mano_random_vs_oracle_case(lib, tables, device, B, seed):
    gen = torch.Generator().manual_seed(seed)
    pose = (0.6 * torch.randn(B, 48, generator=gen)).requires_grad_(True)
    beta = (0.7 * torch.randn(B, 10, generator=gen)).requires_grad_(True)
    wv = torch.randn(B, 778, 3, generator=gen)
    wj = torch.randn(B, 21, 3, generator=gen)
    verts, jtr, _ = mo.mano_forward(tables, pose, beta)
    ((verts * wv).sum() + (jtr * wj).sum()).backward()
    g = dict(pose=pose.detach().numpy(), beta=beta.detach().numpy(), wv=wv.numpy(), wj=wj.numpy(),
             verts=verts.detach().numpy(), jtr=jtr.detach().numpy(), gpose=pose.grad.numpy(), gbeta=beta.grad.numpy())
    mano_fwd_bwd_case(lib, tables, g, device)


def mano_joints_case(lib, tables, device, B, seed, root_id=9):
    gen = torch.Generator().manual_seed(seed)
    verts = (0.1 * torch.randn(B, 778, 3, generator=gen)).requires_grad_(True)
    j = mo.xyz_from_vertice(tables, verts)
    if root_id >= 0:
        jr, vr, root = mo.root_relative(j, verts, root_id)
    else:
        jr, vr, root = j, verts, torch.zeros(B, 1, 3)
    wj = torch.randn(B, 21, 3, generator=gen)
    wv = torch.randn(B, 778, 3, generator=gen)
    wr = torch.randn(B, 3, generator=gen)
    ((jr * wj).sum() + (vr * wv).sum() + (root.reshape(B, 3) * wr).sum()).backward()
    h = lib.mano_create(tables)
    try:
        vd = _dev(verts.detach(), device)
        o_j = torch.empty(B, 21, 3, device=device)
        o_v = torch.empty(B, 778, 3, device=device)
        o_r = torch.empty(B, 3, device=device)
        lib.mano_joints_fwd(h, vd, root_id, o_j, o_v, o_r)
        np.testing.assert_allclose(o_j.cpu().numpy(), jr.detach().numpy(), atol=2e-6)
        np.testing.assert_allclose(o_v.cpu().numpy(), vr.detach().numpy(), atol=2e-6)
        np.testing.assert_allclose(o_r.cpu().numpy(), root.detach().reshape(B, 3).numpy(), atol=2e-6)
        gv = torch.empty(B, 778, 3, device=device)
        lib.mano_joints_bwd(h, _dev(wj, device), _dev(wv, device), _dev(wr, device) if root_id >= 0 else None, root_id, gv)
        np.testing.assert_allclose(gv.cpu().numpy(), verts.grad.numpy(), atol=2e-4, rtol=1e-4)
    finally:
        lib.mano_destroy(h)
