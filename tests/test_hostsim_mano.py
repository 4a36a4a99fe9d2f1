"""Runs the MANO HIP kernel SOURCES on the CPU through tests/hostsim (a HIP execution-model emulator) and
checks them against the reference-pinned golden vectors and the oracle.  The real parity tests are the
`gpu`-marked ones in test_gpu_mano.py; this module exists so kernel bugs are found without a GPU."""
import os

import numpy as np
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


def test_mano_kernels_vs_reference_golden(hostsim_lib, synth_tables, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "mano_synth.npz")))
    kc.mano_fwd_bwd_case(hostsim_lib, synth_tables, g, "cpu")


def test_mano_kernels_vs_oracle_random(hostsim_lib, synth_tables):
    kc.mano_random_vs_oracle_case(hostsim_lib, synth_tables, "cpu", B=3, seed=5)


@pytest.mark.parametrize("root_id", [9, 0, -1])
def test_mano_joints_kernels(hostsim_lib, synth_tables, root_id):
    kc.mano_joints_case(hostsim_lib, synth_tables, "cpu", B=2, seed=3, root_id=root_id)


@pytest.mark.parametrize("B,root_id,with_cam", [(3, 9, True), (2, 0, False), (2, -1, True)])
def test_mano_full_kernels(hostsim_lib, synth_tables, B, root_id, with_cam):
    kc.mano_full_case(hostsim_lib, synth_tables, "cpu", B=B, seed=41 + B, root_id=root_id, with_cam=with_cam)


@pytest.fixture(scope="module")
def real_mano(tmp_path_factory):
    """(tables, golden) for the REAL MANO tables: the reference's own ManoLayer (utils/my_mano.py:315-483) is run on the spot by
    tools/make_golden.py into a temporary directory -- posed real-MANO meshes are derived licensed data and are never committed.
    Container only: needs MANO_RIGHT.pkl AND the reference tree; skipped everywhere else."""
    import subprocess
    import sys
    from conftest import mano_pkl_path
    pkl = mano_pkl_path()
    ref = os.environ.get("HIFIHR_REFERENCE", "/root/reference")
    if pkl is None or not os.path.exists(os.path.join(ref, "utils", "my_mano.py")):
        pytest.skip("real MANO tables + the reference tree are only present in the build container")
    out = str(tmp_path_factory.mktemp("mano_real"))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GOLDEN_ONLY="mano_real", HIFIHR_GOLDEN_OUT=out, HIFIHR_MANO_PKL=pkl)
    subprocess.run([sys.executable, os.path.join(repo, "tools", "make_golden.py")], check=True, env=env, cwd=repo, stdout=subprocess.DEVNULL)
    from hifihr_amd.mano_tables import load_mano_pkl
    return load_mano_pkl(pkl), dict(np.load(os.path.join(out, "mano_real.npz")))


def test_mano_kernels_on_real_tables_vs_reference(hostsim_lib, real_mano):
    """The kernel sources on the REAL tables (up to 6 non-zero skin weights per vertex, a dense 45 x 45 PCA basis; the synthetic tables
    have at most 4) against the reference's ManoLayer output generated a moment ago."""
    tables, g = real_mano
    assert int((np.asarray(tables.weights) != 0).sum(1).max()) > 4          # the case the synthetic tables do not have
    kc.mano_fwd_bwd_case(hostsim_lib, tables, g, "cpu")


@pytest.mark.parametrize("root_id,with_cam", [(9, True), (0, False)])
def test_mano_full_kernels_on_real_tables(hostsim_lib, real_mano, root_id, with_cam):
    kc.mano_full_case(hostsim_lib, real_mano[0], "cpu", B=3, seed=17, root_id=root_id, with_cam=with_cam)
