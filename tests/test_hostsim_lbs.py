"""Generic LBS kernel SOURCES (csrc/lbs.hip) on the hostsim emulator vs oracle/lbs_oracle.py."""
import pytest

import kernel_cases as kc


@pytest.fixture(scope="module")
def hostsim_lib():
    return kc.build_hostsim()


@pytest.mark.parametrize("V,J,S,B", [(300, 7, 5, 2), (777, 25, 20, 1), (64, 1, 0, 2), (257, 32, 32, 1)])
def test_lbs_random_tables(hostsim_lib, V, J, S, B):
    kc.lbs_case(hostsim_lib, "cpu", kc.random_lbs_tables(V, J, S, seed=V + J), B, seed=J)


def test_lbs_on_mano_tables(hostsim_lib, synth_tables):
    t = synth_tables
    kc.lbs_case(hostsim_lib, "cpu", (t.v_template, t.shapedirs, t.J_regressor, t.weights, kc.MANO_PARENTS16), 2, seed=4)


def test_lbs_rejects_bad_tables(hostsim_lib):
    import numpy as np
    vt, sd, jr, w, par = kc.random_lbs_tables(50, 5, 3, seed=1)
    bad = par.copy(); bad[2] = 3
    with pytest.raises(Exception, match="topologically"):
        hostsim_lib.lbs_create(vt, sd, jr, w, bad)
    dense = np.full((50, 12), 1.0 / 12, dtype=np.float32)
    with pytest.raises(Exception, match="non-zero skin weights"):
        hostsim_lib.lbs_create(vt, np.zeros((50, 3, 0), np.float32), np.zeros((12, 50), np.float32), dense, np.array([-1] + list(range(11)), dtype=np.int32))
