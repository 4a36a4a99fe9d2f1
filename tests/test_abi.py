"""The C-ABI library loads and exports every symbol include/hifihr.h declares (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "hifihr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hifihr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib_path = os.path.join(REPO, "hifihr_amd", "libhifihr.so")
    if not os.path.exists(lib_path):
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "hifihr_amd", "csrc"), "-j8"], check=True)
    import torch  # noqa: F401  (libamdhip64 is resolved from torch's copy)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hifihr.h but not exported"
    lib.hifihr_version.restype = ctypes.c_int
    assert lib.hifihr_version() >= 1


def test_python_binding_covers_the_header():
    from hifihr_amd._lib import HifihrLib, LIB_PATH
    lib = HifihrLib(LIB_PATH)
    for n in _declared():
        getattr(lib.c, n)
