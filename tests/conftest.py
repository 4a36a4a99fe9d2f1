import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

pytest.register_assert_rewrite("kernel_cases", "loss_cases")       # the shared case bodies: their assertions report margins too

# ---- margins of the passing assertions ---------------------------------------------------------------------------------------------
# `pytest -o enable_assertion_pass_hook=true` (tools/flake_check.sh does): every PASSING `assert a <= b` / `a < b` whose two sides
# evaluate to non-negative numbers is logged as observed / bound; the session ends with the assertions closest to their bounds
# (and gpurun_out/margins.txt).  A tolerance that holds with a margin below 2 is a flake waiting for another box (GPUTEST_r04).
_MARGINS = {}
_SAFE = {"max": max, "min": min, "abs": abs, "float": float, "int": int, "inf": float("inf"), "nan": float("nan")}


def pytest_assertion_pass(item, lineno, orig, expl):
    import ast
    try:
        tree = ast.parse(expl.split("\n", 1)[0].strip(), mode="eval")
    except (SyntaxError, ValueError, MemoryError):
        return
    for node in ast.walk(tree):
        if not (isinstance(node, ast.Compare) and len(node.ops) == 1 and isinstance(node.ops[0], (ast.Lt, ast.LtE))):
            continue
        try:
            left = eval(compile(ast.Expression(node.left), "<margin>", "eval"), {"__builtins__": {}}, _SAFE)
            right = eval(compile(ast.Expression(node.comparators[0]), "<margin>", "eval"), {"__builtins__": {}}, _SAFE)
        except Exception:
            continue
        if isinstance(left, bool) or isinstance(right, bool) or not isinstance(left, (int, float)) or not isinstance(right, (int, float)):
            continue
        if not (right > 0 and left >= 0):
            continue
        key = (item.nodeid.split("::")[0], lineno, " ".join(orig.split())[:110])
        ratio = left / right
        if ratio > _MARGINS.get(key, (-1.0,))[0]:
            _MARGINS[key] = (ratio, left, right, item.nodeid)


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    rows = sorted(_MARGINS.items(), key=lambda kv: -kv[1][0])
    lines = [f"{len(rows)} passing inequality assertions logged; closest to their bounds (observed / bound; 1.0 = at the bound):"]
    for (f, ln, orig), (ratio, left, right, nodeid) in rows[:60]:
        lines.append(f"  {ratio:7.3f}  {left:.3e} vs {right:.3e}  {nodeid} (line {ln}): {orig}")
    text = "\n".join(lines)
    print("\n" + text)
    out = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "margins.txt"), "w") as fh:
            fh.write(text + "\n")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


# Order of the -m gpu suite: the per-row parity files (one kernel family each, deterministic inputs, tight bounds) run FIRST,
# the whole-step / multi-process / front-end files LAST, so that under `-x` a failure in a composite test can no longer hide
# the parity evidence of every row behind it (GPUTEST_r04: one whole-step tolerance hid 125 kernel tests).
_FILE_ORDER = ["test_gpu_mano", "test_gpu_render", "test_raster_known", "test_shade_known", "test_gpu_losses", "test_gpu_tail", "test_gpu_gemm",
               "test_gpu_nimble", "test_gpu_conv", "test_gpu_e2e", "test_gpu_frontend", "test_gpu_dp"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _FILE_ORDER.index(name) if name in _FILE_ORDER else -1          # CPU files keep their place in front
    items.sort(key=rank)                                                        # stable: order inside a file is unchanged


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_tables():
    from hifihr_amd.mano_tables import synthetic_mano_tables
    return synthetic_mano_tables(0)


def mano_pkl_path():
    """A user-supplied MANO file (never shipped).  In the build container the reference mounts one."""
    for p in (os.environ.get("HIFIHR_MANO_PKL"), "/root/reference/data/MANO_RIGHT.pkl"):
        if p and os.path.exists(p):
            return p
    return None
