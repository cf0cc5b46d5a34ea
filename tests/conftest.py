import json
import os
import sys

# numpy asks for transparent huge pages on every allocation >= 4 MiB; with THP in "madvise" mode and defrag = madvise a first
# touch then compacts memory in the page fault.  On a long-running box that made the fixture generator's temporaries cost
# 16 s of system time per 25k-token input instead of 4 (the whole CPU suite 17 min instead of 8).  Off for the tests.
os.environ.setdefault("NUMPY_MADVISE_HUGEPAGE", "0")
try:                                     # ... and glibc keeps freed blocks up to 32 MiB in the heap instead of handing every
    import ctypes                        #     multi-megabyte temporary back to the kernel (M_MMAP_THRESHOLD, M_TRIM_THRESHOLD,
    _libc = ctypes.CDLL("libc.so.6")     #     M_TOP_PAD): the generator's per-frame temporaries stop page-faulting
    _libc.mallopt(-3, 32 << 20), _libc.mallopt(-1, 1 << 30), _libc.mallopt(-2, 64 << 20)
except Exception:                        # pragma: no cover - not glibc
    pass

import numpy as np
import pytest
import torch

try:                                     # (numpy may have been imported before this file: the switch also exists at run time)
    from numpy._core.multiarray import _set_madvise_hugepage
    _set_madvise_hugepage(False)
except Exception:                        # pragma: no cover - older numpy layouts
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the HIP library and the oracle are built (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()


_master_cache = {}


def make_input(F, N, D, dtype_name, seed, dist):
    """Regenerate a fixture input from its seed (fp32 master cached across dtypes)."""
    from vidcom2_amd import synth
    key = (F, N, D, seed, dist)
    if key not in _master_cache:
        if len(_master_cache) > 2:
            _master_cache.clear()
        _master_cache[key] = synth.make_fp32(F, N, D, seed, dist)
    return synth.to_torch(_master_cache[key], DT[dtype_name]).reshape(F * N, D)


def load_core_cases():
    with open(os.path.join(GOLDEN, "core_cases.json")) as fh:
        cases = json.load(fh)["cases"]
    # order so that cases sharing an fp32 master are adjacent (cache hits)
    cases.sort(key=lambda c: (c["F"] * c["N"] * c["D"], c["name"], c["seed"], c["dist"], c["dtype"]))
    return cases


def case_id(c):
    return f"{c['name']}-{c['dtype']}-{c['dist']}-s{c['seed']}"


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_topk_kat():
    z = np.load(os.path.join(GOLDEN, "topk_kat.npz"))
    out = []
    for i in range(len(z["k"])):
        v = z["values"][z["offs_v"][i]: z["offs_v"][i + 1]]
        o = z["out"][z["offs_o"][i]: z["offs_o"][i + 1]]
        out.append((v, int(z["k"][i]), bool(z["sorted"][i]), ("f32", "bf16", "f16")[int(z["dtype"][i])], o))
    return out
