"""CPU-side checks of the C-ABI boundary: the library loads, exports exactly what include/omni_hip.h declares, the
host-only entry points work, and the product path fails loudly (never falls back) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import match_ref as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(omni):
    c = omni.capi
    assert os.path.exists(c.LIB_PATH), "libomni_hip.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(c.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "omni_hip.h")).read()
    declared = set(re.findall(r"\b(omni_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(c.SYMBOLS), (declared ^ set(c.SYMBOLS))
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/omni_hip.h but not exported"
    assert c.lib().omni_abi_version() == 1


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "omni-swarm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_no_gpu_means_loud_failure_not_fallback(omni):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(omni.capi.OmniError, match="no CPU fallback"):
        omni.capi.Context(0)


def test_topk_merge_host(omni):
    rng = np.random.default_rng(1)
    db = rng.standard_normal((400, 64)).astype(np.float32)
    db[300] = db[7]                                    # a tie across shards
    q = np.stack([db[7], db[100] + 0.01]).astype(np.float32)
    k, world = 6, 4
    Dl, Il = [], []
    for r in range(world):
        rows = np.arange(r, 400, world)
        D, I = M.ip_search(db[rows], q, k)
        Il.append(np.where(I >= 0, I * world + r, -1))
        Dl.append(D)
    D, I = omni.capi.topk_merge(np.stack(Dl), np.stack(Il), k)
    Dr, Ir = M.ip_search(db, q, k)
    assert np.array_equal(I, Ir) and np.array_equal(D, Dr)
    assert I[0, 0] == 7 and I[0, 1] == 300
    # short shards: padding entries are ignored
    D2, I2 = omni.capi.topk_merge(np.full((2, 1, 3), -3.4e38, np.float32), np.full((2, 1, 3), -1, np.int64), 3)
    assert (I2 == -1).all()
