import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import omni_loader  # noqa: E402

omni_loader.load()          # registers omni-swarm_amd/ as `omni_swarm_amd` so that test modules can import its data generators at collection time

# fp32 shards answer batched searches through their fp16 mirror only from 32768 rows on (below, the exact kernels are as fast); the tests want
# that path -- mirror pass, exact re-scoring, certificate, fallback -- on their small databases too (read once by libomni_hip.so)
# (OMNI_TEST_PRODUCTION_DEFAULTS=1 leaves the production threshold in place: tests/test_gpu_index.py re-runs its batched-search cases that way in a subprocess)
if os.environ.get("OMNI_TEST_PRODUCTION_DEFAULTS") != "1":
    os.environ.setdefault("OMNI_INDEX_MIRROR_MIN_ROWS", "0")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def omni():
    return omni_loader.load()


@pytest.fixture(scope="session")
def ctx(omni):
    """HIP context on cuda:0.  GPU tests must FAIL (not skip) when the native library or device is missing."""
    c = omni.capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return lambda name: np.load(os.path.join(GOLDEN, name))
