"""Re-export of the seeded synthetic input generators (they live in the product package so bench.py can use them
without touching oracle/)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import omni_loader  # noqa: E402

omni_loader.load()
from omni_swarm_amd.synth import *  # noqa: F401,F403,E402
from omni_swarm_amd.synth import BASE_SEED  # noqa: F401,E402
