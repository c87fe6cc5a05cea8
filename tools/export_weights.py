#!/usr/bin/env python
"""Writes OMNW1 weight files for the C++ host adapters (omni-swarm_amd/host/omni_swarm.hpp: omni::load_omnw).

    python tools/export_weights.py superpoint <superpoint_v1.pth | synth> out.omnw
    python tools/export_weights.py mobilenetvlad synth out.omnw
    python tools/export_weights.py pca synth components_.csv mean_.csv

OMNW1 = "OMNW1\\0\\0\\0", u32 n, then per tensor: u32 name_len, name, u32 ndim, u32 dims[ndim], float32 data (little endian).
The MobileNetVLAD file carries the (assumed) layer table as tensor "layers" [n][4] = (kind, cin, cout, stride) and the
per-layer tensors as "layer<i>.weight" / "layer<i>.bias".
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_omnw(path, tensors: dict):
    import omni_loader
    omni_loader.load()
    from omni_swarm_amd import weights
    weights.write_omnw(path, tensors)


def vlad_tensors(weights_dict, specs, kinds):
    import omni_loader
    omni_loader.load()
    from omni_swarm_amd import weights
    return weights.vlad_omnw_tensors(weights_dict, specs, kinds)


def main(argv):
    import omni_loader
    omni_loader.load()
    from omni_swarm_amd import capi, synth, weights
    what = argv[1]
    if what == "superpoint":
        w = weights.superpoint_synth_weights(0) if argv[2] == "synth" else weights.load_superpoint_pth(argv[2])
        write_omnw(argv[3], w)
    elif what == "mobilenetvlad":
        w = weights.mobilenetvlad_synth_weights()
        write_omnw(argv[3], vlad_tensors(w, weights.mobilenetvlad_layer_specs(), capi.VLAD_KINDS))
    elif what == "pca":
        comp, mean = synth.pca()
        np.savetxt(argv[3], comp, delimiter=",", fmt="%.9g")
        np.savetxt(argv[4], mean, fmt="%.9g")
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main(sys.argv)
