#!/usr/bin/env python
"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh (gpurun_out/<tag>_pmc3, _pmc4): HBM bytes per
launch of the kernels bench.py quotes a `traffic` for.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KiB) counts half of a wide
coalesced streaming read -> bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024.

    python tools/make_traffic.py r02k 64 > profiles/r02k_traffic.json        (64 = images per SuperPoint launch in that run)
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def mean_by(dirname, counter, match, grid=None):
    vals = []
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and match(r["Kernel_Name"]) and (grid is None or int(r["Grid_Size"]) == grid):
                vals.append(float(r["Counter_Value"]))
    return vals


def main(tag, images):
    out = {"source": f"gpurun_out/{tag}_pmc3 (FETCH_SIZE) and {tag}_pmc4 (WRITE_SIZE): separate rocprofv3 --pmc passes of tools/profile_gpu.sh at HEAD, "
                     f"micro-batches of {images // 8} key frames = {images} images per SuperPoint launch",
           "correction": "gfx950: FETCH_SIZE counts 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM): bytes = FETCH_SIZE_KiB*1024*2 + WRITE_SIZE_KiB*1024"}
    kernels = {
        "conv3x3_c64_pp_kernel<POOL,FUSE1A>": lambda n: "conv3x3_c64_pp_kernelILb1ELi0ELb1" in n,
        "conv3x3_c128_rs_kernel<POOL>": lambda n: "conv3x3_c128_rs_kernelILb1" in n,
        "conv3x3_c128_rs_kernel": lambda n: "conv3x3_c128_rs_kernelILb0" in n,
        "ip_scan_mq_kernel": lambda n: "ip_scan_mq_kernel" in n,
        "vlad_mblock_kernel": lambda n: "vlad_mblock_kernel" in n,
        "vlad_fc_mfma_kernel": lambda n: "vlad_fc_mfma_kernel" in n,
        # OMNI_PREC_SPLIT: conv1b = the FUSE1A instantiation <C128 = 0, POOL = 1, OUT_F32 = 0, TRN = 0, FUSE1A = 1> (conv1a is built inside it from the u8 image)
        "conv3x3_split_kernel<cin64,POOL,FUSE1A> (conv1b)": lambda n: "conv3x3_split_kernelILb0ELb1ELb0ELb0ELb1" in n,
        "conv3x3_split_c128_sparse_kernel (convDa at the key points)": lambda n: "conv3x3_split_c128_sparse_kernel" in n,
        # OMNI_PREC_SPLIT, cin = 128 (template arguments <C128, POOL, OUT_F32, TRN, FUSE1A, FZMIX>): the layers VERDICT r4 asked a read traffic of <= 1.5 x the
        # algorithmic input for (split-64 activations: 512 B per pixel at cin = 128)
        "conv3x3_split_kernel<cin128,POOL> (conv3b)": lambda n: "conv3x3_split_kernelILb1ELb1ELb0ELb0ELb0" in n,
        "conv3x3_split_kernel<cin128,TRN> (conv4a, conv4b)": lambda n: "conv3x3_split_kernelILb1ELb0ELb0ELb1ELb0" in n,
        "conv3x3_split_kernel<cin128,OUT_F32,TRN> (convPa)": lambda n: "conv3x3_split_kernelILb1ELb0ELb1ELb1ELb0" in n,
        "conv3x3_split_kernel<cin64,POOL> (conv2b)": lambda n: "conv3x3_split_kernelILb0ELb1ELb0ELb0ELb0" in n,
        # OMNI_PREC_SPLIT, Winograd F(2x2,3x3) kernels of conv_wino.hip (template arguments <POOL, OUT_SPLIT, FUSE1A>)
        "conv3x3_wino_kernel<POOL,FUSE1A> (conv1b)": lambda n: "conv3x3_wino_kernelILb1ELb0ELb1" in n,
        "conv3x3_wino_kernel (conv2a)": lambda n: "conv3x3_wino_kernelILb0ELb0ELb0" in n,
        "conv3x3_wino_kernel<POOL,OUT_SPLIT> (conv2b)": lambda n: "conv3x3_wino_kernelILb1ELb1ELb0" in n,
    }
    algorithmic_input = {   # bytes of the layer's input tensor per launch of `images` images (what a single pass over it reads)
        "conv3x3_split_kernel<cin128,POOL> (conv3b)": 120 * 150 * 512 * images,
        "conv3x3_split_kernel<cin128,TRN> (conv4a, conv4b)": 60 * 75 * 512 * images,
        "conv3x3_split_kernel<cin128,OUT_F32,TRN> (convPa)": 60 * 75 * 512 * images,
        "conv3x3_split_kernel<cin64,POOL> (conv2b)": 240 * 300 * 256 * images,
        "conv3x3_wino_kernel<POOL,FUSE1A> (conv1b)": 480 * 600 * images,                   # the u8 image: conv1a is built inside the kernel
        "conv3x3_wino_kernel (conv2a)": 240 * 300 * 256 * images,                          # raw-32 frames: 256 B per pixel
        "conv3x3_wino_kernel<POOL,OUT_SPLIT> (conv2b)": 240 * 300 * 256 * images,
    }
    for key, m in kernels.items():
        rd = mean_by(f"gpurun_out/{tag}_pmc3", "FETCH_SIZE", m)
        wr = mean_by(f"gpurun_out/{tag}_pmc4", "WRITE_SIZE", m)
        if rd and wr:
            r, w = sum(rd) / len(rd) * 1024 * 2, sum(wr) / len(wr) * 1024
            out[key] = {"read_bytes": round(r), "write_bytes": round(w), "bytes_per_launch": round(r + w), "dispatches": len(rd)}
            if "conv" in key:
                out[key]["images_per_launch"] = images
            if key in algorithmic_input:
                out[key]["algorithmic_input_bytes"] = algorithmic_input[key]
                out[key]["read_over_algorithmic_input"] = round(r / algorithmic_input[key], 3)
    # single-query scans: one entry per database size seen in the pass (launches clustered by FETCH_SIZE; rows = bytes / 16 384 rounded to 1 000)
    rd = sorted(mean_by(f"gpurun_out/{tag}_pmc3", "FETCH_SIZE", lambda n: "ip_scan_kernel<float, 1>" in n))
    clusters = []
    for x in rd:
        if clusters and x < 1.1 * clusters[-1][0]:
            clusters[-1].append(x)
        else:
            clusters.append([x])
    for c in clusters:
        b = sum(c) / len(c) * 1024 * 2
        rows = int(round(b / 16384 / 1000.0)) * 1000
        if rows >= 50000 and len(c) >= 5:
            out[f"ip_scan_kernel<float,1>@{rows}"] = {"bytes_per_launch": round(b), "dispatches": len(c), "rows": rows,
                                                      "note": "algorithmic = rows x 16384 B (no write traffic to speak of: 8 B per row)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64)
