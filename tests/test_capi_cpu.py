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
    inline = set(re.findall(r"static inline \w+ (omni_[a-z0-9_]+)\s*\(", hdr))          # helpers defined in the header itself: nothing to export
    assert inline == {"omni_fisheye_mask_rows"}
    declared = set(re.findall(r"\b(omni_[a-z0-9_]+)\s*\(", hdr)) - inline
    assert declared == set(c.SYMBOLS), (declared ^ set(c.SYMBOLS))
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/omni_hip.h but not exported"
    assert c.lib().omni_abi_version() == c.ABI_VERSION == 2


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


def test_vlad_split_fp16_weight_blob_layout_and_exactness(omni):
    """omni_vlad_pack_block (host only): the blob vlad_sblock_kernel reads.  Decoded here with the documented layout: every weight of the two
    pointwise convolutions comes back as hi + lo within 2^-21 relative, the expand bias sits in the two constant K slots, padded rows / K slots
    are zero, the depthwise taps and bias are the fp32 values themselves; unsupported shapes report -1."""
    c = omni.capi
    rng = np.random.default_rng(5)
    assert c.vlad_pack_block(8, 50, 8, 1, *(np.zeros(1, np.float32),) * 5) is None            # hidden width not a multiple of 48
    for cin, hid, cout, stride in ((8, 48, 8, 2), (24, 144, 32, 1), (56, 336, 112, 1)):
        we, be = rng.standard_normal((hid, cin)).astype(np.float32), rng.standard_normal(hid).astype(np.float32)
        wd, bd = rng.standard_normal((hid, 9)).astype(np.float32), rng.standard_normal(hid).astype(np.float32)
        wp = rng.standard_normal((cout, hid)).astype(np.float32)
        blob = c.vlad_pack_block(cin, hid, cout, stride, we, be, wd, bd, wp)
        n_chunks, S1, S2, NT = hid // 48, (2 * cin + 2 + 15) // 16, (cin + 15) // 16, {1: 1, 2: 2, 3: 4, 4: 4}[(cout + 31) // 32]
        CB = (S1 + S2) * 2048 + NT * 6144
        assert blob.size == n_chunks * (480 * 4 + CB)
        taps = blob[:n_chunks * 1920].view(np.float32).reshape(n_chunks, 10, 48)
        assert np.array_equal(taps[:, :9].transpose(0, 2, 1).reshape(hid, 9), wd) and np.array_equal(taps[:, 9].reshape(hid), bd)
        frag = blob[n_chunks * 1920:].reshape(n_chunks, CB)

        def a_frag(ch, off):                      # [64 lanes][8] halfs -> [32 rows][16 k]: row = lane & 31, k = 8 (lane >> 5) + e
            f = frag[ch, off:off + 1024].view(np.float16).astype(np.float64).reshape(2, 32, 8)
            return np.concatenate([f[0], f[1]], axis=1)

        for ch in range(n_chunks):
            K1 = np.concatenate([np.concatenate([a_frag(ch, (ks * 2 + m) * 1024) for m in range(2)], axis=0) for ks in range(S1)], axis=1)      # [64][S1*16]
            K2 = np.concatenate([np.concatenate([a_frag(ch, ((S1 + ks) * 2 + m) * 1024) for m in range(2)], axis=0) for ks in range(S2)], axis=1)
            W = we[ch * 48:(ch + 1) * 48].astype(np.float64)
            assert np.array_equal(K1[:48, :cin], K1[:48, cin:2 * cin])                                    # We_hi twice (x_hi and x_lo slots)
            rec = K1[:48, :cin] + K2[:48, :cin]
            assert np.abs(rec - W).max() <= 2.0 ** -21 * np.abs(W).max()
            b = be[ch * 48:(ch + 1) * 48].astype(np.float64)
            assert np.abs(K1[:48, 2 * cin] + K1[:48, 2 * cin + 1] - b).max() <= 2.0 ** -21 * np.abs(b).max()
            assert not K1[48:].any() and not K2[48:].any() and not K1[:, 2 * cin + 2:].any() and not K2[:, cin:].any()
            base = (S1 + S2) * 2048
            for m in range(NT):
                hi = np.concatenate([a_frag(ch, base + ((m * 3 + ks) * 2 + 0) * 1024) for ks in range(3)], axis=1)          # [32 cout][48 hidden]
                lo = np.concatenate([a_frag(ch, base + ((m * 3 + ks) * 2 + 1) * 1024) for ks in range(3)], axis=1)
                rows = min(32, max(0, cout - 32 * m))
                ref = wp[32 * m:32 * m + rows, ch * 48:(ch + 1) * 48].astype(np.float64)
                if rows:
                    assert np.abs(hi[:rows] + lo[:rows] - ref).max() <= 2.0 ** -21 * np.abs(ref).max()
                assert not hi[rows:].any() and not lo[rows:].any()

def test_header_is_plain_c_and_links_from_a_c_program(omni, tmp_path):
    """The boundary is a C ABI: include/omni_hip.h compiles as C99 (-pedantic: no C++ in it), and a C program built with gcc alone links against
    libomni_hip.so and reads its ABI version -- what a cgo / JNI / ctypes binding relies on (no GPU needed: no compute call)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "omni_hip.h"\n'
                   'int main(void) {\n'
                   '    int v = 0;\n'
                   '    if (omni_abi_version() != OMNI_ABI_VERSION) return 1;\n'
                   '    if (omni_config_value("OMNI_SP_SPARSE_DA", &v) != OMNI_OK || v != 1) return 2;\n'
                   '    if (omni_ctx_mfma_ceiling(NULL, 1.0f, NULL, NULL) == OMNI_OK) return 3;      /* argument errors are codes, not aborts */\n'
                   '    printf("abi %d: %s\\n", omni_abi_version(), omni_last_error());\n'
                   '    return 0;\n}\n')
    libdir = os.path.join(ROOT, "omni-swarm_amd", "lib")
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lomni_hip",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("OMNI_")]:
        env.pop(k)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("abi 2:"), (r.returncode, r.stdout, r.stderr)

def test_host_library_exports_what_its_c_header_declares(tmp_path):
    """include/omni_host.h declares the C entry points of libomni_host.so (the C++ key-frame loop behind plain C): valid C99, every declaration exported,
    nothing exported that is not declared (host_capi.cpp includes the header, so a signature mismatch is a compile error), and the Python bindings
    (pipeline.py, flatten.py) use exactly this set."""
    import re
    import subprocess
    hdr_path = os.path.join(ROOT, "include", "omni_host.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = re.sub(r"/\*.*?\*/", "", open(hdr_path).read(), flags=re.S)                       # declarations only
    declared = set(re.findall(r"\b(omni_[a-z0-9_]+)\s*\(", text))
    lib = os.path.join(ROOT, "omni-swarm_amd", "lib", "libomni_host.so")
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if l.split() and l.split()[-1].startswith("omni_") and " T " in l}
    assert declared == exported, (sorted(declared - exported), sorted(exported - declared))
    src = open(os.path.join(ROOT, "omni-swarm_amd", "pipeline.py")).read()
    bound = set(re.findall(r'"(omni_[a-z0-9_]+)"', re.search(r"SYMBOLS\s*=\s*\[(.*?)\]", src, re.S).group(1))) | {"omni_fisheye_maps"}      # (flatten.py binds the last one)
    assert bound == declared, (sorted(bound ^ declared))
    assert len(declared) == 28
