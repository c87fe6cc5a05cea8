"""GPU parity: omni_index_* (HIP) vs the exact-IP oracle (faiss::IndexFlatIP semantics, loop_detector.cpp:166,213).
Bar: ids bit-exact, scores within 1e-5 relative (fp32 dot products in a different association order)."""
import os

import numpy as np
import pytest

from oracle import match_ref as M
from omni_swarm_amd import synth          # seeded synthetic inputs (data generators; shared by bench.py)

pytestmark = pytest.mark.gpu
DIM = 4096


def _check(idx, db, q, k, rtol=1e-5):
    """ids bit-exact wherever the oracle's neighbouring scores are separated by more than the fp32 summation noise
    (2e-6); inside such a near-tie the two ids may swap (the GPU and the scalar oracle add 4096 products in different
    orders), which is the only deviation tolerated."""
    D, I = idx.search(q, k)
    Dr, Ir = M.ip_search(db, q, k)
    valid = Ir >= 0
    assert np.array_equal(I < 0, Ir < 0)
    assert np.allclose(D[valid], Dr[valid], rtol=rtol, atol=2e-6)
    assert (D[~valid] < -1e38).all()
    for qi in range(I.shape[0]):
        bad = np.nonzero(I[qi] != Ir[qi])[0]
        if len(bad) == 0:
            continue
        score = {int(i): float(s) for i, s in zip(Ir[qi], Dr[qi])}
        for pos in bad:
            near = [abs(float(Dr[qi, pos]) - float(Dr[qi, p2])) for p2 in (pos - 1, pos + 1) if 0 <= p2 < k and Ir[qi, p2] >= 0]
            assert near and min(near) < 2e-6, (qi, pos, I[qi, pos], Ir[qi, pos])
            gid = int(I[qi, pos])
            assert gid in score or pos >= k - 2            # the swapped-in id is in the oracle's list (or at the cut)
            if gid in score:
                assert abs(score[gid] - float(Dr[qi, pos])) < 2e-6


def test_golden_db(omni, ctx, golden):
    g = golden("match.npz")
    db = synth.global_db(3000, seed=3)
    q, _ = synth.queries_from_db(db, 8, seed=4)
    idx = omni.capi.IndexFlatIP(ctx, DIM)
    idx.add(db)
    assert idx.ntotal == 3000
    D, I = idx.search(q, 10)
    assert np.array_equal(I, g["ip_I"]) and np.allclose(D, g["ip_D"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n", [1, 5, 2047, 2048, 2049, 6000])
def test_sizes_and_k(omni, ctx, n):
    rng = np.random.default_rng(n)
    db = rng.standard_normal((n, DIM)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[rng.integers(0, n, 3)] + 0.01 * rng.standard_normal((3, DIM)).astype(np.float32)
    idx = omni.capi.IndexFlatIP(ctx, DIM)
    for s in range(0, n, 1500):                       # incremental add, like LoopDetector (1 row at a time there)
        idx.add(db[s:s + 1500])
    assert idx.ntotal == n
    for k in (1, 6, 10, 15):
        _check(idx, db, q, k)
    _check(idx, db, q[:1], 1000)                       # the reference's hard cap (loop_detector.cpp:200)


def test_empty_and_padding(omni, ctx):
    idx = omni.capi.IndexFlatIP(ctx, DIM)
    q = np.ones((2, DIM), np.float32)
    D, I = idx.search(q, 6)
    assert (I == -1).all() and (D < -1e38).all()
    db = synth.global_db(3, seed=1)
    idx.add(db)
    _check(idx, db, q, 6)


def test_ties_go_to_lower_row_and_row_by_row_add(omni, ctx):
    db = synth.global_db(300, seed=2)
    db[150] = db[20]
    db[299] = db[20]
    idx = omni.capi.IndexFlatIP(ctx, DIM)
    for r in db:                                       # IndexFlatIP::add(1, x), loop_detector.cpp:166
        idx.add(r)
    D, I = idx.search(db[20], 5)
    assert I[0, :3].tolist() == [20, 150, 299]
    _check(idx, db, db[[20, 77]], 10)
    idx.reset()
    assert idx.ntotal == 0


def test_massive_ties_and_the_small_k_boundary(omni, ctx):
    """k <= 64 takes the radix-select top-k: thousands of rows tied at the cut-off overflow its gather buffer and fall back to the
    in-place sort; ids must still come out in ascending order.  k = 64 / 65 straddle the two kernels."""
    db = synth.global_db(6000, seed=12)
    db[1000:5000] = db[1000]                           # 4000 identical rows: two whole 2048-key chunks of ties
    idx = omni.capi.IndexFlatIP(ctx, DIM)
    idx.add(db)
    D, I = idx.search(db[1000], 10)
    assert I[0].tolist() == list(range(1000, 1010)) and np.allclose(D[0], D[0, 0])
    for k in (64, 65):
        _check(idx, db, db[[7, 5500]], k)


@pytest.mark.parametrize("nq", [1, 2, 7, 8, 9, 17])
def test_query_batches(omni, ctx, nq):
    db = synth.global_db(4500, seed=6)
    q, _ = synth.queries_from_db(db, nq, seed=7)
    idx = omni.capi.IndexFlatIP(ctx, DIM)
    idx.add(db)
    _check(idx, db, q, 10)


def test_fp16_storage(omni, ctx):
    db = synth.global_db(5000, seed=8)
    q, rows = synth.queries_from_db(db, 6, seed=9)
    idx = omni.capi.IndexFlatIP(ctx, DIM, omni.capi.STORE_F16)
    idx.add(db)
    db16 = db.astype(np.float16).astype(np.float32)    # the oracle sees exactly what the shard stores
    D, I = idx.search(q, 10)
    Dr, Ir = M.ip_search(db16, q, 10)
    assert np.array_equal(I[:, 0], rows)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-5, atol=1e-6)


def test_fp16_row_by_row_add_across_reallocation(omni, ctx):
    """fp16 shards live in 16-row blocks (k-chunks interleaved over the rows of a block): rows appended one at a time, the way
    LoopDetector adds them (loop_detector.cpp:166), across two capacity doublings and a partially filled last block."""
    db = synth.global_db(2500, seed=17)
    idx = omni.capi.IndexFlatIP(ctx, DIM, omni.capi.STORE_F16)            # default capacity 1024 rows
    db16 = db.astype(np.float16).astype(np.float32)
    for i, r in enumerate(db):
        idx.add(r)
        if i in (0, 14, 15, 16, 1023, 1024, 2047, 2048):                   # block and capacity boundaries
            q = db[[i, i // 2]]
            _check(idx, db16[:i + 1], q, min(6, i + 1))
    assert idx.ntotal == 2500
    q, rows = synth.queries_from_db(db, 5, seed=18)
    _check(idx, db16, q, 15)                                               # 5 queries: matrix-core kernel
    _check(idx, db16, q[:2], 15)                                           # 2 queries: VALU kernel on the same blocks


def test_64_concurrent_queries_fp16_storage(omni, ctx):
    """BASELINE config 5 shape in miniature: 64 queries (one per concurrent key frame) against an fp16 shard in one call."""
    db = synth.global_db(9000, seed=13)
    q, rows = synth.queries_from_db(db, 64, seed=14)
    idx = omni.capi.IndexFlatIP(ctx, DIM, omni.capi.STORE_F16)
    idx.add(db)
    D, I = idx.search(q, 15)
    Dr, Ir = M.ip_search(db.astype(np.float16).astype(np.float32), q, 15)
    assert np.array_equal(I[:, 0], rows) and np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,nq", [(5, 9), (16, 12), (1000, 33), (4099, 64), (9000, 65), (20011, 130)])
def test_batched_matrix_core_search_fp16(omni, ctx, n, nq):
    """Four or more queries against an fp16 shard run on the matrix cores (ip_scan_mq_kernel: queries split into fp16 hi + lo, so the
    scores stay fp32-class).  Ragged row counts (tiles of 16, passes of 512 rows), query counts around the 64-slot block, k from the
    reference's 5 + max_index up to the 1000 cap."""
    db = synth.global_db(n, seed=20 + n % 7)
    if n > 100:
        db[n // 2] = db[3]                                  # an exact tie across tiles: lower row first
    rng = np.random.default_rng(nq)
    rows = rng.integers(0, n, nq)
    q = db[rows] + 0.02 * rng.standard_normal((nq, DIM)).astype(np.float32)
    q[0] *= 37.5                                            # per-query power-of-two scaling: magnitudes far from 1 keep their accuracy
    q[1] *= 1e-3
    idx = omni.capi.IndexFlatIP(ctx, DIM, omni.capi.STORE_F16)
    idx.add(db)
    db16 = db.astype(np.float16).astype(np.float32)
    for k in (6, 15) if n < 9000 else (10,):
        _check(idx, db16, q, k)
    if n == 4099:
        _check(idx, db16, q[:9], 1000)


def test_batched_search_equals_the_per_query_path(omni, ctx, monkeypatch):
    """Same shard, same queries: the matrix-core kernel (OMNI_MQ_MIN=1 forces it even for one query) and the VALU kernel agree on
    ids and on scores to fp32 summation noise."""
    db = synth.global_db(7000, seed=31)
    q, rows = synth.queries_from_db(db, 20, seed=32)
    idx = omni.capi.IndexFlatIP(ctx, DIM, omni.capi.STORE_F16)
    idx.add(db)
    monkeypatch.setenv("OMNI_MQ_MIN", "0")
    Dv, Iv = idx.search(q, 12)
    monkeypatch.setenv("OMNI_MQ_MIN", "1")
    Dm, Im = idx.search(q, 12)
    D1, I1 = idx.search(q[:1], 12)
    assert np.array_equal(Iv, Im) and np.allclose(Dv, Dm, rtol=1e-5, atol=2e-6)
    assert np.array_equal(I1, Im[:1]) and np.array_equal(D1, Dm[:1])          # a query's result does not depend on its batch


def _block_sign(seed, j):
    return np.where(np.random.default_rng(seed + 1 + j).random(DIM) < 0.5, np.float32(-1), np.float32(1))


def _as_stored(x, f16_rows):
    """float32 values of the rows as the shard stores them (fp16 shards round to nearest even, as numpy's and torch's conversions do)"""
    if not f16_rows:
        return x
    import torch
    return torch.from_numpy(x).half().float().numpy()


def _derived_blocks(n, block_rows, seed, f16_rows, base):
    """n unit rows in blocks of block_rows, cheap to make at full size: block j = the seeded base block with its columns rolled by 37 j and a per-block
    sign pattern (still unit rows, pairwise ~independent); every 1 000th row of the second half is a near-duplicate (cos ~ 0.8) of a row of the base
    block, as synth.global_db plants them.  Yields (first_row, rows as the shard stores them)."""
    for j, s in enumerate(range(0, n, block_rows)):
        m = min(block_rows, n - s)
        blk = np.roll(base[:m], 37 * j, axis=1) * _block_sign(seed, j) if j else base[:m].copy()
        if s >= n // 2:
            cnt = len(range(0, m, 1000))
            noise = np.random.default_rng(seed + 1000 + j).standard_normal((cnt, DIM), dtype=np.float32)
            noise /= np.linalg.norm(noise, axis=1, keepdims=True)
            v = 0.8 * base[0:m:1000][:cnt] + 0.6 * noise
            blk[0:m:1000] = v / np.linalg.norm(v, axis=1, keepdims=True)
        yield s, _as_stored(blk, f16_rows)


def _full_size_against_the_oracle(omni, ctx, n, storage, block_rows, seed, k=10):
    """A full-size shard against the exact oracle (oracle.match_ref.ip_search_blocked: BLAS per block, float64 re-scoring, ties -> lower row): ids
    identical wherever the oracle's adjacent scores are further apart than 32 ulp (fp32) of the score -- the GPU and the oracle add 4096 products in
    different orders --, scores within 1e-5 relative; 1, 8 and 64 concurrent queries = the single-query scan, the micro-batch of the key-frame loop, and
    BASELINE configs[4]'s 64 concurrent key frames (fp32 rows: fp16 mirror + exact re-scoring + certificate, at the production row threshold; fp16
    rows: the matrix-core scan).  One pass over the rows: each block is appended to the shard and scored by the oracle, then dropped."""
    c = omni.capi
    f16 = storage == c.STORE_F16
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((block_rows, DIM), dtype=np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    n_blocks = -(-n // block_rows)
    qs = []
    for i in range(56):                                                  # queries near rows spread over all blocks (never a planted row: off % 1000 != 0)
        j, off = i % n_blocks, 1 + 7 * i
        row = np.roll(base[off], 37 * j) * _block_sign(seed, j) if j else base[off]
        qs.append(row + 0.02 * rng.standard_normal(DIM).astype(np.float32))
    qs += [base[0], base[1000]]                                          # rows with planted near-duplicates in every block of the second half
    qs += [rng.standard_normal(DIM).astype(np.float32) for _ in range(6)]            # and pure noise
    q = np.stack(qs).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[3] *= 41.0                                                         # magnitudes away from 1 (the batched kernels scale per query)
    q[5] *= 2e-3
    idx = c.IndexFlatIP(ctx, DIM, storage, capacity=n)

    def appended():
        for s, blk in _derived_blocks(n, block_rows, seed, f16, base):
            idx.add(blk)
            yield s, blk
    Dr, Ir, Sr = M.ip_search_blocked(appended(), q, k)
    assert idx.ntotal == n
    assert (Ir[:56, 0] == [(i % n_blocks) * block_rows + 1 + 7 * i for i in range(56)]).all()       # the oracle itself: every query's own row first
    served0, fall0 = idx.cert_stats()
    for nq in (1, 8, 64):
        D, I = idx.search(q[:nq], k)
        assert np.allclose(D, Dr[:nq], rtol=1e-5, atol=0), (nq, np.abs(D / Dr[:nq] - 1).max())
        for qi in range(nq):
            for pos in np.nonzero(I[qi] != Ir[qi])[0]:                   # only inside a near-tie of the oracle's own scores
                tol = 32 * np.spacing(np.float32(max(abs(Sr[qi, pos]), 2.0 ** -6)))
                near = [abs(Sr[qi, pos] - Sr[qi, p2]) for p2 in (pos - 1, pos + 1) if 0 <= p2 < k]
                assert min(near) < tol, (nq, qi, pos, I[qi].tolist(), Ir[qi].tolist(), Sr[qi].tolist())
        assert (np.diff(D, axis=1) <= 0).all() and all(len(set(r)) == k for r in I.tolist())
    served, fallbacks = idx.cert_stats()
    if not f16 and n >= 32768:                                           # fp32 rows, >= 4 queries, production threshold: answered through the mirror
        assert served - served0 == 8 + 64 and fallbacks - fall0 <= 4, (served, fallbacks)
    assert idx.last_scan_ms() > 0
    idx.close()


def test_full_size_100k_fp32_rows_equal_the_oracle(omni, ctx):
    """BASELINE configs[3]'s 100 000 rows (1.6 GB fp32, + the fp16 mirror) -- loop_detector.cpp:199-242's search at the size north_star names."""
    _full_size_against_the_oracle(omni, ctx, 100_000, omni.capi.STORE_F32, 25_000, seed=42)


def test_full_size_125k_fp16_rows_equal_the_oracle(omni, ctx):
    """BASELINE configs[4] per-GPU shard: 1 M key frames / 8 GPUs = 125 000 fp16 rows."""
    _full_size_against_the_oracle(omni, ctx, 125_000, omni.capi.STORE_F16, 25_000, seed=5)


def test_full_size_500k_fp16_rows_equal_the_oracle(omni, ctx):
    """The bench's c5_shard database: 500 000 fp16 rows (4 M rows / 8 GPUs), 4 GB."""
    _full_size_against_the_oracle(omni, ctx, 500_000, omni.capi.STORE_F16, 25_000, seed=6)


@pytest.mark.parametrize("dim", [512, 1024, 8192])
def test_other_dimensions_fp32_and_fp16(omni, ctx, dim):
    """The reference only ever uses 4096; the handle accepts any multiple of 512 up to 8192 (k-slices of 256, 16-row blocks)."""
    rng = np.random.default_rng(dim)
    db = rng.standard_normal((1500, dim)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[[3, 700, 1499, 8, 9]] + 0.05 * rng.standard_normal((5, dim)).astype(np.float32)
    for storage in (omni.capi.STORE_F32, omni.capi.STORE_F16):
        idx = omni.capi.IndexFlatIP(ctx, dim, storage)
        idx.add(db)
        ref = db if storage == omni.capi.STORE_F32 else db.astype(np.float16).astype(np.float32)
        for nq in (1, 5):                                      # 5 queries: matrix cores on the fp16 shard
            D, I = idx.search(q[:nq], 7)
            Dr, Ir = M.ip_search(ref, q[:nq], 7)
            assert np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-5, atol=2e-6)


def test_snapshot_roundtrip(omni, ctx, tmp_path):
    """omni_index_save / omni_index_load: a shard checkpoint restores rows, storage type and search results bit for bit."""
    db = synth.global_db(3000, seed=15)
    q, _ = synth.queries_from_db(db, 3, seed=16)
    for storage in (omni.capi.STORE_F32, omni.capi.STORE_F16):
        idx = omni.capi.IndexFlatIP(ctx, DIM, storage)
        idx.add(db)
        D, I = idx.search(q, 12)
        path = str(tmp_path / f"shard{storage}.omnx")
        idx.save(path)
        idx2 = omni.capi.IndexFlatIP(ctx, DIM, storage)
        idx2.load(path)
        assert idx2.ntotal == 3000
        D2, I2 = idx2.search(q, 12)
        assert np.array_equal(I, I2) and np.array_equal(D, D2)
        idx2.add(db[:5])                                              # still appendable after a restore
        assert idx2.ntotal == 3005
        with pytest.raises(omni.capi.OmniError):
            omni.capi.IndexFlatIP(ctx, DIM, 1 - storage).load(path)   # storage type mismatch is an error, not a reinterpretation
    with pytest.raises(omni.capi.OmniError):
        omni.capi.IndexFlatIP(ctx, DIM).load(str(tmp_path / "missing.omnx"))


def test_shard_ids_and_merge_equals_unsharded(omni, ctx):
    db = synth.global_db(4001, seed=10)
    db[3000] = db[11]
    q = np.stack([db[11], db[2500]])
    world, k = 4, 8
    Dl, Il = [], []
    for r in range(world):
        idx = omni.capi.IndexFlatIP(ctx, DIM)
        idx.set_shard(r, world)
        idx.add(db[r::world])
        D, I = idx.search(q, k)
        assert ((I % world == r) | (I < 0)).all()
        Dl.append(D); Il.append(I)
    D, I = omni.capi.topk_merge(np.stack(Dl), np.stack(Il), k)
    Dr, Ir = M.ip_search(db, q, k)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("storage,nq", [("f32", 8), ("f32", 3), ("f32", 40), ("f32", 64), ("f16", 8), ("f16", 2), ("f16", 40)])
def test_batch_prefix_search_equals_one_prefix_search_per_query(omni, ctx, storage, nq):
    """omni_index_search_batch_prefix_dev (one pass over the shard for a micro-batch of key frames, per-query row limits, queries gathered
    out of a row buffer) == nq separate omni_index_search_prefix_dev calls == the oracle on the truncated database -- including empty
    prefixes, limits beyond ntotal, and limits inside the last 16-row block of the fp16 layout."""
    c = omni.capi
    rng = np.random.default_rng(77)
    n, k = 5003, 10
    db = rng.standard_normal((n, DIM)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    idx = c.IndexFlatIP(ctx, DIM, c.STORE_F16 if storage == "f16" else c.STORE_F32)
    idx.add(db)
    ref_db = db.astype(np.float16).astype(np.float32) if storage == "f16" else db
    rows = rng.standard_normal((4 * nq, DIM)).astype(np.float32)
    rows[1::4] = db[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, DIM)).astype(np.float32)    # the queried direction: row 4j+1
    rows /= np.linalg.norm(rows, axis=1, keepdims=True)
    row_idx = [4 * j + 1 for j in range(nq)]
    limits = [int(x) for x in rng.integers(1, n, nq)]
    limits[0] = 0
    limits[-1] = n + 50
    if nq > 2:
        limits[1] = 4993          # inside a 16-row block
    rows_dev = ctx.to_device(rows)
    buf = ctx.alloc(nq * k * 12)
    idx.search_batch_prefix_dev(rows_dev, row_idx, k, limits, buf + nq * k * 8, buf)
    raw = ctx.from_device(buf, (nq * k * 12,), np.uint8)
    I = raw[:nq * k * 8].view(np.int64).reshape(nq, k)
    D = raw[nq * k * 8:].view(np.float32).reshape(nq, k)
    one = ctx.alloc(k * 12)
    for j in range(nq):
        lim = min(limits[j], n)
        if lim == 0:
            assert (I[j] == -1).all() and (D[j] < -1e38).all()
            continue
        idx.search_prefix_dev(1, rows_dev + row_idx[j] * DIM * 4, k, lim, one + k * 8, one)
        r1 = ctx.from_device(one, (k * 12,), np.uint8)
        if storage == "f32":                           # fp32 rows: batches go through the fp16 mirror + EXACT re-scoring: the single-query scan's bits
            assert np.array_equal(I[j], r1[:k * 8].view(np.int64)) and np.array_equal(D[j], r1[k * 8:].view(np.float32))
        elif nq < 4:                                   # same scan kernel family: same bits
            assert np.array_equal(I[j], r1[:k * 8].view(np.int64)) and np.allclose(D[j], r1[k * 8:].view(np.float32), rtol=1e-6, atol=1e-7)
        Dr, Ir = M.ip_search(ref_db[:lim], rows[row_idx[j]][None], k)
        assert np.array_equal(I[j], Ir[0]) and np.allclose(D[j], Dr[0], rtol=1e-5, atol=2e-6)
    if storage == "f32" and nq >= 4:
        served, fallbacks = idx.cert_stats()
        if os.environ.get("OMNI_INDEX_MIRROR_MIN_ROWS") == "0":
            assert served == nq and fallbacks <= 1, (served, fallbacks)    # random rows: the certificate holds (the empty prefix is trivially exact)
        else:
            assert served == 0                                             # production threshold: 5 003 rows stay on the exact many-query scan
    for p in (rows_dev, buf, one):
        ctx.free(p)


def test_batched_fp32_search_with_the_production_mirror_threshold():
    """The suite sets OMNI_INDEX_MIRROR_MIN_ROWS=0 so that its small databases go through the fp16 mirror + certificate; what ships answers a batch over
    fewer than 32 768 fp32 rows with the exact many-query scan (ip_scan_rows_kernel), and with OMNI_INDEX_MIRROR=0 every batch.  The same cases, in
    subprocesses with those settings (the library reads them once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({}, {"OMNI_INDEX_MIRROR": "0"}):
        env = {k: v for k, v in os.environ.items() if k != "OMNI_INDEX_MIRROR_MIN_ROWS"}
        env.update(extra, OMNI_TEST_PRODUCTION_DEFAULTS="1")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_index.py"), "-m", "gpu", "-q", "-x", "-k",
                            "batch_prefix_search and f32"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (extra, r.stdout[-2000:], r.stderr[-1000:])
        assert "4 passed" in r.stdout, r.stdout[-500:]


def test_fp32_batch_search_certificate_falls_back_to_the_exact_scan_on_near_ties(omni, ctx, tmp_path):
    """fp32 shard, batched search = fp16 mirror pass + exact re-scoring of the k + 24 best + a certificate that nothing else can enter the top k.
    A cluster of 120 rows within 1e-4 of each other defeats the certificate (more near-ties than candidates): those queries must be re-run by
    the exact scan and still return the oracle's ids -- ties resolved to the lower row id -- while queries elsewhere stay certified.  Also:
    truncate + re-add, and a snapshot load, keep the mirror in step with the fp32 rows."""
    c = omni.capi
    rng = np.random.default_rng(123)
    n, k = 3000, 10
    db = rng.standard_normal((n, DIM)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    centre = db[100].copy()
    cluster = rng.choice(np.arange(200, n), 120, replace=False)
    db[cluster] = centre + 1e-4 * rng.standard_normal((120, DIM)).astype(np.float32)      # score spread 1e-4: far above fp32 summation noise, below the fp16 bound
    db[cluster[:7]] = centre                                   # exact duplicates: equal scores, lower row id first
    idx = c.IndexFlatIP(ctx, DIM, c.STORE_F32)
    idx.add(db)
    q = np.stack([centre, db[5], centre * 0.5, db[2222], db[17], centre + 0.01 * db[3]]).astype(np.float32)

    def batch(index, limits):
        rows_dev = ctx.to_device(q)
        buf = ctx.alloc(len(q) * k * 12)
        index.search_batch_prefix_dev(rows_dev, None, k, limits, buf + len(q) * k * 8, buf)
        raw = ctx.from_device(buf, (len(q) * k * 12,), np.uint8)
        ctx.free(rows_dev); ctx.free(buf)
        return raw[len(q) * k * 8:].view(np.float32).reshape(len(q), k), raw[:len(q) * k * 8].view(np.int64).reshape(len(q), k)

    D, I = batch(idx, [n] * len(q))
    Dr, Ir = M.ip_search(db, q, k)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-5, atol=2e-6)
    served, fallbacks = idx.cert_stats()
    assert served == len(q) and 3 <= fallbacks <= 4, (served, fallbacks)     # the three queries on the cluster (the 4th sits 0.01 away)
    D1, I1 = idx.search(q[:1], k)                                            # the single-query scan: same ids, same score bits
    assert np.array_equal(I1[0], I[0]) and np.array_equal(D1[0], D[0])
    # truncate, append other rows, search again: the mirror follows
    idx.truncate(1500)
    extra = rng.standard_normal((40, DIM)).astype(np.float32)
    extra /= np.linalg.norm(extra, axis=1, keepdims=True)
    idx.add(extra)
    db2 = np.concatenate([db[:1500], extra])
    q[1] = extra[7]
    D, I = batch(idx, [len(db2)] * len(q))
    Dr, Ir = M.ip_search(db2, q, k)
    assert np.array_equal(I, Ir) and np.allclose(D, Dr, rtol=1e-5, atol=2e-6) and I[1, 0] == 1507
    path = str(tmp_path / "m.omnx")
    idx.save(path)
    idx2 = c.IndexFlatIP(ctx, DIM, c.STORE_F32)
    idx2.load(path)
    D2, I2 = batch(idx2, [len(db2)] * len(q))
    assert np.array_equal(I2, I) and np.array_equal(D2, D)
    idx.close(); idx2.close()


def test_truncate_and_corrupt_snapshot(omni, ctx, tmp_path):
    c = omni.capi
    rng = np.random.default_rng(5)
    db = rng.standard_normal((300, DIM)).astype(np.float32)
    idx = c.IndexFlatIP(ctx, DIM)
    idx.add(db)
    idx.truncate(200)
    assert idx.ntotal == 200
    D, I = idx.search(db[250][None], 3)
    assert (I < 200).all()
    idx.add(db[200:210] * 2)
    D, I = idx.search(db[205][None] , 1)
    assert I[0, 0] == 205
    with pytest.raises(c.OmniError):
        idx.truncate(1000)
    path = str(tmp_path / "snap.omnx")
    idx.save(path)
    raw = bytearray(open(path, "rb").read())
    # header claims far more rows than the file holds: rejected BEFORE anything is allocated, handle untouched
    bad = bytearray(raw)
    bad[16:24] = (1 << 40).to_bytes(8, "little")
    open(path + ".bad", "wb").write(bad)
    with pytest.raises(c.OmniError):
        idx.load(path + ".bad")
    open(path + ".short", "wb").write(raw[:len(raw) // 2])
    with pytest.raises(c.OmniError):
        idx.load(path + ".short")
    assert idx.ntotal == 210
    D2, I2 = idx.search(db[205][None], 1)
    assert I2[0, 0] == 205
    idx.load(path)
    assert idx.ntotal == 210
