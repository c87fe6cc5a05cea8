"""The row-sharded database with the collective INSIDE the library (csrc/shard.hip: ncclAllGather on device buffers, RCCL resolved by
dlopen, no torch): results must equal the unsharded exact-IP oracle with the same add-before-query order (loop_detector.cpp:89-98).
  * world 1: a real RCCL communicator of one rank in this process (init, two all-gathers per exchange, merge);
  * world 2 / 4 / 8 on ONE GPU: RCCL itself refuses several ranks per device, so OMNI_RCCL_LIB points shard.hip's dlopen at tests/stub_rccl (an
    all-gather through a mapped file + hipMemcpy, TEST INFRASTRUCTURE): every line of omni_shard_step_batch_dev / omni_shard_search / the sharded
    key-frame pipeline runs with world > 1 -- global id numbering, owned-row pick, per-query prefix limits across ranks, the merge over W
    lists -- before an 8-GPU node ever sees them;
  * the real RCCL is covered with world 1 (above); it refuses two ranks on one device;
  * SELF-ACTIVATING on a box with >= 2 GPUs (skipped only when the box has one): the same exchange, the sharded pipeline and `bench.py --gpus N` with
    N = min(8, device count) ranks, one GPU each, over the REAL librccl (xGMI) -- the day an 8-GPU node runs this suite, RCCL with more than one
    rank is covered without anybody changing a line."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import match_ref as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def expected(world, seed):
    """Replays the worker's stream on an unsharded oracle index: per rank, the (D, I) of its queries."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((world * 40, 4096)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    db = [base]
    F, m, k = 3, 4, 10
    out = {r: ([], []) for r in range(world)}
    for it in range(4):
        rows_all = rng.standard_normal((world, F, m, 4096)).astype(np.float32)
        rows_all[:, :, 1] = base[rng.integers(0, len(base), (world, F))] + 0.3 * rows_all[:, :, 1]
        rows_all /= np.linalg.norm(rows_all, axis=-1, keepdims=True)
        for f in range(F):
            db.append(rows_all[:, f].reshape(world * m, 4096))          # step f: rank-major, exactly the global id order
            full = np.concatenate(db)
            for r in range(world):
                D, I = M.ip_search(full, rows_all[r, f, 1][None], k)
                out[r][0].append(D[0]); out[r][1].append(I[0])
    full = np.concatenate(db)
    Ds, Is = M.ip_search(full, base[[3, 17]] + 0.01, k)
    return out, Ds, Is, len(full)


STUB = os.path.join(ROOT, "tests", "stub_rccl", "libstub_rccl.so")


def stub_env():
    if not os.path.exists(STUB):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(STUB)])
    return dict(os.environ, OMNI_RCCL_LIB=STUB)


def run_world(world, tmp_path, seed, worker="shard_rccl_worker.py", env=None, one_gpu_per_rank=False):
    id_file = str(tmp_path / f"uid{world}{'m' if one_gpu_per_rank else ''}")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"w{world}_r{r}{'m' if one_gpu_per_rank else ''}.npz")
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker), str(r), str(world), str(r if one_gpu_per_rank else 0), id_file, out, str(seed)],
                                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env), out))
    logs = []
    for p, _ in procs:
        try:
            logs.append(p.communicate(timeout=400)[0])
        except subprocess.TimeoutExpired:
            for q, _ in procs:
                q.kill()
            raise
    return [(p.returncode, out) for p, out in procs], logs


def check_against_oracle(world, res, logs):
    assert all(rc == 0 for rc, _ in res), "\n".join(logs)[-3000:]
    exp, Ds, Is, ntotal = expected(world, 100 + world)
    for r, (_, out) in enumerate(res):
        z = np.load(out)
        assert int(z["ntotal"]) == ntotal
        assert np.array_equal(z["I"], np.stack(exp[r][1])), r
        assert np.allclose(z["D"], np.stack(exp[r][0]), rtol=1e-5, atol=2e-6)
        assert np.array_equal(z["Is"], Is) and np.allclose(z["Ds"], Ds, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rank_exchange_on_one_gpu_equals_unsharded_oracle(world, tmp_path):
    """omni_shard_* with world ranks as world processes on GPU 0 (stub collective): ids, scores, add-before-query order, ntotal == the oracle."""
    res, logs = run_world(world, tmp_path, seed=100 + world, env=stub_env())
    check_against_oracle(world, res, logs)


def check_pipeline_against_oracle(world, seed, res, logs):
    MB, k, mid, thres = 2, 10, 5, 0.3
    assert all(rc == 0 for rc, _ in res), "\n".join(logs)[-3000:]
    z = [np.load(out) for _, out in res]
    rng = np.random.default_rng(seed)
    db = rng.standard_normal((world * 40, 4096)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    assert all(np.allclose(z[0]["g"], zr["g"], atol=1e-6) for zr in z[1:])    # the same blocks on every rank
    g = z[0]["g"].reshape(2, MB, 4, 4096)                                     # [block][frame][direction]
    rows = [db]
    n_units = len(z[0]["hits"])
    exp = np.zeros((world, n_units), int)
    for u in range(n_units):
        base = sum(len(x) for x in rows)
        unit = [g[(r + u) % 2] for r in range(world)]                         # rank r runs block (r + u) % 2 in unit u
        for f in range(MB):
            for r in range(world):
                rows.append(unit[r][f])
        full = np.concatenate(rows)
        for f in range(MB):
            nt = base + (f + 1) * world * 4
            for r in range(world):
                D, I = M.ip_search(full[:nt], unit[r][f][1][None], k)
                exp[r, u] += int(any(i >= 0 and i <= nt - mid and d > thres for d, i in zip(D[0], I[0])))
    for r in range(world):
        assert int(z[r]["rows_total"]) == len(db) + n_units * MB * world * 4
        assert np.array_equal(z[r]["hits"], exp[r]), (r, z[r]["hits"], exp[r])
    assert exp[:, 1:].sum() >= world * MB * (n_units - 1)                      # from the second unit on every key frame revisits an earlier one
    return z


def test_sharded_pipeline_two_ranks_equals_unsharded_oracle(tmp_path):
    """The sharded C++ key-frame pipeline, two ranks on GPU 0 (stub collective), toy images: every rank's loop candidates per exchange unit equal
    the reference's rule (loop_detector.cpp:232: recency + threshold on the row id) applied to the unsharded database in global insertion order
    (exchange unit -> step -> rank -> direction)."""
    world, seed = 2, 77
    res, logs = run_world(world, tmp_path, seed, worker="shard_pipeline_worker.py", env=stub_env())
    check_pipeline_against_oracle(world, seed, res, logs)


def _gpus():
    import torch
    return torch.cuda.device_count()


def _real_rccl_env():
    env = {k: v for k, v in os.environ.items() if k != "OMNI_RCCL_LIB"}         # the library's own search: the librccl next to the HIP runtime
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"                                     # this driver: dmabuf IPC only
    return env


def test_real_rccl_one_gpu_per_rank_exchange_equals_unsharded_oracle(tmp_path):
    """min(8, device count) ranks, ONE GPU EACH, the real librccl: omni_shard_* == the unsharded oracle.  Skipped only on a one-GPU box."""
    n = _gpus()
    if n < 2:
        pytest.skip(f"{n} GPU on this box: RCCL refuses several ranks per device (the stub cases above cover the multi-rank code on one GPU)")
    world = min(8, n)
    res, logs = run_world(world, tmp_path, seed=100 + world, env=_real_rccl_env(), one_gpu_per_rank=True)
    check_against_oracle(world, res, logs)
    libs = {str(np.load(out)["librccl"]) for _, out in res}
    assert len(libs) == 1 and "stub" not in next(iter(libs)) and "rccl" in next(iter(libs)), libs


def test_real_rccl_one_gpu_per_rank_sharded_pipeline_equals_unsharded_oracle(tmp_path):
    """The sharded C++ key-frame pipeline on min(8, device count) GPUs over the real librccl == the reference's rule on the unsharded database."""
    n = _gpus()
    if n < 2:
        pytest.skip(f"{n} GPU on this box")
    world = min(8, n)
    res, logs = run_world(world, tmp_path, 77, worker="shard_pipeline_worker.py", env=_real_rccl_env(), one_gpu_per_rank=True)
    z = check_pipeline_against_oracle(world, 77, res, logs)
    assert all("stub" not in str(zr["librccl"]) for zr in z)


def test_real_rccl_bench_line_on_every_gpu_of_the_box(tmp_path):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one rank per GPU), N = min(8, device count): the line says the exchange ran in
    libomni_hip.so over the real librccl on N ranks, with per-rank rates and the all-gathers' device time."""
    import json
    n = _gpus()
    if n < 2:
        pytest.skip(f"{n} GPU on this box")
    world = min(8, n)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "16", "--warmup", "8", "--min-time", "0", "--match-db-rows", "80000",
           "--batched-rows", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_real_rccl_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["rccl_ranks"] == world and "stub" not in line["librccl"] and line["torch_distributed_backend"] == "gloo"
    assert len(line["per_rank_keyframes_per_s"]) == world and min(line["per_rank_keyframes_per_s"]) > 0
    assert line["all_gather_us_p50"]["exchange_units"] > 0 and len(line["all_gather_us_p50"]["topk_lists"]) == world
    assert line["db_rows_per_gpu"] > 0 and line["value"] > 0
    assert line["value"] <= sum(line["per_rank_keyframes_per_s"]) * 1.001             # the job's rate = all key frames / the slowest rank's time


def test_sharded_exchange_with_the_real_rccl_equals_unsharded_oracle(tmp_path):
    """A real RCCL communicator (one rank: librccl's ncclCommInitRank / ncclAllGather on device buffers).  Two ranks on one device are refused
    by RCCL itself (ncclCommInitRank: invalid usage -- measured in round 2), which is what the stub cases above are for."""
    res, logs = run_world(1, tmp_path, seed=101)
    check_against_oracle(1, res, logs)
