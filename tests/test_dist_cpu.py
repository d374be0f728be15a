"""CPU, world_size = 2 over gloo: the multi-GPU glue of the sampler (graal_amd/dist.py).

Each rank owns a contiguous shard of the contact list, computes the fixed-point (Q30, rounded per contact) sum of its
contacts' log-likelihood terms with the numpy re-score (the engine's arithmetic convention, oracle/sparse_numpy.py),
and ONE integer all-reduce yields the total: bit-identical to the unsharded sum, for any world size -- the property
that lets every rank draw the same move without a broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from graal_amd import dist as gdist
from graal_amd import synth
from oracle.sparse_numpy import SparseScorer

Q = float(1 << 30)


def _problem():
    par = synth.make_param_simu(fact=200.0, v_inter=0.02)
    return synth.make_problem(n_bins=80, nnz=1500, n_sub=3, seed=9, contig_weights=(5, 4, 3), mean_len_bp=2000.0, accu=9,
                              param=par)


def _q_terms(P, lo, hi):
    sc = SparseScorer(P["coo_row"], P["coo_col"], P["coo_val"], P["np_sub_frags_id"], P["np_sub_frags_len_bp"],
                      P["np_sub_frags_accu"], P["mean_squared_frags_per_bin"], P["param_simu"])
    state = {k: np.asarray(P["S_o_A_frags"][k], np.int32) for k in P["S_o_A_frags"]}
    centres = sc.centres(state)
    ex, _ = sc._ex(state, centres, sc.row[lo:hi], sc.col[lo:hi])
    return np.rint(sc.count[lo:hi] * np.log(ex.astype(np.float64)) * Q).astype(np.int64)


def _worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group = gdist.Group(rank, world)
        P = _problem()
        lo, hi = gdist.shard_range(len(P["coo_row"]), rank, world)
        local = int(_q_terms(P, lo, hi).sum())
        total = group.all_reduce_sum_int(local)
        # the per-candidate vector path: one SUM all-reduce of an int64 tensor (13 * K values)
        vec = torch.arange(65, dtype=torch.int64) * (rank + 1) + local
        group.all_reduce_sum_(vec)
        group.barrier()
        # the node-local exchange segment: every rank maps the same zero-filled host memory (the engine's GPUs write
        # their slots there; here the hosts do), and it needs no name in /dev/shm once everybody has it
        assert group.single_node()
        seg = group.shared_host_segment(2 * world * 1024)
        words = np.frombuffer(seg, dtype=np.int64)
        assert len(words) * 8 >= 2 * world * 1024 and not words.any()
        words[rank * 128:rank * 128 + 4] = [rank + 1, 10 * (rank + 1), local, group.all_reduce_max_int(rank + 5)]
        group.barrier()
        seen = [words[r * 128:r * 128 + 4].tolist() for r in range(world)]
        group.barrier()
        del words
        seg.close()
        leftovers = [f for f in os.listdir("/dev/shm") if f.startswith("graal_x_")]
        out_q.put((rank, lo, hi, local, total, vec.tolist(), seen, leftovers))
    finally:
        td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(180)
def test_sharded_q_sums_are_bit_identical_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    P = _problem()
    want = int(_q_terms(P, 0, len(P["coo_row"])).sum())
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == len(P["coo_row"])   # shards tile the list
    assert res[0][4] == res[1][4] == want                                               # every rank: exact total
    assert res[0][3] + res[1][3] == want
    locals_sum = res[0][3] + res[1][3]
    want_vec = [i * 3 + locals_sum for i in range(65)]
    assert res[0][5] == want_vec and res[1][5] == want_vec
    for r in res:   # every rank saw every rank's slot; MAX all-reduce; no file left behind
        assert r[6] == [[1, 10, res[0][3], world + 4], [2, 20, res[1][3], world + 4]]
        assert r[7] == []


def test_group_world1_needs_no_process_group():
    g = gdist.Group(0, 1)
    t = torch.arange(5, dtype=torch.int64)
    assert g.all_reduce_sum_(t).tolist() == [0, 1, 2, 3, 4]
    assert g.all_reduce_sum_int(7) == 7
    g.barrier()
    from graal_amd.lib import q_to_float
    assert np.array_equal(q_to_float(np.array([1 << 30, -(1 << 29)])), [1.0, -0.5])
    v = q_to_float(np.array([1 << 29, -(1 << 63), 0, 5]), np.array([-3000000000, 5, 7, 0]), np.array([0, 0, 0, 2]))   # coarse + fine / 2^30; the marker / a flag wins
    assert v[0] == -3000000000 + 0.5 and np.isnan(v[1]) and v[2] == 7.0 and np.isnan(v[3])


def test_env_world_defaults(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert gdist.env_world() == (0, 1, 0)
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("LOCAL_RANK", "3")
    assert gdist.env_world() == (3, 8, 3)


@pytest.mark.timeout(300)
def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus N` with no torchrun environment spawns the N ranks (torch.distributed.run, 127.0.0.1) itself,
    relays rank 0's JSON line and the exit code.  --dry-run: rendezvous + sharding only (no GPU in the CPU suite, and the
    engine has no CPU fallback); the GPU suite runs the same command for real (tests/test_multirank_gpu.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--nnz", "100000"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"dry_run": True, "n_gpus": 2, "contacts": 100000, "contacts_rank0": d["contacts_rank0"]}
    assert 0 < d["contacts_rank0"] < 100000
    # a launcher that started the wrong number of ranks is an error, not a silent single-rank run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "started 1 ranks" in (bad.stderr + bad.stdout)
