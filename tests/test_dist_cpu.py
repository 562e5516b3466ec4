"""gloo tests on the CPU (world 2, and world 8 = the rank count configs[3] / configs[4] name) of the N>1 host logic: stream sharding, the BA point partition and
the all-reduce of the reduced normal equations.  The compute under the collective is
the oracle's reduced-system builder; the partition / reduce code is the product's (dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import orb_slam3_rust_amd as P
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1) frame sharding: disjoint, complete
        mine = P.dist.shard_streams(8, rank, world)
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        assert sorted(sum(allv, [])) == list(range(8))
        # 2) BA: partial reduced systems of the point partition, summed over ranks == full system
        # (world 8: 203 points = 25 or 26 per rank — not a multiple of the Schur product's 16-point k-split tiles, nor equal across ranks)
        w = P.synth.ba_window(11, 6, 120 if world == 2 else 203, P.BA_OBS, n_fixed_extra=1)
        cam = O.Camera(**w["camera"]); cfg = O.ba_config()
        pp = np.concatenate([O.se3_to_params(p) for p in w["poses_cw"]])
        local = P.dist.partition_observations(w["obs"], rank, world)
        owners = P.dist.point_owner(local["mp_idx"], world)
        assert np.all(owners == rank) and 0 < len(local) < len(w["obs"])
        U, gp, S, b, chi2 = O.ba_reduced_system(cam, cfg, 1e-3, pp, w["fixed_cw"], w["points"], local)
        buf = torch.from_numpy(np.concatenate([S.ravel(), U.ravel(), gp, b, [chi2]]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        Uf, gpf, Sf, bf, chif = O.ba_reduced_system(cam, cfg, 1e-3, pp, w["fixed_cw"], w["points"], w["obs"])
        full = np.concatenate([Sf.ravel(), Uf.ravel(), gpf, bf, [chif]])
        assert np.allclose(buf.numpy(), full, rtol=1e-11, atol=1e-8)
        # 3) solving the reduced system from the summed buffer gives the same pose step on every rank
        n = 6 * len(w["poses_cw"])
        Ssum = buf.numpy()[:n * n].reshape(n, n)
        Usum = buf.numpy()[n * n:n * n + 36 * (n // 6)].reshape(-1, 6, 6)
        H = -Ssum.copy()
        for k in range(n // 6):
            Ud = Usum[k].copy()
            Ud[np.diag_indices(6)] += 1e-3 * np.maximum(np.diag(Ud), 1e-6)
            H[6 * k:6 * k + 6, 6 * k:6 * k + 6] += Ud
        rhs = -buf.numpy()[n * n + 36 * (n // 6):n * n + 36 * (n // 6) + n] + buf.numpy()[n * n + 36 * (n // 6) + n:n * n + 36 * (n // 6) + 2 * n]
        dp = np.linalg.solve(H, rhs)
        gathered = [None] * world
        dist.all_gather_object(gathered, dp.tolist())
        assert all(np.array_equal(np.array(gathered[0]), np.array(g)) for g in gathered[1:])
        # ... and the partition covers every observation exactly once, every rank holding ALL observations of the points it owns
        counts = [None] * world
        dist.all_gather_object(counts, (len(local), sorted(set(local["mp_idx"].tolist()))))
        assert sum(c[0] for c in counts) == len(w["obs"])
        owned = [set(c[1]) for c in counts]
        assert all(not (owned[a] & owned[b]) for a in range(world) for b in range(a + 1, world))
        assert set().union(*owned) == set(w["obs"]["mp_idx"].tolist())
        # 4) the bench's max-over-ranks timing helper
        assert P.dist.allreduce_max_seconds(1.0 + rank) == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_partition_and_reduce(world):
    """world 2, and world 8 = the rank count configs[3] / configs[4] name (gloo on the CPU: the host side of the partitioned solve)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


def test_bench_line_names_the_communicator_and_both_ba_shardings():
    """bench.py --gpus N (N > 1): `local_ba.transport` carries the rank count the library's own communicator reports (ncclCommCount), and the
    line holds both shardings of BA — one window partitioned over the ranks (lm_iters_per_s) and one window per rank (independent_windows).
    Parsed here from the source and its helper; the N = 2 rehearsal on one GPU runs the whole path (tests/test_properties_gpu.py)."""
    sys.path.insert(0, ROOT)
    import bench
    t = bench.ba_transport_text(8, True, (8, 3), False)
    assert "ncclCommCount" in t and "8 ranks" in t and "rank 3" in t
    assert "REHEARSAL" in bench.ba_transport_text(2, False, None, True) and "nccl = RCCL" in bench.ba_transport_text(8, False, None, False)
    assert bench.ba_transport_text(1, False, None, False) == "one GPU, no collective"
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["independent_windows"]' in src and "ba_solve_partitioned" in src and "h.rccl_world()" in src


def test_partition_properties():
    sys.path.insert(0, ROOT)
    import orb_slam3_rust_amd as P
    w = P.synth.ba_window(5, 8, 500, P.BA_OBS)
    parts = [P.dist.partition_observations(w["obs"], r, 4) for r in range(4)]
    assert sum(len(p) for p in parts) == len(w["obs"])
    for r, p in enumerate(parts):
        assert np.all(p["mp_idx"] % 4 == r)
    sizes = [len(p) for p in parts]
    assert max(sizes) < 1.3 * min(sizes)        # balanced
    assert P.dist.shard_streams(8, 3, 8) == [3] and P.dist.shard_streams(3, 1, 2) == [1]


def test_partitioned_solve_refuses_without_transport():
    """dist.ba_solve_partitioned with world > 1, no hook and no communicator on the handle raises before any solve is issued
    (host logic only: a stub handle stands in for the GPU one)."""
    import sys
    sys.path.insert(0, ROOT)
    import orb_slam3_rust_amd as P

    class Stub:
        calls = 0
        coll = 0
        def has_collective(self): return self.coll
        def ba_solve_visual(self, *a, **k): Stub.calls += 1; return "solved"
    obs = np.zeros(4, P.BA_OBS); obs["mp_idx"] = [0, 1, 2, 3]
    with pytest.raises(RuntimeError, match="no RCCL communicator"):
        P.dist.ba_solve_partitioned(Stub(), None, None, None, None, None, obs, 0, 2)
    assert Stub.calls == 0
    s = Stub(); s.coll = 1
    assert P.dist.ba_solve_partitioned(s, None, None, None, None, None, obs, 1, 2) == "solved" and Stub.calls == 1
    assert P.dist.ba_solve_partitioned(Stub(), None, None, None, None, None, obs, 0, 1) == "solved"
