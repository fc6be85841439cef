"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups run the same protocol
bench.py runs over RCCL -- every rank samples its shard into its slot of the gather buffer,
ONE all_gather_into_tensor, unpack to reference node order -- with the shard compute done by
the CPU wave emulator (no GPU here).  Shard geometry comes from the product's C ABI
(dg_shard_layout), so the bookkeeping under test is the product's own."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dgtest as T


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, res, ret, pieces=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import emu
        import discregrid_amd as dg
        dg.load_library()
        V, F = T.torus()
        dom = T.oracle_default_domain(V)
        grid = dg.grid_desc(dom[:3], dom[3:], res)
        # bench.py's protocol: piece p of this rank is the shard of virtual rank p*world + rank of a
        # (pieces*world)-way deal; piece p is all-gathered (async) while piece p+1 is sampled
        vworld = pieces * world
        mesh = emu.EmuMesh(V, F)
        count = 0
        stride = dg.shard_layout(grid, rank, vworld)[1]              # product bookkeeping (host only)
        gathered = torch.full((vworld * stride,), float("nan"), dtype=torch.float64)
        works, keep = [], []
        for p in range(pieces):
            v = p * world + rank
            c, st = dg.shard_layout(grid, v, vworld)
            assert st == stride and c == emu.shard_count(res, v, vworld)
            mine = torch.zeros(stride, dtype=torch.float64)
            mine[:c] = torch.from_numpy(mesh.sample_shard(dom, res, v, vworld))
            keep.append(mine)
            works.append(dist.all_gather_into_tensor(gathered[p * world * stride:(p + 1) * world * stride], mine,
                                                     async_op=True))
            count += c
        # ... and piece p is unpacked as soon as its gather is done (dg_unpack_shard_range_device's map)
        field = np.full(T.n_nodes(res), np.nan)
        for p, w in enumerate(works):
            w.wait()
            emu.unpack_ranks(res, vworld, gathered.numpy(), stride, p * world, (p + 1) * world, field)
        assert np.array_equal(field, emu.unpack(res, vworld, gathered.numpy(), stride))
        want = T.OracleMesh(V, F).sample_nodes(dom, res)
        ok = bool(np.array_equal(field, want))
        # max-over-ranks reduction as bench.py does for the timing
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ret[rank] = (ok, float(t.item()), count)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,res,pieces", [(2, [10, 7, 13], 1), (3, [6, 9, 5], 1), (2, [9, 17, 21], 4),
                                              (2, [5, 4, 6], 3)])
def test_shard_allgather_unpack_gloo(world, res, pieces):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), res, ret, pieces), nprocs=world, join=True)
    assert len(ret) == world
    assert all(v[0] for v in ret.values())
    assert all(v[1] == float(world) for v in ret.values())
    assert sum(v[2] for v in ret.values()) == T.n_nodes(res)


def test_bench_grid_growth():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(T.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.grid_for(1) == [256, 256, 256]
    assert bench.grid_for(2) == [256, 256, 512]
    assert bench.grid_for(4) == [256, 512, 512]
    assert bench.grid_for(8) == [512, 512, 512]
    # weak scaling: nodes per GPU stay within 1 % of the single-GPU count
    n1 = T.n_nodes(bench.grid_for(1))
    for n in (2, 4, 8):
        assert abs(T.n_nodes(bench.grid_for(n)) / n / n1 - 1) < 0.01


def test_shard_balance_at_scale():
    """Round-robin 4-plane slabs: node counts per rank within 3 % at the judged sizes."""
    import discregrid_amd as dg
    dg.load_library()
    for world, res in ((2, [256, 256, 512]), (4, [256, 512, 512]), (8, [512, 512, 512])):
        g = dg.grid_desc([0] * 3, [1] * 3, res)
        counts = [dg.shard_layout(g, r, world)[0] for r in range(world)]
        assert sum(counts) == dg.n_nodes(g)
        assert max(counts) / min(counts) < 1.03, (world, counts)
        # dealing to 4*world virtual ranks (bench.py --pieces 4) leaves every GPU the same nodes
        pieces = 4
        vcounts = [sum(dg.shard_layout(g, p * world + r, pieces * world)[0] for p in range(pieces))
                   for r in range(world)]
        assert vcounts == counts


def _inplace_worker(rank, world, port, res, pieces, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import discregrid_amd as dg
        dg.load_library()
        V, F = T.torus()
        dom = T.oracle_default_domain(V)
        grid = dg.grid_desc(dom[:3], dom[3:], res)
        om = T.OracleMesh(V, F)
        D = [(res[0] + 1, res[1] + 1, res[2] + 1), (2 * res[0], res[1] + 1, res[2] + 1), (2 * res[1], res[2] + 1, res[0] + 1),
             (2 * res[2], res[0] + 1, res[1] + 1)]
        off = np.concatenate([[0], np.cumsum([int(np.prod(d)) for d in D])])
        vworld = pieces * world
        rng = np.random.default_rng(7)      # (the same skewed cost profile on every rank)
        cost = [rng.uniform(0.5, 4.0, d[2]).astype(np.float32) for d in D]
        cuts = dg.chunk_layout(grid, vworld, cost)
        field = torch.full((int(off[4]),), float("nan"), dtype=torch.float64)
        mine = 0
        for p in range(pieces):
            v = p * world + rank
            for c in range(4):          # this rank's chunks, sampled straight into their places (CPU oracle here)
                a, b = int(off[c]) + int(cuts[c][v]) * D[c][0] * D[c][1], int(off[c]) + int(cuts[c][v + 1]) * D[c][0] * D[c][1]
                if b > a:
                    field[a:b] = torch.from_numpy(om.sample_nodes(dom, res, a, b))
                    mine += b - a
            for o in range(world):      # dg_sdf_sample_exchange_device's grouped broadcasts, in place
                vo = p * world + o
                for c in range(4):
                    a, b = int(off[c]) + int(cuts[c][vo]) * D[c][0] * D[c][1], int(off[c]) + int(cuts[c][vo + 1]) * D[c][0] * D[c][1]
                    if b > a:
                        dist.broadcast(field[a:b], src=o)
        ret[rank] = (bool(np.array_equal(field.numpy(), om.sample_nodes(dom, res))), mine)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,res,pieces", [(2, [10, 7, 13], 2), (3, [6, 9, 5], 1)])
def test_inplace_chunks_broadcast_gloo(world, res, pieces):
    """The in-place exchange (contiguous cost-weighted chunks of dg_chunk_layout, broadcast into place, no unpack) with
    world_size 2 and 3 over gloo: every rank ends with the whole field, every node is sampled exactly once."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_inplace_worker, args=(world, _free_port(), res, pieces, ret), nprocs=world, join=True)
    assert len(ret) == world and all(v[0] for v in ret.values())
    assert sum(v[1] for v in ret.values()) == T.n_nodes(res)
