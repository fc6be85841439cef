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


def _worker(rank, world, port, res, ret):
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
        count, stride = dg.shard_layout(grid, rank, world)          # product bookkeeping (host only)
        assert count == emu.shard_count(res, rank, world)
        gathered = torch.full((world * stride,), float("nan"), dtype=torch.float64)
        mine = gathered[rank * stride:(rank + 1) * stride]
        shard = emu.EmuMesh(V, F).sample_shard(dom, res, rank, world)
        mine[:count] = torch.from_numpy(shard)
        dist.all_gather_into_tensor(gathered, mine.clone())
        field = emu.unpack(res, world, gathered.numpy(), stride)
        want = T.OracleMesh(V, F).sample_nodes(dom, res)
        ok = bool(np.array_equal(field, want))
        # max-over-ranks reduction as bench.py does for the timing
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ret[rank] = (ok, float(t.item()), count)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,res", [(2, [10, 7, 13]), (3, [6, 9, 5])])
def test_shard_allgather_unpack_gloo(world, res):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), res, ret), nprocs=world, join=True)
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
