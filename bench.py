#!/usr/bin/env python3
"""bench.py -- SDF node-sampling throughput of the HIP path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole grid: every lattice node of a
CubicLagrangeDiscreteGrid gets the signed distance to the mesh (the addFunction node loop),
with the mesh/BVH already resident in HBM.  Workload (BASELINE.json configs[2], the one the
metric is quoted on): synthetic class-I geodesic icosphere, nu = 71 -> 100 820 triangles,
unit radius; grid 256^3 over the reference's default domain -> 118 425 857 nodes per GPU.
For N > 1 (weak scaling, BASELINE configs[3]) the grid grows with N -- (256a, 256b, 256c),
abc = N, i.e. 512^3 at N = 8 -- the lattice is dealt to the ranks in 4-plane slabs, every
rank samples its shard, an RCCL all-gather assembles the packed shards on every GPU and an
unpack kernel restores reference node order.  The gather is issued in --pieces C pieces
(default 4): piece p of rank r is the shard of "virtual rank" p*N + r of a C*N-way deal, so
the all-gather of piece p runs on RCCL's stream over xGMI while the kernel samples piece p+1
and a third stream unpacks piece p-1 (C = 1 is the plain sample / gather / unpack sequence).
value = total nodes / time, max over ranks.

Also on the JSON line:
  roofline      achieved = ALGORITHMIC bytes of the reference traversal per launch
                (B_alg = Vbar*72 + Lbar*84 + 8 bytes/node, SURVEY.md 8(d), frozen in
                profiles/balg_icosphere71_256.json) / mean K1 kernel duration measured with
                HIP events on the launch stream; peak = 8 TB/s HBM3E.  (The kernel is VALU-issue
                bound, not HBM bound: `valu_busy` is the measured fraction of VALU issue slots in use.)
  cpu_baseline  the unmodified reference (oracle/_ref, kind "reference") or this repo's CPU
                restatement (kind "port") timed on this box's host cores on a bounded,
                evenly spread sample of the same lattice (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def grid_for(n_gpus):
    dims = [256, 256, 256]
    k, axis = n_gpus, 2
    while k > 1:
        if k % 2:
            raise SystemExit("--gpus must be a power of two")
        dims[axis] *= 2
        axis = (axis - 1) % 3
        k //= 2
    return dims


def load_balg():
    p = os.path.join(ROOT, "profiles", "balg_icosphere71_256.json")
    with open(p) as f:
        return json.load(f)


def cpu_baseline(V, F, dom, res, budget_s):
    """Reference (or port) node loop on the host cores over an evenly spread sample."""
    import dgtest as T
    n = T.n_nodes(res)
    if T.ref_available():
        kind = "reference"
        g = T.RefGrid(V, F, dom, res)

        def run(b, e):
            g.sample_nodes(b, e)
            return g.last_seconds
    else:
        kind = "port"
        om = T.OracleMesh(V, F)

        def run(b, e):
            om.sample_nodes(dom, res, b, e)
            return om.last_seconds
    n_chunks = 32
    # calibrate on a small spread sample, then size the sample for ~budget_s seconds
    probe = 4096
    starts = [int((i + 0.5) * n / n_chunks) for i in range(n_chunks)]
    t = sum(run(s, s + probe) for s in starts)
    rate = n_chunks * probe / max(t, 1e-9)
    per_chunk = int(min(max(rate * budget_s / n_chunks, probe), n // n_chunks))
    t = sum(run(s, min(n, s + per_chunk)) for s in starts)
    nodes = sum(min(n, s + per_chunk) - s for s in starts)
    return {
        "value": nodes / t / 1e6, "unit": "Mnodes/s", "cores": os.cpu_count(), "kind": kind,
        "sample": "%d nodes = %d evenly spaced runs of %d consecutive lattice nodes of the same %s grid, "
                  "OpenMP schedule(static), %.1f s" % (nodes, n_chunks, per_chunk, "x".join(map(str, res)), t),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--pcie", action="store_true", help="also report the PCIe-inclusive rate on stderr")
    ap.add_argument("--pieces", type=int, default=4,
                    help="N > 1: issue the all-gather in this many pieces, overlapped with the sampling kernel")
    ap.add_argument("--force-shard-path", action="store_true",
                    help="run the N > 1 protocol (RCCL init, shard, all_gather, unpack) even at N = 1 (self-test)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts
    import torch
    import torch.distributed as dist
    import dgtest as T
    import discregrid_amd as dg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Self-test of the N > 1 protocol on a box with ONE GPU: all ranks share device 0 and the collective
    # goes through gloo (RCCL refuses two ranks on one device).  Timings of such a run mean nothing.
    selftest = os.environ.get("DG_BENCH_SELFTEST_ONE_GPU") == "1"
    if selftest:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dg.load_library()
    dg.set_device(local_rank)
    sharded = world > 1 or args.force_shard_path
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if selftest:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    V, F = T.icosphere(71)
    dom = dg.default_domain(V)            # cmd/generate_sdf/main.cpp:83-91
    res = grid_for(world)
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n_nodes = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream

    field = torch.empty(n_nodes, dtype=torch.float64, device="cuda")
    if sharded:
        # piece p of this rank = shard of virtual rank p*world + rank in a (pieces*world)-way deal of the
        # 4-plane slabs; all virtual ranks share one slot size, so `gathered` is exactly the buffer a
        # single all-gather among pieces*world ranks would produce and the unpack kernel is unchanged
        pieces = max(1, min(args.pieces, 64 // world))    # dg_shard_layout handles up to 64 (virtual) ranks
        vworld = pieces * world
        counts = []
        stride = 0
        for p in range(pieces):
            c, stride = dg.shard_layout(grid, p * world + rank, vworld)
            counts.append(c)
        gathered = torch.empty(vworld * stride, dtype=torch.float64, device="cuda")
        mine = torch.zeros(pieces * stride, dtype=torch.float64, device="cuda")   # packed pieces (+ padding)
        launch_nodes = sum(counts)
    else:
        launch_nodes = n_nodes

    unpack_stream = torch.cuda.Stream() if sharded else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record(stream)
        if sharded:
            for p in range(pieces):
                mp = mine[p * stride:(p + 1) * stride]
                mesh.sample_shard_device(grid, p * world + rank, vworld, mp.data_ptr(), stream=s)
                # RCCL's stream waits for the kernel just enqueued; this stream goes on with piece p+1
                work = dist.all_gather_into_tensor(gathered[p * world * stride:(p + 1) * world * stride], mp,
                                                   async_op=True)
                with torch.cuda.stream(unpack_stream):
                    work.wait()          # unpack_stream waits for gather p, not the host
                    dg.unpack_shard_range_device(grid, vworld, gathered.data_ptr(), stride, p * world, (p + 1) * world,
                                                 field.data_ptr(), stream=unpack_stream.cuda_stream)
            if i is not None:
                ev[i][1].record(stream)
            stream.wait_stream(unpack_stream)
        else:
            mesh.sample_nodes_device(grid, 0, n_nodes, field.data_ptr(), stream=s)
            if i is not None:
                ev[i][1].record(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if sharded:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # sanity of the result that was just timed (cheap, outside the timed region)
    probe = field[:: max(1, n_nodes // 1000)].cpu().numpy()
    assert np.isfinite(probe).all() and np.abs(probe).max() < 2.0
    if (args.force_shard_path and world == 1) or selftest:
        ref = torch.empty_like(field)
        mesh.sample_nodes_device(grid, 0, n_nodes, ref.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert torch.equal(ref, field), "shard + all_gather + unpack differs from the direct launch"

    if rank == 0:
        balg = load_balg()
        achieved = balg["bytes_per_node"] * launch_nodes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mnodes/s SDF sampling (256\u00b3 grid, 100k-tri mesh) + % HBM roofline, 1/2/4/8 GPU",
            "value": n_nodes * args.steps / elapsed / 1e6,
            "unit": "Mnodes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "icosphere nu=71 (100820 tris) SDF node sampling, grid %s = %d nodes"
                            % ("x".join(map(str, res)), n_nodes),
                "nodes_per_gpu_launch": launch_nodes,
                "sharding": "none" if not sharded else
                            "4-plane slabs round-robin; sample / all_gather / unpack pipelined in %d piece(s)" % pieces,
                "mesh_bvh_build_s": round(mesh.info()["build_seconds"], 4),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": balg.get("measured_hbm_bytes_per_launch") if world == 1 else None,
                # one dg_sdf_sample_*_device call = k_sample_nodes + the two heavy-brick kernels (4 % of it)
                "kernel": "k_sample_nodes (+ k_heavy_subtrees, k_heavy_finish)", "kernel_ms": kernel_ms,
                # the resource that actually binds K1 (PMC, profiles/r01_pmc_summary.txt): VALU issue
                "valu_busy": balg.get("measured_valu_busy") if world == 1 else None,
                "algorithmic_bytes_per_node": balg["bytes_per_node"],
            },
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(V, F, dom, res, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        if args.pcie and world == 1:
            host = torch.empty(n_nodes, dtype=torch.float64).pin_memory()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mesh.sample_nodes_device(grid, 0, n_nodes, field.data_ptr(), stream=s)
            host.copy_(field, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("PCIe-inclusive (kernel + D2H into pinned host memory): %.1f Mnodes/s" % (n_nodes / dt / 1e6),
                  file=sys.stderr)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
