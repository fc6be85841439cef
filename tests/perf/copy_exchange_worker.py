#!/usr/bin/env python3
"""One rank of a DG_EXCHANGE_COPY self-test (TEST TOOLING; launched by tests/test_gpu_multirank.py through
torch.distributed.run): every rank is a process of its own on device 0, the communicator's control plane is gloo
(dg_comm_create_external), the data path is the library's -- IPC handles, peer copies on a stream per peer, barriers.
A small lattice (seconds, not the bench's gigabytes), several steps with cost-weighted cuts, the field of every rank compared
with the direct launch bit for bit.  Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import dgtest as T
    import discregrid_amd as dg
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    res = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "61 47 53").split()]
    pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    vmm = len(sys.argv) > 3 and sys.argv[3] in ("vmm", "shm")   # the field comes from dg_comm_field_alloc (hipMemCreate chunks, no size limit)
    shm = len(sys.argv) > 3 and sys.argv[3] == "shm"            # ... and the control plane lives in shared memory (dg_comm_create_shm)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    torch.cuda.set_device(0)
    dg.load_library()
    dg.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allgather(mine):
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [bytes(o.numpy().tobytes()) for o in outs]

    if shm:
        names = [None]
        if rank == 0:
            names[0] = "dg_ctltest_%d" % os.getpid()
        dist.broadcast_object_list(names, src=0)      # (gloo hands the segment's name round; the exchange itself never uses it)
        comm = dg.Comm.shared_memory(names[0], rank, world)
    else:
        comm = dg.Comm.external(rank, world, allgather, dist.barrier)
    V, F = T.torus()
    dom = T.oracle_default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    s = torch.cuda.current_stream().cuda_stream
    want = torch.empty(n, dtype=torch.float64, device="cuda")
    mesh.sample_nodes_device(grid, 0, n, want.data_ptr(), stream=s)
    if vmm:
        arr = comm.field_alloc(n)
        field = torch.as_tensor(arr, device="cuda")
        assert field.data_ptr() == arr.ptr and field.numel() == n
        field.fill_(float("nan"))
    else:
        field = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    flags = dg.EXCHANGE_INPLACE | dg.EXCHANGE_COPY
    D2 = [res[2] + 1, res[2] + 1, res[0] + 1, res[1] + 1]
    rng = np.random.default_rng(5)           # the same "measured" costs on every rank
    ok = True
    for step in range(steps):
        cost = None if step == 0 else [rng.uniform(0.5, 3.0, d).astype(np.float32) for d in D2]
        field.fill_(float("nan"))
        torch.cuda.synchronize()
        dist.barrier()
        comm.sample_exchange_device(mesh, grid, field.data_ptr(), pieces=pieces, flags=flags, plane_cost=cost, stream=s)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(field, want))
    # ranks that disagree about the cuts must be told so, not left to corrupt the field
    mismatch_caught = None
    if world > 1:
        bad = [rng.uniform(0.5, 3.0, d).astype(np.float32) * (1.0 + rank) for d in D2]
        bad[0][: len(bad[0]) // 2] *= (1.0 + 3.0 * rank)
        try:
            comm.sample_exchange_device(mesh, grid, field.data_ptr(), pieces=pieces, flags=flags, plane_cost=bad, stream=s)
            mismatch_caught = False
        except dg.DiscregridError as e:
            mismatch_caught = "plane_cost must hold the same values" in str(e)
    info = comm.info()
    print(json.dumps({"rank": rank, "world": world, "ok": ok, "vmm": vmm, "shm": shm, "field_gb": n * 8e-9, "mismatch_caught": mismatch_caught, "registered_fields": info["registered_fields"],
                      "rccl_nranks": info["rccl_nranks"], "wait_ms": comm.last_exchange_wait_ms()}), flush=True)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
