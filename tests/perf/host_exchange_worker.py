#!/usr/bin/env python3
"""One rank of a dg_sdf_sample_to_host_field self-test (TEST TOOLING; launched by tests/test_gpu_multirank.py through
torch.distributed.run): every rank is a process of its own on device 0; the coefficient vector is assembled in the library's
shared-memory segment -- no RCCL, no device IPC, the only collective is the barrier inside the segment.  Several steps with
cost-weighted cuts; every rank compares the shared HOST vector with the direct launch bit for bit; ranks that pass different
plane costs must all be told so.  torch.distributed (gloo) is used for ONE thing: handing rank 0's segment name to the others.
Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import dgtest as T
    import discregrid_amd as dg
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    res = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "61 47 53").split()]
    pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    torch.cuda.set_device(0)
    dg.load_library()
    dg.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names = [None]
    if rank == 0:
        names[0] = "dg_hosttest_%d" % os.getpid()
    dist.broadcast_object_list(names, src=0)
    V, F = T.torus()
    dom = T.oracle_default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    hf = dg.HostField(names[0], n, rank, world)
    mesh = dg.Mesh(V, F)
    s = torch.cuda.current_stream().cuda_stream
    want = torch.empty(n, dtype=torch.float64, device="cuda")
    mesh.sample_nodes_device(grid, 0, n, want.data_ptr(), stream=s)
    want = want.cpu().numpy()
    field = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    D2 = [res[2] + 1, res[2] + 1, res[0] + 1, res[1] + 1]
    rng = np.random.default_rng(5)           # the same "measured" costs on every rank
    ok = True
    for step in range(steps):
        cost = None if step == 0 else [rng.uniform(0.5, 3.0, d).astype(np.float32) for d in D2]
        hf.barrier()                         # (nobody still compares the previous step's vector)
        if rank == 0:
            hf.data[:] = np.nan
        hf.barrier()
        hf.sample(mesh, grid, field.data_ptr(), pieces=pieces, plane_cost=cost, stream=s)
        ok = ok and bool(np.array_equal(hf.data, want))
        ms = hf.last_chunk_ms(pieces)
        ok = ok and len(ms) == pieces and all(t > 0 for t in ms)
    mismatch_caught = None
    if world > 1:
        hf.barrier()
        bad = [rng.uniform(0.5, 3.0, d).astype(np.float32) * (1.0 + rank) for d in D2]
        bad[0][: len(bad[0]) // 2] *= (1.0 + 3.0 * rank)
        try:
            hf.sample(mesh, grid, field.data_ptr(), pieces=pieces, plane_cost=bad, stream=s)
            mismatch_caught = False
        except dg.DiscregridError as e:
            mismatch_caught = "plane_cost must hold the same values" in str(e)
    info = hf.info()
    print(json.dumps({"rank": rank, "world": world, "ok": ok, "mismatch_caught": mismatch_caught, "registered": info["registered"],
                      "field_gb": n * 8e-9}), flush=True)
    hf.barrier()
    hf.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
