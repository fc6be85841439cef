#!/usr/bin/env python3
"""scale_preflight.py -- make the first multi-GPU run diagnosable.

Nothing of this repository's multi-rank path has run on more than one GPU where it was developed.  Before bench.py races
the exchange forms at N > 1, every rank runs this script in a CHILD process (so that a step that hangs takes the child with
it, not the bench) and learns which building blocks work on the box:

    devices  this rank's device, the number of visible devices, its row of the hipDeviceCanAccessPeer matrix
    gloo     the control group the other steps use for their hand-shakes (torch.distributed, CPU tensors)
    shm      the library's shared-memory control plane comes up (dg_comm_create_shm: segment open + its barrier)
    vmm      a field of dg_comm_field_alloc (hipMemCreate chunks) is exported, imported and mapped by every peer and written
             by the peers' copy engines (hipMemcpyAsync): one tiny exchange step in the copy form == the direct launch
    rccl     dg_comm_create (ncclCommInitRank), ncclCommCount == world, one tiny all-gather step == the direct launch
    host     the shared-memory host vector opens, its barrier works, one tiny step into it == the direct launch

Each step prints ONE line on stderr --

    preflight[rank R/W] <step>: ok (<detail>) <seconds> s        or
    preflight[rank R/W] <step>: FAILED <error> <seconds> s       or
    preflight[rank R/W] <step>: skipped (<why>)

-- and the results go to --json-out after EVERY step (a parent that had to kill the child reads what was reached).  A failing
step removes from bench.py's race only the forms that need it (FORMS_NEEDING below).  Stand-alone:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/scale_preflight.py

With DG_BENCH_SELFTEST_ONE_GPU=1 all ranks share device 0 (the one-GPU rig): importer and exporter are then the same device,
and the rccl step is skipped (RCCL refuses two ranks on one device).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEPS = ["devices", "gloo", "shm", "vmm", "rccl", "host"]
# exchange forms of bench.py that cannot work without the step
FORMS_NEEDING = {
    "gloo": [],      # (bench.py has its own control group by the time it calls this)
    "devices": [],
    "shm": ["copy-shm"],
    "vmm": ["copy-shm"],
    "rccl": ["slabs", "inplace", "inplace-p2p", "copy", "to-root"],
    "host": ["host"],
}


def forms_removed(results):
    """results: {step: {"ok": bool | None, ...}} of ONE rank (None = skipped) -> the forms this rank cannot run"""
    out = set()
    for step in STEPS:
        r = results.get(step)
        if r is None or r.get("ok") is False:       # never reached (the child was killed before) or failed
            out.update(FORMS_NEEDING[step])
        if r is None and step in ("devices", "gloo"):
            return set()                            # the preflight itself did not start: it says nothing
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json-out", default=None)
    ap.add_argument("--tag", default=None, help="unique name part for the shared-memory segments (every rank the same)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    selftest = os.environ.get("DG_BENCH_SELFTEST_ONE_GPU") == "1"
    if selftest:
        local_rank = 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("DG_COMM_TIMEOUT_S", "60")      # (the library's own set-up steps and barriers give up after this: RCCL's first
                                                          #  initialisation on eight GPUs can take tens of seconds)
    results = {}
    who = "preflight[rank %d/%d]" % (rank, world)

    def flush():
        if args.json_out:
            tmp = args.json_out + ".tmp"
            with open(tmp, "w") as f:
                json.dump(results, f)
            os.replace(tmp, args.json_out)

    def step(name, fn):
        t0 = time.perf_counter()
        try:
            detail = fn()
            if isinstance(detail, tuple) and detail[0] == "skipped":
                results[name] = {"ok": None, "detail": detail[1]}
                print("%s %s: skipped (%s)" % (who, name, detail[1]), file=sys.stderr, flush=True)
            else:
                results[name] = {"ok": True, "detail": detail, "seconds": round(time.perf_counter() - t0, 3)}
                print("%s %s: ok (%s) %.2f s" % (who, name, detail, time.perf_counter() - t0), file=sys.stderr, flush=True)
        except BaseException as exc:  # noqa: BLE001  (reported; the next step runs)
            if isinstance(exc, (KeyboardInterrupt, SystemExit)):
                raise
            results[name] = {"ok": False, "detail": "%s: %s" % (type(exc).__name__, str(exc)[:300]), "seconds": round(time.perf_counter() - t0, 3)}
            print("%s %s: FAILED %s %.2f s" % (who, name, results[name]["detail"], time.perf_counter() - t0), file=sys.stderr, flush=True)
        flush()
        return results[name]["ok"]

    import numpy as np
    import torch
    import torch.distributed as dist
    import discregrid_amd as dg

    state = {}

    def s_devices():
        n = torch.cuda.device_count()
        if n < 1:
            raise RuntimeError("no HIP device visible")
        if local_rank >= n:
            raise RuntimeError("LOCAL_RANK %d but %d device(s) visible" % (local_rank, n))
        torch.cuda.set_device(local_rank)
        dg.load_library()
        dg.set_device(local_rank)
        peers = [1 if j == local_rank else int(torch.cuda.can_device_access_peer(local_rank, j)) for j in range(n)]
        results.setdefault("info", {}).update({"device": local_rank, "visible": n, "name": torch.cuda.get_device_name(local_rank), "peer_access_row": peers})
        return "device %d of %d, %s, hipDeviceCanAccessPeer row %s" % (local_rank, n, torch.cuda.get_device_name(local_rank), "".join(map(str, peers)))

    def s_gloo():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1.0])
        dist.all_reduce(t)
        if int(t.item()) != world * (world + 1) // 2:
            raise RuntimeError("all_reduce over %d ranks returned %g" % (world, t.item()))
        state["gloo"] = True
        return "%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])

    def shared(obj):
        box = [obj if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return box[0]

    def tiny():
        """the small problem every data step runs: a cube's SDF on a 48 x 40 x 44 lattice (0.6 M nodes; enough planes for slabs of four on
        eight ranks), and the direct launch it must equal"""
        if "mesh" not in state:
            V = np.array([[1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, -1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1], [-1, 1, -1]], dtype=np.float64)
            F = np.array([[2, 3, 4], [8, 7, 6], [5, 6, 2], [6, 7, 3], [3, 7, 8], [1, 4, 8], [1, 2, 4], [5, 8, 6], [1, 5, 2], [2, 6, 3], [4, 3, 8], [5, 1, 8]],
                         dtype=np.uint32) - 1    # (a cube: the product is all this tool uses, no test helper, no checker)
            dom = dg.default_domain(V)
            state["grid"] = dg.grid_desc(dom[:3], dom[3:], [48, 40, 44])
            state["n"] = dg.n_nodes(state["grid"])
            state["mesh"] = dg.Mesh(V, F)
            ref = torch.empty(state["n"], dtype=torch.float64, device="cuda")
            state["mesh"].sample_nodes_device(state["grid"], 0, state["n"], ref.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            state["ref"] = ref
        return state["mesh"], state["grid"], state["n"], state["ref"]

    tag = args.tag or ("%d" % os.getppid())

    def s_shm():
        if not state.get("gloo"):
            raise RuntimeError("needs the gloo step")
        name = shared("dg_pfctl_%s" % tag)
        state["shm"] = dg.Comm.shared_memory(name, rank, world)
        return "segment %s, %d ranks at its barrier" % (name, world)

    def s_vmm():
        c = state.get("shm")
        if c is None:
            raise RuntimeError("needs the shm step")
        mesh, grid, n, ref = tiny()
        arr = c.field_alloc(n)
        f = torch.as_tensor(arr, device="cuda")
        f.fill_(float("nan"))
        s = torch.cuda.current_stream().cuda_stream
        c.sample_exchange_device(mesh, grid, f.data_ptr(), pieces=1, flags=dg.EXCHANGE_INPLACE | dg.EXCHANGE_COPY, root=0, stream=s)
        torch.cuda.synchronize()
        same = bool(torch.equal(f, ref))
        del f       # (a field the peers have mapped lives until the communicator goes: dg_comm_destroy at the end frees it)
        if not same:
            raise RuntimeError("the field assembled by peer copies differs from the direct launch")
        return "hipMemCreate chunk exported / imported / mapped by %d peer(s)%s, peer hipMemcpyAsync, field == direct launch" % (
            world - 1, " on the SAME device (one-GPU rig)" if selftest else "")

    def s_rccl():
        if selftest and world > 1:
            return ("skipped", "one-GPU self-test: RCCL refuses two ranks on one device")
        if not state.get("gloo"):
            raise RuntimeError("needs the gloo step")
        uid, err = None, None
        if rank == 0:
            try:
                uid = dg.Comm.unique_id()
            except Exception as exc:  # noqa: BLE001  (shared below: every rank fails this step together)
                err = "%s: %s" % (type(exc).__name__, exc)
        uid, err = shared((uid, err))
        if uid is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0 (%s)" % err)
        c = dg.Comm(uid, rank, world)
        try:
            nr = c.info()["rccl_nranks"]
            if nr != world:
                raise RuntimeError("ncclCommCount says %d, the world is %d" % (nr, world))
            mesh, grid, n, ref = tiny()
            f = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
            c.sample_allgather_device(mesh, grid, f.data_ptr(), pieces=1, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if not torch.equal(f, ref):
                raise RuntimeError("the all-gathered field differs from the direct launch")
        finally:
            c.close()
        return "ncclCommInitRank, ncclCommCount %d, all-gather of %d B per rank, field == direct launch" % (nr, 8 * dg.shard_layout(grid, rank, world)[1])

    def s_host():
        if not state.get("gloo"):
            raise RuntimeError("needs the gloo step")
        mesh, grid, n, ref = tiny()
        name = shared("dg_pfhost_%s" % tag)
        hf = dg.HostField(name, n, rank, world)
        try:
            scratch = torch.empty(n, dtype=torch.float64, device="cuda")
            hf.sample(mesh, grid, scratch.data_ptr(), pieces=1, stream=torch.cuda.current_stream().cuda_stream)
            hf.barrier()
            same = bool(np.array_equal(np.array(hf.data), ref.cpu().numpy()))
            hf.barrier()
        finally:
            hf.close()
        if not same:
            raise RuntimeError("the shared host vector differs from the direct launch")
        return "segment %s, %d B, host vector == direct launch" % (name, 8 * n)

    t_all = time.perf_counter()
    if step("devices", s_devices) is not True:
        flush()
        return 1
    step("gloo", s_gloo)
    step("shm", s_shm)
    step("vmm", s_vmm)
    step("rccl", s_rccl)
    step("host", s_host)
    if state.get("shm") is not None:
        try:
            state["shm"].close()
        except Exception:  # noqa: BLE001
            pass
    results["seconds"] = round(time.perf_counter() - t_all, 3)
    flush()
    bad = sorted(forms_removed(results))
    print("%s done in %.1f s: %s" % (who, time.perf_counter() - t_all, ("forms this rank cannot run: " + ", ".join(bad)) if bad else "every form can run"),
          file=sys.stderr, flush=True)
    if state.get("gloo"):
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
