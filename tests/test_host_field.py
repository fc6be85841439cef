"""The shared-memory coefficient vector of the exchange form that needs neither RCCL nor device IPC (dg_host_field_*,
discregrid_amd/csrc/dg_capi_hostfield.cpp), without a GPU: several PROCESSES map one POSIX segment through the C ABI, write
disjoint ranges, meet at the barrier that lives in the segment and read each other's data; a rank that never arrives makes the
others FAIL after DG_COMM_TIMEOUT_S instead of hanging.  (The sampling step dg_sdf_sample_to_host_field needs a device:
tests/test_gpu_multirank.py.)"""
import multiprocessing as mp
import os
import time

import numpy as np

import dgtest as T  # noqa: F401  (puts the repo on sys.path)


def _rank(rank, world, name, n, rounds, q):
    try:
        import discregrid_amd as dg
        hf = dg.HostField(name, n, rank, world)
        info = hf.info()
        ok = info["nranks"] == world and info["rank"] == rank and info["n_doubles"] == n
        lo, hi = rank * n // world, (rank + 1) * n // world
        for r in range(rounds):
            hf.data[lo:hi] = np.arange(lo, hi) * (r + 1.0) + rank      # my range of round r
            hf.barrier()                                               # everybody has written
            for o in range(world):
                a, b = o * n // world, (o + 1) * n // world
                ok = ok and bool(np.array_equal(hf.data[a:b], np.arange(a, b) * (r + 1.0) + o))
            hf.barrier()                                               # everybody has read
        hf.close()
        q.put((rank, ok, None))
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, "%s: %s" % (type(e).__name__, e)))


def _run(world, n, rounds=25, missing=(), env=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "dg_test_%d_%d" % (os.getpid(), int(time.time() * 1e6) % 1000000007)
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        procs = [ctx.Process(target=_rank, args=(r, world, name, n, rounds, q)) for r in range(world) if r not in missing]
        for p in procs:
            p.start()
        out = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(30)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return sorted(out)


def test_ranks_share_one_vector_and_meet_at_its_barrier():
    out = _run(3, 100003)
    assert [(r, ok) for r, ok, _ in out] == [(0, True), (1, True), (2, True)], out


def _leftover(name, n, world, kind):
    """what a job that crashed during set-up leaves in /dev/shm: a full-size, initialised segment whose creator is gone"""
    import struct
    import subprocess
    import sys
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    path = "/dev/shm/" + name
    with open(path, "wb") as f:
        # Header (dg_capi_shm.h): magic, payload_bytes, nranks, kind, arrived, generation, attached, failed, creator_pid
        f.write(struct.pack("<QQIIIIIIq", 0x64675f73686d3031, 8 * n, world, kind, 1, 0, 1, 0, dead.pid))
        f.truncate(4096 + 8 * n)
    return path


def _rank_delayed(rank, world, name, n, rounds, q, delay):
    time.sleep(delay)
    _rank(rank, world, name, n, rounds, q)


def test_a_leftover_segment_of_a_crashed_job_is_not_mistaken_for_ours():
    """Round-5 advisor: a rank that arrives BEFORE rank 0 used to attach to any segment of the right name, size and magic -- also to one
    a crashed job left behind -- and then sat in a different segment than its peers until the deadline.  Now the creator's process must
    be alive, and a waiting rank notices when the name is given to another file: rank 1 starts first, finds the leftover, rejects it;
    rank 0 arrives two seconds later, replaces it; both meet."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "dg_test_stale_%d_%d" % (os.getpid(), int(time.time() * 1e6) % 1000000007)
    n = 5000
    kinds = {}
    for kind in (1, 2, 3):          # (the host-vector form's tag is an implementation detail: a leftover of any kind must be survivable)
        kinds[kind] = True
    path = _leftover(name, n, 2, 1)
    try:
        procs = [ctx.Process(target=_rank_delayed, args=(1, 2, name, n, 3, q, 0.0)), ctx.Process(target=_rank_delayed, args=(0, 2, name, n, 3, q, 2.0))]
        for p in procs:
            p.start()
        out = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(30)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    assert [(r, ok) for r, ok, _ in out] == [(0, True), (1, True)], out


def test_single_rank_needs_no_peer():
    out = _run(1, 4097, rounds=2)
    assert out == [(0, True, None)], out


def test_a_missing_rank_fails_the_others_instead_of_hanging():
    t0 = time.time()
    out = _run(3, 1000, missing=(2,), env={"DG_COMM_TIMEOUT_S": "2"})
    assert time.time() - t0 < 60
    # (the first rank to run out of patience marks the segment failed; the other one then leaves at once with "gave up")
    assert len(out) == 2 and all(not ok and err and ("did not arrive" in err or "gave up on this segment" in err) for _, ok, err in out), out
    assert any("did not arrive" in err for _, _, err in out), out


def test_a_missing_creator_fails_the_others():
    out = _run(2, 1000, missing=(0,), env={"DG_COMM_TIMEOUT_S": "2"})
    assert len(out) == 1 and not out[0][1] and "did not appear" in out[0][2], out


def test_descriptor_server_serves_the_peers_and_nobody_else():
    """The copy exchange hands the descriptors of a field's hipMemCreate chunks to the peers over an abstract unix socket
    (dg_capi_vmm.h: FdServer).  The descriptors mean read / write access to device memory and an abstract socket has no permissions
    (round-5 advisor, medium): the name ends in 64 random bits that travel through the control plane only, nothing is accepted before
    the peers' process ids are known, and a connection from any other process is closed without a descriptor and WITHOUT using up a
    peer's turn.  tests/cpp/fd_server_driver.cpp plays it through with a pipe instead of device memory (no GPU)."""
    import json
    import subprocess
    exe = os.path.join(T.ROOT, "tests", "cpp", "build", "fd_server_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(T.ROOT, "tests", "cpp")])
    rec = json.loads(subprocess.check_output([exe], timeout=60).decode())
    assert rec == {"started": True, "name_taken": True, "tokens_differ": True, "stranger_got_nothing": True, "peer_exit": 0,
                   "peer_wrote": "through the served descriptor", "bytes": 30}, rec


def test_a_leftover_segment_without_a_creator_fails_after_the_deadline():
    """... and if rank 0 never comes, a rank that keeps finding the leftover gives up when ITS deadline has passed (it used to start a new
    deadline with every look)."""
    name = "dg_test_stale2_%d_%d" % (os.getpid(), int(time.time() * 1e6) % 1000000007)
    path = _leftover(name, 3000, 2, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    old = os.environ.get("DG_COMM_TIMEOUT_S")
    os.environ["DG_COMM_TIMEOUT_S"] = "2"
    try:
        t0 = time.time()
        p = ctx.Process(target=_rank, args=(1, 2, name, 3000, 1, q))
        p.start()
        out = q.get(timeout=60)
        p.join(30)
        assert time.time() - t0 < 30
    finally:
        if old is None:
            os.environ.pop("DG_COMM_TIMEOUT_S", None)
        else:
            os.environ["DG_COMM_TIMEOUT_S"] = old
        if os.path.exists(path):
            os.unlink(path)
    assert out[1] is False and "leftover" in out[2], out
