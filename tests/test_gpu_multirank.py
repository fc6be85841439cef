"""bench.py's N > 1 protocol with real kernels on ONE GPU: two (and four) ranks launched with
torch.distributed.run share device 0, the collectives go through gloo (RCCL refuses two ranks on one
device), everything else is what runs on a multi-GPU node -- virtual-rank shards per piece, async
all-gathers, range unpacks on their own stream.  Every rank asserts that the assembled field equals
the direct unsharded launch bit for bit (bench.py does that in self-test mode)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import dgtest as T

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_library_allgather_one_rank():
    """The N > 1 step of bench.py as the driver runs it -- dg_sdf_sample_allgather_device: the library's own
    RCCL communicator (dg_comm_create), shards, pieced all-gather, range unpacks -- with a world of one
    rank on the one GPU of this box; bench.py asserts field == direct launch, bit for bit."""
    cmd = [sys.executable, os.path.join(T.ROOT, "bench.py"), "--force-shard-path", "--steps", "2", "--warmup", "1",
           "--no-extras", "--pieces", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "dg_sdf_sample_allgather_device" in rec["config"]["sharding"] and rec["value"] > 0


def test_bench_falls_back_to_torch_distributed_when_the_library_communicator_fails():
    """A scaling run must not be lost to plumbing: if dg_comm_create fails on any rank, every rank of bench.py gathers
    through torch.distributed instead (same kernels, same unpack), says so on its line, and the field still equals
    the direct launch bit for bit (bench.py asserts that with --force-shard-path)."""
    cmd = [sys.executable, os.path.join(T.ROOT, "bench.py"), "--force-shard-path", "--steps", "2", "--warmup", "1",
           "--no-extras", "--pieces", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_PORT=str(_free_port()), DG_BENCH_BREAK_LIBRARY_COMM="1"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    sharding = rec["config"]["sharding"]
    assert "torch.distributed (python)" in sharding and "library communicator unavailable" in sharding and rec["value"] > 0


def test_cpp_multi_gpu_tool_one_rank(tmp_path):
    """GenerateSDFMultiGPU (C++, one process per GPU, ncclCommInitRank through dg_comm_create, device-resident
    field on every rank) with one rank writes the file GenerateSDF writes, byte for byte; with --steps it
    prints the whole-job rate.  With two or more GPUs visible the same comparison runs on all of them."""
    import discregrid_amd as dg
    build = os.path.join(T.ROOT, "discregrid_amd", "cpp", "build")
    V, F = T.torus()
    obj = str(tmp_path / "torus.obj")
    T.write_obj(obj, V, F)
    ref = str(tmp_path / "ref.cdf")
    subprocess.check_call([os.path.join(build, "GenerateSDF"), "-r", "24 20 22", "-o", ref, obj], stdout=subprocess.DEVNULL)
    dg.load_library()
    for gpus in sorted({1, min(dg.device_count(), 8)}):
        # interleaved slabs + unpack / contiguous chunks exchanged in place / pushed by peer copies into a field of
        # dg_comm_field_alloc / copied into the shared-memory host vector (no communicator at all)
        for extra in ([], ["--inplace"], ["--p2p"], ["--copy"], ["--copy-shm"], ["--host"]):
            out = str(tmp_path / ("multi%d%s.cdf" % (gpus, "".join(extra))))
            txt = subprocess.check_output([os.path.join(build, "GenerateSDFMultiGPU"), "-g", str(gpus), "-r", "24 20 22", "--steps", "3",
                                           "--pieces", "2", "-o", out, obj] + extra, timeout=600).decode()
            rec = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
            assert rec["n_gpus"] == gpus and rec["value"] > 0 and rec["nodes"] == T.n_nodes([24, 20, 22])
            assert open(out, "rb").read() == open(ref, "rb").read()


@pytest.mark.parametrize("world,pieces", [(2, 4), (4, 2)])
def test_sharded_protocol_with_several_ranks_on_one_gpu(world, pieces):
    env = dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(T.ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--pieces", str(pieces), "--exchange", "slabs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == world and rec["scaling"] == "weak" and rec["value"] > 0
    assert "pipelined in %d piece" % pieces in rec["config"]["sharding"]


@pytest.mark.parametrize("exchange", ["inplace", "inplace-p2p", "to-root", "copy"])
def test_library_inplace_exchange_one_rank(exchange):
    """dg_sdf_sample_exchange_device (contiguous chunks cut by dg_chunk_layout, sampled straight into the field,
    grouped ncclBroadcast / ncclSend + ncclRecv, no unpack) through the library's own RCCL communicator with a world
    of one rank; bench.py asserts field == direct launch, bit for bit."""
    cmd = [sys.executable, os.path.join(T.ROOT, "bench.py"), "--force-shard-path", "--steps", "2", "--warmup", "1",
           "--no-extras", "--pieces", "4", "--exchange", exchange]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "dg_sdf_sample_exchange_device" in rec["config"]["sharding"] and exchange in rec["config"]["sharding"] and rec["value"] > 0


@pytest.mark.parametrize("world,pieces", [(2, 4), (4, 2)])
def test_inplace_exchange_with_several_ranks_on_one_gpu(world, pieces):
    """The in-place protocol with real kernels and several ranks sharing the one GPU (broadcasts through gloo):
    chunks of dg_chunk_layout, re-cut after the warm-up step from the measured sampling times of every rank
    (cost-weighted cuts), sampled into place by dg_sdf_sample_planes_device; every rank asserts field == direct launch."""
    env = dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(T.ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "2",
           "--pieces", str(pieces), "--exchange", "inplace"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == world and "contiguous chunks" in rec["config"]["sharding"]


def _torchrun(world, script, *args, env=None, timeout=300):
    """torch.distributed.run in a process group of its own: on a timeout the WHOLE group is killed (a rank stuck in a
    driver call must not outlive the test and sit on the GPU)."""
    import signal
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script] + [str(a) for a in args]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env or dict(os.environ), start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        return subprocess.CompletedProcess(cmd, -9, out, (err or "") + "\nTIMEOUT after %d s: process group killed" % timeout)
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


@pytest.mark.parametrize("world,res,pieces", [(2, "61 47 53", 3), (3, "40 36 33", 2), (4, "61 47 53", 2), (4, "256 256 512", 2)])
def test_copy_exchange_with_several_ranks_on_one_gpu(world, res, pieces):
    """DG_EXCHANGE_COPY inside the library with several PROCESSES sharing the one GPU (tests/perf/copy_exchange_worker.py): the
    communicator's control plane is gloo (dg_comm_create_external: RCCL refuses two ranks on one device), everything else is what
    runs on a multi-GPU node -- IPC handles of the ranks' fields exchanged and opened (one rank at a time), cuts agreed on by hash
    (ranks that disagree are told so), barrier, every rank's chunks pushed into every peer's field with hipMemcpyAsync on a copy
    stream per peer while the next piece is sampled, barrier; four steps with changing cost-weighted cuts.  Every rank asserts
    field == direct launch, bit for bit.  The last case is the largest lattice of the bench whose field (1.9 GB) the platform's IPC
    opens.  (One device: the copies do not cross xGMI.  UNVERIFIED on more than one GPU.)"""
    out = _torchrun(world, os.path.join(T.ROOT, "tests", "perf", "copy_exchange_worker.py"), res, pieces)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    recs = [json.loads(l) for l in out.stdout.replace("}{", "}\n{").splitlines() if l.startswith("{")]
    assert len(recs) == world and all(r["ok"] and r["mismatch_caught"] and r["registered_fields"] == 1 and r["rccl_nranks"] == -1 for r in recs), recs


@pytest.mark.parametrize("world,res,pieces,steps", [(2, "61 47 53", 3, 4), (4, "40 36 33", 2, 4), (2, "256 256 600", 2, 2), (4, "256 256 600", 2, 2)])
def test_copy_exchange_through_vmm_chunks_with_several_ranks_on_one_gpu(world, res, pieces, steps):
    """DG_EXCHANGE_COPY with fields from dg_comm_field_alloc: hipMemCreate chunks of 512 MiB behind one address range, every
    chunk exported as a POSIX descriptor, the descriptors handed to the peers over a unix socket (SCM_RIGHTS), imported and
    mapped side by side -- no allocation is ever opened as a whole, so the 2 GiB wall of hipIpcOpenMemHandle does not apply.
    The last two cases are a field of 2.2 GB (256 x 256 x 600: five chunks) with 2 and 4 processes; every rank asserts
    field == direct launch, bit for bit, over steps with changing cost-weighted cuts.  (One device.  UNVERIFIED on more.)"""
    out = _torchrun(world, os.path.join(T.ROOT, "tests", "perf", "copy_exchange_worker.py"), res, pieces, "vmm", steps, timeout=420)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    recs = [json.loads(l) for l in out.stdout.replace("}{", "}\n{").splitlines() if l.startswith("{")]
    assert len(recs) == world and all(r["ok"] and r["vmm"] and r["mismatch_caught"] and r["registered_fields"] == 1 for r in recs), recs
    if res == "256 256 600":
        assert all(r["field_gb"] > 2.147 for r in recs)


@pytest.mark.parametrize("world,res,pieces,steps", [(2, "61 47 53", 3, 4), (3, "40 36 33", 2, 4), (4, "256 256 600", 2, 2)])
def test_copy_exchange_without_any_collective_library(world, res, pieces, steps):
    """DG_EXCHANGE_COPY on a communicator whose control plane lives in shared memory (dg_comm_create_shm) with fields of
    dg_comm_field_alloc: copy engines for the data, descriptors over a unix socket for the set-up, two barriers in a shared-memory
    segment per step -- the whole field on every rank's device without RCCL and without caller-supplied collectives.  Several
    processes on the one GPU, the last case with a field of 2.2 GB; every rank asserts field == direct launch, bit for bit, and
    ranks with different plane costs are all told so."""
    out = _torchrun(world, os.path.join(T.ROOT, "tests", "perf", "copy_exchange_worker.py"), res, pieces, "shm", steps, timeout=420)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    recs = [json.loads(l) for l in out.stdout.replace("}{", "}\n{").splitlines() if l.startswith("{")]
    assert len(recs) == world and all(r["ok"] and r["shm"] and r["mismatch_caught"] and r["registered_fields"] == 1 and r["rccl_nranks"] == -1 for r in recs), recs


@pytest.mark.parametrize("world,res,pieces,steps", [(2, "61 47 53", 3, 4), (3, "40 36 33", 2, 4), (4, "61 47 53", 2, 4), (4, "256 256 600", 2, 2)])
def test_host_vector_exchange_with_several_ranks_on_one_gpu(world, res, pieces, steps):
    """dg_sdf_sample_to_host_field (the form that needs neither collective kernels nor device IPC): every rank copies the chunks
    it sampled into a POSIX shared-memory vector all ranks map, a barrier inside the segment says when it is whole; every rank
    asserts shared vector == direct launch, bit for bit, and ranks with different plane costs are all told so.  The last case is
    a vector of 2.2 GB with four processes."""
    out = _torchrun(world, os.path.join(T.ROOT, "tests", "perf", "host_exchange_worker.py"), res, pieces, steps, timeout=420)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    recs = [json.loads(l) for l in out.stdout.replace("}{", "}\n{").splitlines() if l.startswith("{")]
    assert len(recs) == world and all(r["ok"] and r["mismatch_caught"] for r in recs), recs


def test_copy_exchange_refuses_allocations_the_platform_cannot_open():
    """hipIpcOpenMemHandle never returned for allocations above 2 GiB on the development box: every rank refuses such a field
    together (instead of one of them hanging), and bench.py's exchange race then goes on without the copy form."""
    out = _torchrun(2, os.path.join(T.ROOT, "tests", "perf", "copy_exchange_worker.py"), "61 47 53", 2, env=dict(os.environ, DG_IPC_MAX_MB="1"))
    assert out.returncode != 0 and "opening allocations above 1 MB" in out.stdout + out.stderr and "dg_comm_field_alloc" in out.stdout + out.stderr


def test_copy_exchange_in_bench_with_two_ranks_on_one_gpu():
    """bench.py --exchange copy as the driver would run it at N = 2 (256 x 256 x 512 per rank pair), gloo control plane."""
    env = dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1")
    out = _torchrun(2, os.path.join(T.ROOT, "bench.py"), "--gpus", 2, "--steps", 2, "--warmup", 2, "--pieces", 4, "--exchange", "copy", env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and "peer copies on the copy engines" in rec["config"]["sharding"]
    ex = rec["config"]["exchange"]
    assert ex["chosen"] == "copy" and len(ex["per_rank"]["sample_ms"]) == 2 and len(ex["per_rank"]["exchange_wait_ms"]) == 2
    assert all(len(r) == 4 and all(t > 0 for t in r) for r in ex["per_rank"]["sample_ms"])


def test_host_vector_form_in_bench_one_rank():
    """bench.py --exchange host with a world of one rank: dg_sdf_sample_to_host_field into the shared-memory vector; bench.py
    asserts shared vector == direct launch, bit for bit."""
    cmd = [sys.executable, os.path.join(T.ROOT, "bench.py"), "--force-shard-path", "--steps", "2", "--warmup", "1",
           "--no-extras", "--pieces", "4", "--exchange", "host"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_PORT=str(_free_port())))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "dg_sdf_sample_to_host_field" in rec["config"]["sharding"] and rec["value"] > 0


def test_bench_watchdog_reports_the_forms_measured_before_a_hang():
    """A form that never completes must not take the scaling number with it: two ranks on the one GPU, the fourth form (inplace)
    made to hang on every rank, 25 s per form -- rank 0 prints the line of the best form measured before it (host, copy-shm,
    slabs), the line says which form was cut off, and every rank leaves."""
    env = dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1", DG_BENCH_HANG_FORM="inplace")
    out = _torchrun(2, os.path.join(T.ROOT, "bench.py"), "--gpus", 2, "--steps", 2, "--warmup", 1, "--pieces", 2, "--form-timeout", 25, env=env, timeout=300)
    assert "watchdog: exchange form inplace did not complete" in out.stderr, out.stdout[-2000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ex = rec["config"]["exchange"]
    assert rec["value"] > 0 and set(ex["ms_by_form"]) == {"host", "copy-shm", "slabs"} and ex["chosen"] in ("host", "copy-shm", "slabs")
    assert "inplace did not complete within 25 s" in ex["watchdog"]


def test_exchange_auto_times_every_form_and_explains_itself():
    """bench.py's default for N > 1: every exchange form runs its warm-up and timed steps (two ranks on the one GPU here, gloo
    stand-ins for RCCL), the fastest is reported, and the line carries the timings per form, the sampling time
    per rank and piece and the exchange time the sampling did not hide; the field equals the direct launch on every rank."""
    env = dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pieces", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ex = rec["config"]["exchange"]
    assert set(ex["ms_by_form"]) == {"host", "copy-shm", "slabs", "inplace", "inplace-p2p", "copy"} and not ex["errors"], ex
    assert ex["chosen"] == min(ex["ms_by_form"], key=ex["ms_by_form"].get)
    assert len(ex["per_rank"]["sample_ms"]) == 2


def test_sample_planes_and_chunk_layout():
    """dg_chunk_layout + dg_sdf_sample_planes_device: the chunks of a 6-way cut (uniform and with a skewed cost
    profile), sampled one by one into one array, give the direct launch's field; cuts are monotone, cover every plane,
    sit on brick layers, and follow the cost."""
    import numpy as np
    import torch
    import discregrid_amd as dg
    dg.load_library()
    V, F = T.torus()
    dom = T.oracle_default_domain(V)
    res = [30, 41, 52]
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    want = mesh.sample_nodes(grid)
    D2 = [res[2] + 1, res[2] + 1, res[0] + 1, res[1] + 1]
    cost = [np.linspace(1.0, 9.0, d).astype(np.float32) for d in D2]
    for pc in (None, cost):
        cuts = dg.chunk_layout(grid, 6, pc)
        assert (cuts[:, 0] == 0).all() and list(cuts[:, -1]) == D2 and (np.diff(cuts.astype(np.int64), axis=1) >= 0).all()
        assert (cuts[:, 1:-1] % 4 == 0).all()
        if pc is not None:      # the expensive end gets fewer planes
            assert (np.diff(cuts.astype(np.int64), axis=1)[:, 0] > np.diff(cuts.astype(np.int64), axis=1)[:, -1]).all()
        field = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        for v in range(6):
            dg.sample_planes_device(mesh, grid, cuts[:, v], cuts[:, v + 1], field.data_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(field.cpu().numpy(), want)


def test_comm_create_gives_up_when_a_rank_never_arrives():
    """Communicator set-up is a collective; with a rank missing it can never complete.  dg_comm_create must fail
    after DG_COMM_TIMEOUT_S seconds instead of blocking forever (run in a child process, which leaves at once: the
    helper thread that sits in ncclCommInitRank cannot be recalled)."""
    code = ("import os, sys, time; sys.path.insert(0, %r); import discregrid_amd as dg; dg.load_library(); dg.set_device(0)\n"
            "t0 = time.time()\n"
            "try:\n"
            "    dg.Comm(dg.Comm.unique_id(), 0, 2)\n"
            "    print('NO ERROR')\n"
            "except dg.DiscregridError as e:\n"
            "    print('FAILED AFTER %%.1f s: %%s' %% (time.time() - t0, e))\n"
            "sys.stdout.flush(); os._exit(0)\n") % T.ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, DG_COMM_TIMEOUT_S="4"))
    assert "FAILED AFTER" in out.stdout and "did not complete within 4 s" in out.stdout, out.stdout + out.stderr[-2000:]


def _plain_env():
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e["DG_BENCH_SELFTEST_ONE_GPU"] = "1"
    return e


def test_plain_bench_command_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` typed PLAIN -- no torchrun, no WORLD_SIZE -- starts its two ranks itself (round 6; it used to exit
    with rc 1): the preflight runs first (stderr), then the race over every exchange form with two ranks on the one GPU; ONE line
    on stdout, n_gpus 2, the forms' times also as plain scalars inside roofline; every rank asserted field == direct launch."""
    out = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--pieces", "2"],
                         capture_output=True, text=True, timeout=600, env=_plain_env())
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0 and "256x256x512" in rec["config"]["workload"]
    rf = rec["roofline"]
    assert rf["exchange_chosen"] == rec["config"]["exchange"]["chosen"] and rf["compulsory_hbm_frac"] > 0
    assert {"exchange_ms_host", "exchange_ms_copy_shm", "exchange_ms_slabs", "exchange_ms_inplace", "exchange_ms_inplace_p2p", "exchange_ms_copy"} <= set(rf)
    assert "starting the ranks myself" in out.stderr
    for step in ("devices: ok", "shm: ok", "vmm: ok", "rccl: skipped", "host: ok"):
        assert out.stderr.count(step) == 2, (step, out.stderr[-4000:])
    assert "bench.py preflight: every form may run" in out.stderr


def test_plain_bench_command_strong_scaling():
    """--scaling strong: the metric's own 256^3 lattice shared by the ranks; the line says so."""
    out = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pieces", "2",
                          "--scaling", "strong", "--exchange", "copy-shm", "--no-preflight"], capture_output=True, text=True, timeout=600, env=_plain_env())
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and "256x256x256 = 118425857 nodes" in rec["config"]["workload"]
    assert "STRONG scaling" in rec["config"]["workload"] and rec["roofline"]["exchange_chosen"] == "copy-shm"


@pytest.mark.parametrize("world", [2, 4])
def test_scale_preflight_on_one_gpu(world):
    """tools/scale_preflight.py stand-alone with several processes on the one GPU (importer and exporter of the hipMemCreate chunk
    are then the same device): every step reports ok, RCCL is skipped, and the output has the documented format."""
    import re
    out = _torchrun(world, os.path.join(T.ROOT, "tools", "scale_preflight.py"), env=dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1"), timeout=240)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    for step in ("devices", "gloo", "shm", "vmm", "host"):
        assert len(re.findall(r"preflight\[rank \d/%d\] %s: ok \(" % (world, step), out.stderr)) == world, (step, out.stderr[-4000:])
    assert out.stderr.count("rccl: skipped") == world and out.stderr.count("every form can run") == world


def test_race_does_not_retry_rccl_after_it_failed_to_come_up():
    """The first RCCL form that fails BEFORE its first step completes takes the other RCCL forms out of the race (they would fail the same
    way, each after its own time-out on a real node); the forms that need no RCCL are measured and one of them is reported."""
    env = dict(_plain_env(), DG_BENCH_FAIL_FORM="slabs")
    out = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--pieces", "2", "--no-preflight"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    ex = rec["config"]["exchange"]
    assert set(ex["ms_by_form"]) == {"host", "copy-shm"} and ex["chosen"] in ("host", "copy-shm"), ex
    assert "simulated failure" in ex["errors"]["slabs"]
    for form in ("inplace", "inplace-p2p", "copy"):
        assert ex["errors"][form].startswith("not run: RCCL did not come up for form slabs"), ex["errors"]
    assert rec["roofline"]["exchange_error_slabs"].startswith("RuntimeError") and "exchange_ms_host" in rec["roofline"]


def test_eight_ranks_strong_scaling_on_one_gpu():
    """The 8-rank code paths (cuts for 8 x pieces virtual ranks, 7 peers per rank in the copy form, eight processes at the shared-memory
    barriers) with real kernels on the one GPU: the metric's own 256^3 lattice shared by eight processes, the two forms that run with
    real library paths on the rig; every rank asserts field == direct launch."""
    out = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--pieces", "2", "--scaling", "strong",
                          "--exchange", "copy-shm", "--form-timeout", "300"], capture_output=True, text=True, timeout=900, env=_plain_env())
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["value"] > 0 and len(rec["config"]["exchange"]["per_rank"]["sample_ms"]) == 8
    assert out.stderr.count("vmm: ok") == 8 and out.stderr.count("host: ok") == 8
