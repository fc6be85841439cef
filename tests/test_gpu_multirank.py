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
        out = str(tmp_path / ("multi%d.cdf" % gpus))
        txt = subprocess.check_output([os.path.join(build, "GenerateSDFMultiGPU"), "-g", str(gpus), "-r", "24 20 22", "--steps", "3",
                                       "--pieces", "2", "-o", out, obj], timeout=600).decode()
        rec = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
        assert rec["n_gpus"] == gpus and rec["value"] > 0 and rec["nodes"] == T.n_nodes([24, 20, 22])
        assert open(out, "rb").read() == open(ref, "rb").read()


@pytest.mark.parametrize("world,pieces", [(2, 4), (4, 2)])
def test_sharded_protocol_with_several_ranks_on_one_gpu(world, pieces):
    env = dict(os.environ, DG_BENCH_SELFTEST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(T.ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--pieces", str(pieces)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == world and rec["scaling"] == "weak" and rec["value"] > 0
    assert "pipelined in %d piece" % pieces in rec["config"]["sharding"]
