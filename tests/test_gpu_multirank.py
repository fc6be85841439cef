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
