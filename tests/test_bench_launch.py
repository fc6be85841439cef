"""`python bench.py --gpus N` started PLAIN (no torchrun, no WORLD_SIZE): the script starts its own N ranks, forwards rank 0's
record as the only stdout line and returns the ranks' exit code.  Without a GPU the ranks run in DG_BENCH_DRY_RANKS mode
(they meet through gloo and print a record marked dry_run): the launcher plumbing is what is tested here; the same command
with real kernels is tests/test_gpu_multirank.py::test_plain_bench_command_starts_its_own_ranks."""
import json
import os
import subprocess
import sys

import pytest

import dgtest as T

BENCH = os.path.join(T.ROOT, "bench.py")


def test_launch_command_is_the_drivers_form():
    sys.path.insert(0, T.ROOT)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], 23456)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "23456", BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert bench.grid_for(1) == [256, 256, 256] and bench.grid_for(2) == [256, 256, 512] and bench.grid_for(8) == [512, 512, 512]
    assert bench.grid_for(8, "strong") == [256, 256, 256]
    with pytest.raises(SystemExit):
        bench.grid_for(3)


def _plain(n, *extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(DG_BENCH_DRY_RANKS="1", **(env or {}))
    return subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "1", *extra], capture_output=True, text=True,
                          timeout=300, env=e)


@pytest.mark.parametrize("n,scaling,grid", [(2, "weak", [256, 256, 512]), (2, "strong", [256, 256, 256])])
def test_plain_command_starts_its_own_ranks(n, scaling, grid):
    out = _plain(n, "--scaling", scaling)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # rank 0's record and nothing else on stdout
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == n and rec["scaling"] == scaling and rec["grid"] == grid
    assert "starting the ranks myself" in out.stderr and "torch.distributed.run" in out.stderr
    assert out.stderr.count("chatter on stdout") == n  # the ranks' other output went to stderr


def test_plain_command_returns_the_ranks_exit_code():
    out = _plain(2, env={"DG_BENCH_DRY_RC": "7"})
    assert out.returncode != 0
    out = _plain(3)
    assert out.returncode != 0 and "power of two" in out.stderr


def test_scale_preflight_removes_only_the_forms_a_failing_step_serves():
    """A building block that fails takes only the forms that need it out of bench.py's race; a skipped step and a preflight that
    never started say nothing; a child killed in the middle removes what it had not reached."""
    sys.path.insert(0, os.path.join(T.ROOT, "tools"))
    import scale_preflight as pf
    ok = {s: {"ok": True} for s in pf.STEPS}
    assert pf.forms_removed(ok) == set()
    assert pf.forms_removed(dict(ok, rccl={"ok": False})) == {"slabs", "inplace", "inplace-p2p", "copy", "to-root"}
    assert pf.forms_removed(dict(ok, rccl={"ok": None})) == set()                    # skipped says nothing
    assert pf.forms_removed(dict(ok, vmm={"ok": False})) == {"copy-shm"}
    assert pf.forms_removed({k: v for k, v in ok.items() if k not in ("rccl", "host")}) == {"slabs", "inplace", "inplace-p2p", "copy", "to-root", "host"}  # killed after vmm
    assert pf.forms_removed({}) == set()                                             # never started: says nothing
