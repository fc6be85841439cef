"""K1 A/B harness: times k_sample_nodes on the headline workload for several library variants /
BVH settings, one child process per variant (run on the GPU box).
usage: python tests/perf/k1_ab.py variant.so[:ENV=VAL,...] ...   (median of AB_REPS launches)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, numpy as np
import dgtest as T, discregrid_amd as dg
dg.load_library(%(lib)r)
res = [int(os.environ.get("AB_RES", "256"))]*3
name = os.environ.get("AB_MESH", "ico71")
V, F = T.icosphere(71) if name == "ico71" else T.bunny_mesh()
dom = T.oracle_default_domain(V)
g = dg.grid_desc(dom[:3], dom[3:], res); n = dg.n_nodes(g)
m = dg.Mesh(V, F)
out = torch.empty(n, dtype=torch.float64, device="cuda"); s = torch.cuda.current_stream().cuda_stream
m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s); torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("AB_REPS", "4"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
chk = float(out[::9973].sum().item())
print(json.dumps({"ms": sorted(ts)[len(ts)//2], "min": min(ts), "checksum": chk, "heavy": m.last_heavy_bricks()}))
'''


def main():
    for spec in sys.argv[1:]:
        lib, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            env[k] = v
        try:
            out = subprocess.check_output([sys.executable, "-c", CHILD % {"root": ROOT, "lib": os.path.join(ROOT, lib)}],
                                          env=env, stderr=subprocess.STDOUT, timeout=300).decode().strip().splitlines()[-1]
            res = json.loads(out)
        except Exception as e:  # noqa
            res = {"error": str(e)[-300:]}
        print(spec, res, flush=True)


if __name__ == "__main__":
    main()
