import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import dgtest as T, discregrid_amd as dg
dg.load_library()
V, F = T.icosphere(224)
t0 = time.time(); m = dg.Mesh(V, F); print("mesh create %.2f s" % (time.time() - t0), m.info())
dom = dg.default_domain(V)
for res in (128, 256):
    g = dg.grid_desc(dom[:3], dom[3:], [res] * 3); n = dg.n_nodes(g)
    out = torch.empty(n, dtype=torch.float64, device="cuda"); s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s); e1.record(); torch.cuda.synchronize()
    print(res, "ms", e0.elapsed_time(e1), "Mnodes/s", n / e0.elapsed_time(e1) / 1e3, "heavy", m.last_heavy_bricks())
    # analytic check: |phi - (|x| - 1)| small
    samp = out[::997].cpu().numpy()
    import ctypes
    # positions of sampled nodes via oracle helper
    idx = np.arange(0, n, 997)
    pos = np.concatenate([T.oracle_node_positions(dom, [res] * 3, int(i), int(i) + 1) for i in idx[:2000]])
    err = np.abs(samp[:2000] - (np.linalg.norm(pos.reshape(-1, 3), axis=1) - 1.0)).max()
    print("max deviation from the analytic sphere distance on 2000 nodes: %.2e" % err)
