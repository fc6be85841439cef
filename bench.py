#!/usr/bin/env python3
"""bench.py -- SDF node-sampling throughput of the HIP path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work: started PLAIN with --gpus N > 1 (no WORLD_SIZE / RANK in the environment) the script starts its own N ranks
under torch.distributed.run (self_launch below), forwards rank 0's JSON line as the only line on stdout, everything else on
stderr, and returns the ranks' exit code.

A "step" is one pass of the hot path over the whole grid: every lattice node of a
CubicLagrangeDiscreteGrid gets the signed distance to the mesh (the addFunction node loop,
cubic_lagrange_discrete_grid.cpp:806-831), with the mesh/BVH already resident in HBM and the result
left in HBM.  Workload (BASELINE.json configs[2], the one the metric is quoted on): synthetic class-I
geodesic icosphere, nu = 71 -> 100 820 triangles, unit radius; grid 256^3 over the reference's default
domain -> 118 425 857 nodes per GPU.  For N > 1 (weak scaling, BASELINE configs[3]) the grid grows with
N -- (256a, 256b, 256c), abc = N, i.e. 512^3 at N = 8 -- and one step is ONE call of the library's exchange step: every rank
samples its part of the lattice in --pieces C pieces and the parts are exchanged while the next piece is sampled.  Five forms
exist (host vector, slabs, in place, p2p, copy); by default EVERY form is measured -- its own warm-up and
--steps timed steps between barriers, max over ranks, each form under a watchdog -- and `value` is the fastest
(config.exchange lists them all).  value = total nodes / time.

On the JSON line besides the contract's keys:
  roofline      K1 is bound by VALU issue, not by HBM: `frac` = fraction of the VALU issue cycles of the
                1024 SIMDs that were busy (PMC), `hbm` = measured HBM traffic per launch / kernel time
                against the 8 TB/s peak, next to the compulsory traffic (8 B/node) and a copy kernel's
                bandwidth measured here.  Counter-derived numbers come from profiles/counters.json, which
                profiles/collect.sh writes together with a hash of discregrid_amd/csrc: if the sources
                changed since, they are null ("stale").  `algorithmic_gbs` (the reference traversal's
                bytes, SURVEY.md 8(d), / kernel time) is informational and is NOT a fraction of anything.
                The figures an API user sees are repeated INSIDE roofline as plain scalars (d2h_mnodes_s, host_ready_ms,
                k2_*_frac, k3_seconds, k1_*_ms ...): the driver's record keeps roofline and drops the nested objects below.
  value_with_d2h   Mnodes/s of dg_sdf_sample_nodes: kernels + D2H into the caller's host array.
  addfunction_e2e  the C++ CubicLagrangeDiscreteGrid::addFunction(MeshSDF) call, wall time.
  secondary     K2 (10 M queries, uniform and SPH-like shell, value / value+gradient) and K3 (density
                map on the 256^3 SDF) at BASELINE configs[4] sizes.
  cpu_baseline  the unmodified reference (oracle/_ref, kind "reference") or this repo's CPU
                restatement (kind "port") timed on this box's host cores on a bounded,
                evenly spread sample of the same lattice (rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMDS = 1024         # 256 CUs x 4


def grid_for(n_gpus, scaling="weak"):
    """weak: the lattice grows with N (256 a x 256 b x 256 c, abc = N: 512^3 at N = 8 = BASELINE configs[3]);
    strong: the 256^3 lattice the metric is quoted on at every N"""
    dims = [256, 256, 256]
    k, axis = (n_gpus if scaling == "weak" else 1), 2
    if n_gpus < 1 or (n_gpus & (n_gpus - 1)):
        raise SystemExit("--gpus must be a power of two")
    while k > 1:
        if k % 2:
            raise SystemExit("--gpus must be a power of two")
        dims[axis] *= 2
        axis = (axis - 1) % 3
        k //= 2
    return dims


def csrc_hash():
    """SHA-256 over the kernel / ABI sources: ties profiles/counters.json to the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "discregrid_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # (not the host-only sources: the copy pipeline, the CPU point query and the exchange glue do not change what the kernels do)
        if f.endswith((".hip", ".h", ".cpp")) and f not in ("dg_capi_host.cpp", "dg_host_query.cpp", "dg_capi_comm.cpp", "dg_capi_hostfield.cpp", "dg_capi_vmm.h", "dg_capi_shm.h"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def load_counters():
    p = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(p):
        return None, "profiles/counters.json missing"
    with open(p) as f:
        c = json.load(f)
    if c.get("csrc_sha256") != csrc_hash():
        return None, "stale: discregrid_amd/csrc changed since profiles/collect.sh ran"
    return c, "profiles/counters.json (%s)" % c.get("collected", "?")


def load_balg():
    with open(os.path.join(ROOT, "profiles", "balg_icosphere71_256.json")) as f:
        return json.load(f)


def cpu_baseline(V, F, dom, res, budget_s):
    """Reference (or port) node loop on the host cores over an evenly spread sample."""
    import dgtest as T
    n = T.n_nodes(res)
    if T.ref_available():
        kind = "reference"
        g = T.RefGrid(V, F, dom, res)

        def run(b, e):
            g.sample_nodes(b, e)
            return g.last_seconds
    else:
        kind = "port"
        om = T.OracleMesh(V, F)

        def run(b, e):
            om.sample_nodes(dom, res, b, e)
            return om.last_seconds
    n_chunks = 32
    starts = [int((i + 0.5) * n / n_chunks) for i in range(n_chunks)]
    # Sized in two stages towards 40 % of the lattice within budget_s seconds (a small probe underestimates the rate -- thread
    # start-up -- by a factor that varied between 1 and 2.6 from run to run: 14 % ... 44 % of the lattice for the same budget):
    # probe; a tenth of the lattice (less if the probe says that alone would take a third of the budget); then the rest of the
    # 40 %, scaled down only if the rate MEASURED on the first stage says the budget would be exceeded.  Both stages count.
    probe = 4096
    t = sum(run(s, s + probe) for s in starts)
    rate = n_chunks * probe / max(t, 1e-9)
    cap = n // n_chunks // 2        # (a run starts in the middle of its slot of n / 32 nodes and stays inside it)
    first = int(min(max(min(0.10 * n, rate * budget_s / 3.0) / n_chunks, probe), cap))
    t = sum(run(s, min(n, s + first)) for s in starts)
    nodes = sum(min(n, s + first) - s for s in starts)
    rate = nodes / max(t, 1e-9)
    want = 0.40 * n - nodes
    second = int(min(max(min(want, rate * max(budget_s - t, 0.0)) / n_chunks, 0), cap - first))
    if second > 0:
        t += sum(run(s + first, min(n, s + first + second)) for s in starts)
        nodes += sum(min(n, s + first + second) - min(n, s + first) for s in starts)
    per_chunk = first + second
    return {
        "value": nodes / t / 1e6, "unit": "Mnodes/s", "cores": os.cpu_count(), "kind": kind,
        "sample_nodes": nodes, "sample_fraction": nodes / n, "sample_seconds": t,
        "sample": "%d nodes = %d evenly spaced runs of %d consecutive lattice nodes of the same %s grid, "
                  "OpenMP schedule(static), %.1f s" % (nodes, n_chunks, per_chunk, "x".join(map(str, res)), t),
    }


def timed(torch, stream, fn, reps):
    """mean device time of fn() in ms, HIP events on the launch stream"""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def extras(torch, dg, T, mesh, grid, field, V, F, dom, res, kernel_ms):
    """The figures SURVEY.md 8(d) asks for beside the kernel rate (N = 1): D2H-inclusive rate, the C++
    addFunction call end to end, a copy kernel's HBM bandwidth, K2 and K3 at BASELINE configs[4] sizes."""
    out = {}
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    n_nodes = dg.n_nodes(grid)
    # -- achievable HBM bandwidth: a device-to-device copy of 1 GiB (read + write counted)
    src = torch.empty(1 << 27, dtype=torch.float64, device="cuda").normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    ms = timed(torch, stream, lambda: dst.copy_(src), 5)
    out["hbm_copy_gbs"] = 2 * src.numel() * 8 / (ms * 1e-3) / 1e9
    del src, dst
    # -- K1 + D2H into a host array the caller owns (fresh memory every time, like addFunction's vector)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        host = mesh.sample_nodes(grid)
        best = min(best, time.perf_counter() - t0)
        assert host[0] > 0 and np.isfinite(host[::100003]).all()
        del host
    out["value_with_d2h"] = {"value": n_nodes / best / 1e6, "unit": "Mnodes/s", "ms": best * 1e3,
                             "what": "dg_sdf_sample_nodes: K1 + D2H into a freshly allocated pageable host array (best of 3)"}
    # -- the C++ API call a Discregrid user makes
    exe = os.path.join(ROOT, "tests", "cpp", "build", "unchanged_caller")
    if os.path.exists(exe):
        obj = "/tmp/dg_bench_ico71.obj"
        T.write_obj(obj, V, F)
        try:
            j = json.loads(subprocess.check_output([exe, "addfunction", obj, " ".join(map(str, res)), "4"], timeout=300).decode())
            best = min(j["calls"][1:], key=lambda c: c["total_s"])
            out["addfunction_e2e"] = {
                # value / ratio_to_kernel: until the HOST vector is complete -- what the reference guarantees when addFunction
                # returns (comparable with rounds 1 and 2); the device-side figures are secondary
                "value": n_nodes / best["total_s"] / 1e6, "unit": "Mnodes/s",
                "host_ready_ms": best["total_s"] * 1e3, "ratio_to_kernel": best["total_s"] * 1e3 / kernel_ms,
                "return_ms": best["return_s"] * 1e3, "device_ready_ms": best["device_ready_s"] * 1e3,
                "ratio_to_kernel_device_ready": best["device_ready_s"] * 1e3 / kernel_ms,
                "first_call_ms": j["calls"][0]["total_s"] * 1e3,
                "what": "CubicLagrangeDiscreteGrid::addFunction(MeshSDF) on a fresh grid (C++, the best of calls 2-4; the first call of a "
                        "process also sets up streams and buffers).  host_ready_ms (= value) = until the host vector is complete "
                        "(waitForHostData: what the first scalar interpolate / save / nodeData waits for, and what the reference's "
                        "addFunction means by returning).  The field is produced into a device array its handle owns, in seven chunks whose "
                        "copies run under the following chunks; the call itself returns once the work is enqueued (return_ms), and "
                        "device_ready_ms = until a GPU-side consumer of the WHOLE field (a batched interpolate) has run -- what a "
                        "following addDensityMap / batched query waits for"}
        except Exception as e:  # noqa: BLE001  (a missing / failing driver must not void the headline number)
            out["addfunction_e2e"] = {"error": str(e)[:200]}
    else:
        out["addfunction_e2e"] = None
    # -- K2: batched interpolate, 10 M queries on the field just sampled (BASELINE configs[4])
    fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=n_nodes)
    nq = 10_000_000
    P = torch.from_numpy(T.uniform_points(1234, nq, dom[:3], dom[3:])).cuda()
    C = torch.from_numpy(T.uniform_points(4321, 26_000_000, dom[:3], dom[3:])).cuda()
    phic = torch.empty(len(C), dtype=torch.float64, device="cuda")
    fld.interpolate_device(C.data_ptr(), len(C), phic.data_ptr(), stream=s)
    S = C[(phic.abs() < 0.2)][:nq].contiguous()
    del C, phic
    phi = torch.empty(nq, dtype=torch.float64, device="cuda")
    grad = torch.empty(3 * nq, dtype=torch.float64, device="cuda")
    k2 = {}
    for name, Q in (("uniform", P), ("shell", S)):
        if len(Q) < nq:
            continue
        for g in (False, True):
            fn = (lambda Q=Q, g=g: fld.interpolate_device(Q.data_ptr(), nq, phi.data_ptr(), grad.data_ptr() if g else 0, stream=s))
            fn()
            ms = timed(torch, stream, fn, 5)
            bytes_q = 312 if g else 288
            k2["%s_%s" % (name, "grad" if g else "value")] = {
                "gq_s": nq / (ms * 1e-3) / 1e9, "ms": ms, "algorithmic_gbs": nq * bytes_q / (ms * 1e-3) / 1e9,
                "hbm_frac_algorithmic": nq * bytes_q / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    k2["what"] = ("10 M queries (std::mt19937_64 seed 1234 uniform; |phi| < 2h shell, h = 0.1) on the 256^3 field, device-resident, unordered input incl. "
                  "the on-device sort by tile of 8^3 cells and the LDS-staged gather (k_interpolate_tiles, round 6)")
    # the same with the optional tile-major copy of the field (dg_field_build_tile_major: 1.5 GB, built once per field)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fld.build_tile_major(s)
    torch.cuda.synchronize()
    k2["tile_major_build_ms"] = (time.perf_counter() - t0) * 1e3
    for name, Q in (("uniform", P), ("shell", S)):
        if len(Q) < nq:
            continue
        fn = (lambda Q=Q: fld.interpolate_device(Q.data_ptr(), nq, phi.data_ptr(), 0, stream=s))
        fn()
        ms = timed(torch, stream, fn, 5)
        k2["%s_value_tile_major" % name] = {"gq_s": nq / (ms * 1e-3) / 1e9, "ms": ms, "hbm_frac_algorithmic": nq * 288 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    fld.drop_tile_major()
    # ... and with the optional CELL-major copy (dg_field_build_cell_major: 256 B per cell, 4.3 GB at 256^3, built once per
    # field): one contiguous row per query, fetched cooperatively by the wave -- no binning, the order does not matter
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fld.build_cell_major(s)
    torch.cuda.synchronize()
    k2["cell_major_build_ms"] = (time.perf_counter() - t0) * 1e3
    k2["cell_major_bytes"] = 256 * int(np.prod(res))
    for name, Q in (("uniform", P), ("shell", S)):
        if len(Q) < nq:
            continue
        for g in (False, True):
            fn = (lambda Q=Q, g=g: fld.interpolate_device(Q.data_ptr(), nq, phi.data_ptr(), grad.data_ptr() if g else 0, stream=s))
            fn()
            ms = timed(torch, stream, fn, 5)
            bytes_q = 312 if g else 288
            k2["%s_%s_cell_major" % (name, "grad" if g else "value")] = {
                "gq_s": nq / (ms * 1e-3) / 1e9, "ms": ms, "hbm_frac_algorithmic": nq * bytes_q / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    fld.drop_cell_major()
    # ... and with the BAND-LIMITED cell-major copy (round 4, dg_field_build_cell_major_band): rows only for the cells that reach into
    # |phi| <= 2h + cell diagonal -- what SPH boundary handling and GenerateDensityMap query -- the plain gather for the rest, one launch
    diag = float(np.linalg.norm((dom[3:] - dom[:3]) / np.array(res, dtype=np.float64)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    band_rows = fld.build_cell_major_band(-(0.2 + diag), 0.2 + diag, s)
    torch.cuda.synchronize()
    k2["band_copy"] = {"build_ms": (time.perf_counter() - t0) * 1e3, "band": [-(0.2 + diag), 0.2 + diag], "rows": band_rows,
                       "fraction_of_cells": band_rows / float(np.prod(res)),
                       "copy_bytes": band_rows * 256 + 12 * (int(np.prod(res)) // 64 + 1),
                       "copy_bytes_over_field_bytes": (band_rows * 256 + 12 * (int(np.prod(res)) // 64 + 1)) / (8.0 * n_nodes)}
    for name, Q in (("uniform", P), ("shell", S)):
        if len(Q) < nq:
            continue
        for g in (False, True):
            fn = (lambda Q=Q, g=g: fld.interpolate_device(Q.data_ptr(), nq, phi.data_ptr(), grad.data_ptr() if g else 0, stream=s))
            fn()
            torch.cuda.synchronize()   # (the routing probe's verdict -- how many of the batch's queries map into the band -- reaches the
            fn()                       # host through pinned memory: the calls enqueued after this point are routed by it)
            torch.cuda.synchronize()
            ms = timed(torch, stream, fn, 5)
            bytes_q = 312 if g else 288
            k2["%s_%s_band_copy" % (name, "grad" if g else "value")] = {
                "gq_s": nq / (ms * 1e-3) / 1e9, "ms": ms, "hbm_frac_algorithmic": nq * bytes_q / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    fld.drop_cell_major()
    # queries that arrive sorted by cell (what a caller with spatially sorted particles hands over), plain layout
    h3 = torch.tensor((dom[3:] - dom[:3]) / np.array(res, dtype=np.float64), device="cuda")
    cell = ((P - torch.tensor(dom[:3], device="cuda")) / h3).floor().clamp(0, res[0] - 1).long()
    Ps = P[torch.argsort((cell[:, 2] * res[1] + cell[:, 1]) * res[0] + cell[:, 0])].contiguous()
    del cell
    for g in (False, True):
        fn = (lambda g=g: fld.interpolate_device(Ps.data_ptr(), nq, phi.data_ptr(), grad.data_ptr() if g else 0, stream=s))
        fn()
        torch.cuda.synchronize()   # the probe's verdict ("ordered") reaches the host through pinned memory ...
        fn()                       # ... so that this and the timed calls launch no sort
        torch.cuda.synchronize()
        ms = timed(torch, stream, fn, 5)
        bytes_q = 312 if g else 288
        k2["uniform_sorted_%s" % ("grad" if g else "value")] = {"gq_s": nq / (ms * 1e-3) / 1e9, "ms": ms,
                                                                "hbm_frac_algorithmic": nq * bytes_q / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del Ps
    # ... and in z-order of their cells (what SPlisHSPlasH's z-sort hands over): lanes side by side sit in neighbouring cells in all three directions
    cell = ((P - torch.tensor(dom[:3], device="cuda")) / h3).floor().clamp(0, res[0] - 1).long()
    z = torch.zeros(nq, dtype=torch.long, device="cuda")
    for b in range(10):
        for d in range(3):
            z |= ((cell[:, d] >> b) & 1) << (3 * b + d)
    Pz = P[torch.argsort(z)].contiguous()
    del cell, z
    for g in (False, True):
        fn = (lambda g=g: fld.interpolate_device(Pz.data_ptr(), nq, phi.data_ptr(), grad.data_ptr() if g else 0, stream=s))
        fn()
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        ms = timed(torch, stream, fn, 5)
        bytes_q = 312 if g else 288
        k2["uniform_zorder_%s" % ("grad" if g else "value")] = {"gq_s": nq / (ms * 1e-3) / 1e9, "ms": ms,
                                                                "hbm_frac_algorithmic": nq * bytes_q / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del Pz
    out_secondary = {"k2_interpolate": k2}
    del P, S, phi, grad
    # -- K1 on the other judged lattices (BASELINE configs[1] and the lattice of configs[3] on ONE GPU), device-resident
    k1s = {}
    for name, make, r in (("bunny128", T.bunny_mesh, 128), ("bunny256", T.bunny_mesh, 256), ("dragon256", T.dragon_mesh, 256), ("ico512", None, 512)):
        try:
            Vm, Fm = (V, F) if make is None else make()
            m2 = mesh if make is None else dg.Mesh(Vm, Fm)
            d2 = dg.default_domain(Vm)
            g2 = dg.grid_desc(d2[:3], d2[3:], [r] * 3)
            n2 = dg.n_nodes(g2)
            buf = torch.empty(n2, dtype=torch.float64, device="cuda")
            fn = (lambda: m2.sample_nodes_device(g2, 0, n2, buf.data_ptr(), stream=s))
            fn()
            ms = timed(torch, stream, fn, 3)
            k1s[name] = {"ms": ms, "mnodes_s": n2 / ms / 1e3, "nodes": n2, "triangles": len(Fm)}
            del buf
        except Exception as e:  # noqa: BLE001
            k1s[name] = {"error": str(e)[:200]}
    out_secondary["k1"] = k1s
    # -- K3: density map on the same SDF (GenerateDensityMap's node function), whole lattice
    dens = torch.empty(n_nodes, dtype=torch.float64, device="cuda")
    fld.density_map_nodes_device(0.1, 1000.0, True, 0, min(n_nodes, 1 << 20), dens.data_ptr(), stream=s)   # code load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fld.density_map_nodes_device(0.1, 1000.0, True, 0, n_nodes, dens.data_ptr(), stream=s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    integrated = int(((dens != np.finfo(np.float64).max) & (dens != 0.0)).sum().item())
    out_secondary["k3_density_map"] = {
        "seconds": dt, "mnodes_s": n_nodes / dt / 1e6, "nodes_integrated": integrated,
        "ginterpolations_s": integrated * 4097 / dt / 1e9,
        "what": "dg_density_map_nodes_device, h = 0.1, band predicate, all %d nodes of the 256^3 SDF; an integrated node is 1 + 4096 "
                "interpolations in the reference's count" % n_nodes}
    counters, _ = load_counters()
    k3c = ((counters or {}).get("workloads", {}).get("k3") or {})
    k3name = next((k for k in k3c if k.startswith("k_density_cells")), None) or next((k for k in k3c if k.startswith("k_density_rows")), None) or \
        next((k for k in k3c if k.startswith("k_density_pairs")), None) or next((k for k in k3c if k.startswith("k_density_bricks")), None)
    k3k = k3c.get(k3name) if k3name else None
    out_secondary["k3_density_map"]["roofline"] = None if k3k is None else {
        "kernel": k3name,
        "bound": "f64 VALU issue and the vector memory pipeline together (k_density_cells: a lane owns a lattice point with its seven "
                 "nodes -- 3 cell fetches of 256 B per 7 nodes and quadrature point instead of 5, one sweep of the field for all node "
                 "classes; both units near 0.87-0.9 busy at the 3 waves per SIMD the registers allow).  The floor of this formulation: TD "
                 "needs 22 cycles per load instruction where 16 would do (the 256-byte runs of a wave start at arbitrary 16-byte offsets "
                 "and straddle an extra 128-byte line two times out of three: 38 of 64 B/clk/CU delivered); with VALU at 0.87 removing "
                 "that entirely is worth <= 13 %; traffic is 197 x the compulsory bytes because every quadrature point re-fetches its "
                 "cell through L2 (hit rate 0.90), not because HBM binds (0.16 of its peak)",
        # what the launch has to move at least (field read once + x-major copy of the Y / Z classes written and read + result written)
        # against what the HBM counters saw
        "compulsory_bytes": int(8 * n_nodes * (1 + 2 * 0.57 + 1)), "traffic": k3k.get("hbm_bytes_per_launch"),
        "arith_frac": k3k.get("arith_frac"), "salu_slot_frac": k3k.get("salu_slot_frac"),
        "achieved": max(k3k.get("td_busy") or 0.0, k3k.get("ta_busy") or 0.0, k3k["valu_busy"]), "peak": 1.0,
        "unit": "busy fraction of the busiest unit (TA / TD / VALU)",
        "frac": max(k3k.get("td_busy") or 0.0, k3k.get("ta_busy") or 0.0, k3k["valu_busy"]),
        "ta_busy": k3k.get("ta_busy"), "td_busy": k3k.get("td_busy"),
        "valu_busy": k3k["valu_busy"], "hbm_frac": k3k["hbm_frac"], "l2_hit_rate": k3k["l2_hit_rate"],
        "valu_per_wave": k3k["per_wave"]["valu"], "kernel_ms_when_profiled": k3k["kernel_ms"], "replayed": True,
        "td_cycles_per_load_instruction": k3k.get("td_cycles_per_load_instruction")}
    k2bc = ((counters or {}).get("workloads", {}).get("k2b") or {})
    k2bk = next((v for k, v in k2bc.items() if k.startswith("k_interpolate_band<false")), None)
    k2["roofline_band_kernel"] = None if k2bk is None else {
        "bound": "hbm", "achieved": k2bk["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2bk["hbm_frac"],
        "traffic": k2bk["hbm_bytes_per_launch"], "algorithmic_bytes": 288 * nq, "replayed": True,
        "what": "k_interpolate_band, 10 M shell queries (value), through the band-limited cell-major copy.  traffic / algorithmic_bytes = 1.36: "
                "256 B of row + 24 B of point + 8 B of result are the algorithmic 288 B; on top come the look-up's bit and rank words (two "
                "loads per query from 3 MB of tables that do not all stay in L2 beside 2.9 GB of streaming rows: ~64 B of sector traffic per "
                "query) and the sectors of rows that straddle two 128-byte lines as seen by their four 64-byte quarter fetches"}
    k2p = ((counters or {}).get("workloads", {}).get("k2") or {})
    k2t = next((v for k, v in k2p.items() if k.startswith("k_interpolate_tiles<false")), None)
    k2["roofline_tiles_kernel"] = None if k2t is None else {
        "bound": "latency of a block's load chain (item -> query index -> point, rows; 4 blocks per CU by the 37 KB LDS image)", "achieved": k2t["hbm_gbs"],
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2t["hbm_frac"], "traffic": k2t["hbm_bytes_per_launch"], "algorithmic_bytes": 288 * nq,
        "valu_busy": k2t.get("valu_busy"), "l2_hit_rate": k2t.get("l2_hit_rate"), "kernel_ms_when_profiled": k2t["kernel_ms"], "replayed": True,
        "what": "k_interpolate_tiles (round 6), 10 M uniform queries (value), plain layout: the staged gather alone, behind keys + two radix passes + "
                "bounds + items (0.30 ms).  traffic < algorithmic_bytes: a tile's rows are fetched once for all its queries"}
    k2c = ((counters or {}).get("workloads", {}).get("k2r") or {})
    k2k = next((v for k, v in k2c.items() if k.startswith("k_interpolate_rows<false")), None)
    k2["roofline_rows_kernel"] = None if k2k is None else {
        "bound": "hbm", "achieved": k2k["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2k["hbm_frac"],
        "traffic": k2k["hbm_bytes_per_launch"], "algorithmic_bytes": 288 * nq, "replayed": True}
    if T.ref_available() or True:
        out_secondary["cpu_baseline"] = cpu_baseline_k2_k3(T, dom, res, field, 0.1, 1000.0)
    out["secondary"] = out_secondary
    fld.close()
    return out


def cpu_baseline_k2_k3(T, dom, res, field, h, rho0):
    """The reference's own interpolate (cubic_lagrange_discrete_grid.cpp:977-1063) and the density-map node function
    (cmd/generate_density_map/main.cpp:86-133) timed on this box's host cores, on bounded samples of the same
    workloads (kind "reference": oracle/_ref, the unmodified sources; "port": this repo's CPU restatement)."""
    coeffs = field.cpu().numpy()
    n = len(coeffs)
    out = {"cores": os.cpu_count()}
    nq = 10_000_000
    P = T.uniform_points(1234, nq, dom[:3], dom[3:])
    n_runs, per_run = 16, 8192      # (runs of 128 chunks of the dynamic schedule each: every core gets work)
    starts = [int((i + 0.5) * n / n_runs) for i in range(n_runs)]
    if T.ref_available():
        out["kind"] = "reference"
        g = T.RefGrid(None, None, dom, res)
        t0 = time.perf_counter()
        assert g.add_coeffs(coeffs) == 0
        out["reference_cell_table_build_s"] = time.perf_counter() - t0   # (the reference materialises its 32-index table, :833-891)
        g.interpolate(P[:1 << 16])
        g.interpolate(P)
        t_v = g.last_seconds
        g.interpolate(P, grad=True)
        t_g = g.last_seconds
        t_d, nodes, integrated = 0.0, 0, 0
        for b in starts:
            v = g.density_nodes(h, rho0, b, b + per_run)
            t_d += g.last_seconds
            nodes += per_run
            integrated += int(((v != np.finfo(np.float64).max) & (coeffs[b:b + per_run] <= 2 * h)).sum())
    else:
        out["kind"] = "port"
        t0 = time.perf_counter()
        T.oracle_interpolate(dom, res, coeffs, P)
        t_v = time.perf_counter() - t0
        t0 = time.perf_counter()
        T.oracle_interpolate(dom, res, coeffs, P, grad=True)
        t_g = time.perf_counter() - t0
        t_d, nodes, integrated = 0.0, 0, 0
        for b in starts:
            v = T.oracle_density_map(dom, res, coeffs, h, rho0, band=True, begin=b, end=b + per_run)
            t_d += T.oracle_density_map.last_seconds
            nodes += per_run
            integrated += int(((v != np.finfo(np.float64).max) & (coeffs[b:b + per_run] <= 2 * h)).sum())
    out["k2_interpolate"] = {"value": nq / t_v / 1e6, "value_with_gradient": nq / t_g / 1e6, "unit": "Mqueries/s",
                             "sample": "the %d uniform queries (std::mt19937_64 seed 1234), omp parallel for, %.2f / %.2f s" % (nq, t_v, t_g)}
    out["k3_density_map"] = {"value": nodes / t_d / 1e6, "unit": "Mnodes/s", "ginterpolations_s": integrated * 4097 / t_d / 1e9,
                             "sample": "%d nodes = %d evenly spaced runs of %d consecutive lattice nodes of the 256^3 SDF (%d integrated), "
                                       "omp dynamic, %.2f s" % (nodes, n_runs, per_run, integrated, t_d)}
    return out


def user_facing_scalars(out):
    """The figures an API user sees, as plain scalars INSIDE `roofline` (the driver's record keeps config / roofline / cpu_baseline
    of the line and drops the nested secondary / addfunction_e2e objects)."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return d
    k2 = get(out, "secondary", "k2_interpolate") or {}
    k3 = get(out, "secondary", "k3_density_map") or {}
    k1 = get(out, "secondary", "k1") or {}
    traffic, compulsory = get(k3, "roofline", "traffic"), get(k3, "roofline", "compulsory_bytes")
    return {
        "d2h_mnodes_s": get(out, "value_with_d2h", "value"),            # SURVEY 8(d)'s metric: K1 + D2H into the caller's host vector
        "host_ready_ms": get(out, "addfunction_e2e", "host_ready_ms"),   # C++ addFunction(MeshSDF) until the host vector is complete
        "device_ready_ms": get(out, "addfunction_e2e", "device_ready_ms"),
        "k2_rows_frac": get(k2, "uniform_value_cell_major", "hbm_frac_algorithmic"),   # 10 M uniform queries, cell-major copy: 288 B / query / 8 TB/s
        "k2_band_frac": get(k2, "shell_value_band_copy", "hbm_frac_algorithmic"),      # 10 M shell queries, band-limited copy
        "k2_band_uniform_gq_s": get(k2, "uniform_value_band_copy", "gq_s"),            # ... and uniform queries through the same copy (43 % miss the band)
        "k2_plain_frac": get(k2, "uniform_value", "hbm_frac_algorithmic"),             # plain layout incl. the on-device sort (round 6: by tile + staged gather)
        "k2_plain_gq_s": get(k2, "uniform_value", "gq_s"), "k2_plain_grad_gq_s": get(k2, "uniform_grad", "gq_s"),
        "k2_plain_shell_gq_s": get(k2, "shell_value", "gq_s"),
        "k2_plain_zorder_gq_s": get(k2, "uniform_zorder_value", "gq_s"),                # plain layout, queries in z-order of their cells (sorted particles)
        "k2_tiles_kernel_hbm_frac": get(k2, "roofline_tiles_kernel", "frac"),
        "k3_seconds": get(k3, "seconds"),                                              # density map of the 256^3 SDF, whole lattice
        "k3_td_busy": get(k3, "roofline", "td_busy"),
        "k3_traffic_over_compulsory": (traffic / compulsory) if traffic and compulsory else None,
        "k1_bunny128_ms": get(k1, "bunny128", "ms"), "k1_bunny256_ms": get(k1, "bunny256", "ms"),
        "k1_dragon256_ms": get(k1, "dragon256", "ms"), "k1_ico512_ms": get(k1, "ico512", "ms"),
    }


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n_gpus, argv, port):
    """the command `python bench.py --gpus N ...` re-executes itself as when nobody started ranks for it: the driver's own form"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """--gpus N > 1 without a launcher: start the N ranks here.  stdout of the ranks is filtered -- the LAST line that parses as
    the bench's JSON record is printed as the only stdout line once the ranks have ended, everything else goes to stderr as it
    comes -- and the ranks' exit code is returned (3 if they ended well without a line)."""
    import signal
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = launch_command(n_gpus, argv, free_port())
    print("bench.py: --gpus %d without WORLD_SIZE in the environment: starting the ranks myself:\n  %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)

    def forward(signum, _frame):       # (a driver that gives up on us must not leave ranks sitting on the GPUs)
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        finally:
            os._exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, forward)
    line = None
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        rec = None
        if ln.startswith("{"):
            try:
                rec = json.loads(ln)
            except ValueError:
                rec = None
        if isinstance(rec, dict) and "metric" in rec and "value" in rec:
            line = ln
        elif ln:
            print(ln, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 3
    return rc


def run_preflight(dist, rank, world, timeout_s):
    """tools/scale_preflight.py in a child process per rank (a step that hangs is killed with the child, not with the bench);
    returns ({form: reason} for the forms that cannot run on SOME rank, this rank's step results)"""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_preflight as pf
    box = [None]
    if rank == 0:
        box[0] = (free_port(), "%d_%d" % (os.getpid(), int(time.time())))
    dist.broadcast_object_list(box, src=0)
    port, tag = box[0]
    out = os.path.join(tempfile.gettempdir(), "dg_preflight_%s_%d.json" % (tag, rank))
    # (not the launcher's store: under torch.distributed.run TORCHELASTIC_USE_AGENT_STORE makes init_process_group look for the agent's
    # TCPStore at MASTER_PORT -- the children meet at a port of their own, where rank 0 has to host the store itself)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1")
    results, note = {}, None
    try:
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "scale_preflight.py"), "--json-out", out, "--tag", tag],
                                env=env, stdout=sys.stderr, start_new_session=True)
        try:
            proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            note = "killed after %d s" % timeout_s
        if os.path.exists(out):
            with open(out) as f:
                results = json.load(f)
            os.unlink(out)
    except Exception as exc:  # noqa: BLE001  (a preflight that cannot run says nothing; the race decides)
        note = "%s: %s" % (type(exc).__name__, exc)
    mine = {}
    for form in pf.forms_removed(results):
        why = [st for st in pf.STEPS if form in pf.FORMS_NEEDING[st] and (results.get(st) is None or results[st].get("ok") is False)]
        mine[form] = "preflight step %s on rank %d: %s" % (why[0], rank, (results.get(why[0]) or {}).get("detail") or note or "not reached")
    if note:
        print("bench.py rank %d: preflight %s" % (rank, note), file=sys.stderr, flush=True)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    removed = {}
    for d in everyone:
        for form, why in (d or {}).items():
            removed.setdefault(form, why)
    return removed, results


class Watchdog:
    """N > 1: no exchange form has ever run on more than one GPU where this was developed, and a rank that fails inside a
    collective leaves its peers waiting for ever.  Every candidate form therefore runs under a deadline: when it expires, rank 0
    prints the line of the best form measured SO FAR (saying which form was cut off) and every rank leaves at once -- a scaling
    run returns a number as long as one form completed."""

    def __init__(self, seconds, rank, state):
        self.seconds, self.rank, self.state, self.timer = seconds, rank, state, None

    def _fire(self, what):
        if self.rank == 0:
            line = self.state.get("line")
            if line is not None:
                line["config"]["exchange"]["watchdog"] = "%s did not complete within %d s; the forms measured before it are reported" % (what, self.seconds)
                print(json.dumps(line), flush=True)
            print("bench.py watchdog: %s did not complete within %d s%s" % (what, self.seconds, "" if line else "; nothing was measured before it"),
                  file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(0 if self.state.get("line") is not None or self.rank != 0 else 3)

    def arm(self, what):
        import threading
        self.disarm()
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._fire, [what])
            self.timer.daemon = True
            self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak (default) = 118.4 M nodes per GPU, the lattice grows with N (512^3 at N = 8: BASELINE configs[3]); "
                         "strong = the metric's own 256^3 lattice at every N (1.8 ms of kernel per GPU at N = 8: launch, barrier and "
                         "exchange latency decide)")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip tools/scale_preflight.py (run by default before the race, stderr only)")
    ap.add_argument("--preflight-timeout", type=int, default=150, help="N > 1: seconds the preflight child may take before it is killed")
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="ceiling of the cpu_baseline leg, which samples 40 %% of the lattice (about 30 s on 256 cores; 0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="only the timed K1 steps (profiling runs)")
    ap.add_argument("--pieces", type=int, default=4,
                    help="N > 1: issue the exchange in this many pieces, overlapped with the sampling kernel")
    ap.add_argument("--force-shard-path", action="store_true",
                    help="run the N > 1 protocol (communicator, shards, all-gather, unpack) even at N = 1 (self-test)")
    ap.add_argument("--exchange", choices=["auto", "host", "copy-shm", "slabs", "inplace", "inplace-p2p", "copy", "to-root"], default="auto",
                    help="N > 1: auto (default) = run --warmup + --steps steps of EVERY form, each under --form-timeout, and report the "
                         "fastest (the others' times are on the line).  host = every rank copies its chunks into a shared-memory host "
                         "vector (no RCCL, no device IPC: cannot fail for lack of either; measured first); copy-shm = the copy form with "
                         "its control plane in shared memory (dg_comm_create_shm: the whole field on every GPU without RCCL); slabs = interleaved 4-plane "
                         "slabs, packed buffers, all-gather, unpack; inplace = contiguous chunks cut by measured cost, sampled into "
                         "place and exchanged with grouped broadcasts (no unpack pass, no scratch); inplace-p2p: the same with "
                         "send / recv pairs; copy: the same chunks pushed into the peers' fields by the copy engines (fields from "
                         "dg_comm_field_alloc, mapped by the peers chunk by chunk; no collective kernel beside the sampling); "
                         "to-root: only rank 0 gets the whole field")
    ap.add_argument("--form-timeout", type=int, default=120, help="N > 1: seconds a candidate form may take before the run is cut short")
    ap.add_argument("--python-gather", action="store_true",
                    help="N > 1: drive the pieces from here with torch.distributed's all_gather instead of the library's "
                         "dg_sdf_sample_allgather_device (A/B, and the one-GPU self-test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        grid_for(args.gpus)                       # (a bad --gpus fails here, not N times)
        if os.environ.get("DG_BENCH_DRY_RANKS") != "1":
            import torch
            have = torch.cuda.device_count()
            if have < (1 if os.environ.get("DG_BENCH_SELFTEST_ONE_GPU") == "1" else args.gpus):
                raise SystemExit("bench.py: --gpus %d but %d HIP device(s) visible (the product has no CPU path; DG_BENCH_SELFTEST_ONE_GPU=1 runs "
                                 "the N > 1 protocol with every rank on device 0)" % (args.gpus, have))
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts
    import torch
    import torch.distributed as dist
    if os.environ.get("DG_BENCH_DRY_RANKS") == "1":
        # (tests without a GPU: the launcher plumbing only -- ranks meet, chatter on stdout, rank 0 prints a record marked dry_run)
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.ones(1)
        dist.all_reduce(t)
        print("rank %s chatter on stdout" % os.environ["RANK"], flush=True)
        if os.environ["RANK"] == "0":
            print(json.dumps({"metric": "dry run of the launcher, no measurement", "value": 0.0, "dry_run": True, "n_gpus": int(t.item()),
                              "scaling": args.scaling, "grid": grid_for(args.gpus, args.scaling)}), flush=True)
        dist.destroy_process_group()
        sys.exit(int(os.environ.get("DG_BENCH_DRY_RC", "0")))
    import dgtest as T
    import discregrid_amd as dg

    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no HIP device visible -- the product has no CPU path (tests/ -m 'not gpu' is what runs without one)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Self-test of the N > 1 protocol on a box with ONE GPU: all ranks share device 0 and the collective
    # goes through gloo (RCCL refuses two ranks on one device).  Timings of such a run mean nothing.
    selftest = os.environ.get("DG_BENCH_SELFTEST_ONE_GPU") == "1"
    if selftest:
        local_rank = 0
        args.python_gather = True
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dg.load_library()
    dg.set_device(local_rank)
    sharded = world > 1 or args.force_shard_path
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # The bench's OWN collectives (barriers, max over ranks, shared timings) act on CPU tensors and go through gloo: they
        # must work whatever state RCCL is in.  RCCL is what the library's communicator uses (dg_comm_create, the product),
        # and what torch.distributed falls back to for CUDA tensors if that communicator cannot be created.
        dist.init_process_group("gloo" if selftest else "cpu:gloo,cuda:nccl", rank=rank, world_size=world)

    def ctl_barrier():
        dist.all_reduce(torch.zeros(1))

    def ctl_max(x):
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    V, F = T.icosphere(71)
    dom = dg.default_domain(V)            # cmd/generate_sdf/main.cpp:83-91
    res = grid_for(world, args.scaling)
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n_nodes = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream

    field = torch.empty(n_nodes, dtype=torch.float64, device="cuda")
    comm = None          # the library's RCCL communicator
    comm_ext = None      # one-GPU self-test: the library's communicator with gloo as its control plane (copy form only)
    comm_shm = [None]    # the library's communicator with its control plane in shared memory (dg_comm_create_shm; copy form only)
    hostf = [None]       # the shared-memory host vector of the "host" form (opened on first use)
    copy_field = [None]  # the field of the "copy" form: an array of dg_comm_field_alloc (mappable by the peers whatever its size)
    copy_shm_field = [None]
    pieces = 1
    comm_note = None
    launch_nodes = n_nodes
    state = {"line": None}
    progress = {"first_step_done": False}
    dog = Watchdog(args.form_timeout if world > 1 else 0, rank, state)
    if sharded:
        pieces = max(1, min(args.pieces, 64 // world))    # dg_shard_layout handles up to 64 (virtual) ranks
        vworld = pieces * world
        counts = []
        stride = 0
        for p in range(pieces):
            c, stride = dg.shard_layout(grid, p * world + rank, vworld)
            counts.append(c)
        launch_nodes = sum(counts)
        cg_D = [(res[0] + 1, res[1] + 1, res[2] + 1), (2 * res[0], res[1] + 1, res[2] + 1), (2 * res[1], res[2] + 1, res[0] + 1),
                (2 * res[2], res[0] + 1, res[1] + 1)]      # class dims (fastest, middle, slowest = k, k, i, j)
        cg_off = np.concatenate([[0], np.cumsum([int(np.prod(d)) for d in cg_D])])
    gathered = mine = unpack_stream = None

    def make_comm():
        """the library's communicator (collective).  Should it not come up on some rank, EVERY rank takes the torch.distributed
        form of the slabs exchange -- same kernels, same bytes on the links -- and the line says so."""
        nonlocal comm, comm_ext, comm_note, gathered, mine, unpack_stream
        if comm is not None or comm_ext is not None or gathered is not None:
            return
        if selftest:
            def _ag(mine_):
                t = torch.frombuffer(bytearray(mine_), dtype=torch.uint8)
                outs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(outs, t)
                return [bytes(o.numpy().tobytes()) for o in outs]
            comm_ext = dg.Comm.external(rank, world, _ag, ctl_barrier)
        elif not args.python_gather:
            failed = 0
            uid = [None]
            try:
                if rank == 0:
                    uid[0] = dg.Comm.unique_id()
            except Exception as exc:  # noqa: BLE001 (reported on the line)
                comm_note = "%s: %s" % (type(exc).__name__, exc)
                failed = 1
            dist.broadcast_object_list(uid, src=0)      # (gloo: rank 0's unique id, or None)
            if uid[0] is None:
                failed = 1
            if ctl_max(failed) == 0:
                try:
                    if os.environ.get("DG_BENCH_BREAK_LIBRARY_COMM") == "1":   # (tests: exercise the way out)
                        raise RuntimeError("simulated failure of dg_comm_create")
                    comm = dg.Comm(uid[0], rank, world)
                except Exception as exc:  # noqa: BLE001
                    comm_note = "%s: %s" % (type(exc).__name__, exc)
                    failed = 1
            if ctl_max(failed) != 0:
                if comm is not None:
                    comm.close()
                comm = None
                comm_note = "library communicator unavailable (%s)" % (comm_note or "on another rank")
        if comm is None:     # python-driven slabs (no library communicator): buffers
            gathered = torch.empty(vworld * stride, dtype=torch.float64, device="cuda")
            mine = torch.zeros(pieces * stride, dtype=torch.float64, device="cuda")   # packed pieces (+ padding)
            unpack_stream = torch.cuda.Stream()

    plane_cost = [None]            # chunked forms: relative cost per plane of every class (None: uniform), refined from measured times
    FLAGS = {"inplace": dg.EXCHANGE_INPLACE, "inplace-p2p": dg.EXCHANGE_INPLACE | dg.EXCHANGE_P2P,
             "to-root": dg.EXCHANGE_INPLACE | dg.EXCHANGE_TO_ROOT, "copy": dg.EXCHANGE_INPLACE | dg.EXCHANGE_COPY,
             "copy-shm": dg.EXCHANGE_INPLACE | dg.EXCHANGE_COPY}
    CHUNKED = set(FLAGS) | {"host"}
    last_python_ms = [None]

    def rebalance(piece_ms):
        """piece_ms[r][p]: sampling time of piece p on rank r -> plane costs: every plane of the chunks of virtual rank
        v = p * world + r gets (time of that launch / its nodes) x (nodes of the plane).  Every rank computes this from the
        SAME gathered times, so every rank cuts the lattice the same way."""
        cuts = dg.chunk_layout(grid, vworld, plane_cost[0])
        cost = [np.zeros(d[2], dtype=np.float32) for d in cg_D]
        for r in range(world):
            for p in range(pieces):
                v = p * world + r
                nodes = sum(int(cuts[c][v + 1] - cuts[c][v]) * cg_D[c][0] * cg_D[c][1] for c in range(4))
                if nodes == 0:
                    continue
                per_node = piece_ms[r][p] / nodes
                for c in range(4):
                    cost[c][cuts[c][v]:cuts[c][v + 1]] = per_node * cg_D[c][0] * cg_D[c][1]
        plane_cost[0] = cost

    def python_inplace_step():
        """the in-place exchange driven from here (one-GPU self-test over gloo, A/B): same chunks, same kernels"""
        cuts = dg.chunk_layout(grid, vworld, plane_cost[0])
        times = []
        for p in range(pieces):
            v = p * world + rank
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dg.sample_planes_device(mesh, grid, cuts[:, v], cuts[:, v + 1], field.data_ptr(), stream=s)
            e1.record(stream)
            times.append((e0, e1))
            for o in range(world):
                vo = p * world + o
                for c in range(4):
                    a = int(cg_off[c]) + int(cuts[c][vo]) * cg_D[c][0] * cg_D[c][1]
                    b = int(cg_off[c]) + int(cuts[c][vo + 1]) * cg_D[c][0] * cg_D[c][1]
                    if b > a:
                        dist.broadcast(field[a:b], src=o)
        torch.cuda.synchronize()
        last_python_ms[0] = [a.elapsed_time(b) for a, b in times]

    def comm_of(form):
        if form == "copy-shm":
            return comm_shm[0]
        return comm if comm is not None else (comm_ext if form == "copy" else None)

    def field_of(form):
        if form == "copy-shm":
            return copy_shm_field[0]
        return copy_field[0] if (form == "copy" and copy_field[0] is not None) else field

    def prepare(form):
        """what a form needs before its first step (collective: every rank gets here for every candidate)"""
        if os.environ.get("DG_BENCH_FAIL_FORM") == form:    # (tests: a form that cannot come up, e.g. RCCL that does not initialise)
            raise RuntimeError("simulated failure of form %s before its first step" % form)
        if form == "host":
            if hostf[0] is None:
                name = [None]
                if rank == 0:
                    name[0] = "dg_bench_%d_%d" % (os.getpid(), int(time.time()))
                if world > 1:
                    dist.broadcast_object_list(name, src=0)
                hostf[0] = dg.HostField(name[0], n_nodes, rank, world)
            return
        if form == "copy-shm":
            if comm_shm[0] is None:
                name = [None]
                if rank == 0:
                    name[0] = "dg_benchctl_%d_%d" % (os.getpid(), int(time.time()))
                if world > 1:
                    dist.broadcast_object_list(name, src=0)
                comm_shm[0] = dg.Comm.shared_memory(name[0], rank, world)
                copy_shm_field[0] = torch.as_tensor(comm_shm[0].field_alloc(n_nodes), device="cuda")
                copy_shm_field[0].fill_(float("nan"))
            return
        make_comm()
        if form == "copy" and copy_field[0] is None and comm_of("copy") is not None:
            # a field the peers can map whatever its size (512^3 is 7.5 GB; whole-allocation IPC stops at 2 GiB here);
            # if the virtual-memory route is unavailable on ANY rank, every rank keeps its torch allocation
            arr, failed = None, 0
            try:
                arr = comm_of("copy").field_alloc(n_nodes)
            except Exception as exc:  # noqa: BLE001
                print("bench.py rank %d: dg_comm_field_alloc failed (%s); the copy form uses the torch allocation" % (rank, exc), file=sys.stderr)
                failed = 1
            if sharded and world > 1:
                failed = ctl_max(failed)
            if failed == 0:
                copy_field[0] = torch.as_tensor(arr, device="cuda")
                copy_field[0].fill_(float("nan"))
            elif arr is not None:
                comm_of("copy").field_free(arr)

    def run_form(form):
        """one step of the sharded protocol in the given exchange form (enqueued on `stream`; "host" blocks)"""
        if os.environ.get("DG_BENCH_HANG_FORM") == form:    # (tests: a form that never returns, for the watchdog to cut off)
            time.sleep(1e6)
        if form == "host":
            hostf[0].sample(mesh, grid, field.data_ptr(), pieces=pieces, plane_cost=plane_cost[0], stream=s)
        elif form in FLAGS:
            c = comm_of(form)
            if c is not None:
                c.sample_exchange_device(mesh, grid, field_of(form).data_ptr(), pieces=pieces, flags=FLAGS[form], root=0, plane_cost=plane_cost[0],
                                         stream=s)
            else:
                python_inplace_step()
        elif comm is not None:
            comm.sample_allgather_device(mesh, grid, field.data_ptr(), pieces=pieces, stream=s)
        else:
            for p in range(pieces):
                mp = mine[p * stride:(p + 1) * stride]
                mesh.sample_shard_device(grid, p * world + rank, vworld, mp.data_ptr(), stream=s)
                # the collective's stream waits for the kernel just enqueued; this stream goes on with piece p+1
                work = dist.all_gather_into_tensor(gathered[p * world * stride:(p + 1) * world * stride], mp, async_op=True)
                with torch.cuda.stream(unpack_stream):
                    work.wait()          # unpack_stream waits for gather p, not the host
                    dg.unpack_shard_range_device(grid, vworld, gathered.data_ptr(), stride, p * world, (p + 1) * world,
                                                 field.data_ptr(), stream=unpack_stream.cuda_stream)
            stream.wait_stream(unpack_stream)

    def piece_times(form):
        """this rank's sampling time per piece of the step just run (None where the form does not report it)"""
        if form == "host":
            return hostf[0].last_chunk_ms(pieces)
        c = comm_of(form)
        if c is not None:
            return c.last_chunk_ms(pieces)
        return last_python_ms[0] if form in FLAGS else None

    def share_and_rebalance(form):
        torch.cuda.synchronize()
        mine_ms = torch.tensor(piece_times(form), dtype=torch.float32)
        all_ms = [torch.empty_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)
        rebalance([t.tolist() for t in all_ms])

    def check_result(form):
        """sanity of the result that was just timed (cheap, outside the timed region); in the self-test modes: the whole field
        against the direct launch, bit for bit"""
        out = hostf[0].data if form == "host" else None
        if out is not None:
            probe = np.array(out[:: max(1, n_nodes // 1000)])
        else:
            probe = field_of(form)[:: max(1, n_nodes // 1000)].cpu().numpy()
        assert np.isfinite(probe).all() and np.abs(probe).max() < 2.0, "the field of form %s holds values no signed distance can take" % form
        if (args.force_shard_path and world == 1) or selftest:
            ref = torch.empty_like(field)
            mesh.sample_nodes_device(grid, 0, n_nodes, ref.data_ptr(), stream=s)
            torch.cuda.synchronize()
            if form == "host":
                assert np.array_equal(out, ref.cpu().numpy()), "the shared host vector differs from the direct launch"
            elif form != "to-root" or rank == 0 or comm is None:
                assert torch.equal(ref, field_of(form)), "the sharded protocol's field differs from the direct launch"

    def measure(form):
        """--warmup untimed steps (the chunked forms re-cut the lattice from every rank's measured sampling times after each; at
        least two such steps), then EXACTLY --steps steps between barrier + synchronize on both sides; max over ranks."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        if sharded:
            prepare(form)
            plane_cost[0] = None
        chunked = sharded and form in CHUNKED and world > 1

        def step(i=None):
            if i is not None:
                ev[i][0].record(stream)
            if sharded:
                run_form(form)
            else:
                mesh.sample_nodes_device(grid, 0, n_nodes, field.data_ptr(), stream=s)
            if i is not None:
                ev[i][1].record(stream)    # (the sharded calls make `stream` wait for the complete field)

        for w in range(max(args.warmup, 2 if chunked else 0)):
            step()
            if w == 0:
                torch.cuda.synchronize()
                progress["first_step_done"] = True     # (a form that fails before this point failed to come up, not to run)
            if chunked:
                share_and_rebalance(form)
        torch.cuda.synchronize()
        if sharded:
            ctl_barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        progress["first_step_done"] = True
        if sharded:
            ctl_barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if sharded:
            elapsed = ctl_max(elapsed)
        r = {"form": form, "elapsed": elapsed, "kernel_ms": float(np.mean([a.elapsed_time(b) for a, b in ev])), "per_rank": None}
        if sharded and world > 1:
            # what the last timed step looked like on every rank: sampling time per piece and the part of the exchange the
            # sampling did not hide (a bad scaling number must be readable from the line)
            c = comm_of(form) if form != "host" else None
            ms = piece_times(form) or [float("nan")] * pieces
            wait = c.last_exchange_wait_ms() if c is not None else float("nan")
            mine_t = torch.tensor(list(ms) + [wait], dtype=torch.float32)
            all_t = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(all_t, mine_t)
            r["per_rank"] = {"sample_ms": [[round(float(x), 3) for x in t.tolist()[:-1]] for t in all_t],
                             "exchange_wait_ms": [round(float(t.tolist()[-1]), 3) for t in all_t]}
        check_result(form)
        return r

    def sharding_text(form):
        if not sharded:
            return "none"
        if form == "host":
            return ("contiguous chunks cut by measured cost; every rank copies its chunks into ONE shared-memory host vector with its own "
                    "copy engine in %d piece(s) under the sampling, a barrier inside the segment (dg_sdf_sample_to_host_field: no RCCL, no "
                    "device IPC); the result is the reference's host vector m_nodes[field], not a device-resident field" % pieces)
        if form not in FLAGS:
            return ("4-plane slabs round-robin; sample / all_gather / unpack pipelined in %d piece(s) by %s"
                    % (pieces, ("torch.distributed (python)" + ("; " + comm_note if comm_note else "")) if comm is None
                       else "dg_sdf_sample_allgather_device (RCCL inside the library)"))
        return ("contiguous chunks cut by measured cost, sampled in place, %s in %d piece(s) by %s"
                % (form, pieces, "torch.distributed broadcasts (python)" if comm_of(form) is None
                   else ("dg_sdf_sample_exchange_device (%s)" % ("peer copies on the copy engines, fields of dg_comm_field_alloc mapped chunk by chunk, control plane in shared memory: no RCCL"
                                                                 if form == "copy-shm" else
                                                                 ("peer copies on the copy engines, fields of dg_comm_field_alloc mapped chunk by chunk"
                                                                  if copy_field[0] is not None else "peer copies on the copy engines, HIP IPC")
                                                                 if form == "copy" else "RCCL inside the library"))))

    def flat_exchange(form, exchange_report, rccl_nranks):
        """N > 1: which form won, every form's ms per step, what failed -- as plain scalars inside `roofline`"""
        if not sharded:
            return {}
        flat = {"rccl_nranks": rccl_nranks, "exchange_chosen": form, "exchange_pieces": pieces}
        for k, v in ((exchange_report or {}).get("ms_by_form") or {}).items():
            flat["exchange_ms_" + k.replace("-", "_")] = v
        for k, v in ((exchange_report or {}).get("errors") or {}).items():
            flat["exchange_error_" + k.replace("-", "_")] = str(v)[:160]
        return flat

    def build_line(r, exchange_report):
        """the JSON line for the measured form r (no collectives, no GPU work: the watchdog may call for it at any time)"""
        form, elapsed, kernel_ms = r["form"], r["elapsed"], r["kernel_ms"]
        balg = load_balg()
        counters, counters_note = load_counters()
        # (the counters are of the N = 1 launch -- 118.4 M nodes of the 256^3 lattice; at N > 1 with weak scaling every GPU runs the same kernel
        # over as many nodes of a finer lattice: the replayed utilisation figures are reported there too and `counters_of` says what they are)
        k1 = (counters or {}).get("k1")
        rccl_nranks = comm.info()["rccl_nranks"] if (comm is not None and form not in (None, "host", "copy-shm")) else None
        return {
            "metric": "Mnodes/s SDF sampling (256³ grid, 100k-tri mesh) + % HBM roofline, 1/2/4/8 GPU",
            "value": n_nodes * args.steps / elapsed / 1e6,
            "unit": "Mnodes/s",
            "value_is": ("the sampling step with the result left in HBM (at N > 1: incl. the exchange that leaves the whole field on every GPU); "
                         "SURVEY 8(d)'s metric incl. the D2H into the host vector is roofline.d2h_mnodes_s / value_with_d2h / addfunction_e2e") if form != "host" else
                        ("the sampling step incl. the copies that assemble the WHOLE coefficient vector in one host vector shared by all ranks "
                         "(SURVEY 8(d)'s metric: sampling + D2H into m_nodes); no device-resident copy of the whole field"),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "icosphere nu=71 (100820 tris) SDF node sampling, grid %s = %d nodes%s"
                            % ("x".join(map(str, res)), n_nodes,
                               "" if world == 1 else ("; weak scaling: 256^3 per GPU, the lattice grows with N" if args.scaling == "weak" else
                                                      "; STRONG scaling: the metric's own 256^3 lattice shared by the %d GPUs" % world)),
                "nodes_per_gpu_launch": launch_nodes,
                "sharding": sharding_text(form),
                # every multi-rank figure of this path is UNVERIFIED on hardware until a driver-run SCALE file exists
                "exchange": (dict(exchange_report or {}, chosen=form, per_rank=r["per_rank"], rccl_nranks=rccl_nranks, preflight=preflight_summary[0]) if sharded else None),
                "mesh_bvh_build_s": round(mesh.info()["build_seconds"], 4),
            },
            "roofline": {
                # K1's binding resource is VALU issue: frac = busy VALU cycles / (SIMDs x kernel cycles), from PMC
                "bound": "valu_issue",
                "achieved": k1["valu_busy"] if k1 else None, "peak": 1.0,
                "unit": "fraction of the VALU issue cycles of the %d SIMDs" % N_SIMDS,
                "frac": k1["valu_busy"] if k1 else None,
                "traffic": k1["hbm_bytes_per_launch"] if k1 else None,
                # one dg_sdf_sample_*_device call = k_sample_fast (the filtered K1 kernel; k_sample_nodes with DG_FORCE=k1_fast=0)
                # + the two heavy-brick kernels (4 % of it); kernel_ms: HIP events around the call, in the timed region
                "kernel": "%s (+ k_heavy_subtrees, k_heavy_finish)" % ((k1 or {}).get("kernel") or
                          ("k_sample_nodes" if "k1_fast=0" in os.environ.get("DG_FORCE", "") else "k_sample_fast")),
                "kernel_ms": kernel_ms,
                "hbm": {
                    "bound_by_hbm": False,
                    "achieved_gbs": (k1["hbm_bytes_per_launch"] / (k1["kernel_ms"] * 1e-3) / 1e9) if k1 else None,
                    "peak_gbs": HBM_PEAK_GBS,
                    "frac": (k1["hbm_bytes_per_launch"] / (k1["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if k1 else None,
                    "compulsory_bytes_per_launch": 8 * launch_nodes,
                    "kernel_ms_at_hbm_roof": 8 * launch_nodes / (HBM_PEAK_GBS * 1e9) * 1e3,
                },
                "counters": counters_note,
                # the counter-derived figures (achieved / frac / traffic / per_brick) are REPLAYED from profiles/counters.json
                # (rocprofv3 PMC passes cannot run inside the driver's bench); kernel_ms is measured live
                "replayed": True, "counters_sha": (counters or {}).get("csrc_sha256"),
                "counters_of": "the N = 1 launch (118 425 857 nodes of the 256^3 lattice), one GPU",
                "per_brick": k1.get("per_brick") if k1 else None,
                # frac above is UTILISATION of the vector issue slots; these say how much of it is arithmetic and how full the scalar unit is:
                # arith_frac = (f32 + f64 arithmetic instructions) / all vector instructions (the rest: compares, selects, moves, integer);
                # salu_slot_frac = (SALU + SMEM instructions) / (256 CUs x kernel cycles), one scalar issue slot per cycle and CU
                "arith_frac": (k1 or {}).get("arith_frac") or ((((k1 or {}).get("per_brick") or {}).get("valu_f32", 0) + ((k1 or {}).get("per_brick") or {}).get("valu_f64", 0)) /
                                                                 ((k1 or {}).get("per_brick") or {}).get("valu") if ((k1 or {}).get("per_brick") or {}).get("valu") else None),
                "salu_slot_frac": (k1 or {}).get("salu_slot_frac"),
                # informational only: bytes the REFERENCE's traversal would move for these nodes / this kernel's time
                "algorithmic_bytes_per_node": balg["bytes_per_node"],
                "algorithmic_gbs": balg["bytes_per_node"] * launch_nodes / (kernel_ms * 1e-3) / 1e9,
                # the HBM fractions the metric asks for, as plain scalars (the driver's record drops the nested `hbm` object):
                # hbm_frac = HBM traffic per launch by the counters / kernel time / 8 TB/s (replayed, N = 1 only);
                # compulsory_hbm_frac = the 8 B per node this rank's launches have to write / live kernel time / 8 TB/s
                "hbm_frac": (k1["hbm_bytes_per_launch"] / (k1["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if k1 else None,
                "compulsory_hbm_frac": 8 * launch_nodes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                **flat_exchange(form, exchange_report, rccl_nranks),
            },
            "cpu_baseline": None,
        }

    # ---- which forms, in which order ---------------------------------------------------------------------------------
    if not sharded:
        candidates = [None]
    elif args.exchange != "auto":
        candidates = [args.exchange]
    elif world == 1:
        candidates = ["slabs"]    # (--force-shard-path on one GPU: nothing to choose)
    else:
        # Measure, do not guess: none of these has ever run on more than one GPU where this was developed.  "host" first: it needs
        # nothing but shared memory and each GPU's own copy engine, so a number exists before RCCL or device IPC are touched.
        # "copy-shm" second: the whole field on every GPU, still without RCCL.
        candidates = ["host", "copy-shm", "slabs", "inplace", "inplace-p2p", "copy"]
    results, errors = {}, {}
    best = None
    preflight_summary = [None]
    RCCL_FORMS = ("slabs", "inplace", "inplace-p2p", "copy", "to-root")
    rccl_out = None               # set by the first RCCL form that fails before its first step completes: the others are not retried
    removed, preflight = {}, None
    if sharded and world > 1 and not args.no_preflight:
        # tools/scale_preflight.py: which building blocks (shared memory, chunk export / import / peer copy, RCCL) work on this box;
        # a failing step removes only the forms that need it
        pre_dog = Watchdog(args.preflight_timeout + 60, rank, state)   # (the children are killed after --preflight-timeout by themselves)
        pre_dog.arm("the preflight")
        removed, preflight = run_preflight(dist, rank, world, args.preflight_timeout)
        pre_dog.disarm()
        if len(candidates) > 1 and all(c in removed for c in candidates):
            removed = {}          # (nothing would be left: the preflight is then the suspect; race them all)
        preflight_summary[0] = {"rank0_steps": {k: (v.get("ok") if isinstance(v, dict) and "ok" in v else v) for k, v in (preflight or {}).items()},
                                "forms_removed": removed or None}
        if rank == 0:
            print("bench.py preflight: %s" % ("; ".join("%s out (%s)" % kv for kv in sorted(removed.items())) or "every form may run"), file=sys.stderr, flush=True)
    for cand in candidates:
        if len(candidates) > 1 and (cand in removed or (rccl_out is not None and cand in RCCL_FORMS)):
            errors[cand] = ("not run: " + removed[cand]) if cand in removed else ("not run: RCCL did not come up for form %s (%s)" % rccl_out)
            if rank == 0:
                print("bench.py exchange race: %s -> %s" % (cand, errors[cand]), file=sys.stderr, flush=True)
            continue
        dog.arm("exchange form %s" % cand)
        r, note, failed = None, None, 0
        progress["first_step_done"] = False
        try:
            r = measure(cand)
        except Exception as exc:  # noqa: BLE001 (reported on the line; the form is out of the race)
            if len(candidates) == 1:
                raise
            note = "%s: %s" % (type(exc).__name__, str(exc)[:200])
            failed = 1
        early = 1 if (failed and not progress["first_step_done"]) else 0
        if sharded and world > 1:
            failed = ctl_max(failed)     # (a form that failed on any rank is out on all of them)
            early = ctl_max(early)
        dog.disarm()
        if failed and early and cand in RCCL_FORMS and (not selftest or os.environ.get("DG_BENCH_FAIL_FORM")):
            rccl_out = (cand, note or "failed on another rank")
        if failed:
            errors[cand] = note or "failed on another rank"
        else:
            results[cand] = r
            if best is None or r["elapsed"] < best["elapsed"]:
                best = r
        if rank == 0 and len(candidates) > 1:   # (stderr: a later form that hangs must not take the evidence of the earlier ones with it)
            print("bench.py exchange race: %s -> %s" % (cand, ("%.3f ms / step" % (r["elapsed"] / args.steps * 1e3)) if not failed else errors[cand]),
                  file=sys.stderr, flush=True)
        if rank == 0 and best is not None:
            report = None if len(candidates) == 1 else {
                "ms_by_form": {k: v["elapsed"] / args.steps * 1e3 for k, v in results.items()}, "errors": errors or None,
                "how": "every form: >= 2 cost-rebalancing warm-up steps (chunked forms), --warmup steps, then --steps timed steps between "
                       "barriers (max over ranks), each form under a %d s watchdog; value = the fastest form" % args.form_timeout}
            state["line"] = build_line(best, report)
    if best is None:
        raise SystemExit("no exchange form ran: %s" % errors)

    if rank == 0:
        out = state["line"]
        kernel_ms = best["kernel_ms"]
        if world == 1 and not args.no_extras:
            out.update(extras(torch, dg, T, mesh, grid, field, V, F, dom, res, kernel_ms))
            out["roofline"]["hbm"]["copy_kernel_gbs"] = out.pop("hbm_copy_gbs")
            out["roofline"].update(user_facing_scalars(out))
        if world == 1 and args.cpu_seconds > 0 and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(V, F, dom, res, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if hostf[0] is not None:
        hostf[0].close()
    if comm is not None:
        comm.close()
    if comm_ext is not None:
        comm_ext.close()
    if comm_shm[0] is not None:
        comm_shm[0].close()
    if sharded:
        ctl_barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
