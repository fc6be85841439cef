"""GPU-backed parts of the C++ host API, driven like a downstream application would: the
GenerateSDF tool reproduces the reference's golden box.cdf byte for byte, and the API driver
checks MeshSDF addFunction (with predicate / inversion), batched-vs-scalar interpolate and
batched-vs-single signed_distance."""
import os
import subprocess

import numpy as np
import pytest

import dgtest as T

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max
BUILD = os.path.join(T.ROOT, "discregrid_amd", "cpp", "build")
TEST_BUILD = os.path.join(T.ROOT, "tests", "cpp", "build")


def _need(path):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(BUILD)])
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(TEST_BUILD)])
    return path


def test_generate_sdf_cli_reproduces_box_cdf(tmp_path):
    """GenerateSDF -r "5 5 5" box.obj == cmd/generate_sdf/resources/box.cdf (OBJ reader, default
    domain rule, TriangleMeshDistance, GPU addFunction, save)."""
    exe = _need(os.path.join(BUILD, "GenerateSDF"))
    obj = str(tmp_path / "box.obj")
    V, F = T.box_mesh()
    T.write_obj(obj, V, F)
    out = str(tmp_path / "box.cdf")
    log = subprocess.check_output([exe, "-r", "5 5 5", "-o", out, obj]).decode()
    assert "Construction took" in log and "DONE" in log
    assert open(out, "rb").read() == open(os.path.join(T.GOLDEN, "box.cdf"), "rb").read()
    # default output name + inversion
    subprocess.check_call([exe, "-r", "4 5 6", "-i", obj], stdout=subprocess.DEVNULL)
    g = T.read_cdf(str(tmp_path / "box.cdf"))
    want = T.OracleMesh(V, F).sample_nodes(T.oracle_default_domain(V), [4, 5, 6], invert=True)
    np.testing.assert_array_equal(g["nodes"][0], want)
    np.testing.assert_array_equal(g["res"], [4, 5, 6])


def test_generate_sdf_cli_on_several_devices(tmp_path):
    """DG_DEVICES replicates the mesh on the listed devices and addFunction spreads the lattice over
    them (dg_sdf_sample_nodes_multi).  One GPU here, so the list names it twice -- same code path:
    two handles, two host threads, two copy pipelines; the file must not change by a byte."""
    exe = _need(os.path.join(BUILD, "GenerateSDF"))
    obj = str(tmp_path / "box.obj")
    V, F = T.box_mesh()
    T.write_obj(obj, V, F)
    out = str(tmp_path / "box.cdf")
    env = T.force_env(dict(os.environ, DG_DEVICES="0,0,0"), host_chunk_nodes=1024)
    subprocess.check_call([exe, "-r", "5 5 5", "-o", out, obj], stdout=subprocess.DEVNULL, env=env)
    assert open(out, "rb").read() == open(os.path.join(T.GOLDEN, "box.cdf"), "rb").read()
    tor = str(tmp_path / "torus.obj")
    Vt, Ft = T.torus()
    T.write_obj(tor, Vt, Ft)
    subprocess.check_call([exe, "-r", "21 18 26", "-o", out, tor], stdout=subprocess.DEVNULL, env=dict(os.environ, DG_DEVICES="all,0"))
    a = open(out, "rb").read()
    subprocess.check_call([exe, "-r", "21 18 26", "-o", out, tor], stdout=subprocess.DEVNULL)
    assert a == open(out, "rb").read()
    bad = subprocess.run([exe, "-r", "5 5 5", "-o", out, obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                         env=dict(os.environ, DG_DEVICES="0,63"))
    assert bad.returncode != 0


def test_generate_sdf_cli_explicit_domain(tmp_path):
    exe = _need(os.path.join(BUILD, "GenerateSDF"))
    obj = str(tmp_path / "torus.obj")
    V, F = T.torus()
    T.write_obj(obj, V, F)
    out = str(tmp_path / "t.cdf")
    subprocess.check_call([exe, "--resolution", "7 6 5", "-d", "-1.5 -1.6 -0.5 1.7 1.5 0.6", "--output=" + out, obj],
                          stdout=subprocess.DEVNULL)
    g = T.read_cdf(out)
    dom = np.array([-1.5, -1.6, -0.5, 1.7, 1.5, 0.6])
    np.testing.assert_array_equal(g["domain"], dom)
    np.testing.assert_array_equal(g["nodes"][0], T.OracleMesh(V, F).sample_nodes(dom, [7, 6, 5]))
    np.testing.assert_array_equal(g["cells"][0], T.oracle_cell_table([7, 6, 5]))


def test_host_api_gpu_driver(tmp_path):
    exe = _need(os.path.join(TEST_BUILD, "host_api_driver"))
    obj = str(tmp_path / "torus.obj")
    V, F = T.torus()
    T.write_obj(obj, V, F)
    dom = T.oracle_default_domain(V)
    res = [9, 7, 8]
    n = T.n_nodes(res)
    ext = dom[3:] - dom[:3]
    P = np.random.default_rng(2).uniform(dom[:3] - 0.05 * ext, dom[3:] + 0.05 * ext, size=(5000, 3))
    pts = str(tmp_path / "pts.bin")
    P.tofile(pts)
    out = str(tmp_path / "out.bin")
    print(subprocess.check_output([exe, "gpu", obj, pts, out]).decode())
    got = np.fromfile(out)
    assert got[0] == 0.0                              # batched == scalar, batch == single
    f0, f1 = got[1:1 + n], got[1 + n:1 + 2 * n]
    d, phi = got[1 + 2 * n:1 + 2 * n + len(P)], got[1 + 2 * n + len(P):]
    om = T.OracleMesh(V, F)
    want = om.sample_nodes(dom, res)
    np.testing.assert_array_equal(f0, want)
    pos = T.oracle_node_positions(dom, res)
    np.testing.assert_array_equal(f1, np.where(pos[:, 2] > 0.0, -1.0 * want, DBL_MAX))
    np.testing.assert_array_equal(d, om.signed_distance(P))
    np.testing.assert_array_equal(phi, T.oracle_interpolate(dom, res, want, P))


@pytest.mark.parametrize("src,bound,golden", [("box.cdf", 0.25, "box_reduced_0p25.cdf"),
                                               ("torus_9_14_6.cdf", 0.08, "torus_9_14_6_reduced_0p08.cdf")])
def test_reduce_field_on_device_matches_reference_file(tmp_path, src, bound, golden):
    """reduceField with a typed ValuePredicate runs on the GPU (node flags, surviving cells, compaction,
    Morton keys with the reference's arithmetic, radix sort, renumbering) and reproduces the file the
    REFERENCE wrote for the same reduction, byte for byte; dg_reduce_field through ctypes gives the same
    arrays."""
    import discregrid_amd as dg
    exe = _need(os.path.join(TEST_BUILD, "host_api_driver"))
    out = str(tmp_path / "red.cdf")
    path = subprocess.check_output([exe, "reduceband", os.path.join(T.GOLDEN, src), str(-bound), str(bound), out, "gpu"]).decode().strip()
    assert open(out, "rb").read() == open(os.path.join(T.GOLDEN, golden), "rb").read()
    full, want = T.read_cdf(os.path.join(T.GOLDEN, src)), T.read_cdf(os.path.join(T.GOLDEN, golden))
    dg.load_library()
    g = dg.grid_desc(full["domain"][:3], full["domain"][3:], full["res"])
    coeffs, cells, cmap, tied = dg.reduce_field(g, full["nodes"][0], -bound, bound)
    # the torus lattice (9 x 14 x 6 over an oblong box) is anisotropic enough for tied Morton keys: the
    # device says so and the class runs the host algorithm; the cube's lattice has unique keys
    assert tied == (src == "torus_9_14_6.cdf") and path == ("host" if tied else "gpu")
    if tied:
        return
    np.testing.assert_array_equal(coeffs, want["nodes"][0])
    np.testing.assert_array_equal(cells, want["cells"][0].reshape(-1, 32))
    np.testing.assert_array_equal(cmap, want["cell_map"][0])


def test_reduce_field_on_device_random_fields_and_ties(tmp_path):
    """Device reduceField == host reduceField (which reproduces the reference's files) on random fields incl.
    DBL_MAX nodes, closed-range predicates, everything / nothing removed; on a strongly anisotropic lattice
    the Morton keys tie, the device reports it and the C++ class runs the host algorithm."""
    import discregrid_amd as dg
    exe = _need(os.path.join(TEST_BUILD, "host_api_driver"))
    dg.load_library()
    rng = np.random.default_rng(31)
    cases = [([7, 9, 6], [-1.0, -0.5, 0.0, 1.5, 0.75, 2.0], False), ([12, 12, 12], [0.0, 0.0, 0.0, 1.0, 1.0, 1.0], False),
             ([40, 3, 3], [0.0, 0.0, 0.0, 1.0, 1.0, 1.0], True)]
    for res, dom, expect_tie in cases:
        dom = np.array(dom)
        n = T.n_nodes(res)
        coeffs = rng.normal(size=n)
        coeffs[rng.integers(0, n, n // 50)] = DBL_MAX
        src = str(tmp_path / "src.cdf")
        T.oracle_write_cdf(src, dom, res, [coeffs])
        for lo, hi in ((-0.3, 0.2), (-100.0, 100.0), (5.0, 6.0)):
            outs = {}
            for mode in ("gpu", "host"):
                out = str(tmp_path / (mode + ".cdf"))
                path = subprocess.check_output([exe, "reduceband", src, str(lo), str(hi), out, mode]).decode().strip()
                outs[mode] = (path, open(out, "rb").read())
            assert outs["host"][0] == "host"
            g = dg.grid_desc(dom[:3], dom[3:], res)
            _, _, _, tied = dg.reduce_field(g, coeffs, lo, hi)
            assert outs["gpu"][0] == ("host" if tied else "gpu")
            assert outs["gpu"][1] == outs["host"][1]
            if expect_tie and lo < 0:
                assert tied
        # closed range through the C ABI vs a numpy restatement of the node / cell selection
        c2, cells, cmap, tied = dg.reduce_field(g, coeffs, -0.25, 0.5, closed=True)
        if not tied:
            keep = (coeffs >= -0.25) & (coeffs <= 0.5) & (coeffs != DBL_MAX)
            table = T.oracle_cell_table(res)
            alive = keep[table].any(axis=1)
            assert len(cells) == alive.sum() and ((cmap != 0xFFFFFFFF) == alive).all()
            used = np.zeros(n, dtype=bool)
            used[table[alive].ravel()] = True
            assert len(c2) == used.sum()
            np.testing.assert_array_equal(np.sort(c2), np.sort(coeffs[used]))
            np.testing.assert_array_equal(c2[cells[cmap[alive]]], coeffs[table[alive]])


def test_unchanged_reference_caller(tmp_path):
    """The reference's own idiom -- addFunction with the lambda of cmd/generate_sdf/main.cpp:97-101, an
    opaque std::function -- compiled unchanged against this repository's headers: the grid takes its host
    loop and every node is one TriangleMeshDistance::signed_distance(point) on the calling OpenMP thread
    (dg_signed_distance_point).  Same coefficients as the typed MeshSDF path (GPU) and as the oracle, and
    not slower than the unmodified reference on the same host cores (bunny 32^3, BASELINE config 1's size)."""
    exe = _need(os.path.join(TEST_BUILD, "unchanged_caller"))
    V, F = T.bunny_mesh()
    obj = str(tmp_path / "bunny.obj")
    T.write_obj(obj, V, F)
    out = str(tmp_path / "out.bin")
    subprocess.check_call([exe, "lambda", obj, "32 32 32", out])
    got = np.fromfile(out)
    n = int(got[0])
    t_lambda, t_typed, bad = got[1], got[2], got[3]
    assert n == T.n_nodes([32, 32, 32]) and bad == 0.0
    dom = T.oracle_default_domain(V)
    om = T.OracleMesh(V, F)
    want = om.sample_nodes(dom, [32, 32, 32])
    np.testing.assert_array_equal(got[4:], want)
    if T.ref_available():
        g = T.RefGrid(V, F, dom, [32, 32, 32])
        g.sample_nodes(0, n)
        g.sample_nodes(0, n)                # timed with the OpenMP team already running, like the caller
        t_ref = g.last_seconds
        print("bunny 32^3: unchanged lambda caller %.3f s, reference node loop %.3f s, MeshSDF (GPU) %.4f s"
              % (t_lambda, t_ref, t_typed))
        # same league as the reference (measured: several times faster).  The bound is loose on purpose: both times are
        # tens of milliseconds of 256 OpenMP threads on a shared box, and what it guards against is the 150 x cliff of a
        # per-point GPU round trip, not a factor of two
        assert t_lambda <= 5.0 * t_ref + 0.25


def test_dg_devices_all_spreads_addfunction(tmp_path):
    """DG_DEVICES=all: the C++ addFunction deals the lattice to every visible GPU (dg_sdf_sample_nodes_multi,
    one copy pipeline per device writing into the one host array).  Needs two or more devices."""
    import discregrid_amd as dg
    dg.load_library()
    if dg.device_count() < 2:
        pytest.skip("one GPU visible")
    exe = _need(os.path.join(BUILD, "GenerateSDF"))
    V, F = T.torus()
    obj = str(tmp_path / "torus.obj")
    T.write_obj(obj, V, F)
    outs = []
    for env in (dict(os.environ), dict(os.environ, DG_DEVICES="all")):
        out = str(tmp_path / ("o%d.cdf" % len(outs)))
        subprocess.check_call([exe, "-r", "40 36 30", "-o", out, obj], env=env, stdout=subprocess.DEVNULL)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]


def test_concurrent_single_point_callers(tmp_path):
    """64 threads call signed_distance(point) on one const TriangleMeshDistance at once (declared thread
    safe in the reference, TriangleMeshDistance.h:188,199); every result equals the batched GPU query."""
    exe = _need(os.path.join(TEST_BUILD, "unchanged_caller"))
    V, F = T.icosphere(12)
    obj = str(tmp_path / "ico.obj")
    T.write_obj(obj, V, F)
    out = str(tmp_path / "out.bin")
    subprocess.check_call([exe, "threads", obj, "64", "20000", out])
    assert np.fromfile(out)[0] == 0.0


def _bitmap_cases():
    import sys
    sys.path.insert(0, T.GOLDEN)
    from make_golden_bitmaps import CASES
    return CASES


@pytest.mark.parametrize("name", sorted(_bitmap_cases()))
def test_discrete_field_to_bitmap_cli_reproduces_reference_bitmaps(tmp_path, name):
    """DiscreteFieldToBitmap (slice sampled by ONE batched GPU interpolate) == the bitmap the
    unmodified reference tool wrote for the same options (cmd/discrete_field_to_bitmap), byte
    for byte; offsets 34..37 (uninitialised biSizeImage in the reference) are 0 on both sides."""
    exe = _need(os.path.join(BUILD, "DiscreteFieldToBitmap"))
    src, opts = _bitmap_cases()[name]
    out = str(tmp_path / "out.bmp")
    log = subprocess.check_output([exe] + opts + ["-o", out, os.path.join(T.GOLDEN, src)]).decode()
    assert "bmp resolution" in log
    got = np.frombuffer(open(out, "rb").read(), np.uint8)
    want = np.frombuffer(open(os.path.join(T.GOLDEN, "bitmap_%s.bmp" % name), "rb").read(), np.uint8)
    assert got.shape == want.shape
    diff = np.flatnonzero(got != want)
    assert diff.size == 0, "first differing byte at offset %d" % diff[0]


def test_discrete_field_to_bitmap_cli_default_output_and_errors(tmp_path):
    import shutil
    exe = _need(os.path.join(BUILD, "DiscreteFieldToBitmap"))
    src = str(tmp_path / "field.cdf")
    shutil.copyfile(os.path.join(T.GOLDEN, "box.cdf"), src)
    subprocess.check_call([exe, "--samples=16", "--plane", "yz", src], stdout=subprocess.DEVNULL)
    b = open(str(tmp_path / "field.bmp"), "rb").read()
    assert b[:2] == b"BM" and len(b) == 54 + 16 * 16 * 3
    assert subprocess.call([exe], stdout=subprocess.DEVNULL) == 1
    assert subprocess.call([exe, "-p", "xyz", src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 1
    assert subprocess.call([exe, "-f", "3", src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 1


# ---- one device-resident copy per field (dg_sdf_sample_field / dg_density_map_field / dg_reduce_field_device) ----
@pytest.mark.parametrize("lazy", ["1", "0"])
def test_flow_device_resident_fields_reproduce_the_reference_files(tmp_path, golden, lazy):
    """addFunction(MeshSDF) -> addDensityMap -> batched interpolate -> first host reads -> copies / moves of the
    grid -> reduceField x 2 -> save, in ONE process: every field is produced into a device array its handle owns
    and never uploaded; the host vectors fill asynchronously (DG_LAZY_HOST=1, default) or before the call returns
    (=0).  The .cdm written at the end == the file the REFERENCE's GenerateSDF + GenerateDensityMap flow wrote
    (tests/golden/torus_16_16_6_density_reduced.cdm), byte for byte, in both modes."""
    exe = _need(os.path.join(TEST_BUILD, "host_api_driver"))
    obj = str(tmp_path / "torus.obj")
    V, F = T.torus()
    T.write_obj(obj, V, F)
    dom = T.oracle_default_domain(V)
    ext = dom[3:] - dom[:3]
    P = np.random.default_rng(5).uniform(dom[:3] - 0.02 * ext, dom[3:] + 0.02 * ext, size=(6000, 3))
    pts = str(tmp_path / "pts.bin")
    P.tofile(pts)
    prefix = str(tmp_path / "flow")
    log = subprocess.check_output([exe, "flow", obj, "16 16 6", "0.1", pts, prefix], env=dict(os.environ, DG_LAZY_HOST=lazy)).decode()
    print(log)
    got = np.fromfile(prefix + ".bin")
    assert got[0] == 0.0            # batched == scalar, copies == original, reduced batched == reduced scalar
    ref_sdf = T.read_cdf(os.path.join(T.GOLDEN, "torus_16_16_6.cdf"))
    g = T.read_cdf(prefix + ".cdf")
    np.testing.assert_array_equal(g["domain"], ref_sdf["domain"])
    np.testing.assert_array_equal(g["nodes"][0], ref_sdf["nodes"][0])
    np.testing.assert_array_equal(g["nodes"][1], golden["torus16_density_h01"])
    assert open(prefix + ".cdf", "rb").read() == open(prefix + "_copy.cdf", "rb").read()
    assert open(prefix + ".cdm", "rb").read() == open(os.path.join(T.GOLDEN, "torus_16_16_6_density_reduced.cdm"), "rb").read()
    n = len(P)
    np.testing.assert_array_equal(got[1:1 + n], T.oracle_interpolate(dom, [16, 16, 6], ref_sdf["nodes"][0], P))
    np.testing.assert_array_equal(got[1 + n:1 + 2 * n], T.oracle_interpolate(dom, [16, 16, 6], golden["torus16_density_h01"], P))


@pytest.mark.parametrize("res,host_memory", [([9, 7, 8], "fresh"), ([160, 150, 140], "fresh"), ([161, 150, 140], "fresh"), ([160, 150, 140], "resident"),
                                             ([112, 96, 120], "pinned"), ([112, 96, 120], "none")])
def test_sample_field_is_device_resident_and_fills_the_host_array(res, host_memory):
    """dg_sdf_sample_field: K1 into a device array the new field handle owns; the host array is filled by a
    worker (direct DMA into fresh / pinned memory, blocking copies into resident 4 KiB pages) while consumers on
    the device already run.  Chunked (>= 2^22 nodes), scheduled (>= 2^24) and single-launch forms; mask; K2 / K3 /
    reduceField straight off the handle == the host-pointer entry points, bit for bit."""
    import torch
    import discregrid_amd as dg
    V, F = T.icosphere(10)
    dom = T.oracle_default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    want = mesh.sample_nodes(grid)
    if host_memory == "fresh":
        host = np.empty(n)
    elif host_memory == "resident":
        host = np.zeros(n) + 1.0
    elif host_memory == "pinned":
        pin = torch.empty(n, dtype=torch.float64).pin_memory()
        host = pin.numpy()
    else:
        host = None
    fld = mesh.sample_field(grid, host_out=host, host_first=(host_memory != "fresh" or res[0] != 160))
    info = fld.info()
    assert info["owns_coefficients"] == 1 and info["n_coeffs"] == n and info["device_bytes"] >= 8 * n
    # device consumers need no host data
    P = np.random.default_rng(1).uniform(dom[:3], dom[3:], size=(30000, 3))
    phi, grad = fld.interpolate(P, grad=True)
    wphi, wgrad = dg.Field(grid, want).interpolate(P, grad=True)
    np.testing.assert_array_equal(phi, wphi)
    np.testing.assert_array_equal(grad, wgrad)
    if host is not None:
        got = fld.host_wait()
        assert fld.info()["host_copy_pending"] == 0
        np.testing.assert_array_equal(got, want)
        assert fld.host_wait() is got            # idempotent
    dev = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    import ctypes
    assert ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(info["d_coeffs"]),
                                                                ctypes.c_size_t(8 * n), 3) == 0
    np.testing.assert_array_equal(dev.cpu().numpy(), want)
    if n < 1 << 20:
        mask = (np.arange(n) % 3 != 1).astype(np.uint8)
        mf = mesh.sample_field(grid, invert=True, mask=mask, host_out=True)
        np.testing.assert_array_equal(mf.host_wait(), mesh.sample_nodes(grid, invert=True, mask=mask))
        # K3 and the device reduction straight off the handle
        h = 0.2
        rho = fld.density_map_field(h, 1000.0, True, host_out=True)
        wrho = dg.Field(grid, want).density_map_nodes(n, h, 1000.0, True)
        np.testing.assert_array_equal(rho.host_wait(), wrho)
        np.testing.assert_array_equal(rho.interpolate(P), dg.Field(grid, wrho).interpolate(P))
        a = fld.reduce(-0.15, 0.15, as_field=True)
        b = dg.reduce_field(grid, want, -0.15, 0.15)
        assert a[3] == b[3]
        if not a[3]:
            for x, y in zip(a[:3], b[:3]):
                np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(a[4].interpolate(P), dg.Field(grid, b[0], b[1], b[2]).interpolate(P))


def test_sample_field_destroyed_while_the_copy_runs():
    """dg_field_destroy on a handle whose host copy is still in flight waits for it (no write after free)."""
    import discregrid_amd as dg
    V, F = T.icosphere(10)
    dom = T.oracle_default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], [150, 150, 150])
    mesh = dg.Mesh(V, F)
    host = np.empty(dg.n_nodes(grid))
    fld = mesh.sample_field(grid, host_out=host)
    fld.close()
    np.testing.assert_array_equal(host, mesh.sample_nodes(grid))
