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


def _need(path):
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(BUILD)])
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
    env = dict(os.environ, DG_DEVICES="0,0,0", DG_HOST_CHUNK_NODES="1024")
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
    exe = _need(os.path.join(BUILD, "host_api_driver"))
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
