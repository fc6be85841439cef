"""The Discregrid-compatible C++ host API (discregrid_amd/cpp): file format, scalar evaluator,
generic-callback addFunction, reduceField -- against the reference's golden files and the
oracle.  CPU only (the GPU-backed parts are in test_gpu_host_api.py)."""
import os
import subprocess

import numpy as np
import pytest

import dgtest as T

DBL_MAX = np.finfo(np.float64).max
CPP = os.path.join(T.ROOT, "discregrid_amd", "cpp")
DRIVER = os.path.join(T.ROOT, "tests", "cpp", "build", "host_api_driver")


@pytest.fixture(scope="module")
def driver():
    from discregrid_amd.build import build
    build()
    subprocess.check_call(["make", "-s", "-C", CPP])
    subprocess.check_call(["make", "-s", "-C", os.path.join(T.ROOT, "tests", "cpp")])
    return DRIVER


def run(driver, *args):
    subprocess.check_call([driver] + [str(a) for a in args])


def read(path):
    with open(path, "rb") as f:
        return f.read()


@pytest.mark.parametrize("name", ["box.cdf", "box_reduced_0p25.cdf", "torus_9_14_6.cdf", "torus_9_14_6_reduced_0p08.cdf"])
def test_load_save_roundtrip_is_byte_identical(driver, tmp_path, name):
    """Loads files written by the reference (unreduced and reduced) and writes them back."""
    src = os.path.join(T.GOLDEN, name)
    out = str(tmp_path / "rt.cdf")
    run(driver, "roundtrip", src, out)
    assert read(out) == read(src)


def test_generic_callback_addfunction(driver, tmp_path):
    """addFunction with an arbitrary host callable (+ SamplePredicate) == reference semantics."""
    out = str(tmp_path / "poly.cdf")
    run(driver, "poly", out)
    dom = np.array([-1.0, -0.5, 0.0, 1.5, 0.75, 2.0])
    res = [3, 4, 2]
    p = T.oracle_node_positions(dom, res)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    f0 = 1 + x - 2 * y + 0.5 * z + x * y - y * z + x * x * z - 0.3 * y * y * y + 0.7 * x * y * z
    f1 = np.where(x < 0.25, f0, DBL_MAX)
    want = str(tmp_path / "want.cdf")
    T.oracle_write_cdf(want, dom, res, [f0, f1])
    got = T.read_cdf(out)
    np.testing.assert_array_equal(got["nodes"][0], f0)
    np.testing.assert_array_equal(got["nodes"][1], f1)
    assert read(out) == read(want)


@pytest.mark.parametrize("src,bound,golden", [("box.cdf", 0.25, "box_reduced_0p25.cdf"),
                                               ("torus_9_14_6.cdf", 0.08, "torus_9_14_6_reduced_0p08.cdf")])
def test_reduce_field_matches_reference_file(driver, tmp_path, src, bound, golden):
    """reduceField (cell drop, node compaction, Morton re-sort, renumbering) reproduces the
    file the reference writes, byte for byte."""
    out = str(tmp_path / "red.cdf")
    run(driver, "reduce", os.path.join(T.GOLDEN, src), bound, out)
    assert read(out) == read(os.path.join(T.GOLDEN, golden))
    g = T.read_cdf(out)
    assert len(g["nodes"][0]) < T.n_nodes(g["res"]) and (g["cell_map"][0] == 0xFFFFFFFF).any()


@pytest.mark.skipif(not T.ref_available(), reason="oracle/_ref not built")
def test_reduce_field_against_live_reference(driver, tmp_path):
    V, F = T.icosphere(8)
    dom = T.ref_default_domain(V)
    g = T.RefGrid(V, F, dom, [11, 9, 10])
    g.add_sdf()
    src = str(tmp_path / "src.cdf")
    g.save(src)
    g.reduce_abs_lt(0, 0.11)
    want = str(tmp_path / "want.cdf")
    g.save(want)
    out = str(tmp_path / "got.cdf")
    run(driver, "reduce", src, 0.11, out)
    assert read(out) == read(want)


@pytest.mark.parametrize("name", ["torus_9_14_6.cdf", "torus_9_14_6_reduced_0p08.cdf", "box_reduced_0p25.cdf"])
@pytest.mark.parametrize("mode", ["eval", "evalsplit"])
def test_scalar_interpolate(driver, tmp_path, name, mode):
    """interpolate(field, x, grad*) and the determineShapeFunctions / split-interpolate pair on
    unreduced (implicit table) and reduced (explicit table) fields vs the oracle."""
    src = os.path.join(T.GOLDEN, name)
    g = T.read_cdf(src)
    dom = g["domain"]
    ext = dom[3:] - dom[:3]
    rng = np.random.default_rng(21)
    P = rng.uniform(dom[:3] - 0.05 * ext, dom[3:] + 0.05 * ext, size=(3000, 3))
    P[:4] = dom[3:]
    P[4:8] = dom[:3]
    pts = str(tmp_path / "pts.bin")
    P.tofile(pts)
    out = str(tmp_path / "out.bin")
    run(driver, mode, src, pts, out)
    got = np.fromfile(out).reshape(-1, 5)
    phi, grad = T.oracle_interpolate(dom, g["res"], g["nodes"][0], P, grad=True, cells=g["cells"][0],
                                     cell_map=g["cell_map"][0])
    np.testing.assert_array_equal(got[:, 0], phi)
    np.testing.assert_array_equal(got[:, 1], phi)
    ok = phi != DBL_MAX
    assert ok.any() and (~ok).any()
    np.testing.assert_array_equal(got[ok, 2:], grad[ok])
    assert (got[~ok, 2:] == 0).all()


def test_headers_compile_standalone(tmp_path):
    """Every public header is self-contained (compiles on its own against the Eigen stand-in)."""
    inc = os.path.join(CPP, "include")
    for h in ("Discregrid/All", "Discregrid/discrete_grid.hpp", "Discregrid/cubic_lagrange_discrete_grid.hpp",
              "Discregrid/geometry/TriangleMeshDistance.h", "Discregrid/mesh/triangle_mesh.hpp"):
        src = tmp_path / "t.cpp"
        src.write_text("#include <%s>\nint main() { return 0; }\n" % h)
        subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-I" + inc,
                               "-I" + os.path.join(CPP, "third_party", "eigen_min"), str(src)])


def _no_gpu_env(how="force"):
    """the C++ API without a HIP device: on the build box there is none; on a GPU box either DG_FORCE_CPU=1 makes the mesh
    handle host-only (include/discregrid_hip.h dg_mesh_device()) or HIP_VISIBLE_DEVICES= hides the devices from the runtime."""
    if how == "hidden":
        return dict(os.environ, HIP_VISIBLE_DEVICES="")
    return dict(os.environ, DG_FORCE_CPU="1")


@pytest.mark.parametrize("how", ["force", "hidden"])
def test_generate_sdf_cli_without_a_device_reproduces_box_cdf(driver, tmp_path, how):
    """without a HIP device the typed MeshSDF functor falls back to the reference's OpenMP node loop over the
    per-point query (the product's own BVH and arithmetic on the host).  GenerateSDF -r "5 5 5" box.obj still writes the
    reference's box.cdf, byte for byte."""
    exe = os.path.join(CPP, "build", "GenerateSDF")
    obj = str(tmp_path / "box.obj")
    V, F = T.box_mesh()
    T.write_obj(obj, V, F)
    out = str(tmp_path / "box.cdf")
    log = subprocess.check_output([exe, "-r", "5 5 5", "-o", out, obj], env=_no_gpu_env(how)).decode()
    assert "Construction took" in log and "DONE" in log
    assert read(out) == read(os.path.join(T.GOLDEN, "box.cdf"))


@pytest.mark.parametrize("how", ["force", "hidden"])
def test_unchanged_reference_caller_and_typed_functor_without_a_device(driver, tmp_path, how):
    """tests/cpp/unchanged_caller.cpp (the body of the reference's GenerateSDF with the reference's own lambda) and the typed
    MeshSDF functor on a box without a GPU: same coefficients from both, equal to the oracle bit for bit; the batched
    signed_distance falls back to the per-point query as well."""
    exe = os.path.join(T.ROOT, "tests", "cpp", "build", "unchanged_caller")
    V, F = T.torus()
    obj = str(tmp_path / "torus.obj")
    T.write_obj(obj, V, F)
    out = str(tmp_path / "out.bin")
    subprocess.check_call([exe, "lambda", obj, "9 7 8", out], env=_no_gpu_env(how))
    got = np.fromfile(out)
    n = int(got[0])
    assert n == T.n_nodes([9, 7, 8]) and got[3] == 0.0          # lambda path == typed path
    dom = T.oracle_default_domain(V)
    om = T.OracleMesh(V, F)
    np.testing.assert_array_equal(got[4:], om.sample_nodes(dom, [9, 7, 8]))
    out2 = str(tmp_path / "cpu.bin")
    n2 = T.n_nodes([6, 5, 7])
    P = T.oracle_node_positions(dom, [6, 5, 7])[(np.arange(1000) * 7919) % n2] + np.array([0.01, -0.02, 0.005])
    pts = str(tmp_path / "pts.bin")
    P.tofile(pts)
    subprocess.check_call([driver, "cpu", obj, "6 5 7", pts, out2], env=_no_gpu_env(how))
    got = np.fromfile(out2)
    assert got[0] == 0.0                                        # lastAddFunctionUsedGpu() == false
    want = om.sample_nodes(dom, [6, 5, 7])
    np.testing.assert_array_equal(got[2:2 + n2], want)
    np.testing.assert_array_equal(got[2 + n2:], om.signed_distance(P))


def test_a_failed_addfunction_leaves_the_grid_consistent(driver, tmp_path):
    """A MeshSDF addFunction that throws (here: host-only mesh under DG_REQUIRE_GPU=1) unwinds the entries it pushed:
    nFields() is unchanged, the next addFunction returns the id that indexes its data, scalar interpolate and save / load see
    exactly the fields that exist (round-3 review: the GPU branch threw after the push_backs without popping)."""
    V, F = T.torus()
    obj = str(tmp_path / "torus.obj")
    T.write_obj(obj, V, F)
    out = str(tmp_path / "two_fields.cdf")
    r = subprocess.run([driver, "fault", obj, out], env=dict(os.environ, DG_FORCE_CPU="1", DG_REQUIRE_GPU="1"),
                       stdout=subprocess.PIPE)
    assert r.returncode == 0, (r.returncode, r.stdout.decode())
    assert b"failed as arranged" in r.stdout
    g = T.read_cdf(out)
    assert len(g["nodes"]) == 2
    np.testing.assert_array_equal(g["nodes"][1], 2.0 * g["nodes"][0])


STRICT_EIGEN = os.path.join(T.ROOT, "tests", "cpp", "eigen_strict")


def _syntax_only(source, eigen):
    cpp = os.path.join(T.ROOT, "discregrid_amd", "cpp")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-fopenmp", "-D__HIP_PLATFORM_AMD__", "-I" + eigen, "-I" + os.path.join(cpp, "include"),
           "-I" + os.path.join(T.ROOT, "include"), "-I" + os.path.join(T.ROOT, "discregrid_amd", "csrc"), "-I/opt/rocm/include", "-x", "c++", source]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_host_api_compiles_against_a_strict_eigen_stand_in(tmp_path):
    """Eigen3 is absent from the image: the public headers have only ever met the EAGER stand-in (third_party/eigen_min), whose
    accidents -- operators that return finished matrices, public members -- code could silently lean on.  Every source of the
    host library, the four tools, the C++ test drivers and the UNCHANGED reference caller must also compile against a second
    stand-in that behaves like real Eigen at compile time (tests/cpp/eigen_strict: arithmetic returns expression proxies,
    storage and box members are private, no pointer / bool / scalar conversions, no vector * vector) ..."""
    import glob
    cpp = os.path.join(T.ROOT, "discregrid_amd", "cpp")
    sources = sorted(glob.glob(os.path.join(cpp, "src", "*.cpp")) + glob.glob(os.path.join(cpp, "cmd", "*.cpp")) +
                     glob.glob(os.path.join(T.ROOT, "tests", "cpp", "*.cpp")))
    assert len(sources) >= 10
    umbrella = tmp_path / "umbrella.cpp"
    umbrella.write_text("#include <Discregrid/All>\nint main() { Discregrid::CubicLagrangeDiscreteGrid g(Eigen::AlignedBox3d(Eigen::Vector3d(0, 0, 0), "
                        "Eigen::Vector3d(1, 1, 1)), {{2u, 2u, 2u}}); return (int)g.nCells(); }\n")
    for src in sources + [str(umbrella)]:
        out = _syntax_only(src, STRICT_EIGEN)
        assert out.returncode == 0, "%s does not compile against the strict Eigen stand-in:\n%s" % (src, out.stderr[-3000:])


@pytest.mark.parametrize("snippet", [
    "struct D { template <int N> static double f(Eigen::Matrix<double, N, 1> const& v) { return v[0]; } }; double d = D::f(a + b);",  # an expression is not a Matrix<...>: nothing to deduce
    "double* p = a;",                                        # no conversion to a pointer
    "Eigen::Vector3d c = a * b;",                            # no product of two column vectors
    "double m = box.m_min[0];",                              # box members are private
    "Eigen::Vector3d c = 0.0;",                              # no construction from a scalar
    "Eigen::Matrix<double, 32, 3> dn; double v = dn[5];",    # operator[] is for vectors
    "Eigen::Matrix<double, 2, 1> c = a + b;",                # sizes must agree
])
def test_the_strict_stand_in_rejects_what_eigen_rejects(tmp_path, snippet):
    """... and the strict stand-in is strict: constructs the eager stand-in accepts and real Eigen refuses do not compile."""
    src = tmp_path / "neg.cpp"
    src.write_text("#include <Eigen/Dense>\nint main(int argc, char**) { bool flag = argc > 1; Eigen::Vector3d a(1, 2, 3), b(4, 5, 6); "
                   "Eigen::AlignedBox3d box(a, b); (void)flag; (void)box;\n" + snippet + "\nreturn 0; }\n")
    assert _syntax_only(str(src), STRICT_EIGEN).returncode != 0, "the strict stand-in accepted: " + snippet
