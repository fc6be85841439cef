"""The C-ABI shared library: builds for gfx950, loads, exports every symbol that
include/discregrid_hip.h declares, does its host-side arithmetic like the reference, and
FAILS LOUDLY (no CPU fallback) when there is no HIP device.  CPU only, no compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import dgtest as T


@pytest.fixture(scope="module")
def dg():
    from discregrid_amd.build import build
    build()
    import discregrid_amd
    discregrid_amd.load_library()
    return discregrid_amd


def header_symbols():
    src = open(os.path.join(T.ROOT, "include", "discregrid_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(dg):
    names = header_symbols()
    assert len(names) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", dg.LIB_PATH]).decode()
    exported = set(re.findall(r" T (dg_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(dg.SYMBOLS), set(names) ^ set(dg.SYMBOLS)
    # only dg_* is exported from the C ABI surface (C++ internals stay in namespace dg)
    assert all(n.startswith("dg_") for n in exported if not n.startswith("_"))


def test_library_exports_nothing_but_the_c_abi(dg):
    """Thin C ABI: the dynamic symbol table of the shipped library defines the entry points of the header and NOTHING else --
    no internal C++ function (fail, valid_grid, env_int, TraceRange ...), no data, no libstdc++ instantiation a host
    application of SPlisHSPlasH's size could interpose (-fvisibility=hidden + csrc/exports.map)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", dg.LIB_PATH]).decode()
    rows = [ln.split() for ln in out.splitlines() if ln.strip()]
    foreign = [r for r in rows if not r[-1].startswith("dg_")]
    assert not foreign, foreign[:10]
    assert all(r[-2] == "T" for r in rows), [r for r in rows if r[-2] != "T"][:10]
    assert sorted(r[-1] for r in rows) == header_symbols()
    # every prototype of the header carries the export attribute
    src = open(os.path.join(T.ROOT, "include", "discregrid_hip.h")).read()
    protos = re.findall(r"^(\S[^\n]*?)\bdg_[a-z0-9_]+\(", src, flags=re.M)
    assert len(protos) == len(header_symbols()) and all(p.startswith("DG_API ") for p in protos), [p for p in protos if not p.startswith("DG_API ")]


def test_library_contains_gfx950_code_object(dg):
    blob = open(dg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"k_sample_nodes" in blob and b"k_interpolate" in blob


def test_grid_desc_matches_reference_ctor(dg, golden):
    for name in ("box", "bunny", "torus"):
        dom, res = golden[name + "_domain"], golden[name + "_res"]
        g = dg.grid_desc(dom[:3], dom[3:], res)
        cell = np.empty(3)
        inv = np.empty(3)
        T.oracle_lib().dgo_grid_header(T.dp(np.ascontiguousarray(dom)), T.up(np.ascontiguousarray(res)), T.dp(cell),
                                       T.dp(inv))
        np.testing.assert_array_equal(np.array(g.cell_size), cell)
        np.testing.assert_array_equal(np.array(g.inv_cell_size), inv)
        assert dg.n_nodes(g) == T.n_nodes(res) == len(golden[name + "_coeffs"])
    assert dg.n_nodes(dg.grid_desc([0] * 3, [1] * 3, [256] * 3)) == 118425857
    assert dg.n_nodes(dg.grid_desc([0] * 3, [1] * 3, [512] * 3)) == 943460865


def test_default_domain_matches_reference_rule(dg, golden):
    """dg_default_domain == cmd/generate_sdf/main.cpp:83-91 (the oracle's restatement and the domains
    the unmodified reference computed for the golden vectors), bit for bit."""
    for name, (V, _) in (("box", T.box_mesh()), ("torus", T.torus()), ("bunny", T.bunny_mesh()),
                         ("ico", T.icosphere(7))):
        got = dg.default_domain(V)
        np.testing.assert_array_equal(got, T.oracle_default_domain(V))
        if name + "_domain" in golden:
            np.testing.assert_array_equal(got, golden[name + "_domain"])
    with pytest.raises(dg.DiscregridError):
        dg.default_domain(np.zeros((0, 3)))


def test_invalid_arguments_are_reported(dg):
    with pytest.raises(dg.DiscregridError) as e:
        dg.grid_desc([0] * 3, [1] * 3, [4, 0, 4])
    assert e.value.status == dg.DG_ERR_INVALID
    g = dg.grid_desc([0] * 3, [1] * 3, [4, 4, 4])
    with pytest.raises(dg.DiscregridError):
        dg.shard_layout(g, 3, 2)
    counts = [dg.shard_layout(g, r, 3)[0] for r in range(3)]
    assert sum(counts) == dg.n_nodes(g)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_fails_loudly(dg):
    """No CPU path behind the ABI: on a machine without a HIP device every entry point that would launch a kernel returns
    DG_ERR_NO_DEVICE instead of silently computing on the host.  What works without one (round 4): a HOST-ONLY
    mesh handle (BVH + pseudonormals in host memory, dg_mesh_device() == -1) and the per-point query on it -- the
    reference's TriangleMeshDistance::signed_distance(point) -- with the bits of the oracle."""
    assert dg.device_count() == 0
    V, F = T.torus()
    m = dg.Mesh(V, F)
    assert m.device() == -1 and m.info()["device_bytes"] == 0 and m.info()["n_triangles"] == len(F)
    g = dg.grid_desc([-1.5] * 3, [1.5] * 3, [3, 3, 3])
    for call in (lambda: m.sample_nodes(g), lambda: m.signed_distance(np.zeros((4, 3)))):
        with pytest.raises(dg.DiscregridError) as e:
            call()
        assert e.value.status == dg.DG_ERR_NO_DEVICE
        assert "no CPU path" in str(e.value)
    P = np.random.default_rng(4).uniform(-1.5, 1.5, size=(300, 3))
    got = np.array([m.signed_distance_point(p) for p in P])
    np.testing.assert_array_equal(got, T.OracleMesh(V, F).signed_distance(P))
    with pytest.raises(dg.DiscregridError) as e:
        dg.Field(g, np.zeros(dg.n_nodes(g)))
    assert e.value.status == dg.DG_ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    """The product tree must not import/link/execute anything under oracle/ or tests/."""
    bad = []
    for top in (os.path.join(T.ROOT, "discregrid_amd"), os.path.join(T.ROOT, "include")):
        for root, dirs, files in os.walk(top):
            dirs[:] = [d for d in dirs if d not in ("build", "variants", "__pycache__")]
            for f in files:
                txt = None
                if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp")) or f in ("All", "Dense", "Core"):
                    txt = open(os.path.join(root, f), errors="ignore").read()
                    pattern = r"oracle/|libdiscregrid_oracle|libdiscregrid_ref|wave_emu|dgtest"
                elif f.endswith((".mk", ".sh", ".cmake")) or f in ("Makefile", "CMakeLists.txt"):
                    # build recipes must not even mention the test trees
                    txt = open(os.path.join(root, f), errors="ignore").read()
                    pattern = r"oracle|tests/|wave_emu|dgtest"
                if txt is not None and re.search(pattern, txt):
                    bad.append(os.path.join(root, f))
    assert not bad, bad
    out = subprocess.check_output(["ldd", os.path.join(T.ROOT, "discregrid_amd", "libdiscregrid_hip.so")]).decode()
    assert "oracle" not in out and "libamdhip64" in out
