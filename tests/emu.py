"""ctypes binding of tests/emu/wave_emu.cpp (CPU lock-step emulation of the HIP kernels,
TEST INFRASTRUCTURE ONLY -- see the header of that file)."""
import ctypes as C
import os
import subprocess

import numpy as np

import dgtest as T

HERE = os.path.join(T.ROOT, "tests", "emu")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libwave_emu.so")
        csrc = os.path.join(T.ROOT, "discregrid_amd", "csrc")
        deps = [os.path.join(HERE, "wave_emu.cpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc)]
        def stale():
            return not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps)
        if stale():
            # (pytest -n: several workers arrive here at once; one builds -- into a temporary name, renamed when complete -- the others wait)
            import fcntl
            with open(so + ".lock", "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if stale():
                    tmp = so + ".%d.tmp" % os.getpid()
                    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                                           os.path.join(HERE, "wave_emu.cpp"), os.path.join(csrc, "dg_build.cpp"), "-o", tmp])
                    os.replace(tmp, so)
        L = C.CDLL(so)
        L.emu_mesh_create.restype = C.c_void_p
        L.emu_mesh_create.argtypes = [T.c_dp, C.c_size_t, T.c_up, C.c_size_t, C.c_int]
        L.emu_mesh_free.argtypes = [C.c_void_p]
        L.emu_mesh_info.argtypes = [C.c_void_p, T.c_u64p, T.c_up, T.c_up]
        L.emu_mesh_pseudonormals.argtypes = [C.c_void_p, T.c_dp]
        L.emu_mesh_check.argtypes = [C.c_void_p, T.c_dp, T.c_up]
        L.emu_sample_nodes.argtypes = [C.c_void_p, T.c_dp, T.c_dp, T.c_up, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                       C.c_void_p, T.c_dp, C.c_void_p, T.c_u64p]
        L.emu_set_heavy.argtypes = [C.c_uint32, C.c_int]
        L.emu_n_subtrees.argtypes = [C.c_void_p]
        L.emu_subtree_triangles.argtypes = [C.c_void_p]
        L.emu_subtree_triangles.restype = C.c_longlong
        L.emu_signed_distance.argtypes = [C.c_void_p, T.c_dp, C.c_uint64, T.c_dp, T.c_ip, T.c_ip, T.c_dp]
        L.emu_host_signed_distance.argtypes = [C.c_void_p, T.c_dp, C.c_uint64, C.c_int, T.c_dp, T.c_ip, T.c_ip, T.c_dp]
        L.emu_shard_count.restype = C.c_uint64
        L.emu_shard_count.argtypes = [T.c_up, C.c_int, C.c_int]
        L.emu_unpack.argtypes = [T.c_up, C.c_int, T.c_dp, C.c_uint64, T.c_dp]
        L.emu_density_map.argtypes = [T.c_dp, T.c_dp, T.c_dp, T.c_up, T.c_dp, T.c_up, T.c_up, C.c_double, C.c_double,
                                      C.c_int, C.c_uint64, C.c_uint64, T.c_dp]
        L.emu_interpolate.argtypes = [T.c_dp, T.c_dp, T.c_dp, T.c_up, T.c_dp, T.c_up, T.c_up, T.c_dp, C.c_uint64,
                                      T.c_dp, T.c_dp]
        _lib = L
    return _lib


class EmuMesh:
    def __init__(self, V, F, max_leaf=6):
        self.L = lib()
        self.V = np.ascontiguousarray(V, dtype=np.float64)
        self.F = np.ascontiguousarray(F, dtype=np.uint32)
        self.h = self.L.emu_mesh_create(T.dp(self.V), len(self.V), T.up(self.F), len(self.F), max_leaf)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emu_mesh_free(self.h)
            self.h = None

    def n_subtrees(self):
        return self.L.emu_n_subtrees(self.h)

    def subtree_triangles(self):
        return self.L.emu_subtree_triangles(self.h)

    def check(self):
        return self.L.emu_mesh_check(self.h, T.dp(self.V), T.up(self.F))

    def info(self):
        n = C.c_uint64()
        d = C.c_uint()
        f = C.c_uint()
        self.L.emu_mesh_info(self.h, C.byref(n), C.byref(d), C.byref(f))
        return dict(n_nodes=n.value, depth=d.value, flags=f.value)

    def pseudonormals(self):
        pn = np.zeros((len(self.F), 8, 3))
        self.L.emu_mesh_pseudonormals(self.h, T.dp(pn))
        return pn

    def _grid(self, domain, res):
        domain = np.ascontiguousarray(domain, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.uint32)
        cell = np.empty(3)
        inv = np.empty(3)
        T.oracle_lib().dgo_grid_header(T.dp(domain), T.up(res), T.dp(cell), T.dp(inv))
        return domain, res, cell, inv

    def sample_range(self, domain, res, begin=0, end=None, invert=False, mask=None, stats=False):
        domain, res, cell, _ = self._grid(domain, res)
        if end is None:
            end = T.n_nodes(res)
        out = np.full(end - begin, np.nan)
        written = np.zeros(end - begin, dtype=np.uint8)
        st = np.zeros(12, dtype=np.uint64)
        dmin = np.ascontiguousarray(domain[:3])
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.emu_sample_nodes(self.h, T.dp(dmin), T.dp(cell), T.up(res), int(invert), 0, begin, end,
                                None if m is None else m.ctypes.data_as(C.c_void_p), T.dp(out),
                                written.ctypes.data_as(C.c_void_p), st.ctypes.data_as(T.c_u64p))
        self.written = written
        self.stats = dict(zip(("bricks", "node_visits", "leaf_visits", "tri_tests", "descent_nodes", "slab_tests", "lane_interest", "useful_tests", "leaf_groups", "pops", "stale_pops", "heavy_bricks"), st.tolist()))
        return out

    def sample_shard(self, domain, res, rank, nranks, invert=False):
        domain, res, cell, _ = self._grid(domain, res)
        cnt = self.L.emu_shard_count(T.up(res), rank, nranks)
        out = np.full(cnt, np.nan)
        written = np.zeros(cnt, dtype=np.uint8)
        dmin = np.ascontiguousarray(domain[:3])
        self.L.emu_sample_nodes(self.h, T.dp(dmin), T.dp(cell), T.up(res), int(invert), 1, rank, nranks, None,
                                T.dp(out), written.ctypes.data_as(C.c_void_p), None)
        self.written = written
        return out

    def signed_distance(self, P, full=False):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
        n = len(P)
        d = np.empty(n)
        tri = np.empty(n, dtype=np.int32)
        ent = np.empty(n, dtype=np.int32)
        near = np.empty((n, 3))
        self.L.emu_signed_distance(self.h, T.dp(P), n, T.dp(d), T.ip(tri), T.ip(ent), T.dp(near))
        return (d, tri, ent, near) if full else d

    def host_signed_distance(self, P, full=False, threads=8):
        """The product's single-point host evaluator (dg_host_query.h), `threads` concurrent callers."""
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
        n = len(P)
        d = np.empty(n)
        tri = np.empty(n, dtype=np.int32)
        ent = np.empty(n, dtype=np.int32)
        near = np.empty((n, 3))
        self.L.emu_host_signed_distance(self.h, T.dp(P), n, threads, T.dp(d), T.ip(tri), T.ip(ent), T.dp(near))
        return (d, tri, ent, near) if full else d


def density_map(domain, res, coeffs, h, rho0, band=True, begin=0, end=None, cells=None, cell_map=None):
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    cell = np.empty(3)
    inv = np.empty(3)
    T.oracle_lib().dgo_grid_header(T.dp(domain), T.up(res), T.dp(cell), T.dp(inv))
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    if end is None:
        end = T.n_nodes(res)
    out = np.empty(end - begin)
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
        cell_map = np.ascontiguousarray(cell_map, dtype=np.uint32)
    lib().emu_density_map(T.dp(domain), T.dp(cell), T.dp(inv), T.up(res), T.dp(coeffs), T.up(cells), T.up(cell_map),
                          h, rho0, int(band), begin, end, T.dp(out))
    return out


def unpack(res, nranks, gathered, stride):
    res = np.ascontiguousarray(res, dtype=np.uint32)
    gathered = np.ascontiguousarray(gathered, dtype=np.float64)
    field = np.empty(T.n_nodes(res))
    lib().emu_unpack(T.up(res), nranks, T.dp(gathered), stride, T.dp(field))
    return field


def unpack_ranks(res, nranks, gathered, stride, r0, r1, field):
    res = np.ascontiguousarray(res, dtype=np.uint32)
    lib().emu_unpack_ranks.argtypes = [T.c_up, C.c_int, T.c_dp, C.c_uint64, C.c_int, C.c_int, T.c_dp]
    lib().emu_unpack_ranks(T.up(res), nranks, T.dp(gathered), stride, r0, r1, T.dp(field))


def shard_count(res, rank, nranks):
    res = np.ascontiguousarray(res, dtype=np.uint32)
    return lib().emu_shard_count(T.up(res), rank, nranks)


def interpolate(domain, res, coeffs, P, grad=False, cells=None, cell_map=None):
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    cell = np.empty(3)
    inv = np.empty(3)
    T.oracle_lib().dgo_grid_header(T.dp(domain), T.up(res), T.dp(cell), T.dp(inv))
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
    phi = np.empty(len(P))
    g = np.empty((len(P), 3)) if grad else None
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
        cell_map = np.ascontiguousarray(cell_map, dtype=np.uint32)
    lib().emu_interpolate(T.dp(domain), T.dp(cell), T.dp(inv), T.up(res), T.dp(coeffs), T.up(cells), T.up(cell_map),
                          T.dp(P), len(P), T.dp(phi), T.dp(g))
    return (phi, g) if grad else phi


def interpolate_rows(domain, res, coeffs, P, grad=False, cells=None, cell_map=None):
    """K2 through the cooperative row kernel's data path (k_interpolate_rows): cell-major copy, staged rows, split
    evaluation.  Same arguments and results as interpolate()."""
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    cell = np.empty(3)
    inv = np.empty(3)
    T.oracle_lib().dgo_grid_header(T.dp(domain), T.up(res), T.dp(cell), T.dp(inv))
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
    phi = np.empty(len(P))
    g = np.empty((len(P), 3)) if grad else None
    n_rows = int(np.prod(res.astype(np.uint64)))
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
        cell_map = np.ascontiguousarray(cell_map, dtype=np.uint32)
        n_rows = len(cells)
    lib().emu_interpolate_rows(T.dp(domain), T.dp(cell), T.dp(inv), T.up(res), T.dp(coeffs), T.up(cells), C.c_uint64(n_rows),
                               T.up(cell_map), T.dp(P), C.c_uint64(len(P)), T.dp(phi), T.dp(g))
    return (phi, g) if grad else phi


def set_fast(on=1):
    """Filtered K1 kernel (float filter + per-lane candidate lists) in the emulated launches; 0 = exact kernel
    only (DG_FORCE=k1_fast=0).  Resets the counters of fast_stats()."""
    lib().emu_set_fast(int(on))


def set_pool_cap(cap=0x7fffffff):
    """DG_FORCE=pool_cap: the most (owner, triangle) pairs the pooled epilogue of the filtered kernel takes (default: its LDS capacity)"""
    L = lib()
    L.emu_set_pool_cap.argtypes = [C.c_uint32]
    L.emu_set_pool_cap(int(cap))


def pool_stats():
    """(waves whose tails were pooled, waves whose tails ran lane by lane) since the last set_fast()"""
    st = np.zeros(2, dtype=np.uint64)
    lib().emu_pool_stats(st.ctypes.data_as(T.c_u64p))
    return int(st[0]), int(st[1])


def fast_stats():
    st = np.zeros(28, dtype=np.uint64)
    lib().emu_fast_stats(st.ctypes.data_as(T.c_u64p))
    names = ("bricks", "pair_steps", "leaf_visits", "tri_pairs", "appends", "resets", "redo_bricks", "sum_max_list",
             "sum_list", "lanes", "parked")
    d = dict(zip(names, st[:11].tolist()))
    d["hist"] = st[11:].tolist()
    return d


def filter_interval_check(rng, n_tri=2000, n_pts=64):
    """Random + adversarial (triangle, point) sets through emu_filter_check; returns the totals."""
    L = lib()
    L.emu_filter_check.restype = C.c_uint64
    L.emu_filter_check.argtypes = [T.c_dp, C.c_size_t, T.c_dp, C.c_size_t, T.c_dp, T.c_dp, T.c_u64p]
    tot = dict(violations=0, checked=0, worst=0.0)
    for case in range(12):
        scale = 10.0 ** rng.integers(-3, 4)
        shift = rng.uniform(-1, 1, size=3) * scale * (0.0 if case % 3 == 0 else 10.0 ** rng.integers(0, 4))
        tri = rng.uniform(-1, 1, size=(n_tri, 3, 3)) * scale
        k = n_tri // 4
        tri[:k, 2] = tri[:k, 0] + (tri[:k, 1] - tri[:k, 0]) * rng.uniform(-0.5, 1.5, size=(k, 1)) + \
            rng.normal(size=(k, 3)) * scale * 10.0 ** rng.uniform(-6, -1, size=(k, 1))      # slivers / needles
        tri[k:2 * k, 1] = tri[k:2 * k, 0] + rng.normal(size=(k, 3)) * scale * 1e-4              # one very short side
        tri += shift
        origin = 0.5 * (tri.reshape(-1, 3).min(0) + tri.reshape(-1, 3).max(0))
        for t0 in range(0, n_tri, 2 * 250):
            sub = np.ascontiguousarray(tri[t0:t0 + 500]).reshape(-1, 9)
            base = sub.reshape(-1, 3, 3)
            w = rng.dirichlet([0.3, 0.3, 0.3], size=n_pts)                                     # near vertices / sides
            on = np.einsum("pk,pkd->pd", w, base[rng.integers(0, len(base), n_pts)])
            pts = np.concatenate([
                on + rng.normal(size=(n_pts, 3)) * scale * 10.0 ** rng.uniform(-9, 0, size=(n_pts, 1)),
                rng.uniform(-3, 3, size=(n_pts // 2, 3)) * scale + shift,
                rng.uniform(-1, 1, size=(n_pts // 4, 3)) * scale * 1e3 + shift,
            ])
            pts = np.ascontiguousarray(pts)
            worst = C.c_double(0.0)
            checked = C.c_uint64(0)
            bad = L.emu_filter_check(T.dp(sub), len(sub), T.dp(pts), len(pts), T.dp(np.ascontiguousarray(origin)),
                                     C.byref(worst), C.byref(checked))
            tot["violations"] += int(bad)
            tot["checked"] += int(checked.value)
            tot["worst"] = max(tot["worst"], worst.value)
    return tot


def set_brick_blocking(on=1):
    """Emulated K1 launches enumerate the bricks in K3's blocked order (SampleParams::brick_blocking)."""
    lib().emu_set_brick_blocking(int(on))


def udiv_mismatches(n, d):
    """udiv_by(n, d, udiv_magic(d)) of dg_kernels.h (the brick map's scalar division) against n // d, n % d."""
    n = np.ascontiguousarray(n, dtype=np.uint32)
    d = np.ascontiguousarray(d, dtype=np.uint32)
    L = lib()
    L.emu_udiv_check.restype = C.c_uint64
    return int(L.emu_udiv_check(n.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_uint64(len(n))))


def set_tile_major(on=1):
    """K2 / K3 bodies read unreduced fields through a tile-major copy (dg_lattice.h) built as k_expand_tiles does
    (on = 2: through the x-major copy of the Y and Z classes, built as k_xmajor_copy / k_xmajor_flags do)."""
    lib().emu_set_tile_major(int(on))


def density_cells(domain, res, coeffs, h, rho0, band=True, begin=0, end=None, mask=None, block=(1, 16, 8)):
    """One emulated k_density_cells launch (a lane owns a lattice point with its seven nodes, dg_density_cells.h) over the
    node range: the values and, per node, how many lanes wrote it."""
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    block = np.ascontiguousarray(block, dtype=np.uint32)
    cell = np.empty(3)
    inv = np.empty(3)
    T.oracle_lib().dgo_grid_header(T.dp(domain), T.up(res), T.dp(cell), T.dp(inv))
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    if end is None:
        end = T.n_nodes(res)
    out = np.empty(end - begin)
    hits = np.zeros(end - begin, dtype=np.uint32)
    L = lib()
    L.emu_density_cells.restype = None
    L.emu_density_cells.argtypes = [T.c_dp, T.c_dp, T.c_dp, T.c_up, T.c_dp, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint64,
                                    C.c_void_p, T.c_up, T.c_dp, C.c_void_p]
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
    L.emu_density_cells(T.dp(domain), T.dp(cell), T.dp(inv), T.up(res), T.dp(coeffs), h, rho0, int(band), begin, end,
                        mask.ctypes.data_as(C.c_void_p) if mask is not None else None, T.up(block), T.dp(out),
                        hits.ctypes.data_as(C.c_void_p))
    return out, hits


def interpolate_band(domain, res, coeffs, lo, hi, P, grad=False, cells=None, cell_map=None):
    """K2 through a band-limited cell-major copy built on the host (bit / rank words as the device builds them);
    returns (phi, grad or None, rows in the copy, queries served from the copy)."""
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    cell = np.empty(3)
    inv = np.empty(3)
    T.oracle_lib().dgo_grid_header(T.dp(domain), T.up(res), T.dp(cell), T.dp(inv))
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
    n_rows = int(np.prod(res.astype(np.int64)))
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
        cell_map = np.ascontiguousarray(cell_map, dtype=np.uint32)
        n_rows = len(cells.reshape(-1, 32))
    phi = np.empty(len(P))
    g = np.empty((len(P), 3)) if grad else None
    rows = C.c_uint64(0)
    mapped = C.c_uint64(0)
    L = lib()
    L.emu_interpolate_band.restype = None
    L.emu_interpolate_band.argtypes = [T.c_dp, T.c_dp, T.c_dp, T.c_up, T.c_dp, T.c_up, T.c_up, C.c_uint64, C.c_double, C.c_double, T.c_dp,
                                       C.c_uint64, T.c_dp, T.c_dp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.emu_interpolate_band(T.dp(domain), T.dp(cell), T.dp(inv), T.up(res), T.dp(coeffs), T.up(cells), T.up(cell_map), n_rows, lo, hi, T.dp(P),
                           len(P), T.dp(phi), T.dp(g) if grad else None, C.byref(rows), C.byref(mapped))
    return phi, g, int(rows.value), int(mapped.value)


def div_h_mismatches(h, n, seed):
    """quotients of k3c_div_h() (dg_density_cells.h: d / h without the division) that differ from d / h."""
    L = lib()
    L.emu_div_h_check.restype = C.c_uint64
    L.emu_div_h_check.argtypes = [C.c_double, C.c_uint64, C.c_uint64]
    return int(L.emu_div_h_check(float(h), int(n), int(seed)))


def set_heavy(slots=0xffffffff, work=0):
    """Heavy-brick settings of the emulated K1 launches (dg_kernels.h: overflow_slots_for(), kHeavyWork);
    slots = 0 disables the split; the defaults size slots and budget as the product does."""
    lib().emu_set_heavy(slots, work)
