"""discregrid_amd -- MI355X (gfx950) implementation of Discregrid's SDF-discretisation hot path.

The product is the C-ABI shared library ``libdiscregrid_hip.so`` (hand-written HIP kernels,
``include/discregrid_hip.h``) plus the Discregrid-compatible C++ host API under
``discregrid_amd/cpp``.  This Python module is a thin ``ctypes`` binding of that C ABI used
by the tests and by ``bench.py``; it adds no compute of its own and has NO fallback: if the
library is missing or no HIP device is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiscregrid_hip.so")
NO_VALUE = float(np.finfo(np.float64).max)

DG_OK, DG_ERR_INVALID, DG_ERR_NO_DEVICE, DG_ERR_HIP, DG_ERR_ALLOC = range(5)


class DiscregridError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("discregrid_hip status %d: %s" % (status, message))
        self.status = status


class GridDesc(C.Structure):
    """dg_grid_desc: the serialised members of Discregrid::DiscreteGrid."""
    _fields_ = [("domain_min", C.c_double * 3), ("domain_max", C.c_double * 3), ("resolution", C.c_uint32 * 3),
                ("reserved_", C.c_uint32), ("cell_size", C.c_double * 3), ("inv_cell_size", C.c_double * 3)]


class MeshInfo(C.Structure):
    _fields_ = [("n_vertices", C.c_uint64), ("n_triangles", C.c_uint64), ("n_bvh_nodes", C.c_uint64),
                ("bvh_depth", C.c_uint32), ("not_watertight", C.c_uint32), ("device_bytes", C.c_uint64),
                ("build_seconds", C.c_double)]


class FieldInfo(C.Structure):
    _fields_ = [("n_coeffs", C.c_uint64), ("n_cell_rows", C.c_uint64), ("device_bytes", C.c_uint64),
                ("d_coeffs", C.c_void_p), ("device", C.c_int32), ("owns_coefficients", C.c_int32),
                ("has_cell_major", C.c_int32), ("has_tile_major", C.c_int32), ("immutable", C.c_int32),
                ("host_copy_pending", C.c_int32), ("band_rows", C.c_uint64)]


class ShardInfo(C.Structure):
    _fields_ = [("count", C.c_uint64), ("stride", C.c_uint64)]


_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

# every symbol include/discregrid_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "dg_version": (C.c_char_p, []),
    "dg_last_error": (C.c_char_p, []),
    "dg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dg_set_device": (C.c_int, [C.c_int]),
    "dg_current_device": (C.c_int, [C.POINTER(C.c_int)]),
    "dg_grid_desc_init": (C.c_int, [_dp, _dp, _u32p, C.POINTER(GridDesc)]),
    "dg_default_domain": (C.c_int, [_dp, C.c_uint64, _dp]),
    "dg_sdf_sample_nodes_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(GridDesc), C.c_int, C.c_uint64,
                                            C.c_uint64, C.c_void_p, _dp]),
    "dg_grid_n_nodes": (C.c_uint64, [C.POINTER(GridDesc)]),
    "dg_grid_n_cells": (C.c_uint64, [C.POINTER(GridDesc)]),
    "dg_mesh_create": (C.c_int, [_dp, C.c_size_t, _u32p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dg_mesh_get_info": (C.c_int, [C.c_void_p, C.POINTER(MeshInfo)]),
    "dg_mesh_device": (C.c_int, [C.c_void_p]),
    "dg_mesh_destroy": (None, [C.c_void_p]),
    "dg_sdf_sample_nodes": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, C.c_uint64, C.c_uint64, _u8p, _dp]),
    "dg_sdf_sample_nodes_device": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, C.c_uint64, C.c_uint64,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_signed_distance": (C.c_int, [C.c_void_p, _dp, C.c_uint64, _dp, _i32p, _i32p, _dp]),
    "dg_signed_distance_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "dg_signed_distance_point": (C.c_int, [C.c_void_p, _dp, _dp, _i32p, _i32p, _dp]),
    "dg_set_progress_callback": (None, [C.c_void_p, C.c_void_p]),
    "dg_shard_layout": (C.c_int, [C.POINTER(GridDesc), C.c_int, C.c_int, C.POINTER(ShardInfo)]),
    "dg_sdf_sample_shard_device": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p]),
    "dg_unpack_shards_device": (C.c_int, [C.POINTER(GridDesc), C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                          C.c_void_p]),
    "dg_unpack_shard_range_device": (C.c_int, [C.POINTER(GridDesc), C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p]),
    "dg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dg_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dg_comm_adopt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dg_comm_destroy": (None, [C.c_void_p]),
    "dg_sdf_sample_allgather_device": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p]),
    "dg_chunk_layout": (C.c_int, [C.POINTER(GridDesc), C.c_int, C.c_void_p, _u32p]),
    "dg_sdf_sample_planes_device": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, _u32p, _u32p, C.c_void_p, C.c_void_p]),
    "dg_sdf_sample_exchange_device": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_comm_last_chunk_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "dg_comm_last_exchange_wait_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "dg_comm_create_external": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dg_comm_get_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dg_comm_create_shm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dg_comm_field_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "dg_comm_field_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dg_host_field_open": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dg_host_field_data": (C.c_void_p, [C.c_void_p]),
    "dg_host_field_barrier": (C.c_int, [C.c_void_p]),
    "dg_host_field_get_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dg_host_field_close": (None, [C.c_void_p]),
    "dg_sdf_sample_to_host_field": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "dg_host_field_last_chunk_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "dg_field_create": (C.c_int, [C.POINTER(GridDesc), _dp, C.c_uint64, _u32p, C.c_uint64, _u32p,
                                  C.POINTER(C.c_void_p)]),
    "dg_field_attach_device": (C.c_int, [C.POINTER(GridDesc), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.POINTER(C.c_void_p)]),
    "dg_field_destroy": (None, [C.c_void_p]),
    "dg_field_get_info": (C.c_int, [C.c_void_p, C.POINTER(FieldInfo)]),
    "dg_field_set_immutable": (C.c_int, [C.c_void_p, C.c_int]),
    "dg_sdf_sample_field": (C.c_int, [C.c_void_p, C.POINTER(GridDesc), C.c_int, _u8p, _dp, C.c_int, C.POINTER(C.c_void_p)]),
    "dg_density_map_field": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, _u8p, _dp, C.POINTER(C.c_void_p)]),
    "dg_field_host_wait": (C.c_int, [C.c_void_p]),
    "dg_field_cache_trim": (None, []),
    "dg_reduce_field_device": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "dg_reduction_to_field": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "dg_field_build_cell_major": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dg_field_drop_cell_major": (C.c_int, [C.c_void_p]),
    "dg_field_build_cell_major_band": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_uint64)]),
    "dg_field_build_tile_major": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dg_field_drop_tile_major": (C.c_int, [C.c_void_p]),
    "dg_interpolate_batch": (C.c_int, [C.c_void_p, _dp, C.c_uint64, _dp, _dp]),
    "dg_interpolate_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "dg_reduce_field": (C.c_int, [C.POINTER(GridDesc), _dp, C.c_uint64, C.c_int, C.c_double, C.c_double, C.c_double,
                                  C.POINTER(C.c_void_p)]),
    "dg_reduction_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "dg_reduction_fetch": (C.c_int, [C.c_void_p, _dp, _u32p, _u32p]),
    "dg_reduction_destroy": (None, [C.c_void_p]),
    "dg_density_map_nodes": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint64, _u8p, _dp]),
    "dg_density_map_nodes_device": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_uint64, C.c_uint64,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "dg_last_kernel_ms": (C.c_double, []),
    "dg_mesh_last_heavy_bricks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "dg_mesh_last_epilogue_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}

_lib = None


def load_library(path=None):
    """Loads libdiscregrid_hip.so and types every entry point.  Raises (never falls back)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("DG_LIB") or LIB_PATH  # DG_LIB: experiment builds (tools/)
    # PyTorch wheels bundle their own libamdhip64.so.7; a process must use ONE HIP runtime.  When
    # torch is installed (tests, bench.py use it for device memory / streams / torch.distributed)
    # let it load its runtime first so that this library binds to the same one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -m discregrid_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(status):
    if status != DG_OK:
        raise DiscregridError(status, load_library().dg_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    _check(load_library().dg_device_count(C.byref(n)))
    return n.value


def set_device(i):
    _check(load_library().dg_set_device(int(i)))


def grid_desc(domain_min, domain_max, resolution):
    """dg_grid_desc_init: cell_size = (max - min) / n, inv_cell_size = 1 / cell_size."""
    g = GridDesc()
    mn, mx = _f64(domain_min), _f64(domain_max)
    res = np.ascontiguousarray(resolution, dtype=np.uint32)
    _check(load_library().dg_grid_desc_init(mn.ctypes.data_as(_dp), mx.ctypes.data_as(_dp), res.ctypes.data_as(_u32p),
                                            C.byref(g)))
    return g


def default_domain(verts):
    """dg_default_domain: GenerateSDF's default domain (cmd/generate_sdf/main.cpp:83-91) -> 6 doubles."""
    v = _f64(verts).reshape(-1, 3)
    out = np.empty(6)
    _check(load_library().dg_default_domain(v.ctypes.data_as(_dp), len(v), out.ctypes.data_as(_dp)))
    return out


def n_nodes(grid):
    return int(load_library().dg_grid_n_nodes(C.byref(grid)))


def n_cells(grid):
    return int(load_library().dg_grid_n_cells(C.byref(grid)))


def shard_layout(grid, rank, nranks):
    s = ShardInfo()
    _check(load_library().dg_shard_layout(C.byref(grid), rank, nranks, C.byref(s)))
    return int(s.count), int(s.stride)


class Mesh:
    """dg_mesh handle: pseudonormals + flattened BVH + triangle packets on the current device."""

    def __init__(self, vertices, triangles):
        self._lib = load_library()
        V = _f64(vertices).reshape(-1, 3)
        F = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        h = C.c_void_p()
        _check(self._lib.dg_mesh_create(V.ctypes.data_as(_dp), len(V), F.ctypes.data_as(_u32p), len(F), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.dg_mesh_destroy(self.handle)
            self.handle = None

    __del__ = close

    def info(self):
        i = MeshInfo()
        _check(self._lib.dg_mesh_get_info(self.handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in MeshInfo._fields_}

    def device(self):
        """dg_mesh_device: the handle's device, -1 for a host-only handle (no HIP device / DG_FORCE_CPU=1)."""
        return int(self._lib.dg_mesh_device(self.handle))

    def last_heavy_bricks(self):
        """dg_mesh_last_heavy_bricks: (bricks over budget, bricks actually split) of the last launch."""
        heavy, split = C.c_uint32(0), C.c_uint32(0)
        _check(self._lib.dg_mesh_last_heavy_bricks(self.handle, C.byref(heavy), C.byref(split)))
        return int(heavy.value), int(split.value)

    def last_epilogue_stats(self):
        """dg_mesh_last_epilogue_stats: (waves that pooled their tails, waves that ran them lane by lane) of the last launch
        made with DG_FORCE=pool_stats=1."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _check(self._lib.dg_mesh_last_epilogue_stats(self.handle, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- host-pointer entry points -----------------------------------------------------------
    def sample_nodes(self, grid, begin=0, end=None, invert=False, mask=None):
        if end is None:
            end = n_nodes(grid)
        out = np.empty(end - begin, dtype=np.float64)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == end - begin
        _check(self._lib.dg_sdf_sample_nodes(self.handle, C.byref(grid), int(invert), begin, end,
                                             None if m is None else m.ctypes.data_as(_u8p), out.ctypes.data_as(_dp)))
        return out

    def signed_distance(self, points, full=False):
        P = _f64(points).reshape(-1, 3)
        n = len(P)
        d = np.empty(n)
        tri = np.empty(n, dtype=np.int32) if full else None
        ent = np.empty(n, dtype=np.int32) if full else None
        near = np.empty((n, 3)) if full else None
        _check(self._lib.dg_signed_distance(self.handle, P.ctypes.data_as(_dp), n, d.ctypes.data_as(_dp),
                                            None if tri is None else tri.ctypes.data_as(_i32p),
                                            None if ent is None else ent.ctypes.data_as(_i32p),
                                            None if near is None else near.ctypes.data_as(_dp)))
        return (d, tri, ent, near) if full else d

    def signed_distance_point(self, p, full=False):
        """dg_signed_distance_point: one point, evaluated on the calling host thread."""
        x = _f64(p).reshape(3)
        d = C.c_double()
        tri, ent = C.c_int32(), C.c_int32()
        near = np.empty(3)
        _check(self._lib.dg_signed_distance_point(self.handle, x.ctypes.data_as(_dp), C.byref(d), C.byref(tri),
                                                  C.byref(ent), near.ctypes.data_as(_dp)))
        return (d.value, tri.value, ent.value, near) if full else d.value

    def sample_field(self, grid, invert=False, mask=None, host_out=None, host_first=True):
        """dg_sdf_sample_field: K1 into a new device-resident Field; host_out (a float64 array of n_nodes, or True to
        have one allocated) is filled asynchronously -- Field.host_wait() returns it complete."""
        n = n_nodes(grid)
        if host_out is True:
            host_out = np.empty(n, dtype=np.float64)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert len(m) == n
        h = C.c_void_p()
        _check(self._lib.dg_sdf_sample_field(self.handle, C.byref(grid), int(invert),
                                             None if m is None else m.ctypes.data_as(_u8p),
                                             None if host_out is None else host_out.ctypes.data_as(_dp), int(host_first),
                                             C.byref(h)))
        return Field._adopt(h, host_out)

    # ---- device-pointer entry points (ints = device addresses, stream = hipStream_t address) ---
    def sample_nodes_device(self, grid, begin, end, d_out, invert=False, d_mask=None, stream=0):
        _check(self._lib.dg_sdf_sample_nodes_device(self.handle, C.byref(grid), int(invert), begin, end,
                                                    C.c_void_p(d_mask or 0), C.c_void_p(d_out), C.c_void_p(stream)))

    def sample_shard_device(self, grid, rank, nranks, d_packed, invert=False, stream=0):
        _check(self._lib.dg_sdf_sample_shard_device(self.handle, C.byref(grid), int(invert), rank, nranks,
                                                    C.c_void_p(d_packed), C.c_void_p(stream)))

    def signed_distance_device(self, d_xyz, n, d_dist, d_tri=0, d_entity=0, d_nearest=0, stream=0):
        _check(self._lib.dg_signed_distance_device(self.handle, C.c_void_p(d_xyz), n, C.c_void_p(d_dist),
                                                   C.c_void_p(d_tri), C.c_void_p(d_entity), C.c_void_p(d_nearest),
                                                   C.c_void_p(stream)))


def sample_nodes_multi(meshes, grid, begin=0, end=None, invert=False, mask=None):
    """dg_sdf_sample_nodes_multi: one Mesh per device (or several on one), results in host memory."""
    if end is None:
        end = n_nodes(grid)
    out = np.empty(end - begin, dtype=np.float64)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    handles = (C.c_void_p * len(meshes))(*[x.handle for x in meshes])
    _check(load_library().dg_sdf_sample_nodes_multi(handles, len(meshes), C.byref(grid), int(invert), begin, end,
                                                    None if m is None else m.ctypes.data_as(C.c_void_p),
                                                    out.ctypes.data_as(_dp)))
    return out


EXCHANGE_INPLACE, EXCHANGE_P2P, EXCHANGE_TO_ROOT, EXCHANGE_COPY = 1, 2, 4, 8


class CommInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("device", C.c_int32), ("rccl_nranks", C.c_int32),
                ("registered_fields", C.c_int32)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


def _plane_cost_arg(plane_cost):
    """four float32 arrays (one per class, nullable entries) -> (const float* const[4], keep-alive list)"""
    if plane_cost is None:
        return None, []
    keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in plane_cost]
    arr = (C.c_void_p * 4)(*[None if a is None else a.ctypes.data for a in keep])
    return arr, keep


def chunk_layout(grid, nchunks, plane_cost=None):
    """dg_chunk_layout: cuts[4][nchunks + 1], first plane of every contiguous chunk of every class."""
    arr, keep = _plane_cost_arg(plane_cost)
    cuts = np.empty((4, nchunks + 1), dtype=np.uint32)
    _check(load_library().dg_chunk_layout(C.byref(grid), nchunks, arr, cuts.ctypes.data_as(_u32p)))
    return cuts


def sample_planes_device(mesh, grid, plane_begin, plane_end, d_field, invert=False, stream=0):
    b = np.ascontiguousarray(plane_begin, dtype=np.uint32)
    e = np.ascontiguousarray(plane_end, dtype=np.uint32)
    _check(load_library().dg_sdf_sample_planes_device(mesh.handle, C.byref(grid), int(invert), b.ctypes.data_as(_u32p),
                                                      e.ctypes.data_as(_u32p), C.c_void_p(d_field), C.c_void_p(stream)))


def unpack_shards_device(grid, nranks, d_gathered, stride, d_field, stream=0):
    _check(load_library().dg_unpack_shards_device(C.byref(grid), nranks, C.c_void_p(d_gathered), stride,
                                                  C.c_void_p(d_field), C.c_void_p(stream)))


def unpack_shard_range_device(grid, nranks, d_gathered, stride, rank_begin, rank_end, d_field, stream=0):
    _check(load_library().dg_unpack_shard_range_device(C.byref(grid), nranks, C.c_void_p(d_gathered), stride,
                                                       rank_begin, rank_end, C.c_void_p(d_field), C.c_void_p(stream)))


def reduce_field(grid, coeffs, lo, hi, offset=0.0, closed=False):
    """dg_reduce_field: returns (coeffs, cells[rows, 32], cell_map, tied_keys); the arrays are None when tied."""
    lib = load_library()
    c = _f64(coeffs)
    h = C.c_void_p()
    _check(lib.dg_reduce_field(C.byref(grid), c.ctypes.data_as(_dp), len(c), int(closed), lo, hi, offset, C.byref(h)))
    try:
        return _fetch_reduction(lib, h, n_cells(grid))
    finally:
        lib.dg_reduction_destroy(h)


def _fetch_reduction(lib, h, ncells):
    m, rows, tied = C.c_uint64(), C.c_uint64(), C.c_int()
    _check(lib.dg_reduction_info(h, C.byref(m), C.byref(rows), C.byref(tied)))
    if tied.value:
        return None, None, None, True
    out = np.empty(m.value)
    cells = np.empty((rows.value, 32), dtype=np.uint32)
    cmap = np.empty(ncells, dtype=np.uint32)
    _check(lib.dg_reduction_fetch(h, out.ctypes.data_as(_dp), cells.ctypes.data_as(_u32p), cmap.ctypes.data_as(_u32p)))
    return out, cells, cmap, False


class Comm:
    """dg_comm handle: an RCCL communicator of one-process-per-GPU ranks, created inside the library."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(load_library().dg_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, rank, nranks):
        self._lib = load_library()
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(self._lib.dg_comm_create(buf, rank, nranks, C.byref(h)))
        self.handle = h
        self.rank, self.nranks = rank, nranks

    @classmethod
    def external(cls, rank, nranks, allgather, barrier):
        """dg_comm_create_external: a communicator whose two small host-side collectives the caller supplies (RCCL-free; runs
        EXCHANGE_INPLACE | EXCHANGE_COPY only).  allgather(mine: bytes) -> list of every rank's bytes; barrier() -> None."""
        self = cls.__new__(cls)
        self._lib = load_library()

        def _ag(mine, out, nbytes, _user):
            try:
                parts = allgather(C.string_at(mine, nbytes))
                assert len(parts) == nranks and all(len(b) == nbytes for b in parts)
                C.memmove(out, b"".join(parts), nbytes * nranks)
                return 0
            except Exception:  # noqa: BLE001 (reported through the status code)
                return 1

        def _bar(_user):
            try:
                barrier()
                return 0
            except Exception:  # noqa: BLE001
                return 1

        self._callbacks = (ALLGATHER_FN(_ag), BARRIER_FN(_bar))   # kept alive with the handle
        h = C.c_void_p()
        _check(self._lib.dg_comm_create_external(rank, nranks, C.cast(self._callbacks[0], C.c_void_p), C.cast(self._callbacks[1], C.c_void_p),
                                                 None, C.byref(h)))
        self.handle = h
        self.rank, self.nranks = rank, nranks
        return self

    @classmethod
    def shared_memory(cls, name, rank, nranks):
        """dg_comm_create_shm: a communicator whose control plane lives in a POSIX shared-memory segment (one node; RCCL-free;
        runs EXCHANGE_INPLACE | EXCHANGE_COPY only).  Collective; `name` must be unique to the job."""
        self = cls.__new__(cls)
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.dg_comm_create_shm(name.encode(), rank, nranks, C.byref(h)))
        self.handle = h
        self.rank, self.nranks = rank, nranks
        return self

    def info(self):
        i = CommInfo()
        _check(self._lib.dg_comm_get_info(self.handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in CommInfo._fields_}

    def last_exchange_wait_ms(self):
        ms = C.c_float(0.0)
        _check(self._lib.dg_comm_last_exchange_wait_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.dg_comm_destroy(self.handle)
            self.handle = None

    __del__ = close

    def sample_exchange_device(self, mesh, grid, d_field, pieces=4, flags=EXCHANGE_INPLACE, root=0, plane_cost=None, invert=False,
                               stream=0):
        arr, keep = _plane_cost_arg(plane_cost)
        _check(self._lib.dg_sdf_sample_exchange_device(mesh.handle, C.byref(grid), int(invert), self.handle, pieces, flags, root,
                                                       arr, C.c_void_p(d_field), C.c_void_p(stream)))

    def last_chunk_ms(self, pieces=64):
        ms = (C.c_float * pieces)()
        n = C.c_int(pieces)
        _check(self._lib.dg_comm_last_chunk_ms(self.handle, ms, C.byref(n)))
        return [float(ms[i]) for i in range(n.value)]

    def sample_allgather_device(self, mesh, grid, d_field, pieces=4, invert=False, stream=0):
        _check(self._lib.dg_sdf_sample_allgather_device(mesh.handle, C.byref(grid), int(invert), self.handle, pieces,
                                                        C.c_void_p(d_field), C.c_void_p(stream)))

    def field_alloc(self, n_doubles):
        """dg_comm_field_alloc: a device array of hipMemCreate chunks that the peers can map whatever its size; returns a
        DeviceArray (pointer + __cuda_array_interface__, so torch.as_tensor(a, device="cuda") views it)."""
        p = C.c_void_p()
        _check(self._lib.dg_comm_field_alloc(self.handle, n_doubles, C.byref(p)))
        return DeviceArray(p.value, n_doubles)

    def field_free(self, array):
        _check(self._lib.dg_comm_field_free(self.handle, C.c_void_p(array.ptr)))


class DeviceArray:
    """n float64 values of device memory the library allocated (not owned by this object)."""

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None}

    def data_ptr(self):
        return self.ptr


class HostFieldInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("nranks", C.c_int32), ("device", C.c_int32), ("registered", C.c_int32), ("n_doubles", C.c_uint64)]


class HostField:
    """dg_host_field: a coefficient vector in POSIX shared memory that all ranks of one node map (the exchange form without
    collective kernels and without device IPC).  `data` is a numpy view of the shared vector."""

    def __init__(self, name, n_doubles, rank, nranks):
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.dg_host_field_open(name.encode(), n_doubles, rank, nranks, C.byref(h)))
        self.handle = h
        self.rank, self.nranks = rank, nranks
        ptr = self._lib.dg_host_field_data(h)
        self.data = np.ctypeslib.as_array(C.cast(ptr, _dp), shape=(n_doubles,))

    def info(self):
        i = HostFieldInfo()
        _check(self._lib.dg_host_field_get_info(self.handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in HostFieldInfo._fields_}

    def barrier(self):
        _check(self._lib.dg_host_field_barrier(self.handle))

    def sample(self, mesh, grid, d_field, pieces=4, plane_cost=None, invert=False, stream=0):
        arr, keep = _plane_cost_arg(plane_cost)
        _check(self._lib.dg_sdf_sample_to_host_field(mesh.handle, C.byref(grid), int(invert), self.handle, pieces, arr,
                                                     C.c_void_p(d_field), C.c_void_p(stream)))

    def last_chunk_ms(self, pieces=64):
        ms = (C.c_float * pieces)()
        n = C.c_int(pieces)
        _check(self._lib.dg_host_field_last_chunk_ms(self.handle, ms, C.byref(n)))
        return [float(ms[i]) for i in range(n.value)]

    def close(self):
        if getattr(self, "handle", None):
            self.data = None
            self._lib.dg_host_field_close(self.handle)
            self.handle = None

    __del__ = close


class Field:
    """dg_field handle: one coefficient vector (+ optional cell table / cell map)."""

    def __init__(self, grid, coeffs=None, cells=None, cell_map=None, d_coeffs=None, n_coeffs=None, d_cells=None,
                 n_cell_rows=0, d_cell_map=None):
        self._lib = load_library()
        h = C.c_void_p()
        if d_coeffs is not None:  # non-owning attach of device arrays
            _check(self._lib.dg_field_attach_device(C.byref(grid), C.c_void_p(d_coeffs), n_coeffs,
                                                    C.c_void_p(d_cells or 0), n_cell_rows,
                                                    C.c_void_p(d_cell_map or 0), C.byref(h)))
        else:
            c = _f64(coeffs)
            ce = None if cells is None else np.ascontiguousarray(cells, dtype=np.uint32)
            cm = None if cell_map is None else np.ascontiguousarray(cell_map, dtype=np.uint32)
            _check(self._lib.dg_field_create(C.byref(grid), c.ctypes.data_as(_dp), len(c),
                                             None if ce is None else ce.ctypes.data_as(_u32p),
                                             0 if ce is None else len(ce.reshape(-1, 32)),
                                             None if cm is None else cm.ctypes.data_as(_u32p), C.byref(h)))
        self.handle = h

    @classmethod
    def _adopt(cls, handle, host_out=None):
        f = cls.__new__(cls)
        f._lib = load_library()
        f.handle = handle
        f._host_out = host_out  # keeps the array alive while the copy runs
        return f

    def close(self):
        if getattr(self, "handle", None):
            self._lib.dg_field_destroy(self.handle)
            self.handle = None

    __del__ = close

    def info(self):
        i = FieldInfo()
        _check(self._lib.dg_field_get_info(self.handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in FieldInfo._fields_}

    def has_cell_major(self):
        return bool(self.info()["has_cell_major"])

    def set_immutable(self, immutable=True):
        _check(self._lib.dg_field_set_immutable(self.handle, int(immutable)))

    def host_wait(self):
        """dg_field_host_wait: blocks until the host array of the producing call is complete and returns it."""
        _check(self._lib.dg_field_host_wait(self.handle))
        return getattr(self, "_host_out", None)

    def density_map_field(self, support_radius, rho0, band_predicate=True, mask=None, host_out=None):
        """dg_density_map_field: K3 over this SDF's whole lattice into a new device-resident Field."""
        n = int(self.info()["n_coeffs"])
        if host_out is True:
            host_out = np.empty(n, dtype=np.float64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        h = C.c_void_p()
        _check(self._lib.dg_density_map_field(self.handle, support_radius, rho0, int(band_predicate),
                                              None if m is None else m.ctypes.data_as(_u8p),
                                              None if host_out is None else host_out.ctypes.data_as(_dp), C.byref(h)))
        return Field._adopt(h, host_out)

    def reduce(self, lo, hi, offset=0.0, closed=False, as_field=False):
        """dg_reduce_field_device on this handle's coefficients: (coeffs, cells, cell_map, tied) like reduce_field();
        with as_field a fifth element, the reduced field as a device handle (dg_reduction_to_field)."""
        h = C.c_void_p()
        _check(self._lib.dg_reduce_field_device(self.handle, int(closed), lo, hi, offset, C.byref(h)))
        try:
            i = self.info()
            out = _fetch_reduction(self._lib, h, int(i["n_cell_rows"]))
            if as_field and not out[3]:
                fh = C.c_void_p()
                _check(self._lib.dg_reduction_to_field(h, C.byref(fh)))
                return out + (Field._adopt(fh),)
            return out + ((None,) if as_field else ())
        finally:
            self._lib.dg_reduction_destroy(h)

    def build_cell_major(self, stream=0):
        _check(self._lib.dg_field_build_cell_major(self.handle, C.c_void_p(stream)))

    def drop_cell_major(self):
        _check(self._lib.dg_field_drop_cell_major(self.handle))

    def build_cell_major_band(self, lo, hi, stream=0):
        """dg_field_build_cell_major_band: 256-byte rows for the cells whose coefficients reach into [lo, hi]; returns the row count."""
        rows = C.c_uint64(0)
        _check(self._lib.dg_field_build_cell_major_band(self.handle, float(lo), float(hi), C.c_void_p(stream), C.byref(rows)))
        return int(rows.value)

    def build_tile_major(self, stream=0):
        _check(self._lib.dg_field_build_tile_major(self.handle, C.c_void_p(stream)))

    def drop_tile_major(self):
        _check(self._lib.dg_field_drop_tile_major(self.handle))

    def interpolate(self, points, grad=False):
        P = _f64(points).reshape(-1, 3)
        phi = np.empty(len(P))
        g = np.empty((len(P), 3)) if grad else None
        _check(self._lib.dg_interpolate_batch(self.handle, P.ctypes.data_as(_dp), len(P), phi.ctypes.data_as(_dp),
                                              None if g is None else g.ctypes.data_as(_dp)))
        return (phi, g) if grad else phi

    def density_map_nodes(self, n_nodes_total, support_radius, rho0, band_predicate=True, begin=0, end=None,
                          mask=None):
        """K3: the GenerateDensityMap node function on this SDF field's own lattice."""
        if end is None:
            end = n_nodes_total
        out = np.empty(end - begin)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _check(self._lib.dg_density_map_nodes(self.handle, support_radius, rho0, int(band_predicate), begin, end,
                                              None if m is None else m.ctypes.data_as(_u8p), out.ctypes.data_as(_dp)))
        return out

    def density_map_nodes_device(self, support_radius, rho0, band_predicate, begin, end, d_out, d_mask=0, stream=0):
        _check(self._lib.dg_density_map_nodes_device(self.handle, support_radius, rho0, int(band_predicate), begin, end,
                                                     C.c_void_p(d_mask), C.c_void_p(d_out), C.c_void_p(stream)))

    def interpolate_device(self, d_xyz, n, d_phi, d_grad=0, stream=0):
        _check(self._lib.dg_interpolate_batch_device(self.handle, C.c_void_p(d_xyz), n, C.c_void_p(d_phi),
                                                     C.c_void_p(d_grad), C.c_void_p(stream)))


def last_kernel_ms():
    return float(load_library().dg_last_kernel_ms())
