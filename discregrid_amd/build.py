"""Build recipe for libdiscregrid_hip.so (HIP kernels for gfx950 + the C ABI).

    python -m discregrid_amd.build          # or: from discregrid_amd.build import build; build()

hipcc cross-compiles for gfx950 without a GPU.  The shared library is written IN-TREE
(discregrid_amd/libdiscregrid_hip.so) so that it travels to the GPU box with the repo
snapshot.  -ffp-contract=off everywhere: the 1e-10 parity bar is FMA-sensitive.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdiscregrid_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

COMMON = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function"]
SOURCES_HIP = ["dg_kernels_k1.hip", "dg_kernels_k2.hip", "dg_kernels_k3.hip", "dg_kernels_aux.hip"]
SOURCES_CXX = ["dg_capi.cpp", "dg_capi_field.cpp", "dg_capi_host.cpp", "dg_host_query.cpp", "dg_capi_comm.cpp", "dg_capi_hostfield.cpp", "dg_build.cpp"]
HEADERS = ["dg_geom.h", "dg_lattice.h", "dg_density.h", "dg_density_cells.h", "dg_build.h", "dg_kernels.h", "dg_layout.h", "dg_gauss16.h", "dg_capi_internal.h", "dg_capi_vmm.h", "dg_capi_shm.h", "dg_host_query.h", "dg_traverse.h", "dg_device.h", os.path.join("..", "..", "include", "discregrid_hip.h")]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES_HIP + SOURCES_CXX + HEADERS + ["exports.map"]] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out=None):
    """Compile every HIP/C++ source for gfx950 and link the shared library.  `defines`/`out`
    build an experiment variant (e.g. defines=("-DDG_TRI_BOX=0",), out=".../libdg_x.so")."""
    if out is None and not force and not _stale():
        return OUT
    if out is not None:
        return _build(verbose, tuple(defines), os.path.join(HERE, "build", os.path.basename(out) + ".d"), out, force)
    return _build(verbose, tuple(defines), os.path.join(HERE, "build"), OUT, force)


# headers each source includes (directly or not): an object is rebuilt only when one of these is newer
_KERNEL_HEADERS = ["dg_geom.h", "dg_lattice.h", "dg_density.h", "dg_density_cells.h", "dg_kernels.h", "dg_layout.h", "dg_gauss16.h", "dg_device.h"]
DEPS = {
    "dg_kernels_k1.hip": _KERNEL_HEADERS + ["dg_traverse.h"],
    "dg_kernels_k2.hip": _KERNEL_HEADERS,
    "dg_kernels_k3.hip": _KERNEL_HEADERS,
    "dg_kernels_aux.hip": _KERNEL_HEADERS,
    "dg_build.cpp": ["dg_build.h", "dg_geom.h", "dg_kernels.h", "dg_lattice.h", "dg_density.h", "dg_density_cells.h"],
    "dg_host_query.cpp": ["dg_host_query.h", "dg_traverse.h", "dg_build.h", "dg_geom.h", "dg_kernels.h", "dg_lattice.h", "dg_density.h", "dg_density_cells.h",
                          "dg_capi_internal.h", "dg_layout.h", os.path.join("..", "..", "include", "discregrid_hip.h")],
    "dg_capi_hostfield.cpp": ["dg_capi_internal.h", "dg_capi_shm.h", "dg_build.h", "dg_geom.h", "dg_kernels.h", "dg_layout.h", os.path.join("..", "..", "include", "discregrid_hip.h")],
}


def _object_stale(obj, src, defines, force):
    if force or not os.path.exists(obj):
        return True
    stamp = obj + ".flags"
    flags = " ".join(COMMON + list(defines))
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src), os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in DEPS.get(src, HEADERS)]
    return any(os.path.getmtime(d) > t for d in deps)


def _build(verbose, defines, objdir, target, force=False):
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES_HIP + SOURCES_CXX:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if not _object_stale(obj, src, defines, force):
            continue
        if src in SOURCES_HIP:
            cmd = [HIPCC, "--offload-arch=gfx950", *COMMON, *defines, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        else:
            cmd = [HIPCC, "-x", "c++", "-D__HIP_PLATFORM_AMD__", *COMMON, *defines, "-I" + os.path.join(ROCM, "include"), "-c",
                   os.path.join(CSRC, src), "-o", obj]
        jobs.append((obj, cmd))

    def compile_one(job):
        obj, cmd = job
        if verbose:
            print(" ".join(cmd))
        out = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or out.returncode != 0:
            sys.stderr.write(out.stdout + out.stderr)
        if out.returncode != 0:
            raise subprocess.CalledProcessError(out.returncode, cmd)
        with open(obj + ".flags", "w") as fh:
            fh.write(" ".join(COMMON + list(defines)))

    # the translation units are independent: compile them side by side (the kernel files dominate: K1 / K2 / K3 / aux)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        list(pool.map(compile_one, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", target]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
