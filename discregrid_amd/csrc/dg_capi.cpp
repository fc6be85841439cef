// dg_capi.cpp -- the C ABI of include/discregrid_hip.h: handles, host-side preparation
// (BVH/pseudonormal construction, lattice decomposition into bricks, shard bookkeeping) and
// kernel launches.  No CPU compute path exists here: without a gfx950 device every compute
// entry point fails with DG_ERR_NO_DEVICE.
#include "../../include/discregrid_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <cmath>

#include "dg_build.h"
#include "dg_kernels.h"
#include "dg_layout.h"

// Stream-ordered scratch buffers kept with a handle: a buffer is handed out again once the work that
// used it has finished (or to the same stream, where work is ordered anyway).
struct ScratchPool
{
	struct Buf
	{
		void* mem = nullptr;
		size_t bytes = 0;
		hipEvent_t done = nullptr;
		hipStream_t stream = nullptr;
		bool busy = false;
	};
	std::mutex mutex;
	std::vector<Buf> bufs;

	int acquire(size_t bytes, hipStream_t stream, void** mem)
	{
		std::lock_guard<std::mutex> lock(mutex);
		int idx = -1;
		for (size_t i = 0; i < bufs.size() && idx < 0; ++i)
			if (!bufs[i].busy && bufs[i].bytes >= bytes && (bufs[i].stream == stream || hipEventQuery(bufs[i].done) == hipSuccess))
				idx = (int)i;
		if (idx < 0)
		{
			Buf b;
			b.bytes = bytes;
			if (hipMalloc(&b.mem, bytes) != hipSuccess || hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess)
			{
				(void)hipGetLastError();
				if (b.mem) (void)hipFree(b.mem);
				return -1;
			}
			bufs.push_back(b);
			idx = (int)bufs.size() - 1;
		}
		bufs[(size_t)idx].busy = true;
		bufs[(size_t)idx].stream = stream;
		*mem = bufs[(size_t)idx].mem;
		return idx;
	}
	void release(int idx, hipStream_t stream)
	{
		if (idx < 0)
			return;
		std::lock_guard<std::mutex> lock(mutex);
		(void)hipEventRecord(bufs[(size_t)idx].done, stream);
		bufs[(size_t)idx].busy = false;
	}
	void destroy()
	{
		for (Buf& b : bufs)
		{
			if (b.done) (void)hipEventDestroy(b.done);
			if (b.mem) (void)hipFree(b.mem);
		}
		bufs.clear();
	}
};

// Scratch of one K1 launch for its heavy bricks (dg_kernels.h: OverflowBuf).  Buffers are kept with
// the mesh and handed out again once the launch that used them has finished (or to the same stream,
// where launches are ordered anyway), so steady-state launches allocate nothing.
struct HeavyScratch
{
	void* mem = nullptr;
	hipEvent_t done = nullptr;
	hipStream_t stream = nullptr;
	uint32_t slots = 0;      // capacity the buffer was laid out for
	uint32_t used_slots = 0; // slots the most recent launch was given
	bool busy = false; // between acquire and the event record
	uint64_t serial = 0; // order of use
};

struct dg_mesh
{
	dg::MeshDev dev;
	void* d_pairs = nullptr;
	void* d_tri_pairs = nullptr;
	void* d_tris = nullptr;
	void* d_pn = nullptr;
	int device = -1;
	dg_mesh_info info;
	mutable std::mutex scratch_mutex;
	mutable std::vector<HeavyScratch> scratch;
	mutable uint64_t scratch_serial = 0;
	mutable uint64_t unsplit_serial = 0; // serial of the last launch that ran without the split path
	mutable ScratchPool bin_scratch;     // K1p point binning
	double bbox_lo[3], bbox_hi[3];       // of the vertices
};

struct dg_field
{
	dg::FieldDev dev;
	mutable ScratchPool scratch; // K2 query binning
	void* owned[3] = {nullptr, nullptr, nullptr};
	void* d_cell_major = nullptr;
	void* d_wtab = nullptr;    // K3: 4096 kernel values for support radius wtab_h
	void* d_unsafe = nullptr;  // K3: flag written by k_field_check
	double wtab_h = -1.0;
	dg_grid_desc grid;
	uint64_t n_coeffs = 0;
	uint64_t n_rows = 0; // rows of the cell table (= grid cells for an unreduced field)
	int device = -1;
};

namespace
{

thread_local std::string g_error;
thread_local double g_last_ms = -1.0;

dg_status fail(dg_status s, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_error = buf;
	return s;
}

#define DG_HIP(call)                                                                                         \
	do                                                                                                       \
	{                                                                                                        \
		hipError_t e_ = (call);                                                                              \
		if (e_ != hipSuccess)                                                                                \
			return fail(DG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

dg_status require_device()
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
	{
		(void)hipGetLastError();
		return fail(DG_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
					e == hipSuccess ? "count = 0" : hipGetErrorString(e));
	}
	return DG_OK;
}

bool valid_grid(const dg_grid_desc* g)
{
	if (!g)
		return false;
	for (int d = 0; d < 3; ++d)
		if (g->resolution[d] == 0)
			return false;
	// the reference indexes nodes with unsigned int (cubic_lagrange_discrete_grid.cpp:796-809)
	dg::ClassGeom cg[4];
	return dg::class_geometry(g->resolution, cg) < (1ull << 32);
}

} // namespace

// ---- helpers of the host-pointer entry points --------------------------------------------------------------
// Device allocations and the timing events of ONE call: released when the call returns, whichever
// way it returns.  The first failing HIP call is remembered in `err`; later steps become no-ops.
struct HostCall
{
	std::vector<void*> allocations;
	hipEvent_t begin = nullptr, end = nullptr;
	hipError_t err = hipSuccess;

	~HostCall()
	{
		for (void* p : allocations)
			(void)hipFree(p);
		if (begin) (void)hipEventDestroy(begin);
		if (end) (void)hipEventDestroy(end);
	}
	template <class T>
	T* device(uint64_t count, bool wanted = true)
	{
		if (!wanted || err != hipSuccess)
			return nullptr;
		void* p = nullptr;
		err = hipMalloc(&p, std::max<size_t>(count * sizeof(T), 1));
		if (err != hipSuccess)
			return nullptr;
		allocations.push_back(p);
		return static_cast<T*>(p);
	}
	void upload(void* dst, const void* src, size_t bytes)
	{
		if (err == hipSuccess && dst)
			err = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
	}
	void download(void* dst, const void* src, size_t bytes)
	{
		if (err == hipSuccess && dst)
			err = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
	}
	void start_timer()
	{
		if (err == hipSuccess) err = hipEventCreate(&begin);
		if (err == hipSuccess) err = hipEventCreate(&end);
		if (err == hipSuccess) err = hipEventRecord(begin, nullptr);
	}
	void stop_timer()
	{
		if (err == hipSuccess) err = hipEventRecord(end, nullptr);
	}
	void publish_time() // after the downloads (they synchronise with the null stream)
	{
		float ms = -1.f;
		if (err == hipSuccess && begin && end && hipEventElapsedTime(&ms, begin, end) == hipSuccess)
			g_last_ms = ms;
	}
	dg_status status(const char* what) const
	{
		if (err == hipSuccess)
			return DG_OK;
		return fail(err == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
	}
};

extern "C"
{

const char* dg_version(void) { return "discregrid_hip 0.1 (gfx950)"; }
const char* dg_last_error(void) { return g_error.c_str(); }
double dg_last_kernel_ms(void) { return g_last_ms; }

dg_status dg_device_count(int* count)
{
	if (!count)
		return fail(DG_ERR_INVALID, "count is null");
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
	{
		(void)hipGetLastError();
		n = 0;
	}
	*count = n;
	return DG_OK;
}

dg_status dg_set_device(int device)
{
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	DG_HIP(hipSetDevice(device));
	return DG_OK;
}

dg_status dg_current_device(int* device)
{
	if (!device)
		return fail(DG_ERR_INVALID, "null argument");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	DG_HIP(hipGetDevice(device));
	return DG_OK;
}

// XCD chunk size of the K1 launches (dg_kernels.h: logical_block()); tuning knob, default kXcdChunk.
// DG_XCD_CHUNK=-1 gives every XCD one contiguous eighth of the launch.
static uint32_t env_xcd_chunk()
{
	if (const char* e = std::getenv("DG_XCD_CHUNK"))
	{
		const long v = std::atol(e);
		if (v < 0)
			return 0xffffffffu;
		if (v > 0)
			return (uint32_t)std::min(v, 1L << 24);
	}
	return 0;
}

dg_status dg_grid_desc_init(const double domain_min[3], const double domain_max[3], const uint32_t resolution[3],
							dg_grid_desc* out)
{
	if (!domain_min || !domain_max || !resolution || !out)
		return fail(DG_ERR_INVALID, "null argument");
	std::memset(out, 0, sizeof(*out));
	for (int d = 0; d < 3; ++d)
	{
		if (resolution[d] == 0)
			return fail(DG_ERR_INVALID, "resolution[%d] == 0", d);
		out->domain_min[d] = domain_min[d];
		out->domain_max[d] = domain_max[d];
		out->resolution[d] = resolution[d];
		// discrete_grid.hpp:26-27: cell_size = diagonal ./ n ; inv_cell_size = 1 ./ cell_size
		out->cell_size[d] = (domain_max[d] - domain_min[d]) / (double)resolution[d];
		out->inv_cell_size[d] = 1.0 / out->cell_size[d];
	}
	return DG_OK;
}

dg_status dg_default_domain(const double* verts, uint64_t n_vertices, double out_min_max[6])
{
	if (!verts || !out_min_max || n_vertices == 0)
		return fail(DG_ERR_INVALID, "dg_default_domain: no vertices");
	double* lo = out_min_max;
	double* hi = out_min_max + 3;
	for (int d = 0; d < 3; ++d)
		lo[d] = hi[d] = verts[d];
	for (uint64_t v = 1; v < n_vertices; ++v)
		for (int d = 0; d < 3; ++d)
		{
			const double x = verts[3 * v + d];
			lo[d] = x < lo[d] ? x : lo[d];
			hi[d] = x > hi[d] ? x : hi[d];
		}
	// |diagonal| with Eigen's association for fixed 3-vectors: x^2 + (y^2 + z^2)
	for (int pass = 0; pass < 2; ++pass)
	{
		const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
		const double grow = 1.0e-3 * std::sqrt(ex * ex + (ey * ey + ez * ez));
		for (int d = 0; d < 3; ++d)
		{
			if (pass == 0)
				hi[d] += grow;
			else
				lo[d] -= grow;
		}
	}
	return DG_OK;
}

uint64_t dg_grid_n_nodes(const dg_grid_desc* grid)
{
	if (!grid)
		return 0;
	dg::ClassGeom cg[4];
	return dg::class_geometry(grid->resolution, cg);
}

uint64_t dg_grid_n_cells(const dg_grid_desc* grid)
{
	return grid ? (uint64_t)grid->resolution[0] * grid->resolution[1] * grid->resolution[2] : 0;
}

// ---- mesh ------------------------------------------------------------------------------------------
dg_status dg_mesh_create(const double* verts, size_t n_vertices, const uint32_t* tris, size_t n_triangles,
						 dg_mesh** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!verts || !tris || n_vertices == 0 || n_triangles == 0)
		return fail(DG_ERR_INVALID, "empty triangle list"); // reference: message + exit(-1), TriangleMeshDistance.h:338-341
	dg_status s = require_device();
	if (s != DG_OK)
		return s;

	auto t0 = std::chrono::high_resolution_clock::now();
	dg::MeshBuild B;
	int max_leaf = 6; // measured on MI355X (profiles/r01_k1_ab.txt): 4..8 within 2 %, smaller leaves cost dependent node steps, fatter ones exact tests
	if (const char* e = std::getenv("DG_MAX_LEAF")) // tuning knob (1..16)
		max_leaf = std::max(1, std::min(dg::kMaxLeaf, std::atoi(e)));
	if (!dg::build_mesh(verts, n_vertices, tris, n_triangles, max_leaf, B))
		return fail(DG_ERR_INVALID, "invalid mesh (vertex index out of range or too many triangles)");
	auto t1 = std::chrono::high_resolution_clock::now();

	dg_mesh* m = new (std::nothrow) dg_mesh;
	if (!m)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	std::memset(&m->info, 0, sizeof(m->info));
	const size_t nb = B.pairs.size() * sizeof(dg::PairRec);
	const size_t sb = B.tri_pairs.size() * sizeof(dg::PairRec);
	const size_t tb = B.tris.size() * sizeof(dg::TriPacket);
	const size_t pb = B.pn.size() * sizeof(double);
	hipError_t e = hipGetDevice(&m->device);
	if (e == hipSuccess) e = hipMalloc(&m->d_pairs, std::max<size_t>(nb, 128));
	if (e == hipSuccess) e = hipMalloc(&m->d_tri_pairs, sb);
	if (e == hipSuccess) e = hipMalloc(&m->d_tris, tb);
	if (e == hipSuccess) e = hipMalloc(&m->d_pn, pb);
	if (e == hipSuccess && nb) e = hipMemcpy(m->d_pairs, B.pairs.data(), nb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(m->d_tri_pairs, B.tri_pairs.data(), sb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(m->d_tris, B.tris.data(), tb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(m->d_pn, B.pn.data(), pb, hipMemcpyHostToDevice);
	if (e != hipSuccess)
	{
		dg_mesh_destroy(m);
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "mesh upload failed: %s",
					hipGetErrorString(e));
	}
	m->dev.pairs = static_cast<const dg::PairRec*>(m->d_pairs);
	m->dev.tri_pairs = static_cast<const dg::PairRec*>(m->d_tri_pairs);
	m->dev.tris = static_cast<const dg::TriPacket*>(m->d_tris);
	m->dev.pn = static_cast<const double*>(m->d_pn);
	m->dev.root_info = B.root_info;
	m->dev.n_positions = (int32_t)B.tris.size();
	m->dev.stack_levels = (int32_t)std::min<uint32_t>(B.depth + 1, dg::kStackDepth);
	m->dev.n_sub = (int32_t)B.sub_roots.size();
	for (size_t i = 0; i < (size_t)dg::kSubtrees; ++i)
		m->dev.sub_roots[i] = i < B.sub_roots.size() ? B.sub_roots[i] : B.root_info;
	for (int d = 0; d < 3; ++d)
		m->dev.origin[d] = B.origin[d];
	m->dev.mesh_l1 = B.mesh_l1;
	m->dev.pad_ = 0.0f;
	for (int d = 0; d < 3; ++d)
	{
		m->bbox_lo[d] = m->bbox_hi[d] = verts[d];
		for (uint64_t v = 1; v < n_vertices; ++v)
		{
			m->bbox_lo[d] = std::min(m->bbox_lo[d], verts[3 * v + d]);
			m->bbox_hi[d] = std::max(m->bbox_hi[d], verts[3 * v + d]);
		}
	}
	m->info.n_vertices = n_vertices;
	m->info.n_triangles = n_triangles;
	m->info.n_bvh_nodes = 2 * B.pairs.size() + 1;
	m->info.bvh_depth = B.depth;
	m->info.not_watertight = B.not_watertight;
	m->info.device_bytes = nb + tb + pb + sb;
	m->info.build_seconds = std::chrono::duration<double>(t1 - t0).count();
	*out = m;
	return DG_OK;
}

dg_status dg_mesh_get_info(const dg_mesh* mesh, dg_mesh_info* info)
{
	if (!mesh || !info)
		return fail(DG_ERR_INVALID, "null argument");
	*info = mesh->info;
	return DG_OK;
}

void dg_mesh_destroy(dg_mesh* m)
{
	if (!m)
		return;
	if (m->d_pairs) (void)hipFree(m->d_pairs);
	if (m->d_tri_pairs) (void)hipFree(m->d_tri_pairs);
	if (m->d_tris) (void)hipFree(m->d_tris);
	if (m->d_pn) (void)hipFree(m->d_pn);
	for (HeavyScratch& h : m->scratch)
	{
		if (h.done) (void)hipEventDestroy(h.done);
		if (h.mem) (void)hipFree(h.mem);
	}
	m->bin_scratch.destroy();
	delete m;
}

// ---- K1 ----------------------------------------------------------------------------------------------
static int env_int(const char* name, int fallback, int lo, int hi)
{
	if (const char* e = std::getenv(name))
		return std::max(lo, std::min(hi, std::atoi(e)));
	return fallback;
}

// Attaches heavy-brick scratch to a K1 launch (tuning knobs DG_HEAVY_SLOTS, 0 = no splitting, and
// DG_HEAVY_WORK).  Returns the index of the scratch buffer in use, or -1 when the launch runs
// without splitting (tiny tree, knob, or no memory -- splitting only shortens the launch).
static int acquire_heavy_scratch(const dg_mesh* mesh, dg::SampleParams& P, hipStream_t stream)
{
	std::memset(&P.ovf, 0, sizeof(P.ovf));
	const uint32_t slots = (uint32_t)env_int("DG_HEAVY_SLOTS", (int)dg::overflow_slots_for(P.total_bricks), 0, dg::kOverflowSlots);
	if (slots == 0 || mesh->dev.n_sub < 2)
	{
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		mesh->unsplit_serial = ++mesh->scratch_serial;
		return -1;
	}
	int idx = -1;
	{
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		for (size_t i = 0; i < mesh->scratch.size() && idx < 0; ++i)
		{
			HeavyScratch& h = mesh->scratch[i];
			if (!h.busy && h.slots >= slots && (h.stream == stream || hipEventQuery(h.done) == hipSuccess))
				idx = (int)i;
		}
		if (idx < 0)
		{
			HeavyScratch h;
			h.slots = slots;
			size_t unused[6];
			if (hipMalloc(&h.mem, dg::overflow_bytes(slots, unused)) != hipSuccess || hipEventCreateWithFlags(&h.done, hipEventDisableTiming) != hipSuccess)
			{
				(void)hipGetLastError();
				if (h.mem) (void)hipFree(h.mem);
				return -1;
			}
			mesh->scratch.push_back(h);
			idx = (int)mesh->scratch.size() - 1;
		}
		mesh->scratch[(size_t)idx].busy = true;
		mesh->scratch[(size_t)idx].stream = stream;
		mesh->scratch[(size_t)idx].serial = ++mesh->scratch_serial;
		mesh->scratch[(size_t)idx].used_slots = slots;
	}
	char* base = static_cast<char*>(mesh->scratch[(size_t)idx].mem);
	size_t off[6];
	dg::overflow_bytes(mesh->scratch[(size_t)idx].slots, off); // the layout the buffer was allocated with
	P.ovf.count = reinterpret_cast<uint32_t*>(base + off[0]);
	P.ovf.brick = reinterpret_cast<uint32_t*>(base + off[1]);
	P.ovf.saved_d2 = reinterpret_cast<double*>(base + off[2]);
	P.ovf.saved_tri = reinterpret_cast<int32_t*>(base + off[3]);
	P.ovf.cand_d2 = reinterpret_cast<double*>(base + off[4]);
	P.ovf.cand_tri = reinterpret_cast<int32_t*>(base + off[5]);
	P.ovf.slots = slots;
	P.ovf.heavy_work = env_int("DG_HEAVY_WORK", dg::heavy_work_for(mesh->dev.n_positions), 1, 1 << 30);
	if (hipMemsetAsync(P.ovf.count, 0, sizeof(uint32_t), stream) != hipSuccess)
	{
		(void)hipGetLastError();
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		mesh->scratch[(size_t)idx].busy = false;
		std::memset(&P.ovf, 0, sizeof(P.ovf));
		return -1;
	}
	return idx;
}
static void release_heavy_scratch(const dg_mesh* mesh, int idx, hipStream_t stream)
{
	if (idx < 0)
		return;
	std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
	HeavyScratch& h = mesh->scratch[(size_t)idx];
	(void)hipEventRecord(h.done, stream);
	h.busy = false;
}
dg_status dg_mesh_last_heavy_bricks(const dg_mesh* mesh, uint32_t* heavy, uint32_t* split)
{
	if (!mesh || !heavy || !split)
		return fail(DG_ERR_INVALID, "null argument");
	*heavy = *split = 0;
	void* mem = nullptr;
	hipEvent_t done = nullptr;
	uint32_t slots = 0;
	{
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		uint64_t newest = mesh->unsplit_serial;
		for (const HeavyScratch& h : mesh->scratch)
			if (!h.busy && h.serial > newest)
			{
				newest = h.serial;
				mem = h.mem;
				done = h.done;
				slots = h.used_slots;
			}
	}
	if (!mem)
		return DG_OK;
	DG_HIP(hipEventSynchronize(done));
	uint32_t count = 0;
	DG_HIP(hipMemcpy(&count, mem, sizeof(count), hipMemcpyDeviceToHost)); // the counter is the first word
	*heavy = count;
	*split = std::min(count, slots);
	return DG_OK;
}

static hipError_t launch_k1(const dg_mesh* mesh, dg::SampleParams& P, hipStream_t stream)
{
	const int scratch = acquire_heavy_scratch(mesh, P, stream);
	const hipError_t e = dg::launch_sample_nodes(P, stream);
	release_heavy_scratch(mesh, scratch, stream);
	return e;
}

dg_status dg_sdf_sample_nodes_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, uint64_t node_begin,
									 uint64_t node_end, const uint8_t* d_pred_mask, double* d_out, void* stream)
{
	if (!mesh || !grid || !d_out)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid (zero resolution or >= 2^32 nodes)");
	const uint64_t total = dg_grid_n_nodes(grid);
	if (node_begin > node_end || node_end > total)
		return fail(DG_ERR_INVALID, "node range [%llu, %llu) outside [0, %llu)", (unsigned long long)node_begin,
					(unsigned long long)node_end, (unsigned long long)total);
	if (node_begin == node_end)
		return DG_OK;

	dg::SampleParams P;
	dg::init_params(P, mesh->dev, grid->domain_min, grid->cell_size, invert);
	P.xcd_chunk = env_xcd_chunk();
	dg::layout_range(P, grid->resolution, node_begin, node_end);
	P.mask = d_pred_mask;
	P.out = d_out;
	DG_HIP(launch_k1(mesh, P, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

// ---- host-pointer K1: device buffers, kernel and the copy back to pageable memory, pipelined --------------
// The caller's array is ordinary pageable memory (a std::vector in the C++ API), which the runtime
// can only fill at ~13 GB/s in one blocking hipMemcpy -- three times the kernel time at 256^3.  The
// range is therefore cut into chunks of whole 4-plane slabs of one node class (= whole bricks, so no
// brick is traversed twice): while K1 samples chunk k into one of two device buffers, the copy
// stream moves chunk k-1 into pinned staging memory and host threads move chunk k-2 from there into
// the caller's array.  Staging memory is kept for the lifetime of the process.
namespace
{
struct HostPipe
{
	std::mutex mutex; // one host-pointer launch at a time uses the staging buffers
	int device = -1;
	size_t chunk_bytes = 0;
	void* d_buf[2] = {nullptr, nullptr};
	void* h_buf[2] = {nullptr, nullptr};
	hipStream_t compute = nullptr, copy = nullptr;
	hipEvent_t k_begin[2] = {nullptr, nullptr}, k_end[2] = {nullptr, nullptr}, c_end[2] = {nullptr, nullptr};

	void release()
	{
		for (int i = 0; i < 2; ++i)
		{
			if (d_buf[i]) (void)hipFree(d_buf[i]);
			if (h_buf[i]) (void)hipHostFree(h_buf[i]);
			if (k_begin[i]) (void)hipEventDestroy(k_begin[i]);
			if (k_end[i]) (void)hipEventDestroy(k_end[i]);
			if (c_end[i]) (void)hipEventDestroy(c_end[i]);
			d_buf[i] = h_buf[i] = nullptr;
			k_begin[i] = k_end[i] = c_end[i] = nullptr;
		}
		if (compute) (void)hipStreamDestroy(compute);
		if (copy) (void)hipStreamDestroy(copy);
		compute = copy = nullptr;
		chunk_bytes = 0;
		device = -1;
	}
	hipError_t prepare(size_t bytes)
	{
		int dev = 0;
		hipError_t e = hipGetDevice(&dev);
		if (e != hipSuccess)
			return e;
		if (dev == device && bytes <= chunk_bytes)
			return hipSuccess;
		release();
		device = dev;
		e = hipStreamCreateWithFlags(&compute, hipStreamNonBlocking);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
		for (int i = 0; i < 2 && e == hipSuccess; ++i)
		{
			e = hipMalloc(&d_buf[i], bytes);
			if (e == hipSuccess) e = hipHostMalloc(&h_buf[i], bytes, hipHostMallocDefault);
			if (e == hipSuccess) e = hipEventCreate(&k_begin[i]);
			if (e == hipSuccess) e = hipEventCreate(&k_end[i]);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&c_end[i], hipEventDisableTiming);
		}
		if (e == hipSuccess)
			chunk_bytes = bytes;
		else
			release();
		return e;
	}
};
const int kMaxPipes = 16;
HostPipe g_pipes[kMaxPipes]; // [0]: single-mesh calls; [i]: worker i of dg_sdf_sample_nodes_multi

// dst <- src with a few threads (one thread tops out near 10 GB/s, the PCIe link delivers 50+)
void parallel_copy(void* dst, const void* src, size_t bytes)
{
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, hw), std::max<size_t>(1, bytes >> 22));
	if (nt <= 1)
	{
		std::memcpy(dst, src, bytes);
		return;
	}
	std::vector<std::thread> th;
	const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
	for (unsigned t = 0; t < nt; ++t)
	{
		const size_t b = std::min(bytes, per * t), e = std::min(bytes, per * (t + 1));
		if (e > b)
			th.emplace_back([=]() { std::memcpy((char*)dst + b, (const char*)src + b, e - b); });
	}
	for (auto& t : th)
		t.join();
}

// [node_begin, node_end) cut at multiples of `slabs` 4-plane slabs of each node class
void chunk_cuts(const uint32_t res[3], uint64_t node_begin, uint64_t node_end, uint64_t target_nodes,
				std::vector<uint64_t>& cuts)
{
	dg::ClassGeom cg[4];
	dg::class_geometry(res, cg);
	cuts.assign(1, node_begin);
	for (int c = 0; c < 4; ++c)
	{
		const uint64_t slab = (uint64_t)dg::kSlabPlanes * cg[c].D[0] * cg[c].D[1];
		const uint64_t step = std::max<uint64_t>(1, target_nodes / slab) * slab;
		for (uint64_t at = cg[c].off; at < cg[c].off + cg[c].size; at += step)
			if (at > cuts.back() && at < node_end)
				cuts.push_back(at);
	}
	cuts.push_back(node_end);
}
} // namespace

// One array of a pipelined host-pointer call: read from the host (`in`) or written back to it (`out`),
// item_bytes per item; null in and out = absent (optional outputs).  Host pointers address item cuts[0].
struct PipeArray
{
	const void* in;
	void* out;
	size_t item_bytes;
};
// launch(begin, count, d_arrays, stream): enqueue the device work for items [begin, begin + count);
// d_arrays[i] is the device copy of array i for exactly those items (null if the array is absent)
typedef std::function<dg_status(uint64_t, uint64_t, void* const*, hipStream_t)> PipeLaunch;

// Chunks first, first + stride, ... of `cuts` through one pipeline (the caller holds pipe.mutex and has
// made the right device current).  Per chunk: host threads copy the inputs into pinned staging
// memory, the compute stream uploads them and runs `launch`, the copy stream brings the outputs back
// into pinned memory, host threads move them into the caller's arrays -- while the GPU is already
// busy with the next chunk.  kernel_ms accumulates upload + kernel time of the chunks.
static dg_status run_pipeline(HostPipe& pipe, const std::vector<uint64_t>& cuts, size_t first, size_t stride,
							  const std::vector<PipeArray>& arrays, const PipeLaunch& launch, const char* what, double* kernel_ms,
							  double* t_wait, double* t_copy)
{
	const size_t n_chunks = cuts.size() - 1;
	uint64_t longest = 0;
	for (size_t k = first; k < n_chunks; k += stride)
		longest = std::max(longest, cuts[k + 1] - cuts[k]);
	if (longest == 0)
		return DG_OK;
	std::vector<size_t> off(arrays.size() + 1, 0);
	for (size_t i = 0; i < arrays.size(); ++i)
	{
		const bool present = arrays[i].in != nullptr || arrays[i].out != nullptr;
		off[i + 1] = off[i] + (present ? ((longest * arrays[i].item_bytes + 255) & ~(size_t)255) : 0);
	}
	hipError_t e = pipe.prepare(off.back());
	dg_status st = DG_OK;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	auto drain = [&](size_t k, int b) -> hipError_t { // chunk k: wait for its copies, move the outputs into the caller's arrays
		const double t0 = now();
		hipError_t err = hipEventSynchronize(pipe.c_end[b]);
		if (err != hipSuccess)
			return err;
		const double t1 = now();
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, pipe.k_begin[b], pipe.k_end[b]) == hipSuccess)
			*kernel_ms += ms;
		for (size_t i = 0; i < arrays.size(); ++i)
			if (arrays[i].out)
				parallel_copy(static_cast<char*>(arrays[i].out) + (cuts[k] - cuts[0]) * arrays[i].item_bytes,
							  static_cast<char*>(pipe.h_buf[b]) + off[i], (cuts[k + 1] - cuts[k]) * arrays[i].item_bytes);
		*t_wait += t1 - t0;
		*t_copy += now() - t1;
		return hipSuccess;
	};
	std::vector<void*> d_arrays(arrays.size(), nullptr);
	size_t prev = n_chunks; // chunk whose results still sit in the other pair of buffers
	int turn = 0;
	for (size_t k = first; k < n_chunks && e == hipSuccess && st == DG_OK; k += stride, turn ^= 1)
	{
		const int b = turn; // buffers b were last used by the chunk before `prev`, which has been drained
		const uint64_t cn = cuts[k + 1] - cuts[k];
		const double t0 = now();
		for (size_t i = 0; i < arrays.size(); ++i)
		{
			const bool present = arrays[i].in != nullptr || arrays[i].out != nullptr;
			d_arrays[i] = present ? static_cast<char*>(pipe.d_buf[b]) + off[i] : nullptr;
			if (arrays[i].in)
				parallel_copy(static_cast<char*>(pipe.h_buf[b]) + off[i],
							  static_cast<const char*>(arrays[i].in) + (cuts[k] - cuts[0]) * arrays[i].item_bytes, cn * arrays[i].item_bytes);
		}
		*t_copy += now() - t0;
		e = hipEventRecord(pipe.k_begin[b], pipe.compute);
		for (size_t i = 0; i < arrays.size() && e == hipSuccess; ++i)
			if (arrays[i].in)
				e = hipMemcpyAsync(d_arrays[i], static_cast<char*>(pipe.h_buf[b]) + off[i], cn * arrays[i].item_bytes,
								   hipMemcpyHostToDevice, pipe.compute);
		if (e != hipSuccess)
			break;
		st = launch(cuts[k], cn, d_arrays.data(), pipe.compute);
		if (st != DG_OK)
			break;
		e = hipEventRecord(pipe.k_end[b], pipe.compute);
		if (e == hipSuccess) e = hipStreamWaitEvent(pipe.copy, pipe.k_end[b], 0);
		for (size_t i = 0; i < arrays.size() && e == hipSuccess; ++i)
			if (arrays[i].out)
				e = hipMemcpyAsync(static_cast<char*>(pipe.h_buf[b]) + off[i], d_arrays[i], cn * arrays[i].item_bytes,
								   hipMemcpyDeviceToHost, pipe.copy);
		if (e == hipSuccess) e = hipEventRecord(pipe.c_end[b], pipe.copy);
		if (e == hipSuccess && prev < n_chunks)
			e = drain(prev, b ^ 1);
		prev = k;
	}
	if (e == hipSuccess && st == DG_OK && prev < n_chunks)
		e = drain(prev, turn ^ 1);
	else
	{
		(void)hipStreamSynchronize(pipe.compute);
		(void)hipStreamSynchronize(pipe.copy);
	}
	if (st != DG_OK)
		return st;
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
	return DG_OK;
}

// K1 through the pipeline: node range cuts, optional predicate mask in, coefficients out
static dg_status run_k1_chunks(HostPipe& pipe, const dg_mesh* mesh, const dg_grid_desc* grid, int invert,
							   const std::vector<uint64_t>& cuts, size_t first, size_t stride, const uint8_t* pred_mask, double* out,
							   double* kernel_ms, double* t_wait, double* t_copy)
{
	const std::vector<PipeArray> arrays = {{pred_mask, nullptr, 1}, {nullptr, out, sizeof(double)}};
	const PipeLaunch launch = [&](uint64_t begin, uint64_t count, void* const* d, hipStream_t stream) {
		return dg_sdf_sample_nodes_device(mesh, grid, invert, begin, begin + count, static_cast<const uint8_t*>(d[0]),
										  static_cast<double*>(d[1]), stream);
	};
	return run_pipeline(pipe, cuts, first, stride, arrays, launch, "dg_sdf_sample_nodes", kernel_ms, t_wait, t_copy);
}

// items [0, n) in uniform chunks (K1p, K2): big enough to amortise the launches, small enough to overlap
static void uniform_cuts(uint64_t n, int default_chunk, std::vector<uint64_t>& cuts)
{
	const uint64_t chunk = (uint64_t)env_int("DG_HOST_CHUNK_ITEMS", default_chunk, 1 << 8, 1 << 28);
	cuts.clear();
	for (uint64_t at = 0; at < n; at += chunk)
		cuts.push_back(at);
	cuts.push_back(n);
}

static dg_status check_host_range(const dg_grid_desc* grid, uint64_t node_begin, uint64_t node_end)
{
	if (node_begin > node_end)
		return fail(DG_ERR_INVALID, "node_begin > node_end");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (node_end > dg_grid_n_nodes(grid))
		return fail(DG_ERR_INVALID, "node range [%llu, %llu) outside [0, %llu)", (unsigned long long)node_begin,
					(unsigned long long)node_end, (unsigned long long)dg_grid_n_nodes(grid));
	return DG_OK;
}

dg_status dg_sdf_sample_nodes(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, uint64_t node_begin,
							  uint64_t node_end, const uint8_t* pred_mask, double* out)
{
	if (!mesh || !grid || !out)
		return fail(DG_ERR_INVALID, "null argument");
	dg_status s = check_host_range(grid, node_begin, node_end);
	if (s != DG_OK)
		return s;
	const uint64_t n = node_end - node_begin;
	if (n == 0)
		return DG_OK;
	s = require_device();
	if (s != DG_OK)
		return s;
	// ~10 chunks per call (every chunk costs a kernel tail, ~0.4 ms), 32..256 MiB of results each
	const uint64_t auto_target = std::min<uint64_t>(std::max<uint64_t>(n / 10, 1u << 22), 1u << 25);
	const uint64_t target = (uint64_t)env_int("DG_HOST_CHUNK_NODES", (int)auto_target, 1 << 10, 1 << 28);
	std::vector<uint64_t> cuts;
	chunk_cuts(grid->resolution, node_begin, node_end, target, cuts);
	double kernel_ms = 0, t_wait = 0, t_copy = 0;
	std::lock_guard<std::mutex> lock(g_pipes[0].mutex);
	s = run_k1_chunks(g_pipes[0], mesh, grid, invert, cuts, 0, 1, pred_mask, out, &kernel_ms, &t_wait, &t_copy);
	if (s != DG_OK)
		return s;
	g_last_ms = kernel_ms;
	if (std::getenv("DG_HOST_DEBUG"))
		std::fprintf(stderr, "dg_sdf_sample_nodes: %zu chunks, kernels %.1f ms, host waited %.1f ms, host copies %.1f ms\n",
					 cuts.size() - 1, kernel_ms, t_wait * 1e3, t_copy * 1e3);
	return DG_OK;
}

dg_status dg_sdf_sample_nodes_multi(const dg_mesh* const* meshes, int n_meshes, const dg_grid_desc* grid, int invert,
									uint64_t node_begin, uint64_t node_end, const uint8_t* pred_mask, double* out)
{
	if (!meshes || !grid || !out || n_meshes < 1 || n_meshes > kMaxPipes)
		return fail(DG_ERR_INVALID, "null argument or mesh count outside 1..%d", kMaxPipes);
	for (int i = 0; i < n_meshes; ++i)
		if (!meshes[i])
			return fail(DG_ERR_INVALID, "meshes[%d] is null", i);
	dg_status s = check_host_range(grid, node_begin, node_end);
	if (s != DG_OK)
		return s;
	const uint64_t n = node_end - node_begin;
	if (n == 0)
		return DG_OK;
	s = require_device();
	if (s != DG_OK)
		return s;
	// chunks are dealt round-robin: thin interleaved pieces equalise the very uneven cost per node
	const uint64_t auto_target = std::min<uint64_t>(std::max<uint64_t>(n / (10ull * (uint64_t)n_meshes), 1u << 21), 1u << 25);
	const uint64_t target = (uint64_t)env_int("DG_HOST_CHUNK_NODES", (int)auto_target, 1 << 10, 1 << 28);
	std::vector<uint64_t> cuts;
	chunk_cuts(grid->resolution, node_begin, node_end, target, cuts);
	std::vector<dg_status> status((size_t)n_meshes, DG_OK);
	std::vector<std::string> message((size_t)n_meshes);
	std::vector<double> kernel_ms((size_t)n_meshes, 0.0);
	std::vector<std::thread> workers;
	int caller_device = 0;
	(void)hipGetDevice(&caller_device);
	for (int i = 0; i < n_meshes; ++i)
		workers.emplace_back([&, i]() {
			if (hipSetDevice(meshes[i]->device) != hipSuccess)
			{
				status[(size_t)i] = DG_ERR_HIP;
				message[(size_t)i] = "hipSetDevice failed";
				return;
			}
			double t_wait = 0, t_copy = 0;
			std::lock_guard<std::mutex> lock(g_pipes[i].mutex);
			status[(size_t)i] = run_k1_chunks(g_pipes[i], meshes[i], grid, invert, cuts, (size_t)i, (size_t)n_meshes, pred_mask, out,
											  &kernel_ms[(size_t)i], &t_wait, &t_copy);
			if (status[(size_t)i] != DG_OK)
				message[(size_t)i] = dg_last_error(); // thread-local: carry it to the caller
		});
	for (auto& w : workers)
		w.join();
	(void)hipSetDevice(caller_device);
	for (int i = 0; i < n_meshes; ++i)
		if (status[(size_t)i] != DG_OK)
			return fail(status[(size_t)i], "mesh %d (device %d): %s", i, meshes[i]->device, message[(size_t)i].c_str());
	g_last_ms = *std::max_element(kernel_ms.begin(), kernel_ms.end());
	return DG_OK;
}

dg_status dg_signed_distance_device(const dg_mesh* mesh, const double* d_xyz, uint64_t n, double* d_dist,
									int32_t* d_tri, int32_t* d_entity, double* d_nearest, void* stream)
{
	if (!mesh || (n && (!d_xyz || !d_dist)))
		return fail(DG_ERR_INVALID, "null argument");
	if (n == 0)
		return DG_OK;
	hipStream_t st = static_cast<hipStream_t>(stream);
	dg::SampleParams P;
	const double zero[3] = {0.0, 0.0, 0.0};
	dg::init_params(P, mesh->dev, zero, zero, 0);
	P.xcd_chunk = env_xcd_chunk();
	P.pts.xyz = d_xyz;
	P.pts.n = n;
	P.pts.dist = d_dist;
	P.pts.tri = d_tri;
	P.pts.entity = d_entity;
	P.pts.nearest = d_nearest;
	dg::layout_points(P, n);
	// Batches are binned first: a wave's traversal costs the union of what its 64 points need, so points
	// that arrive in arbitrary order are grouped into compact tiles (decided on the device; ordered
	// inputs run as they are).  The tile grid covers the mesh's bounding box grown by its own size;
	// points farther out are clamped into the border tiles.  DG_K1P_BINNING=0: off.
	dg::TileGrid tiles;
	dg::BinScratch S;
	int bin_idx = -1;
	if (n >= 4096 && n < 0xffffffffull && env_int("DG_K1P_BINNING", 1, 0, 1) != 0)
	{
		double lo[3], hi[3];
		for (int d = 0; d < 3; ++d)
		{
			const double ext = mesh->bbox_hi[d] - mesh->bbox_lo[d];
			lo[d] = mesh->bbox_lo[d] - 0.5 * ext;
			hi[d] = mesh->bbox_hi[d] + 0.5 * ext;
		}
		tiles = dg::point_tiles(lo, hi, n);
		size_t off[4];
		const size_t bytes = dg::bin_scratch_bytes(dg::tile_count(tiles), n, off);
		void* mem = nullptr;
		bin_idx = mesh->bin_scratch.acquire(bytes, st, &mem);
		if (bin_idx >= 0)
		{
			char* base = static_cast<char*>(mem);
			S.flag = reinterpret_cast<uint32_t*>(base + off[0]);
			S.start = reinterpret_cast<uint32_t*>(base + off[1]);
			S.cursor = reinterpret_cast<uint32_t*>(base + off[2]);
			S.perm = reinterpret_cast<uint32_t*>(base + off[3]);
			P.pts.bin_flag = S.flag;
			P.pts.perm = S.perm;
		}
	}
	const int heavy_idx = acquire_heavy_scratch(mesh, P, st);
	const hipError_t e = dg::launch_signed_distance(P, bin_idx >= 0 ? &tiles : nullptr, bin_idx >= 0 ? &S : nullptr, st);
	release_heavy_scratch(mesh, heavy_idx, st);
	mesh->bin_scratch.release(bin_idx, st);
	DG_HIP(e);
	return DG_OK;
}

dg_status dg_signed_distance(const dg_mesh* mesh, const double* xyz, uint64_t n, double* dist, int32_t* tri,
							 int32_t* entity, double* nearest)
{
	if (!mesh || (n && (!xyz || !dist)))
		return fail(DG_ERR_INVALID, "null argument");
	if (n == 0)
		return DG_OK;
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	std::vector<uint64_t> cuts;
	uniform_cuts(n, 1 << 23, cuts); // K1p launches end with a long tail (a few waves with costly points): few, big chunks
	const std::vector<PipeArray> arrays = {{xyz, nullptr, 3 * sizeof(double)},
										   {nullptr, dist, sizeof(double)},
										   {nullptr, tri, sizeof(int32_t)},
										   {nullptr, entity, sizeof(int32_t)},
										   {nullptr, nearest, 3 * sizeof(double)}};
	const PipeLaunch launch = [&](uint64_t, uint64_t count, void* const* d, hipStream_t stream) {
		return dg_signed_distance_device(mesh, static_cast<const double*>(d[0]), count, static_cast<double*>(d[1]),
										 static_cast<int32_t*>(d[2]), static_cast<int32_t*>(d[3]), static_cast<double*>(d[4]), stream);
	};
	double kernel_ms = 0, t_wait = 0, t_copy = 0;
	std::lock_guard<std::mutex> lock(g_pipes[0].mutex);
	s = run_pipeline(g_pipes[0], cuts, 0, 1, arrays, launch, "dg_signed_distance", &kernel_ms, &t_wait, &t_copy);
	if (s == DG_OK)
		g_last_ms = kernel_ms;
	return s;
}

// ---- sharding -------------------------------------------------------------------------------------------
dg_status dg_shard_layout(const dg_grid_desc* grid, int rank, int nranks, dg_shard_info* out)
{
	if (!grid || !out)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "rank %d / nranks %d out of range (max %d ranks)", rank, nranks, dg::kMaxRanks);
	uint64_t mx = 0;
	for (int r = 0; r < nranks; ++r)
	{
		const uint64_t cnt = dg::shard_count(grid->resolution, r, nranks);
		if (r == rank)
			out->count = cnt;
		mx = std::max(mx, cnt);
	}
	out->stride = (mx + 63) / 64 * 64;
	return DG_OK;
}

dg_status dg_sdf_sample_shard_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, int rank, int nranks,
									 double* d_packed, void* stream)
{
	if (!mesh || !grid || !d_packed)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "rank %d / nranks %d out of range", rank, nranks);
	dg::SampleParams P;
	dg::init_params(P, mesh->dev, grid->domain_min, grid->cell_size, invert);
	P.xcd_chunk = env_xcd_chunk();
	dg::layout_shard(P, grid->resolution, rank, nranks);
	P.mask = nullptr;
	P.out = d_packed;
	DG_HIP(launch_k1(mesh, P, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

dg_status dg_unpack_shards_device(const dg_grid_desc* grid, int nranks, const double* d_gathered, uint64_t stride,
								  double* d_field, void* stream)
{
	if (!grid || !d_gathered || !d_field)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (nranks < 1 || nranks > dg::kMaxRanks)
		return fail(DG_ERR_INVALID, "nranks %d out of range", nranks);
	dg::UnpackParams U;
	dg::layout_unpack(U, grid->resolution, nranks);
	for (int r = 0; r < nranks; ++r)
		if (dg::shard_count(grid->resolution, r, nranks) > stride)
			return fail(DG_ERR_INVALID, "stride %llu smaller than rank %d's shard", (unsigned long long)stride, r);
	U.nranks = nranks;
	U.stride = stride;
	U.gathered = d_gathered;
	U.field = d_field;
	DG_HIP(dg::launch_unpack(U, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

dg_status dg_unpack_shard_range_device(const dg_grid_desc* grid, int nranks, const double* d_gathered, uint64_t stride,
									   int rank_begin, int rank_end, double* d_field, void* stream)
{
	if (!grid || !d_gathered || !d_field)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (nranks < 1 || nranks > dg::kMaxRanks || rank_begin < 0 || rank_begin > rank_end || rank_end > nranks)
		return fail(DG_ERR_INVALID, "ranks [%d, %d) of %d out of range", rank_begin, rank_end, nranks);
	dg::UnpackParams U;
	dg::layout_unpack(U, grid->resolution, nranks);
	for (int r = rank_begin; r < rank_end; ++r)
		if (U.count[r] > stride)
			return fail(DG_ERR_INVALID, "stride %llu smaller than rank %d's shard", (unsigned long long)stride, r);
	U.stride = stride;
	U.gathered = d_gathered;
	U.field = d_field;
	U.rank_begin = rank_begin;
	U.rank_end = rank_end;
	DG_HIP(dg::launch_unpack_ranks(U, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

// ---- field + K2 ---------------------------------------------------------------------------------------------
static void fill_field(dg::FieldDev& F, const dg_grid_desc* g)
{
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = g->domain_min[d];
		F.dmax[d] = g->domain_max[d];
		F.cell[d] = g->cell_size[d];
		F.inv_cell[d] = g->inv_cell_size[d];
		F.res[d] = g->resolution[d];
	}
}

dg_status dg_field_attach_device(const dg_grid_desc* grid, const double* d_coeffs, uint64_t n_coeffs,
								 const uint32_t* d_cells, uint64_t n_cell_rows, const uint32_t* d_cell_map,
								 dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!grid || !d_coeffs)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if ((d_cells == nullptr) != (d_cell_map == nullptr))
		return fail(DG_ERR_INVALID, "cells and cell_map must be given together");
	if (!d_cells && n_coeffs != dg_grid_n_nodes(grid))
		return fail(DG_ERR_INVALID, "an unreduced field needs %llu coefficients, got %llu",
					(unsigned long long)dg_grid_n_nodes(grid), (unsigned long long)n_coeffs);
	(void)n_cell_rows;
	dg_field* f = new (std::nothrow) dg_field;
	if (!f)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	fill_field(f->dev, grid);
	f->dev.coeffs = d_coeffs;
	f->dev.cells = d_cells;
	f->dev.cell_map = d_cell_map;
	f->dev.cell_major = nullptr;
	f->grid = *grid;
	f->n_coeffs = n_coeffs;
	f->n_rows = d_cells ? n_cell_rows : dg_grid_n_cells(grid);
	(void)hipGetDevice(&f->device);
	*out = f;
	return DG_OK;
}

dg_status dg_field_create(const dg_grid_desc* grid, const double* coeffs, uint64_t n_coeffs, const uint32_t* cells,
						  uint64_t n_cell_rows, const uint32_t* cell_map, dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!grid || !coeffs)
		return fail(DG_ERR_INVALID, "null argument");
	if ((cells == nullptr) != (cell_map == nullptr))
		return fail(DG_ERR_INVALID, "cells and cell_map must be given together");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	void *d_c = nullptr, *d_cells = nullptr, *d_map = nullptr;
	const uint64_t ncell = dg_grid_n_cells(grid);
	hipError_t e = hipMalloc(&d_c, n_coeffs * sizeof(double));
	if (e == hipSuccess) e = hipMemcpy(d_c, coeffs, n_coeffs * sizeof(double), hipMemcpyHostToDevice);
	if (e == hipSuccess && cells)
	{
		e = hipMalloc(&d_cells, std::max<uint64_t>(n_cell_rows, 1) * 32 * sizeof(uint32_t));
		if (e == hipSuccess && n_cell_rows)
			e = hipMemcpy(d_cells, cells, n_cell_rows * 32 * sizeof(uint32_t), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMalloc(&d_map, ncell * sizeof(uint32_t));
		if (e == hipSuccess) e = hipMemcpy(d_map, cell_map, ncell * sizeof(uint32_t), hipMemcpyHostToDevice);
	}
	dg_status st = DG_OK;
	if (e == hipSuccess)
		st = dg_field_attach_device(grid, (const double*)d_c, n_coeffs, (const uint32_t*)d_cells, n_cell_rows,
									(const uint32_t*)d_map, out);
	if (e != hipSuccess || st != DG_OK)
	{
		if (d_c) (void)hipFree(d_c);
		if (d_cells) (void)hipFree(d_cells);
		if (d_map) (void)hipFree(d_map);
		if (st != DG_OK)
			return st;
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "dg_field_create: %s", hipGetErrorString(e));
	}
	(*out)->owned[0] = d_c;
	(*out)->owned[1] = d_cells;
	(*out)->owned[2] = d_map;
	return DG_OK;
}

void dg_field_destroy(dg_field* f)
{
	if (!f)
		return;
	for (void* p : f->owned)
		if (p)
			(void)hipFree(p);
	if (f->d_cell_major)
		(void)hipFree(f->d_cell_major);
	if (f->d_wtab)
		(void)hipFree(f->d_wtab);
	if (f->d_unsafe)
		(void)hipFree(f->d_unsafe);
	f->scratch.destroy();
	delete f;
}

dg_status dg_field_build_cell_major(dg_field* field, void* stream)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	if (field->d_cell_major)
		return DG_OK;
	if (field->n_rows == 0)
		return DG_OK;
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, field->n_rows * 32 * sizeof(double));
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "cell-major allocation of %llu bytes: %s",
					(unsigned long long)(field->n_rows * 256), hipGetErrorString(e));
	e = dg::launch_expand_cells(field->dev, field->n_rows, static_cast<double*>(p), static_cast<hipStream_t>(stream));
	if (e != hipSuccess)
	{
		(void)hipFree(p);
		return fail(DG_ERR_HIP, "k_expand_cells: %s", hipGetErrorString(e));
	}
	field->d_cell_major = p;
	field->dev.cell_major = static_cast<const double*>(p);
	return DG_OK;
}

dg_status dg_field_drop_cell_major(dg_field* field)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	if (field->d_cell_major)
	{
		DG_HIP(hipDeviceSynchronize());
		(void)hipFree(field->d_cell_major);
		field->d_cell_major = nullptr;
		field->dev.cell_major = nullptr;
	}
	return DG_OK;
}

// ---- K3 ---------------------------------------------------------------------------------------------------
dg_status dg_density_map_nodes_device(dg_field* sdf, double support_radius, double rho0, int band_predicate,
									  uint64_t node_begin, uint64_t node_end, const uint8_t* d_pred_mask,
									  double* d_out, void* stream)
{
	if (!sdf || !d_out)
		return fail(DG_ERR_INVALID, "null argument");
	if (!(support_radius > 0.0))
		return fail(DG_ERR_INVALID, "support radius must be positive");
	const uint64_t total = dg_grid_n_nodes(&sdf->grid);
	if (node_begin > node_end || node_end > total)
		return fail(DG_ERR_INVALID, "node range outside the lattice");
	if (node_begin == node_end)
		return DG_OK;
	hipStream_t st = static_cast<hipStream_t>(stream);
	dg::DensityParams P;
	std::vector<double> w;
	dg::init_density_params(P, support_radius, rho0, sdf->grid.cell_size, band_predicate, w);
	if (sdf->wtab_h != support_radius)
	{
		if (!sdf->d_wtab)
			DG_HIP(hipMalloc(&sdf->d_wtab, 4096 * sizeof(double)));
		DG_HIP(hipMemcpy(sdf->d_wtab, w.data(), 4096 * sizeof(double), hipMemcpyHostToDevice));
		sdf->wtab_h = support_radius;
	}
	P.wtab = static_cast<const double*>(sdf->d_wtab);
	// K1's lattice decomposition: one wave per 4x4x4 brick of nodes
	dg::SampleParams L;
	dg::MeshDev none;
	std::memset(&none, 0, sizeof(none));
	dg::init_params(L, none, sdf->grid.domain_min, sdf->grid.cell_size, 0);
	dg::layout_range(L, sdf->grid.resolution, node_begin, node_end);
	L.mask = d_pred_mask;
	L.out = d_out;
	P.wtab = static_cast<const double*>(sdf->d_wtab);
	// zero-weight quadrature points are skipped unless the field holds non-finite / huge values (checked
	// on the device before every launch: an attached device array may have changed); DG_K3_SKIP=0: never
	if (env_int("DG_K3_SKIP", 1, 0, 1) != 0 && support_radius >= 1.0e-12)
	{
		if (!sdf->d_unsafe)
			DG_HIP(hipMalloc(&sdf->d_unsafe, sizeof(uint32_t)));
		P.skip_mode = 2;
		P.unsafe = static_cast<const uint32_t*>(sdf->d_unsafe);
	}
	DG_HIP(dg::launch_density_bricks(L, sdf->dev, sdf->n_coeffs, P, st));
	return DG_OK;
}

dg_status dg_density_map_nodes(dg_field* sdf, double support_radius, double rho0, int band_predicate,
							   uint64_t node_begin, uint64_t node_end, const uint8_t* pred_mask, double* out)
{
	if (!sdf || !out)
		return fail(DG_ERR_INVALID, "null argument");
	if (node_begin > node_end)
		return fail(DG_ERR_INVALID, "node_begin > node_end");
	const uint64_t n = node_end - node_begin;
	if (n == 0)
		return DG_OK;
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	HostCall call;
	double* d_out = call.device<double>(n);
	uint8_t* d_mask = call.device<uint8_t>(n, pred_mask != nullptr);
	call.upload(d_mask, pred_mask, n);
	call.start_timer();
	if (call.err == hipSuccess)
	{
		s = dg_density_map_nodes_device(sdf, support_radius, rho0, band_predicate, node_begin, node_end, d_mask, d_out, nullptr);
		if (s != DG_OK)
			return s;
	}
	call.stop_timer();
	call.download(out, d_out, n * sizeof(double));
	call.publish_time();
	return call.status("dg_density_map_nodes");
}

dg_status dg_interpolate_batch_device(const dg_field* field, const double* d_xyz, uint64_t n, double* d_phi,
									  double* d_grad, void* stream)
{
	if (!field || (n && (!d_xyz || !d_phi)))
		return fail(DG_ERR_INVALID, "null argument");
	hipStream_t st = static_cast<hipStream_t>(stream);
	// Large batches against a field that does not fit the L2s go through the binned path (queries in
	// arbitrary order are then processed tile by tile; ordered inputs are detected on the device and
	// run as they are).  DG_K2_BINNING=0 switches it off, =2 forces it for any size.
	const int binning = env_int("DG_K2_BINNING", 1, 0, 2);
	const bool big = n >= (1u << 18) && field->n_coeffs * sizeof(double) >= (32u << 20);
	if (binning != 0 && (big || binning == 2) && n < 0xffffffffull)
	{
		size_t off[4];
		const size_t bytes = dg::bin_scratch_bytes(dg::tile_count(dg::field_tiles(field->dev)), n, off);
		void* mem = nullptr;
		const int idx = field->scratch.acquire(bytes, st, &mem);
		if (idx >= 0)
		{
			char* base = static_cast<char*>(mem);
			dg::BinScratch S;
			S.flag = reinterpret_cast<uint32_t*>(base + off[0]);
			S.start = reinterpret_cast<uint32_t*>(base + off[1]);
			S.cursor = reinterpret_cast<uint32_t*>(base + off[2]);
			S.perm = reinterpret_cast<uint32_t*>(base + off[3]);
			const hipError_t e = dg::launch_interpolate_binned(field->dev, d_xyz, n, d_phi, d_grad, S, st);
			field->scratch.release(idx, st);
			DG_HIP(e);
			return DG_OK;
		}
	}
	DG_HIP(dg::launch_interpolate(field->dev, d_xyz, n, d_phi, d_grad, st));
	return DG_OK;
}

dg_status dg_interpolate_batch(const dg_field* field, const double* xyz, uint64_t n, double* phi, double* grad)
{
	if (!field || (n && (!xyz || !phi)))
		return fail(DG_ERR_INVALID, "null argument");
	if (n == 0)
		return DG_OK;
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	std::vector<uint64_t> cuts;
	uniform_cuts(n, 1 << 20, cuts);
	const std::vector<PipeArray> arrays = {{xyz, nullptr, 3 * sizeof(double)}, {nullptr, phi, sizeof(double)}, {nullptr, grad, 3 * sizeof(double)}};
	const PipeLaunch launch = [&](uint64_t, uint64_t count, void* const* d, hipStream_t stream) {
		return dg_interpolate_batch_device(field, static_cast<const double*>(d[0]), count, static_cast<double*>(d[1]),
										   static_cast<double*>(d[2]), stream);
	};
	double kernel_ms = 0, t_wait = 0, t_copy = 0;
	std::lock_guard<std::mutex> lock(g_pipes[0].mutex);
	s = run_pipeline(g_pipes[0], cuts, 0, 1, arrays, launch, "dg_interpolate_batch", &kernel_ms, &t_wait, &t_copy);
	if (s == DG_OK)
		g_last_ms = kernel_ms;
	return s;
}

} // extern "C"
