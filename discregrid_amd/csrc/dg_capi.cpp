// dg_capi.cpp -- the C ABI of include/discregrid_hip.h: runtime and grid helpers, the mesh handle
// (host-side BVH / pseudonormal construction, upload), the K1 / K1p device entry points with their
// heavy-brick scratch, and the shard bookkeeping.  Fields (K2, K3) are in dg_capi_field.cpp, the
// host-pointer entry points in dg_capi_host.cpp.  No CPU compute path exists anywhere: without a
// gfx950 device every compute entry point fails with DG_ERR_NO_DEVICE.
#include "dg_capi_internal.h"
#include "dg_host_query.h"

#include <dlfcn.h>

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


// ---- ROCTx (dlopen'ed) ---------------------------------------------------------------------------------
namespace
{
struct Roctx
{
	int (*push)(const char*) = nullptr;
	int (*pop)() = nullptr;
	Roctx()
	{
		if (const char* e = std::getenv("DG_ROCTX"))
			if (std::atoi(e) == 0)
				return;
		// rocprofv3 (--marker-trace) listens to the ROCTx of the rocprofiler SDK; the older libroctx64 is what roctracer-based
		// tools see.  Same entry points: take the first that loads.
		void* lib = nullptr;
		for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
			if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr)
				break;
		if (!lib)
			return;
		push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
		pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
		if (!push || !pop)
			push = nullptr, pop = nullptr;
	}
};
Roctx& roctx()
{
	static Roctx r;
	return r;
}
} // namespace
TraceRange::TraceRange(const char* name)
{
	Roctx& r = roctx();
	if (r.push)
	{
		r.push(name);
		on = true;
	}
}
TraceRange::~TraceRange()
{
	if (on)
		roctx().pop();
}

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
// DG_FORCE=xcd_chunk=-1 gives every XCD one contiguous eighth of the launch.
extern "C++" uint32_t env_xcd_chunk()
{
	std::string forced;
	if (dg::force_lookup("xcd_chunk", forced))
	{
		const long v = std::atol(forced.c_str());
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
	// Without a HIP device (or under DG_FORCE_CPU=1) the handle is HOST-ONLY: BVH, pseudonormals and filter records are
	// built as always and kept in host memory, dg_signed_distance_point() -- the reference's per-point signed_distance,
	// TriangleMeshDistance.h:269-328 -- works, and every entry point that would launch a kernel returns
	// DG_ERR_NO_DEVICE (dg_mesh_device() tells).  What a caller does then is its decision: the C++ host API runs the
	// reference's OpenMP node loop over the point query (cpp/src/cubic_lagrange_discrete_grid.cpp), like the reference
	// does everywhere (cubic_lagrange_discrete_grid.cpp:806-831).
	bool host_only = env_int("DG_FORCE_CPU", 0, 0, 1) != 0;
	if (!host_only)
	{
		int n_dev = 0;
		if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
		{
			(void)hipGetLastError();
			host_only = true;
		}
	}

	auto t0 = std::chrono::high_resolution_clock::now();
	dg::MeshBuild B;
	int max_leaf = 6; // measured on MI355X (profiles/r01_k1_ab.txt): 4..8 within 2 %, smaller leaves cost dependent node steps, fatter ones exact tests
	max_leaf = force_int("max_leaf", max_leaf, 1, dg::kMaxLeaf); // (tuning knob)
	if (!dg::build_mesh(verts, n_vertices, tris, n_triangles, max_leaf, B))
		return fail(DG_ERR_INVALID, "invalid mesh (vertex index out of range or too many triangles)");
	auto t1 = std::chrono::high_resolution_clock::now();
	// the traversals (device and host) keep one postponed subtree per tree level on a stack of kStackDepth entries:
	// a deeper tree -- impossible with median splits below 2^31 triangles -- would lose subtrees silently
	if (B.depth + 1 > (uint32_t)dg::kStackDepth)
		return fail(DG_ERR_INVALID, "BVH depth %u exceeds the traversal stack (%d levels)", B.depth, dg::kStackDepth);

	dg_mesh* m = new (std::nothrow) dg_mesh;
	if (!m)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	std::memset(&m->info, 0, sizeof(m->info));
	const size_t nb = B.pairs.size() * sizeof(dg::PairRec);
	const size_t sb = B.tri_pairs.size() * sizeof(dg::PairRec);
	const size_t tb = B.tris.size() * sizeof(dg::TriPacket);
	const size_t pb = B.pn.size() * sizeof(double);
	const size_t ab = B.tri_approx.size() * sizeof(dg::TriApproxPair);
	hipError_t e = hipSuccess;
	if (host_only)
		m->device = -1;
	else
		e = hipGetDevice(&m->device);
	if (!host_only)
	{
	if (e == hipSuccess) e = hipMalloc(&m->d_pairs, std::max<size_t>(nb, 128));
	if (e == hipSuccess) e = hipMalloc(&m->d_tri_pairs, sb);
	if (e == hipSuccess) e = hipMalloc(&m->d_tris, tb);
	if (e == hipSuccess) e = hipMalloc(&m->d_pn, pb);
	if (e == hipSuccess) e = hipMalloc(&m->d_tri_approx, ab);
	if (e == hipSuccess) e = hipMemcpy(m->d_tri_approx, B.tri_approx.data(), ab, hipMemcpyHostToDevice);
	if (e == hipSuccess && nb) e = hipMemcpy(m->d_pairs, B.pairs.data(), nb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(m->d_tri_pairs, B.tri_pairs.data(), sb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(m->d_tris, B.tris.data(), tb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(m->d_pn, B.pn.data(), pb, hipMemcpyHostToDevice);
	}
	if (e != hipSuccess)
	{
		dg_mesh_destroy(m);
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "mesh upload failed: %s",
					hipGetErrorString(e));
	}
	m->dev.pairs = static_cast<const dg::PairRec*>(m->d_pairs);
	m->dev.tri_pairs = static_cast<const dg::PairRec*>(m->d_tri_pairs);
	m->dev.tris = static_cast<const dg::TriPacket*>(m->d_tris);
	m->dev.tri_approx = static_cast<const dg::TriApproxPair*>(m->d_tri_approx);
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
	m->info.device_bytes = host_only ? 0 : nb + tb + pb + sb + ab;
	m->info.build_seconds = std::chrono::duration<double>(t1 - t0).count();
	m->host = std::move(B);
	dg::host::mesh_view(m->host, m->host_view);
	*out = m;
	return DG_OK;
}

int dg_mesh_device(const dg_mesh* mesh) { return mesh ? mesh->device : -1; }

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
	if (m->device < 0) // host-only handle: nothing of it lives in the HIP runtime
	{
		delete m;
		return;
	}
	DeviceGuard guard(m->device);
	if (m->d_pairs) (void)hipFree(m->d_pairs);
	if (m->d_tri_pairs) (void)hipFree(m->d_tri_pairs);
	if (m->d_tri_approx) (void)hipFree(m->d_tri_approx);
	if (m->d_tris) (void)hipFree(m->d_tris);
	if (m->d_pn) (void)hipFree(m->d_pn);
	for (HeavyScratch& h : m->scratch)
	{
		if (h.done) (void)hipEventDestroy(h.done);
		if (h.mem) (void)hipFree(h.mem);
	}
	m->bin_scratch.destroy();
	if (m->bin_flag_host) (void)hipHostFree(m->bin_flag_host);
	delete m;
}

// ---- K1 ----------------------------------------------------------------------------------------------
extern "C++" int env_int(const char* name, int fallback, int lo, int hi)
{
	if (const char* e = std::getenv(name))
		return std::max(lo, std::min(hi, std::atoi(e)));
	return fallback;
}

// Attaches heavy-brick scratch to a K1 launch (DG_FORCE keys heavy_slots, 0 = no splitting, and
// heavy_work).  Returns the index of the scratch buffer in use, or -1 when the launch runs
// without splitting (tiny tree, knob, or no memory -- splitting only shortens the launch).
extern "C++" int acquire_bin_scratch(ScratchPool& pool, uint32_t** flag_host, const dg::TileGrid& tiles, uint64_t n, hipStream_t stream,
									 dg::BinScratch& S)
{
	std::memset(&S, 0, sizeof(S));
	if (*flag_host == nullptr)
	{
		void* p = nullptr;
		if (hipHostMalloc(&p, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
		{
			(void)hipGetLastError();
			return -1;
		}
		*flag_host = static_cast<uint32_t*>(p);
		**flag_host = 1u; // nothing known yet: assume the first batch is unordered
	}
	size_t off[6];
	const uint32_t n_tiles = dg::tile_count(tiles);
	const size_t bytes = dg::bin_scratch_bytes(n_tiles, n, off);
	void* mem = nullptr;
	const int idx = pool.acquire(bytes, stream, &mem);
	if (idx < 0)
		return -1;
	dg::bin_scratch_assign(S, mem, off, n_tiles, n);
	S.flag_host = *flag_host;
	S.sort_launched = *reinterpret_cast<volatile uint32_t*>(*flag_host) != 0u ? 1 : 0;
	return idx;
}

extern "C++" int acquire_heavy_scratch(const dg_mesh* mesh, dg::SampleParams& P, hipStream_t stream)
{
	std::memset(&P.ovf, 0, sizeof(P.ovf));
	P.ovf.pool_cap = (uint32_t)force_int("pool_cap", 0x7fffffff, 0, 0x7fffffff); // (the epilogue's pool: its LDS capacity unless a test lowers it)
	// Which K1 kernel: the filtered one (dg_kernels_k1.hip: k_sample_fast) or the exact one only.  By default: from dg::kFastMinTriangles triangles up, and for lattices only where a brick (3 cells) is not much
	// smaller than a triangle -- the filter pays through the exact tests it saves, and a brick smaller than the
	// triangles around it needs few (icosphere 100 820 triangles: 128^3 -27 %, 256^3 -9.5 %, 512^3 -2.3 %; brick /
	// mean triangle edge = 2.8, 1.4, 0.7; bunny 512^3, 0.6: +4 %; dg_kernels.h: kFastMinBrickRatio).  DG_FORCE=k1_fast=0 / 1 force the exact / the filtered kernel.
	int fast_default = mesh->info.n_triangles >= dg::kFastMinTriangles ? 1 : 0;
	if (fast_default && P.pts.xyz == nullptr && mesh->host.mean_edge > 0.0)
	{
		const double brick = 3.0 * std::cbrt(P.cell[0] * P.cell[1] * P.cell[2]);
		if (!(brick >= dg::kFastMinBrickRatio * mesh->host.mean_edge))
			fast_default = 0;
	}
	P.filtered = (force_int("k1_fast", fast_default, 0, 1) != 0 && mesh->dev.n_positions < (1 << 26)) ? 1 : 0;
	const uint32_t slots = (uint32_t)force_int("heavy_slots", (int)dg::overflow_slots_for(P.total_bricks), 0, dg::kOverflowSlots);
	if (slots == 0 || mesh->dev.n_sub < 2)
	{
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		mesh->unsplit_serial = ++mesh->scratch_serial;
		return -1;
	}
	int idx = -1;
	char* base = nullptr;
	uint32_t laid_out_slots = 0;
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
		base = static_cast<char*>(mesh->scratch[(size_t)idx].mem);
		laid_out_slots = mesh->scratch[(size_t)idx].slots;
	}
	size_t off[6];
	dg::overflow_bytes(laid_out_slots, off); // the layout the buffer was allocated with
	P.ovf.count = reinterpret_cast<uint32_t*>(base + off[0]);
	P.ovf.brick = reinterpret_cast<uint32_t*>(base + off[1]);
	P.ovf.saved_d2 = reinterpret_cast<double*>(base + off[2]);
	P.ovf.saved_tri = reinterpret_cast<int32_t*>(base + off[3]);
	P.ovf.cand_d2 = reinterpret_cast<double*>(base + off[4]);
	P.ovf.cand_tri = reinterpret_cast<int32_t*>(base + off[5]);
	P.ovf.slots = slots;
	P.ovf.heavy_work = force_int("heavy_work", dg::heavy_work_for(mesh->dev.n_positions), 1, 1 << 30);
	// (test hooks of the pooled epilogue: the two counters live behind the slot counter, in the padding of its 256 bytes)
	P.ovf.stats = force_int("pool_stats", 0, 0, 1) != 0 ? P.ovf.count + 1 : nullptr;
	P.ovf.pool_cap = (uint32_t)force_int("pool_cap", 0x7fffffff, 0, 0x7fffffff);
	if (hipMemsetAsync(P.ovf.count, 0, 4 * sizeof(uint32_t), stream) != hipSuccess)
	{
		(void)hipGetLastError();
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		mesh->scratch[(size_t)idx].busy = false;
		std::memset(&P.ovf, 0, sizeof(P.ovf));
		P.ovf.pool_cap = 0x7fffffffu;
		return -1;
	}
	return idx;
}
extern "C++" void release_heavy_scratch(const dg_mesh* mesh, int idx, hipStream_t stream)
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
	DG_ON_DEVICE_OF(mesh);
	DG_HIP(hipEventSynchronize(done));
	uint32_t count = 0;
	DG_HIP(hipMemcpy(&count, mem, sizeof(count), hipMemcpyDeviceToHost)); // the counter is the first word
	*heavy = count;
	*split = std::min(count, slots);
	return DG_OK;
}

dg_status dg_mesh_last_epilogue_stats(const dg_mesh* mesh, uint32_t* pooled, uint32_t* lane_by_lane)
{
	if (!mesh || !pooled || !lane_by_lane)
		return fail(DG_ERR_INVALID, "null argument");
	*pooled = *lane_by_lane = 0;
	void* mem = nullptr;
	hipEvent_t done = nullptr;
	{
		std::lock_guard<std::mutex> lock(mesh->scratch_mutex);
		uint64_t newest = mesh->unsplit_serial;
		for (const HeavyScratch& h : mesh->scratch)
			if (!h.busy && h.serial > newest)
			{
				newest = h.serial;
				mem = h.mem;
				done = h.done;
			}
	}
	if (!mem)
		return DG_OK;
	DG_ON_DEVICE_OF(mesh);
	DG_HIP(hipEventSynchronize(done));
	uint32_t words[4] = {0, 0, 0, 0};
	DG_HIP(hipMemcpy(words, mem, sizeof(words), hipMemcpyDeviceToHost)); // slot counter, then the two test counters
	*pooled = words[1];
	*lane_by_lane = words[2];
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
	TraceRange trace_range_("dg K1 sample_nodes");
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
	DG_ON_DEVICE_OF(mesh);

	dg::SampleParams P;
	dg::init_params(P, mesh->dev, grid->domain_min, grid->cell_size, invert);
	P.xcd_chunk = env_xcd_chunk();
	dg::layout_range(P, grid->resolution, node_begin, node_end);
	P.mask = d_pred_mask;
	P.out = d_out;
	DG_HIP(launch_k1(mesh, P, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

dg_status dg_signed_distance_device(const dg_mesh* mesh, const double* d_xyz, uint64_t n, double* d_dist,
									int32_t* d_tri, int32_t* d_entity, double* d_nearest, void* stream)
{
	TraceRange trace_range_("dg K1p signed_distance");
	if (!mesh || (n && (!d_xyz || !d_dist)))
		return fail(DG_ERR_INVALID, "null argument");
	if (n == 0)
		return DG_OK;
	DG_ON_DEVICE_OF(mesh);
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
	// points farther out are clamped into the border tiles.  DG_FORCE=k1p_binning=0: off.
	dg::TileGrid tiles;
	dg::BinScratch S;
	int bin_idx = -1;
	if (n >= 4096 && n < 0xffffffffull && force_int("k1p_binning", 1, 0, 1) != 0)
	{
		double lo[3], hi[3];
		for (int d = 0; d < 3; ++d)
		{
			const double ext = mesh->bbox_hi[d] - mesh->bbox_lo[d];
			lo[d] = mesh->bbox_lo[d] - 0.5 * ext;
			hi[d] = mesh->bbox_hi[d] + 0.5 * ext;
		}
		tiles = dg::point_tiles(lo, hi, n);
		bin_idx = acquire_bin_scratch(mesh->bin_scratch, &mesh->bin_flag_host, tiles, n, st, S);
		if (bin_idx >= 0)
		{
			P.pts.bin_flag = S.flag;
			P.pts.perm = S.sort_launched ? S.perm : nullptr;
		}
	}
	const int heavy_idx = acquire_heavy_scratch(mesh, P, st);
	const hipError_t e = dg::launch_signed_distance(P, bin_idx >= 0 ? &tiles : nullptr, bin_idx >= 0 ? &S : nullptr, st);
	release_heavy_scratch(mesh, heavy_idx, st);
	mesh->bin_scratch.release(bin_idx, st);
	DG_HIP(e);
	return DG_OK;
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
	TraceRange trace_range_("dg K1 sample_shard");
	if (!mesh || !grid || !d_packed)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "rank %d / nranks %d out of range", rank, nranks);
	DG_ON_DEVICE_OF(mesh);
	dg::SampleParams P;
	dg::init_params(P, mesh->dev, grid->domain_min, grid->cell_size, invert);
	P.xcd_chunk = env_xcd_chunk();
	dg::layout_shard(P, grid->resolution, rank, nranks);
	P.mask = nullptr;
	P.out = d_packed;
	DG_HIP(launch_k1(mesh, P, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

dg_status dg_chunk_layout(const dg_grid_desc* grid, int nchunks, const float* const plane_cost[4], uint32_t* cuts)
{
	if (!grid || !cuts)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (nchunks < 1 || nchunks > dg::kMaxRanks)
		return fail(DG_ERR_INVALID, "nchunks %d out of range (max %d)", nchunks, dg::kMaxRanks);
	uint32_t c4[4][dg::kMaxRanks + 1];
	dg::chunk_planes(grid->resolution, nchunks, plane_cost, c4);
	for (int c = 0; c < 4; ++c)
		for (int v = 0; v <= nchunks; ++v)
			cuts[c * (nchunks + 1) + v] = c4[c][v];
	return DG_OK;
}

dg_status dg_sdf_sample_planes_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, const uint32_t plane_begin[4],
									  const uint32_t plane_end[4], double* d_field, void* stream)
{
	TraceRange trace_range_("dg K1 sample_planes");
	if (!mesh || !grid || !plane_begin || !plane_end || !d_field)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	DG_ON_DEVICE_OF(mesh);
	dg::SampleParams P;
	dg::init_params(P, mesh->dev, grid->domain_min, grid->cell_size, invert);
	P.xcd_chunk = env_xcd_chunk();
	dg::layout_class_planes(P, grid->resolution, plane_begin, plane_end);
	if (P.total_bricks == 0)
		return DG_OK;
	P.mask = nullptr;
	P.out = d_field;
	DG_HIP(launch_k1(mesh, P, static_cast<hipStream_t>(stream)));
	return DG_OK;
}

dg_status dg_unpack_shards_device(const dg_grid_desc* grid, int nranks, const double* d_gathered, uint64_t stride,
								  double* d_field, void* stream)
{
	TraceRange trace_range_("dg U unpack_shards");
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
	TraceRange trace_range_("dg U unpack_shard_range");
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

} // extern "C"
