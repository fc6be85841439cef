// dg_capi_host.cpp -- the host-pointer entry points of include/discregrid_hip.h.  The caller's arrays
// are ordinary pageable memory; every call is cut into chunks that run through one pipeline: host
// threads stage the inputs in pinned memory, the compute stream uploads them and runs the device entry
// point, the copy stream brings the outputs back, host threads move them into the caller's arrays --
// while the GPU is already busy with the next chunk.
#include "dg_capi_internal.h"

extern "C"
{

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
