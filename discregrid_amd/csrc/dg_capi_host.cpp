// dg_capi_host.cpp -- the host-pointer entry points of include/discregrid_hip.h.  The caller's arrays
// are ordinary pageable memory; every call is cut into chunks that run through a pipeline.  Two forms:
//   * direct (K1 results into memory nobody has touched yet, e.g. the freshly allocated coefficient
//     vector of addFunction): the pages are faulted in as huge pages by a few host threads and pinned
//     (hipHostRegister) while the first chunk is already being sampled, and the copy engine then writes
//     every chunk straight into the caller's array while the next one is computed;
//   * staged (everything else): host threads stage the inputs in pinned memory, the compute stream
//     uploads them and runs the device entry point, the copy stream brings the outputs back into pinned
//     memory, host threads move them into the caller's arrays -- while the GPU is busy with the next chunk.
#include "dg_capi_internal.h"

#include <sys/mman.h>
#include <unistd.h>


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
	std::mutex mutex; // one host-pointer call at a time uses a pipe (see lease_pipe())
	int device = -1;
	size_t chunk_bytes = 0, staging_bytes = 0;
	void* d_buf[2] = {nullptr, nullptr};
	void* h_buf[2] = {nullptr, nullptr}; // pinned staging memory: allocated only when the staged form runs
	hipStream_t compute = nullptr, copy = nullptr;
	hipEvent_t k_begin[2] = {nullptr, nullptr}, k_end[2] = {nullptr, nullptr}, c_end[2] = {nullptr, nullptr};
	std::vector<hipEvent_t> ev; // direct form: three events per chunk (kernel begin, kernel end, copy end)

	void release_staging()
	{
		for (int i = 0; i < 2; ++i)
		{
			if (h_buf[i]) (void)hipHostFree(h_buf[i]);
			h_buf[i] = nullptr;
		}
		staging_bytes = 0;
	}
	void release()
	{
		release_staging();
		for (int i = 0; i < 2; ++i)
		{
			if (d_buf[i]) (void)hipFree(d_buf[i]);
			if (k_begin[i]) (void)hipEventDestroy(k_begin[i]);
			if (k_end[i]) (void)hipEventDestroy(k_end[i]);
			if (c_end[i]) (void)hipEventDestroy(c_end[i]);
			d_buf[i] = nullptr;
			k_begin[i] = k_end[i] = c_end[i] = nullptr;
		}
		for (hipEvent_t e : ev)
			(void)hipEventDestroy(e);
		ev.clear();
		if (compute) (void)hipStreamDestroy(compute);
		if (copy) (void)hipStreamDestroy(copy);
		compute = copy = nullptr;
		chunk_bytes = 0;
		device = -1;
	}
	// device buffers, streams and events for chunks of `bytes`; pinned staging buffers only if `staging`
	hipError_t prepare(size_t bytes, bool staging)
	{
		int dev = 0;
		hipError_t e = hipGetDevice(&dev);
		if (e != hipSuccess)
			return e;
		if (dev != device || bytes > chunk_bytes)
		{
			release();
			device = dev;
			e = hipStreamCreateWithFlags(&compute, hipStreamNonBlocking);
			if (e == hipSuccess) e = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
			for (int i = 0; i < 2 && e == hipSuccess; ++i)
			{
				e = hipMalloc(&d_buf[i], bytes);
				if (e == hipSuccess) e = hipEventCreate(&k_begin[i]);
				if (e == hipSuccess) e = hipEventCreate(&k_end[i]);
				if (e == hipSuccess) e = hipEventCreateWithFlags(&c_end[i], hipEventDisableTiming);
			}
			if (e == hipSuccess)
				chunk_bytes = bytes;
		}
		if (e == hipSuccess && staging && bytes > staging_bytes)
		{
			release_staging();
			for (int i = 0; i < 2 && e == hipSuccess; ++i)
				e = hipHostMalloc(&h_buf[i], chunk_bytes, hipHostMallocDefault);
			if (e == hipSuccess)
				staging_bytes = chunk_bytes;
		}
		if (e != hipSuccess)
			release();
		return e;
	}
	hipError_t events(size_t n)
	{
		while (ev.size() < n)
		{
			hipEvent_t e = nullptr;
			const hipError_t err = hipEventCreate(&e);
			if (err != hipSuccess)
				return err;
			ev.push_back(e);
		}
		return hipSuccess;
	}
};
const int kMaxPipes = 16;
HostPipe g_pipes[kMaxPipes];

// dg_set_progress_callback: per calling thread
thread_local dg_progress_fn t_progress = nullptr;
thread_local void* t_progress_user = nullptr;
struct Progress
{
	dg_progress_fn cb;
	void* user;
	uint64_t total;
	std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
	void report(uint64_t done)
	{
		if (!cb)
			return;
		const auto now = std::chrono::steady_clock::now();
		if (done >= total || now - last > std::chrono::milliseconds(1000))
		{
			last = now;
			cb(done, total, user);
		}
	}
};

// A pipe for the calling thread: an idle one already set up for `device` if there is one, else any idle
// one; concurrent callers therefore run side by side (up to kMaxPipes of them), and only when every
// pipe is busy does a caller wait.
struct PipeLease
{
	HostPipe* pipe = nullptr;
	std::unique_lock<std::mutex> lock;
};
PipeLease lease_pipe(int device)
{
	PipeLease L;
	for (int pass = 0; pass < 3; ++pass)
		for (int i = 0; i < kMaxPipes; ++i)
		{
			std::unique_lock<std::mutex> lk(g_pipes[i].mutex, std::try_to_lock);
			if (!lk.owns_lock())
				continue;
			const int d = g_pipes[i].device;
			if ((pass == 0 && d != device) || (pass == 1 && d != -1))
				continue;
			L.pipe = &g_pipes[i];
			L.lock = std::move(lk);
			return L;
		}
	const size_t slot = std::hash<std::thread::id>()(std::this_thread::get_id()) % kMaxPipes;
	L.pipe = &g_pipes[slot];
	L.lock = std::unique_lock<std::mutex>(g_pipes[slot].mutex);
	return L;
}

// ---- the caller's output array as a DMA target ----------------------------------------------------------------
// Pinning 0.95 GB that already sits in 4 KiB pages costs 40+ ms, more than the staged pipeline loses;
// pinning memory that was faulted in as 2 MiB pages costs 3 ms, and faulting it in with 16 threads 5 ms
// [MI355X host, tests/perf/probes/host_mem_probe.cpp].  So the direct form is taken when the array is
// (a) already pinned by the caller, (b) still untouched -- then this code decides how it is faulted in --
// or (c) an array this code has faulted in itself on an earlier call.
struct PreparedRanges
{
	std::mutex mutex;
	std::vector<std::pair<uintptr_t, size_t>> ranges;
	bool contains(const void* p, size_t bytes)
	{
		std::lock_guard<std::mutex> lock(mutex);
		for (auto const& r : ranges)
			if ((uintptr_t)p >= r.first && (uintptr_t)p + bytes <= r.first + r.second)
				return true;
		return false;
	}
	void forget(const void* p, size_t bytes)
	{
		std::lock_guard<std::mutex> lock(mutex);
		for (size_t i = 0; i < ranges.size();)
			if (ranges[i].first < (uintptr_t)p + bytes && (uintptr_t)p < ranges[i].first + ranges[i].second)
				ranges.erase(ranges.begin() + (long)i);
			else
				++i;
	}
	void add(const void* p, size_t bytes)
	{
		std::lock_guard<std::mutex> lock(mutex);
		if (ranges.size() >= 64)
			ranges.erase(ranges.begin());
		ranges.emplace_back((uintptr_t)p, bytes);
	}
} g_prepared;

// fraction of (sampled) pages of [p, p + bytes) that are resident
double resident_fraction(const void* p, size_t bytes)
{
	const size_t page = (size_t)sysconf(_SC_PAGESIZE);
	const uintptr_t lo = (uintptr_t)p & ~(uintptr_t)(page - 1);
	const size_t pages = ((uintptr_t)p + bytes - lo + page - 1) / page;
	const size_t samples = std::min<size_t>(pages, 128);
	size_t resident = 0, seen = 0;
	for (size_t i = 0; i < samples; ++i)
	{
		unsigned char v = 0;
		const uintptr_t at = lo + (pages * i / samples) * page;
		if (mincore(reinterpret_cast<void*>(at), page, &v) == 0)
		{
			++seen;
			resident += v & 1u;
		}
	}
	return seen ? (double)resident / (double)seen : 1.0;
}

struct HostTarget // RAII: the registrations end with the call
{
	char* out = nullptr;
	size_t bytes = 0;
	bool caller_pinned = false; // the caller's array already is a DMA target
	bool fresh = false;         // untouched so far: this code faults it in (2 MiB pages, several threads)
	size_t prepared = 0;        // bytes made a DMA target so far
	std::vector<void*> registered;
	void release()
	{
		for (void* p : registered)
			(void)hipHostUnregister(p);
		registered.clear();
	}
	~HostTarget() { release(); }
};
// May [out, out + bytes) -- an array this call overwrites completely -- become a DMA target?  Returns false if
// the staged form should run instead (array already resident in pages of unknown size, or switched off with
// DG_FORCE=host_direct=0; =2 takes the direct form whatever the state of the pages).  Nothing is touched or pinned yet:
// prepare_host_piece() does that range by range, so that the first copies run while the rest is prepared.
bool begin_host_target(void* out, size_t bytes, HostTarget& T)
{
	const int mode = force_int("host_direct", 1, 0, 2);
	if (mode == 0 || (mode != 2 && bytes < (1u << 22)))
		return false;
	T.out = static_cast<char*>(out);
	T.bytes = bytes;
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, out) == hipSuccess)
	{
		if (attr.type == hipMemoryTypeHost)
		{
			T.caller_pinned = true; // pinned or registered by the caller
			return true;
		}
		if (attr.type != hipMemoryTypeUnregistered)
			return false;
	}
	else
		(void)hipGetLastError(); // ordinary memory the runtime has never seen
	T.fresh = resident_fraction(out, bytes) < 0.1;
	if (mode != 2 && !T.fresh && !g_prepared.contains(out, bytes))
		return false;
	if (T.fresh)
	{
		// huge pages for the 2 MiB-aligned interior (every byte of the range is overwritten by this call)
		const uintptr_t a = ((uintptr_t)out + (1u << 21) - 1) & ~(uintptr_t)((1u << 21) - 1);
		const uintptr_t b = ((uintptr_t)out + bytes) & ~(uintptr_t)((1u << 21) - 1);
		if (b > a)
			(void)madvise(reinterpret_cast<void*>(a), b - a, MADV_HUGEPAGE);
	}
	return true;
}
// Bytes [lo, hi) of the array become a DMA target: first touch from several threads if the array is fresh, then
// hipHostRegister.  Interior boundaries must be multiples of the page size (the callers use 2 MiB) so that no page
// belongs to two registrations.
bool prepare_host_piece(HostTarget& T, size_t lo, size_t hi)
{
	if (T.caller_pinned || hi <= lo)
		return true;
	char* base = T.out + lo;
	const size_t bytes = hi - lo;
	if (T.fresh)
	{
		const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
		const unsigned nt = (unsigned)std::min<size_t>(std::min(16u, hw), std::max<size_t>(1, bytes >> 24));
		const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
		volatile char* c = base;
		auto touch = [=](unsigned t) {
			const size_t b = std::min(bytes, per * t), e = std::min(bytes, per * (t + 1));
			for (size_t o = b; o < e; o += 4096)
				c[o] = 0;
			if (e > b)
				c[e - 1] = 0;
		};
		std::vector<std::thread> th;
		for (unsigned t = 1; t < nt; ++t)
			th.emplace_back(touch, t);
		touch(0);
		for (auto& t : th)
			t.join();
	}
	if (hipHostRegister(base, bytes, hipHostRegisterPortable) != hipSuccess)
	{
		(void)hipGetLastError();
		g_prepared.forget(T.out, T.bytes); // (an address range remembered from an earlier call may have been freed and mapped anew)
		return false;
	}
	T.registered.push_back(base);
	T.prepared += bytes;
	if (T.fresh && T.prepared >= T.bytes)
		g_prepared.add(T.out, T.bytes);
	return true;
}
// the whole array at once (calls that deal one array to several devices)
bool prepare_host_target(void* out, size_t bytes, HostTarget& T)
{
	return begin_host_target(out, bytes, T) && prepare_host_piece(T, 0, bytes);
}

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
// [node_begin, node_end) cut at slab boundaries into chunks whose sizes follow `fractions` (of the whole
// range) as closely as the slabs allow.  The direct form wants few chunks -- every chunk ends with a
// kernel tail of ~0.4 ms -- and a small LAST one, whose copy is the only one nothing overlaps.
void schedule_cuts(const uint32_t res[3], uint64_t node_begin, uint64_t node_end, const std::vector<double>& fractions,
				   std::vector<uint64_t>& cuts)
{
	dg::ClassGeom cg[4];
	dg::class_geometry(res, cg);
	std::vector<uint64_t> cand;
	for (int c = 0; c < 4; ++c)
	{
		const uint64_t slab = (uint64_t)dg::kSlabPlanes * cg[c].D[0] * cg[c].D[1];
		for (uint64_t at = cg[c].off; at < cg[c].off + cg[c].size; at += slab)
			if (at > node_begin && at < node_end)
				cand.push_back(at);
	}
	std::sort(cand.begin(), cand.end());
	cuts.assign(1, node_begin);
	const double n = (double)(node_end - node_begin);
	double acc = 0.0;
	for (size_t i = 0; i + 1 < fractions.size(); ++i)
	{
		acc += fractions[i];
		const uint64_t want = node_begin + (uint64_t)(acc * n);
		auto it = std::lower_bound(cand.begin(), cand.end(), want);
		if (it != cand.begin() && (it == cand.end() || *it - want > want - *(it - 1)))
			--it;
		if (it != cand.end() && *it > cuts.back())
			cuts.push_back(*it);
	}
	cuts.push_back(node_end);
}
// chunk sizes of the direct form as fractions of the range (DG_FORCE=host_direct_fractions=0.1,0.2,...: experiments)
std::vector<double> parse_fractions(std::vector<double> f, const char* force_key)
{
	std::string forced;
	if (dg::force_lookup(force_key, forced))
	{
		const char* e = forced.c_str();
		std::vector<double> g;
		double sum = 0;
		for (const char* p = e; *p;)
		{
			char* end = nullptr;
			const double v = std::strtod(p, &end);
			if (end == p)
				break;
			if (v > 0)
			{
				g.push_back(v);
				sum += v;
			}
			p = (*end == ',') ? end + 1 : end;
		}
		if (g.size() >= 1 && sum > 0)
		{
			for (double& v : g)
				v /= sum;
			f = g;
		}
	}
	return f;
}
std::vector<double> direct_fractions() { return parse_fractions({0.22, 0.22, 0.20, 0.16, 0.11, 0.06, 0.03}, "host_direct_fractions"); }
// dg_sdf_sample_field: ONE profile for callers that read the field on the device next and for callers that wait for the
// host vector (round 4).  The copy engine moves 57 GB/s, K1 produces 64 GB/s: the copy must start early and never run
// dry, so the chunks are fine and shrink towards the end (the last copy is what the host waits for after the last kernel).
// Measured at 256^3 on one stream (ms until the field is complete on the device / in the host vector; one launch then
// eight copy pieces: 16.4 / 34.7): 3 chunks 17.2 / 27.2, 4 chunks 19.0 / 25.3, 5 chunks 18.1 / 23.1, these seven 18.3 / 21.6.
std::vector<double> field_fractions() { return parse_fractions({0.22, 0.22, 0.20, 0.16, 0.11, 0.06, 0.03}, "field_fractions"); }
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
	hipError_t e = pipe.prepare(off.back(), true);
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

// K1, direct form: the copy engine writes every chunk straight into the caller's array.  Everything is enqueued up
// front: the compute stream samples chunk i into device buffer i & 1 as soon as the copy of chunk i - 2 has left it, the
// copy stream writes chunk i into the caller's array as soon as it is sampled, and the host only waits for the last
// copy.  Meanwhile the host makes the array a DMA target PIECE BY PIECE (DirectHost::piece: first touch + registration
// of the bytes chunk i ends in), one piece ahead of the copy that needs it: preparing the whole array first took 6 ms
// of a 23 ms call at 256^3 during which no copy ran; piece by piece the preparation costs more in total (~1.4 ms per
// registration) but hides behind the sampling, and the call ends one kernel tail after the last chunk: 23.0 -> 21.9 ms
// on the same box (DG_FORCE=host_pieces=0: the whole array at once, as before).  Measured and not kept: the chunks alternating
// between two compute streams so that a chunk's tail runs under the next chunk's bulk (25.0 ms: the two launches slow
// each other down by more than the tails they hide), other chunk size profiles (DG_FORCE=host_direct_fractions=...).
// `begin` (cheap: mode, size, page state) runs before anything is enqueued: may the direct form run at all?  If it
// says no the caller runs the staged form and nothing was sampled twice; only if a PIECE cannot be prepared later
// (registration fails) are the chunks in flight given up and the staged form run over the whole range.
struct DirectHost
{
	std::function<bool()> begin;
	std::function<bool(size_t, size_t)> piece; // bytes [lo, hi) of `out`; empty: the whole array already is a DMA target
};
static dg_status run_k1_direct(HostPipe& pipe, const dg_mesh* mesh, const dg_grid_desc* grid, int invert,
							   const std::vector<uint64_t>& cuts, size_t first, size_t stride, const uint8_t* pred_mask, double* out,
							   const DirectHost& host, bool* went_direct, double* kernel_ms, Progress* progress = nullptr)
{
	*went_direct = true;
	const size_t n_chunks = cuts.size() - 1;
	std::vector<size_t> mine;
	uint64_t longest = 0;
	for (size_t k = first; k < n_chunks; k += stride)
	{
		mine.push_back(k);
		longest = std::max(longest, cuts[k + 1] - cuts[k]);
	}
	if (longest == 0)
		return DG_OK;
	const size_t out_bytes = (longest * sizeof(double) + 255) & ~(size_t)255;
	const size_t mask_bytes = pred_mask ? ((longest + 255) & ~(size_t)255) : 0;
	hipError_t e = pipe.prepare(out_bytes + mask_bytes, false);
	if (e == hipSuccess) e = pipe.events(3 * mine.size());
	dg_status st = DG_OK;
	// piece i = bytes [bound[i], bound[i + 1]) of the array: chunk i's bytes end in it.  Interior bounds are the chunk
	// starts rounded up to 2 MiB (of the address), so a chunk may begin in the last 2 MiB of the piece before.
	const bool by_piece = force_int("host_pieces", 1, 0, 1) != 0;
	const bool pieces = (bool)host.piece && stride == 1 && first == 0 && by_piece;
	const size_t total_bytes = (size_t)(cuts[n_chunks] - cuts[0]) * sizeof(double);
	std::vector<size_t> bound(mine.size() + 1, 0);
	bound[mine.size()] = total_bytes;
	for (size_t i = 1; i < mine.size(); ++i)
	{
		const uintptr_t at = (uintptr_t)out + (size_t)(cuts[mine[i]] - cuts[0]) * sizeof(double);
		const uintptr_t up = (at + ((uintptr_t)1 << 21) - 1) & ~(((uintptr_t)1 << 21) - 1);
		bound[i] = std::max(bound[i - 1], std::min<size_t>(total_bytes, (size_t)(up - (uintptr_t)out)));
	}

	auto enqueue_kernel = [&](size_t i) {
		const size_t k = mine[i];
		const int b = (int)(i & 1);
		const hipStream_t cs = pipe.compute;
		const uint64_t cn = cuts[k + 1] - cuts[k];
		double* d_out = static_cast<double*>(pipe.d_buf[b]);
		uint8_t* d_mask = pred_mask ? reinterpret_cast<uint8_t*>(static_cast<char*>(pipe.d_buf[b]) + out_bytes) : nullptr;
		if (i >= 2)
			e = hipStreamWaitEvent(cs, pipe.ev[3 * (i - 2) + 2], 0);
		if (e == hipSuccess) e = hipEventRecord(pipe.ev[3 * i], cs);
		if (e == hipSuccess && pred_mask)
			e = hipMemcpyAsync(d_mask, pred_mask + (cuts[k] - cuts[0]), cn, hipMemcpyHostToDevice, cs);
		if (e == hipSuccess)
			st = dg_sdf_sample_nodes_device(mesh, grid, invert, cuts[k], cuts[k + 1], d_mask, d_out, cs);
		if (e == hipSuccess && st == DG_OK) e = hipEventRecord(pipe.ev[3 * i + 1], cs);
	};
	auto enqueue_copy = [&](size_t i) {
		const size_t k = mine[i];
		const size_t lo = (size_t)(cuts[k] - cuts[0]) * sizeof(double), hi = (size_t)(cuts[k + 1] - cuts[0]) * sizeof(double);
		e = hipStreamWaitEvent(pipe.copy, pipe.ev[3 * i + 1], 0);
		// one copy per registration the chunk's bytes lie in (a copy must not straddle two registrations)
		size_t at = lo;
		while (e == hipSuccess && at < hi)
		{
			size_t end = hi;
			if (pieces)
				for (size_t j = 1; j < mine.size(); ++j)
					if (bound[j] > at && bound[j] < end)
						end = bound[j];
			e = hipMemcpyAsync(reinterpret_cast<char*>(out) + at, static_cast<const char*>(pipe.d_buf[i & 1]) + (at - lo), end - at,
							   hipMemcpyDeviceToHost, pipe.copy);
			at = end;
		}
		if (e == hipSuccess) e = hipEventRecord(pipe.ev[3 * i + 2], pipe.copy);
	};
	auto give_up = [&]() { // the chunks already in flight are simply sampled again by the staged form
		(void)hipStreamSynchronize(pipe.compute);
		(void)hipStreamSynchronize(pipe.copy);
		*went_direct = false;
		return DG_OK;
	};
	// The cheap tests (mode, size, who owns the pages, are they resident) run BEFORE anything is enqueued: a caller
	// whose array cannot take the direct form goes straight to the staged form and samples nothing twice.  Only
	// the expensive part -- first touch and registration, piece by piece -- runs under the first chunks.
	if (e == hipSuccess && !host.begin())
	{
		*went_direct = false;
		return DG_OK;
	}
	const size_t head = std::min<size_t>(2, mine.size());
	for (size_t i = 0; i < head && e == hipSuccess && st == DG_OK; ++i)
		enqueue_kernel(i);
	if (e == hipSuccess && st == DG_OK && !pieces && (bool)host.piece && !host.piece(0, total_bytes)) // the whole array at once
		return give_up();
	for (size_t i = 0; i < mine.size() && e == hipSuccess && st == DG_OK; ++i)
	{
		if (i >= head)
			enqueue_kernel(i);
		if (e == hipSuccess && st == DG_OK && pieces && !host.piece(bound[i], bound[i + 1]))
			return give_up();
		if (e == hipSuccess && st == DG_OK)
			enqueue_copy(i);
	}
	if (progress && progress->cb && e == hipSuccess && st == DG_OK)
	{
		uint64_t done = 0;
		for (size_t i = 0; i < mine.size(); ++i)
		{
			if (hipEventSynchronize(pipe.ev[3 * i + 2]) != hipSuccess)
				break;
			done += cuts[mine[i] + 1] - cuts[mine[i]];
			progress->report(done);
		}
	}
	const hipError_t e1 = hipStreamSynchronize(pipe.compute), e2 = hipStreamSynchronize(pipe.copy);
	if (st != DG_OK)
		return st;
	if (e == hipSuccess) e = e1;
	if (e == hipSuccess) e = e2;
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "dg_sdf_sample_nodes (direct): %s", hipGetErrorString(e));
	for (size_t i = 0; i < mine.size(); ++i)
	{
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, pipe.ev[3 * i], pipe.ev[3 * i + 1]) == hipSuccess)
			*kernel_ms += ms;
	}
	return DG_OK;
}

// Streams of destroyed produced fields and finished copy jobs, kept for the next ones: creating a stream costs
// milliseconds on this runtime (the first use of a new stream sets up a hardware queue), more than the launch
// work it then carries.  A stream goes back only when everything enqueued on it has finished.
namespace
{
// Two kinds, never mixed: producer streams (kernels) and copy streams.  Copy streams are created with the highest
// priority: the runtime multiplexes the streams of a process onto a few hardware queues, and a copy stream that
// lands on the queue of the stream that samples the field has its copies ordered BEHIND the kernels still queued
// there -- the host array was then complete after kernels + copies (36 ms at 256^3) instead of kernels || copies
// (22 ms) on every second call [MI355X]; streams of another priority get hardware queues of their own.
struct StreamPool
{
	std::mutex mutex;
	struct Idle
	{
		int device, kind;
		hipStream_t s;
	};
	std::vector<Idle> idle;
	hipError_t take(int device, int kind, hipStream_t* out)
	{
		{
			std::lock_guard<std::mutex> lock(mutex);
			for (size_t i = 0; i < idle.size(); ++i)
				if (idle[i].device == device && idle[i].kind == kind)
				{
					*out = idle[i].s;
					idle.erase(idle.begin() + (long)i);
					return hipSuccess;
				}
		}
		if (kind == 1)
		{
			int least = 0, greatest = 0;
			if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
				hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest) == hipSuccess)
				return hipSuccess;
			(void)hipGetLastError();
		}
		return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
	}
	void give(int device, int kind, hipStream_t s)
	{
		std::lock_guard<std::mutex> lock(mutex);
		if (idle.size() < 16)
			idle.push_back({device, kind, s});
		else
			(void)hipStreamDestroy(s);
	}
} g_streams;
} // namespace
void recycle_stream(int device, hipStream_t s) { g_streams.give(device, 0, s); }

// ---- fields produced on the device: the asynchronous copy into the caller's host array ---------------------------
// dg_sdf_sample_field / dg_density_map_field leave the coefficients in a device array the new field handle owns and
// return once the kernels are enqueued.  If the caller wants the coefficients on the host as well, a worker thread
// does what run_k1_direct does inline: it makes the caller's array a DMA target piece by piece (first touch +
// hipHostRegister, ~1.4 ms per piece) and enqueues, on a copy stream of its own, the copy of every segment behind
// the event that says the segment is final on the device; then it waits for the copies and unregisters.  An array
// that cannot take the direct form (resident 4 KiB pages: pinning would cost more than it saves) is filled by
// blocking copies, segment by segment, still from the worker.  dg_field_host_wait() joins the worker.
struct HostCopyJob
{
	std::thread worker;
	int device = -1;
	const char* d_src = nullptr;
	char* h_dst = nullptr;
	std::vector<hipEvent_t> ev; // owned; ev[i]: bytes [seg[i], seg[i + 1]) are final on the device
	std::vector<size_t> seg;
	dg_status status = DG_OK;
	std::string message;
	bool direct = false;
	bool after_kernel = false; // nothing can be copied before ev[0] anyway: keep out of the runtime until then (see run())
	double seconds = 0.0;

	void run()
	{
		const auto t0 = std::chrono::steady_clock::now();
		DeviceGuard guard(device);
		hipError_t e = guard.err;
		// One launch produces the whole field: the copies cannot start before it ends, and a hipHostRegister issued now
		// would hold the runtime for ~2 ms at a time while the caller is enqueueing the consumer of the field on the
		// device (measured: that consumer started 2 ms late).  Wait for the kernel first; the first piece is prepared then.
		if (after_kernel && e == hipSuccess && !ev.empty())
			e = hipEventSynchronize(ev[0]);
		hipStream_t copy = nullptr;
		if (e == hipSuccess) e = g_streams.take(device, 1, &copy);
		const size_t total = seg.back();
		HostTarget T;
		direct = e == hipSuccess && begin_host_target(h_dst, total, T);
		const size_t n = ev.size();
		// piece i = bytes [bound[i], bound[i + 1]): interior bounds are the segment starts rounded up to 2 MiB of the address
		std::vector<size_t> bound(n + 1, 0);
		bound[n] = total;
		for (size_t i = 1; i < n; ++i)
		{
			const uintptr_t at = (uintptr_t)h_dst + seg[i];
			const uintptr_t up = (at + ((uintptr_t)1 << 21) - 1) & ~(((uintptr_t)1 << 21) - 1);
			bound[i] = std::max(bound[i - 1], std::min<size_t>(total, (size_t)(up - (uintptr_t)h_dst)));
		}
		size_t next = 0; // first segment that has not been enqueued / copied
		const bool debug = dg::force_set("host_debug");
		if (debug)
			std::fprintf(stderr, "  host copy job: started after %.2f ms, %s\n",
						 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3, direct ? "direct" : "blocking copies");
		for (; direct && next < n && e == hipSuccess; ++next)
		{
			const auto tp = std::chrono::steady_clock::now();
			const bool ok = prepare_host_piece(T, bound[next], bound[next + 1]);
			if (debug)
				std::fprintf(stderr, "  host copy job: piece %zu [%zu, %zu) MiB prepared in %.2f ms (at %.2f ms)\n", next, bound[next] >> 20,
							 bound[next + 1] >> 20, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count() * 1e3,
							 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
			if (!ok)
			{
				direct = false;
				break;
			}
			e = hipStreamWaitEvent(copy, ev[next], 0);
			size_t at = seg[next];
			while (e == hipSuccess && at < seg[next + 1]) // a copy must not straddle two registrations
			{
				size_t end = seg[next + 1];
				for (size_t j = 1; j < n; ++j)
					if (bound[j] > at && bound[j] < end)
						end = bound[j];
				e = hipMemcpyAsync(h_dst + at, d_src + at, end - at, hipMemcpyDeviceToHost, copy);
				at = end;
			}
		}
		if (copy && e == hipSuccess) e = hipStreamSynchronize(copy);
		if (next < n && e == hipSuccess)
		{
			// the rest (or everything) by blocking copies into ordinary memory
			T.release();
			for (; next < n && e == hipSuccess; ++next)
			{
				e = hipEventSynchronize(ev[next]);
				if (e == hipSuccess)
					e = hipMemcpy(h_dst + seg[next], d_src + seg[next], seg[next + 1] - seg[next], hipMemcpyDeviceToHost);
			}
		}
		if (copy)
		{
			(void)hipStreamSynchronize(copy);
			g_streams.give(device, 1, copy);
		}
		if (e != hipSuccess)
		{
			status = e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP;
			message = std::string("copy of a device-resident field into the host array: ") + hipGetErrorString(e);
			(void)hipGetLastError();
		}
		seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}
	~HostCopyJob()
	{
		if (worker.joinable())
			worker.join();
		for (hipEvent_t x : ev)
			(void)hipEventDestroy(x);
	}
};

dg_status finish_host_job(dg_field* field)
{
	HostCopyJob* job = nullptr;
	{
		std::lock_guard<std::mutex> lock(field->host_mutex);
		job = field->host_job;
		if (job && job->worker.joinable())
			job->worker.join(); // under the lock: concurrent waiters all return only when the array is complete
		field->host_job = nullptr;
	}
	if (!job)
		return DG_OK;
	const dg_status st = job->status;
	const std::string msg = job->message;
	if (dg::force_set("host_debug"))
		std::fprintf(stderr, "host copy job: %s, %zu segments, %.1f ms\n", job->direct ? "direct" : "blocking copies", job->ev.size(),
					 job->seconds * 1e3);
	{
		DeviceGuard guard(field->device);
		delete job;
	}
	return st == DG_OK ? DG_OK : fail(st, "%s", msg.c_str());
}

// Coefficient arrays of destroyed produced fields, kept for the next one of the same size on the same device: a
// 0.95 GB hipMalloc costs 2.5-3 ms, a fifth of the sampling it precedes [MI355X].  At most DG_FIELD_CACHE_MB
// megabytes (default 2048; 0: nothing is kept) stay cached per process; dg_field_cache_trim() releases them.
namespace
{
struct FieldBufferCache
{
	struct Buf
	{
		void* p;
		size_t bytes;
		int device;
	};
	std::mutex mutex;
	std::vector<Buf> bufs;
	size_t cached = 0;
	void* take(size_t bytes, int device)
	{
		std::lock_guard<std::mutex> lock(mutex);
		for (size_t i = 0; i < bufs.size(); ++i)
			if (bufs[i].device == device && bufs[i].bytes == bytes)
			{
				void* p = bufs[i].p;
				cached -= bytes;
				bufs.erase(bufs.begin() + (long)i);
				return p;
			}
		return nullptr;
	}
	bool give(void* p, size_t bytes, int device)
	{
		const size_t cap = (size_t)env_int("DG_FIELD_CACHE_MB", 2048, 0, 1 << 20) << 20;
		std::lock_guard<std::mutex> lock(mutex);
		if (bytes > cap)
			return false;
		while (cached + bytes > cap && !bufs.empty()) // oldest first
		{
			DeviceGuard guard(bufs.front().device);
			(void)hipFree(bufs.front().p);
			cached -= bufs.front().bytes;
			bufs.erase(bufs.begin());
		}
		bufs.push_back({p, bytes, device});
		cached += bytes;
		return true;
	}
	void trim()
	{
		std::lock_guard<std::mutex> lock(mutex);
		for (Buf& b : bufs)
		{
			DeviceGuard guard(b.device);
			(void)hipFree(b.p);
		}
		bufs.clear();
		cached = 0;
	}
} g_field_buffers;
} // namespace

// (called by dg_field_destroy for the coefficient array of a produced field; the device is current and idle for it)
bool recycle_field_buffer(void* p, size_t bytes, int device) { return g_field_buffers.give(p, bytes, device); }


// a new field on `grid` whose coefficient array (n doubles, uninitialised) it owns, with a producer stream and event
static dg_status new_produced_field(const dg_grid_desc* grid, uint64_t n, dg_field** out)
{
	int device = 0;
	(void)hipGetDevice(&device);
	const size_t bytes = std::max<uint64_t>(n, 1) * sizeof(double);
	void* d_c = g_field_buffers.take(bytes, device);
	hipError_t e = d_c ? hipSuccess : hipMalloc(&d_c, bytes);
	if (e != hipSuccess)
	{
		g_field_buffers.trim(); // the cache must never be the reason an allocation fails
		(void)hipGetLastError();
		e = hipMalloc(&d_c, bytes);
	}
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "device allocation of %llu bytes: %s",
					(unsigned long long)(n * sizeof(double)), hipGetErrorString(e));
	dg_status st = dg_field_attach_device(grid, static_cast<const double*>(d_c), n, nullptr, 0, nullptr, out);
	if (st != DG_OK)
	{
		(void)hipFree(d_c);
		return st;
	}
	dg_field* f = *out;
	f->owned[0] = d_c;
	f->recyclable_bytes = bytes;
	e = g_streams.take(device, 0, &f->producer_stream);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&f->produced, hipEventDisableTiming);
	if (e != hipSuccess)
	{
		dg_field_destroy(f);
		*out = nullptr;
		return fail(DG_ERR_HIP, "stream / event creation: %s", hipGetErrorString(e));
	}
	return DG_OK;
}

// uploads the predicate mask of a producing call (kept with the field until it is destroyed)
static dg_status upload_producer_mask(dg_field* f, const uint8_t* pred_mask, uint64_t n)
{
	if (!pred_mask)
		return DG_OK;
	DG_HIP(hipMalloc(&f->d_producer_mask, n));
	DG_HIP(hipMemcpy(f->d_producer_mask, pred_mask, n, hipMemcpyHostToDevice));
	return DG_OK;
}

// items [0, n) in uniform chunks (K1p, K2): big enough to amortise the launches, small enough to overlap
static void uniform_cuts(uint64_t n, int default_chunk, std::vector<uint64_t>& cuts)
{
	const uint64_t chunk = (uint64_t)force_int("host_chunk_items", default_chunk, 1 << 8, 1 << 28);
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

extern "C"
{

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
	const uint64_t target = (uint64_t)force_int("host_chunk_nodes", (int)auto_target, 1 << 10, 1 << 28);
	std::vector<uint64_t> cuts, direct_cuts;
	chunk_cuts(grid->resolution, node_begin, node_end, target, cuts);
	// direct form: seven chunks, the last one small -- its copy is the only one nothing overlaps
	// (DG_FORCE=host_chunk_nodes set: the uniform chunks above, as in the staged form)
	if (dg::force_set("host_chunk_nodes") || n < (1u << 24))
		direct_cuts = cuts;
	else
		schedule_cuts(grid->resolution, node_begin, node_end, direct_fractions(), direct_cuts);
	double kernel_ms = 0, t_wait = 0, t_copy = 0;
	DG_ON_DEVICE_OF(mesh);
	PipeLease lease = lease_pipe(mesh->device);
	HostTarget host_target;
	bool direct = false;
	double t_prepare = 0;
	Progress progress{t_progress, t_progress_user, n};
	DirectHost host;
	host.begin = [&]() { return begin_host_target(out, n * sizeof(double), host_target); };
	host.piece = [&](size_t lo, size_t hi) {
		const auto t0 = std::chrono::steady_clock::now();
		const bool ok = prepare_host_piece(host_target, lo, hi);
		const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (dg::force_set("host_debug"))
			std::fprintf(stderr, "  piece [%zu, %zu) MiB prepared in %.2f ms\n", lo >> 20, hi >> 20, dt * 1e3);
		t_prepare += dt;
		return ok;
	};
	s = run_k1_direct(*lease.pipe, mesh, grid, invert, direct_cuts, 0, 1, pred_mask, out, host, &direct, &kernel_ms, &progress);
	if (s == DG_OK && !direct)
	{
		kernel_ms = 0;
		s = run_k1_chunks(*lease.pipe, mesh, grid, invert, cuts, 0, 1, pred_mask, out, &kernel_ms, &t_wait, &t_copy);
	}
	if (s != DG_OK)
		return s;
	progress.report(n);
	g_last_ms = kernel_ms;
	if (dg::force_set("host_debug"))
		std::fprintf(stderr, "dg_sdf_sample_nodes: %s, %zu chunks, kernels %.1f ms, host prepared the array in %.1f ms, waited %.1f ms, copied %.1f ms\n",
					 direct ? "direct" : "staged", (direct ? direct_cuts : cuts).size() - 1, kernel_ms, t_prepare * 1e3, t_wait * 1e3, t_copy * 1e3);
	return DG_OK;
}

dg_status dg_sdf_sample_field(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, const uint8_t* pred_mask,
							  double* host_out, int host_first, dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!mesh || !grid)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	const uint64_t n = dg_grid_n_nodes(grid);
	DG_ON_DEVICE_OF(mesh);
	const bool debug = dg::force_set("host_debug");
	const auto t_begin = std::chrono::steady_clock::now();
	dg_field* f = nullptr;
	s = new_produced_field(grid, n, &f);
	if (s == DG_OK) s = upload_producer_mask(f, pred_mask, n);
	if (s != DG_OK)
	{
		dg_field_destroy(f);
		return s;
	}
	const auto t_alloc = std::chrono::steady_clock::now();
	double* d_c = static_cast<double*>(f->owned[0]);
	const uint8_t* d_mask = static_cast<const uint8_t*>(f->d_producer_mask);
	// Without a host array: one launch over the whole lattice.  With one: the chunk profile of the direct form, so
	// that the copy of chunk i runs under the sampling of chunk i + 1 (every chunk costs a kernel tail of ~0.4 ms).
	std::vector<uint64_t> cuts, copy_cuts;
	if (host_out == nullptr || n < (1u << 22))
		cuts = {0, n};
	else if (!host_first && env_int("DG_FIELD_ONE_LAUNCH", 0, 0, 1) != 0)
	{
		// (round 3's lazy form, kept for A/B: one launch, no chunk tails; the copy follows in eight pieces so that the
		// registration of piece i + 1 runs under the copy of piece i -- host vector complete 2.4 x the kernel time after the call)
		cuts = {0, n};
		const uint64_t pieces = std::max<uint64_t>(1, std::min<uint64_t>(8, (n * sizeof(double)) >> 24));
		for (uint64_t i = 0; i <= pieces; ++i)
			copy_cuts.push_back(i == pieces ? n : ((n / pieces * i) & ~(uint64_t)511));
	}
	else if (dg::force_set("host_chunk_nodes") || n < (1u << 24))
	{
		const uint64_t auto_target = std::min<uint64_t>(std::max<uint64_t>(n / 10, 1u << 22), 1u << 25);
		chunk_cuts(grid->resolution, 0, n, (uint64_t)force_int("host_chunk_nodes", (int)auto_target, 1 << 10, 1 << 28), cuts);
	}
	else
	{
		// the copy of chunk i can only start when chunk i is sampled and runs under the sampling of chunk i + 1
		// (field_fractions() above; host_first is a hint without effect since round 4: one policy serves both kinds of caller)
		schedule_cuts(grid->resolution, 0, n, field_fractions(), cuts);
	}
	hipEvent_t dbg0 = nullptr, dbg1 = nullptr;
	if (debug)
	{
		(void)hipEventCreate(&dbg0);
		(void)hipEventCreate(&dbg1);
		(void)hipEventRecord(dbg0, f->producer_stream);
	}
	HostCopyJob* job = nullptr;
	if (host_out)
	{
		job = new (std::nothrow) HostCopyJob;
		if (!job)
		{
			dg_field_destroy(f);
			return fail(DG_ERR_ALLOC, "host allocation failed");
		}
		job->device = mesh->device;
		job->d_src = reinterpret_cast<const char*>(d_c);
		job->h_dst = reinterpret_cast<char*>(host_out);
		job->after_kernel = !copy_cuts.empty();
	}
	hipError_t e = hipSuccess;
	// DG_FORCE=field_streams=2: the chunks alternate between the field's stream and a second one, so that a chunk's last waves
	// (and its two small heavy-brick kernels) do not run alone.  Measured at 256^3: 18.5 / 21.7 ms (device / host complete)
	// against 18.6 / 21.8 on one stream -- nothing: the 2.2 ms the chunks cost a consumer on the device against one launch
	// (16.3 ms) are not tails.  Off by default.
	hipStream_t second = nullptr;
	hipEvent_t second_done = nullptr;
	if (cuts.size() > 2 && force_int("field_streams", 1, 1, 2) == 2)
	{
		if (g_streams.take(mesh->device, 0, &second) != hipSuccess)
			second = nullptr;
		if (second && hipEventCreateWithFlags(&second_done, hipEventDisableTiming) != hipSuccess)
		{
			g_streams.give(mesh->device, 0, second);
			second = nullptr;
		}
	}
	size_t launch_no = 0;
	for (size_t k = 0; k + 1 < cuts.size() && e == hipSuccess && s == DG_OK; ++k)
	{
		if (cuts[k + 1] == cuts[k])
			continue;
		hipStream_t chunk_stream = (second && (launch_no++ & 1u)) ? second : f->producer_stream;
		s = dg_sdf_sample_nodes_device(mesh, grid, invert, cuts[k], cuts[k + 1], d_mask ? d_mask + cuts[k] : nullptr, d_c + cuts[k],
									   chunk_stream);
		// the segments of the host copy: the chunk itself, or (one launch) the pieces of copy_cuts behind the whole launch
		const std::vector<uint64_t> segs = copy_cuts.empty() ? std::vector<uint64_t>{cuts[k], cuts[k + 1]} : copy_cuts;
		for (size_t g = 0; g + 1 < segs.size() && s == DG_OK && job && e == hipSuccess; ++g)
		{
			hipEvent_t ev = nullptr;
			e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
			if (e == hipSuccess)
			{
				job->ev.push_back(ev);
				if (job->seg.empty())
					job->seg.push_back(segs[g] * sizeof(double));
				job->seg.push_back(segs[g + 1] * sizeof(double));
				e = hipEventRecord(ev, chunk_stream);
			}
		}
	}
	if (second)
	{
		// the field is complete when both streams are through: the field's stream waits for the second one
		if (e == hipSuccess && s == DG_OK) e = hipEventRecord(second_done, second);
		if (e == hipSuccess && s == DG_OK) e = hipStreamWaitEvent(f->producer_stream, second_done, 0);
		if (e != hipSuccess || s != DG_OK)
			(void)hipStreamSynchronize(second);
		(void)hipEventDestroy(second_done); // (released by the runtime once the recorded work has completed)
		g_streams.give(mesh->device, 0, second);
	}
	if (e == hipSuccess && s == DG_OK) e = hipEventRecord(f->produced, f->producer_stream);
	if (e != hipSuccess || s != DG_OK)
	{
		(void)hipStreamSynchronize(f->producer_stream);
		delete job;
		dg_field_destroy(f);
		return s != DG_OK ? s : fail(DG_ERR_HIP, "dg_sdf_sample_field: %s", hipGetErrorString(e));
	}
	if (job)
	{
		f->host_job = job;
		job->worker = std::thread([job]() { job->run(); });
	}
	if (debug)
	{
		const auto t_enq = std::chrono::steady_clock::now();
		(void)hipEventRecord(dbg1, f->producer_stream);
		(void)hipEventSynchronize(dbg1); // (debug only: waits for the sampling)
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, dbg0, dbg1);
		std::fprintf(stderr, "dg_sdf_sample_field: %zu chunks, allocation %.2f ms, enqueue %.2f ms, kernels on the device %.2f ms, done %.2f ms after the call began\n",
					 cuts.size() - 1, std::chrono::duration<double>(t_alloc - t_begin).count() * 1e3,
					 std::chrono::duration<double>(t_enq - t_alloc).count() * 1e3, ms,
					 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3);
		(void)hipEventDestroy(dbg0);
		(void)hipEventDestroy(dbg1);
	}
	*out = f;
	return DG_OK;
}

dg_status dg_density_map_field(dg_field* sdf, double support_radius, double rho0, int band_predicate,
							   const uint8_t* pred_mask, double* host_out, dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!sdf)
		return fail(DG_ERR_INVALID, "null argument");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	const uint64_t n = dg_grid_n_nodes(&sdf->grid);
	DG_ON_DEVICE_OF(sdf);
	dg_field* f = nullptr;
	s = new_produced_field(&sdf->grid, n, &f);
	if (s == DG_OK) s = upload_producer_mask(f, pred_mask, n);
	if (s == DG_OK)
		s = dg_density_map_nodes_device(sdf, support_radius, rho0, band_predicate, 0, n, static_cast<const uint8_t*>(f->d_producer_mask),
										static_cast<double*>(f->owned[0]), f->producer_stream);
	hipError_t e = hipSuccess;
	if (s == DG_OK) e = hipEventRecord(f->produced, f->producer_stream);
	HostCopyJob* job = nullptr;
	if (s == DG_OK && e == hipSuccess && host_out)
	{
		// one kernel, one event: the copy starts when K3 ends; eight pieces so that the registration of piece i + 1
		// runs under the copy of piece i
		job = new (std::nothrow) HostCopyJob;
		if (!job)
			s = fail(DG_ERR_ALLOC, "host allocation failed");
		else
		{
			job->device = sdf->device;
			job->d_src = static_cast<const char*>(f->owned[0]);
			job->h_dst = reinterpret_cast<char*>(host_out);
			job->after_kernel = true;
			const size_t total = n * sizeof(double);
			const size_t pieces = std::max<size_t>(1, std::min<size_t>(8, total >> 24));
			job->seg.push_back(0);
			for (size_t i = 1; i <= pieces && e == hipSuccess; ++i)
			{
				hipEvent_t ev = nullptr;
				e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
				if (e != hipSuccess)
					break;
				job->ev.push_back(ev);
				job->seg.push_back(i == pieces ? total : ((total / pieces * i) & ~(size_t)4095));
				e = hipEventRecord(ev, f->producer_stream);
			}
		}
	}
	if (s != DG_OK || e != hipSuccess)
	{
		if (f && f->producer_stream) (void)hipStreamSynchronize(f->producer_stream);
		delete job;
		dg_field_destroy(f);
		return s != DG_OK ? s : fail(DG_ERR_HIP, "dg_density_map_field: %s", hipGetErrorString(e));
	}
	if (job)
	{
		f->host_job = job;
		job->worker = std::thread([job]() { job->run(); });
	}
	*out = f;
	return DG_OK;
}

void dg_field_cache_trim(void) { g_field_buffers.trim(); }

dg_status dg_field_host_wait(dg_field* field)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	return finish_host_job(field);
}

void dg_set_progress_callback(dg_progress_fn cb, void* user)
{
	t_progress = cb;
	t_progress_user = user;
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
	for (int i = 0; i < n_meshes; ++i)
		if (meshes[i]->device < 0)
			return fail(DG_ERR_NO_DEVICE, "meshes[%d] is a host-only handle (dg_mesh_device() < 0)", i);
	// chunks are dealt round-robin: thin interleaved pieces equalise the very uneven cost per node
	const uint64_t auto_target = std::min<uint64_t>(std::max<uint64_t>(n / (10ull * (uint64_t)n_meshes), 1u << 21), 1u << 25);
	const uint64_t target = (uint64_t)force_int("host_chunk_nodes", (int)auto_target, 1 << 10, 1 << 28);
	std::vector<uint64_t> cuts;
	chunk_cuts(grid->resolution, node_begin, node_end, target, cuts);
	std::vector<dg_status> status((size_t)n_meshes, DG_OK);
	std::vector<std::string> message((size_t)n_meshes);
	std::vector<double> kernel_ms((size_t)n_meshes, 0.0);
	std::vector<std::thread> workers;
	// the array becomes a DMA target once, for all devices, before the workers start (3-8 ms; the staged
	// form runs if it cannot)
	HostTarget host_target;
	const bool direct = prepare_host_target(out, n * sizeof(double), host_target);
	for (int i = 0; i < n_meshes; ++i)
		workers.emplace_back([&, i]() {
			DeviceGuard guard(meshes[i]->device);
			if (guard.err != hipSuccess)
			{
				status[(size_t)i] = DG_ERR_HIP;
				message[(size_t)i] = "hipSetDevice failed";
				return;
			}
			double t_wait = 0, t_copy = 0;
			PipeLease lease = lease_pipe(meshes[i]->device);
			bool went = false;
			if (direct)
			{
				DirectHost whole; // the array was prepared above, for all devices
				whole.begin = []() { return true; };
				status[(size_t)i] = run_k1_direct(*lease.pipe, meshes[i], grid, invert, cuts, (size_t)i, (size_t)n_meshes, pred_mask, out,
												  whole, &went, &kernel_ms[(size_t)i]);
			}
			else
				status[(size_t)i] = run_k1_chunks(*lease.pipe, meshes[i], grid, invert, cuts, (size_t)i, (size_t)n_meshes, pred_mask, out,
												  &kernel_ms[(size_t)i], &t_wait, &t_copy);
			if (status[(size_t)i] != DG_OK)
				message[(size_t)i] = dg_last_error(); // thread-local: carry it to the caller
		});
	for (auto& w : workers)
		w.join();
	for (int i = 0; i < n_meshes; ++i)
		if (status[(size_t)i] != DG_OK)
			return fail(status[(size_t)i], "mesh %d (device %d): %s", i, meshes[i]->device, message[(size_t)i].c_str());
	g_last_ms = *std::max_element(kernel_ms.begin(), kernel_ms.end());
	if (t_progress)
		t_progress(n, n, t_progress_user);
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
	DG_ON_DEVICE_OF(mesh);
	PipeLease lease = lease_pipe(mesh->device);
	s = run_pipeline(*lease.pipe, cuts, 0, 1, arrays, launch, "dg_signed_distance", &kernel_ms, &t_wait, &t_copy);
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
	DG_ON_DEVICE_OF(sdf);
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
	DG_ON_DEVICE_OF(field);
	PipeLease lease = lease_pipe(field->device);
	s = run_pipeline(*lease.pipe, cuts, 0, 1, arrays, launch, "dg_interpolate_batch", &kernel_ms, &t_wait, &t_copy);
	if (s == DG_OK)
		g_last_ms = kernel_ms;
	return s;
}

} // extern "C"
