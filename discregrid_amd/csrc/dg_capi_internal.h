// dg_capi_internal.h -- shared by the translation units that implement include/discregrid_hip.h:
//   dg_capi.cpp        runtime, grid helpers, mesh handle, K1 / K1p device entry points, sharding
//   dg_capi_field.cpp  field handle, K2 and K3 device entry points
//   dg_capi_host.cpp   the host-pointer entry points (pinned staging pipeline)
#pragma once
#include "../../include/discregrid_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "dg_build.h"
#include "dg_force.h"
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
	void* d_tri_approx = nullptr;
	void* d_tris = nullptr;
	void* d_pn = nullptr;
	int device = -1;
	dg_mesh_info info;
	mutable std::mutex scratch_mutex;
	mutable std::deque<HeavyScratch> scratch; // deque: a push_back never moves the entries other launches hold
	mutable uint64_t scratch_serial = 0;
	mutable uint64_t unsplit_serial = 0; // serial of the last launch that ran without the split path
	mutable ScratchPool bin_scratch;     // K1p point binning
	mutable uint32_t* bin_flag_host = nullptr; // pinned: was the previous batch unordered? (prediction, starts at 1)
	double bbox_lo[3], bbox_hi[3];       // of the vertices
	dg::MeshBuild host;                  // the arrays that were uploaded, kept for dg_signed_distance_point (immutable)
	dg::MeshDev host_view;               // ... as the kernels' MeshDev (host pointers): what the one-lane traversal walks
};

struct HostCopyJob; // dg_capi_host.cpp: the asynchronous copy of a device-resident field into the caller's host array

struct dg_field
{
	dg::FieldDev dev;
	mutable ScratchPool scratch; // K2 query binning
	mutable ScratchPool tile_scratch; // K3: per-launch tile-major copy of an unreduced field
	mutable uint32_t* bin_flag_host = nullptr; // pinned: was the previous batch unordered? (prediction, starts at 1)
	void* owned[3] = {nullptr, nullptr, nullptr};
	void* d_tile_major = nullptr;
	void* d_cell_major = nullptr;
	void* d_band_rows = nullptr; // band-limited cell-major copy (dg_field_build_cell_major_band): rows and row map
	void* d_band_map = nullptr;
	uint64_t band_rows = 0;
	hipEvent_t band_ready = nullptr;
	mutable uint32_t* band_probe_host = nullptr; // pinned, 2 words: of the previous large batch's sampled queries, how many had a cell / a row in the band copy
	hipEvent_t cell_major_ready = nullptr; // recorded behind k_expand_cells: launches on other streams wait for it
	hipEvent_t tile_major_ready = nullptr; // the same for k_expand_tiles (one event per copy: they may be built on different streams)
	// A field whose coefficients a kernel of this library produces (dg_sdf_sample_field, dg_density_map_field): the
	// device array is owned, `produced` is recorded behind the last producing kernel on `producer_stream`, and every
	// consumer on another stream waits for it.  The copy into the caller's host array, if one was asked for, runs
	// from a worker thread (host_job) until dg_field_host_wait() / dg_field_destroy() collects it.
	hipEvent_t produced = nullptr;
	hipStream_t producer_stream = nullptr;
	size_t recyclable_bytes = 0;     // != 0: owned[0] goes back to the field-buffer cache (dg_capi_host.cpp) when the field dies
	void* d_producer_mask = nullptr; // the predicate mask of the producing launch (freed with the job / the field)
	mutable std::mutex host_mutex;   // guards host_job
	HostCopyJob* host_job = nullptr;
	bool immutable = false;          // dg_field_set_immutable: an attached array that will not change (K2 may build its copy)
	mutable ScratchPool flag_scratch;     // K3: the flag word k_field_check writes, one per launch in flight
	std::mutex wtab_mutex;
	std::map<double, void*> wtabs;        // K3: support radius -> immutable device table of 4096 kernel values
	mutable std::mutex copy_mutex;        // guards d_cell_major / d_tile_major / dev.{cell,tile}_major (built on demand by K2)
	mutable bool auto_copy_tried = false; // K2 built (or could not build) the cell-major copy of an owned field by itself
	dg_grid_desc grid;
	uint64_t n_coeffs = 0;
	uint64_t n_rows = 0; // rows of the cell table (= grid cells for an unreduced field)
	int device = -1;
};


// Every entry point that takes a handle runs on the handle's device, whatever device is current on the
// calling thread (the HIP current device is per thread and starts at 0: an OpenMP worker or any thread
// other than the one that called dg_set_device() would otherwise launch on device 0 with pointers into
// the handle's device).  The caller's current device is restored on return.
struct DeviceGuard
{
	int prev = -1;
	bool changed = false;
	hipError_t err = hipSuccess;
	explicit DeviceGuard(int want)
	{
		err = hipGetDevice(&prev);
		if (err == hipSuccess && want >= 0 && want != prev)
		{
			err = hipSetDevice(want);
			changed = err == hipSuccess;
		}
	}
	~DeviceGuard()
	{
		if (changed)
			(void)hipSetDevice(prev);
	}
	DeviceGuard(const DeviceGuard&) = delete;
	DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DG_ON_DEVICE_OF(handle)                                                                       \
	if ((handle)->device < 0)                                                                         \
		return fail(DG_ERR_NO_DEVICE, "the handle has no HIP device (a host-only mesh handle -- created without a device or under " \
									  "DG_FORCE_CPU=1 -- answers dg_signed_distance_point only; there is no CPU path for batches)"); \
	DeviceGuard device_guard_((handle)->device);                                                      \
	if (device_guard_.err != hipSuccess)                                                              \
		return fail(DG_ERR_HIP, "cannot switch to device %d: %s", (handle)->device, hipGetErrorString(device_guard_.err))

// ROCTx ranges around the launches of the device entry points (tracing): `rocprofv3 --marker-trace` shows
// "dg K1 sample_nodes" / "dg K2 interpolate" / "dg K3 density_map" / "dg U unpack" around the kernels they enqueue.
// The ROCTx library (the rocprofiler SDK's, else libroctx64) is bound at run time (no hard dependency; DG_ROCTX=0: off);
// without a tool attached a range costs a call.
struct TraceRange
{
	explicit TraceRange(const char* name);
	~TraceRange();
	TraceRange(const TraceRange&) = delete;
	TraceRange& operator=(const TraceRange&) = delete;
	bool on = false;
};
bool recycle_field_buffer(void* p, size_t bytes, int device); // dg_capi_host.cpp
void recycle_stream(int device, hipStream_t s);               // dg_capi_host.cpp: an idle stream for the next produced field
// collects the field's host copy job, if any (dg_capi_host.cpp); returns its status
dg_status finish_host_job(dg_field* field);
// makes `stream` wait for the kernels that produce the field's coefficients (no-op for ordinary fields)
inline hipError_t wait_produced(const dg_field* field, hipStream_t stream)
{
	if (field->produced && stream != field->producer_stream)
		return hipStreamWaitEvent(stream, field->produced, 0);
	return hipSuccess;
}

// ---- shared internals (defined in dg_capi.cpp) ---------------------------------------------------------------
extern thread_local std::string g_error;  // dg_last_error()
extern thread_local double g_last_ms;     // dg_last_kernel_ms()
dg_status fail(dg_status s, const char* fmt, ...);
dg_status require_device();
bool valid_grid(const dg_grid_desc* g);
int env_int(const char* name, int fallback, int lo, int hi); // a DOCUMENTED variable; test hooks / tuning: force_int (dg_force.h)
using dg::force_int;
uint32_t env_xcd_chunk();
// binning scratch of a K1p / K2 batch (dg_kernels.h: BinScratch): fills S from `pool`; *flag_host is the
// handle's pinned prediction word (allocated on first use).  Returns the pool index or -1 (no binning).
int acquire_bin_scratch(ScratchPool& pool, uint32_t** flag_host, const dg::TileGrid& tiles, uint64_t n, hipStream_t stream,
						dg::BinScratch& S);
// heavy-brick scratch of a K1 / K1p launch (dg_kernels.h: OverflowBuf)
int acquire_heavy_scratch(const dg_mesh* mesh, dg::SampleParams& P, hipStream_t stream);
void release_heavy_scratch(const dg_mesh* mesh, int idx, hipStream_t stream);

#define DG_HIP(call)                                                                                         \
	do                                                                                                       \
	{                                                                                                        \
		hipError_t e_ = (call);                                                                              \
		if (e_ != hipSuccess)                                                                                \
			return fail(DG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

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

