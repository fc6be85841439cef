// dg_capi_hostfield.cpp -- the exchange step of the multi-GPU path that needs NEITHER collective kernels NOR device IPC:
// the coefficient vector is assembled in a POSIX shared-memory segment every rank of the node maps, each rank copies the
// chunks it sampled into their places with its own copy engine (hipMemcpyAsync D2H over its own PCIe link), and a barrier
// that lives in the segment itself says when the vector is whole.  This is SURVEY 8(e)'s "if only the host copy is needed,
// N independent D2H copies into disjoint ranges of m_nodes[field] avoid the collective entirely": what the reference's
// addFunction leaves behind is exactly that host vector (cubic_lagrange_discrete_grid.cpp:806-831).  Nothing here touches
// RCCL, HIP IPC or the virtual-memory API, so it is the form that still works when those do not.
// No reference counterpart (the reference is one process).
#include "dg_capi_internal.h"

#include <atomic>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace
{
constexpr uint64_t kMagic = 0x64675f686f737466ull; // "dg_hostf"
constexpr size_t kHeaderBytes = 4096;
struct ShmHeader // lives at the start of the segment; every member is address-free
{
	std::atomic<uint64_t> magic;     // set last by the creating rank
	uint64_t n_doubles;
	uint32_t nranks;
	uint32_t pad;
	std::atomic<uint32_t> arrived;    // barrier: ranks that have arrived in the current generation
	std::atomic<uint32_t> generation; // barrier: bumped by the last arrival
	std::atomic<uint32_t> attached;   // ranks that mapped the segment
	uint32_t pad2;
	uint64_t cuts_hash[2][dg::kMaxRanks]; // [call parity][rank]: the cuts every rank derived for the call (a rank is at most one call ahead)
};
static_assert(sizeof(ShmHeader) <= kHeaderBytes, "header page");
static_assert(std::atomic<uint32_t>::is_always_lock_free && std::atomic<uint64_t>::is_always_lock_free, "shared-memory atomics");
} // namespace

struct dg_host_field
{
	std::string name;
	int fd = -1;
	void* map = nullptr;
	size_t map_bytes = 0;
	ShmHeader* hdr = nullptr;
	double* data = nullptr;
	uint64_t n_doubles = 0;
	int rank = 0, nranks = 1, device = -1;
	bool registered = false; // hipHostRegister took the mapping: the copies are direct DMA
	uint64_t calls = 0;
	hipStream_t copy = nullptr;
	std::vector<hipEvent_t> sampled, t_begin, t_end;
	int last_pieces = 0;
	double timeout_s = 180.0;
};

static dg_status shm_barrier(dg_host_field* hf)
{
	if (hf->nranks <= 1)
		return DG_OK;
	ShmHeader* h = hf->hdr;
	const uint32_t gen = h->generation.load(std::memory_order_acquire);
	if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)hf->nranks)
	{
		h->arrived.store(0, std::memory_order_relaxed);
		h->generation.store(gen + 1, std::memory_order_release);
		return DG_OK;
	}
	const auto t0 = std::chrono::steady_clock::now();
	for (uint64_t spins = 0; h->generation.load(std::memory_order_acquire) == gen; ++spins)
	{
		if (spins < 2000)
			continue;
		(void)sched_yield();
		if ((spins & 1023) == 0 && hf->timeout_s > 0 &&
			std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > hf->timeout_s)
			return fail(DG_ERR_HIP, "shared-memory barrier: the other ranks did not arrive within %.0f s (rank %d of %d): is every rank running?",
						hf->timeout_s, hf->rank, hf->nranks);
	}
	return DG_OK;
}

extern "C"
{

dg_status dg_host_field_open(const char* name, uint64_t n_doubles, int rank, int nranks, dg_host_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!name || !*name || n_doubles == 0 || nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "bad name, size, rank or nranks (max %d ranks)", dg::kMaxRanks);
	dg_host_field* hf = new (std::nothrow) dg_host_field;
	if (!hf)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	hf->name = name[0] == '/' ? std::string(name) : "/" + std::string(name);
	hf->rank = rank;
	hf->nranks = nranks;
	hf->n_doubles = n_doubles;
	hf->timeout_s = (double)env_int("DG_COMM_TIMEOUT_S", 180, 0, 86400);
	hf->map_bytes = kHeaderBytes + (size_t)n_doubles * sizeof(double);
	const auto t0 = std::chrono::steady_clock::now();
	auto expired = [&]() { return hf->timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > hf->timeout_s; };
	if (rank == 0)
	{
		(void)shm_unlink(hf->name.c_str()); // (a segment a crashed job left behind)
		hf->fd = shm_open(hf->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
		// (posix_fallocate: the pages are claimed NOW -- a tmpfs that is too small answers ENOSPC here instead of SIGBUS at the
		// first touch of a page it cannot back)
		int falloc = 0;
		if (hf->fd < 0 || ftruncate(hf->fd, (off_t)hf->map_bytes) != 0 || (falloc = posix_fallocate(hf->fd, 0, (off_t)hf->map_bytes)) != 0)
		{
			const int err = falloc != 0 ? falloc : errno;
			if (hf->fd >= 0)
			{
				(void)close(hf->fd);
				(void)shm_unlink(hf->name.c_str());
			}
			delete hf;
			return fail(DG_ERR_ALLOC, "shared-memory segment %s of %.2f GB: %s", name, (double)n_doubles * 8e-9, std::strerror(err));
		}
	}
	else
	{
		// the creating rank may be later than this one: wait for the segment to exist at its full size
		while (true)
		{
			hf->fd = shm_open(hf->name.c_str(), O_RDWR, 0600);
			struct stat st;
			if (hf->fd >= 0 && fstat(hf->fd, &st) == 0 && (size_t)st.st_size == hf->map_bytes)
				break;
			if (hf->fd >= 0)
				(void)close(hf->fd);
			hf->fd = -1;
			if (expired())
			{
				delete hf;
				return fail(DG_ERR_HIP, "shared-memory segment %s did not appear within the deadline (rank %d of %d): is rank 0 running?", name, rank, nranks);
			}
			(void)usleep(2000);
		}
	}
	hf->map = mmap(nullptr, hf->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, hf->fd, 0);
	if (hf->map == MAP_FAILED)
	{
		const int err = errno;
		hf->map = nullptr;
		(void)close(hf->fd);
		if (rank == 0)
			(void)shm_unlink(hf->name.c_str());
		delete hf;
		return fail(DG_ERR_ALLOC, "mapping the shared-memory segment %s: %s", name, std::strerror(err));
	}
	hf->hdr = static_cast<ShmHeader*>(hf->map);
	hf->data = reinterpret_cast<double*>(static_cast<char*>(hf->map) + kHeaderBytes);
	if (rank == 0)
	{
		hf->hdr->n_doubles = n_doubles; // (fresh pages are zero: counters and hashes start at 0)
		hf->hdr->nranks = (uint32_t)nranks;
		hf->hdr->magic.store(kMagic, std::memory_order_release);
	}
	else
	{
		while (hf->hdr->magic.load(std::memory_order_acquire) != kMagic)
		{
			if (expired())
			{
				dg_host_field_close(hf);
				return fail(DG_ERR_HIP, "shared-memory segment %s was never initialised by rank 0", name);
			}
			(void)usleep(1000);
		}
		if (hf->hdr->n_doubles != n_doubles || hf->hdr->nranks != (uint32_t)nranks)
		{
			dg_host_field_close(hf);
			return fail(DG_ERR_INVALID, "shared-memory segment %s was created for another size or rank count", name);
		}
	}
	hf->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
	// with a device: the mapping becomes a DMA target (without one -- CPU-only tests of the barrier -- it stays plain memory)
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 && hipGetDevice(&hf->device) == hipSuccess)
	{
		if (hipHostRegister(hf->data, (size_t)n_doubles * sizeof(double), hipHostRegisterDefault) == hipSuccess)
			hf->registered = true;
		else
			(void)hipGetLastError(); // (pageable copies still work, through the runtime's staging buffers)
		if (hipStreamCreateWithFlags(&hf->copy, hipStreamNonBlocking) != hipSuccess)
		{
			(void)hipGetLastError();
			hf->copy = nullptr;
		}
	}
	else
	{
		(void)hipGetLastError();
		hf->device = -1;
	}
	// everybody has mapped the segment: its name can go (the memory lives until the last rank unmaps it)
	const dg_status s = shm_barrier(hf);
	if (rank == 0)
		(void)shm_unlink(hf->name.c_str());
	if (s != DG_OK)
	{
		dg_host_field_close(hf);
		return s;
	}
	*out = hf;
	return DG_OK;
}

double* dg_host_field_data(dg_host_field* hf) { return hf ? hf->data : nullptr; }

dg_status dg_host_field_barrier(dg_host_field* hf)
{
	if (!hf)
		return fail(DG_ERR_INVALID, "null argument");
	return shm_barrier(hf);
}

void dg_host_field_close(dg_host_field* hf)
{
	if (!hf)
		return;
	{
		DeviceGuard guard(hf->device);
		if (hf->copy)
		{
			(void)hipStreamSynchronize(hf->copy);
			(void)hipStreamDestroy(hf->copy);
		}
		for (hipEvent_t e : hf->sampled) (void)hipEventDestroy(e);
		for (hipEvent_t e : hf->t_begin) (void)hipEventDestroy(e);
		for (hipEvent_t e : hf->t_end) (void)hipEventDestroy(e);
		if (hf->registered)
			(void)hipHostUnregister(hf->data);
	}
	if (hf->map)
		(void)munmap(hf->map, hf->map_bytes);
	if (hf->fd >= 0)
		(void)close(hf->fd);
	delete hf;
}

dg_status dg_host_field_get_info(dg_host_field* hf, dg_host_field_info* info)
{
	if (!hf || !info)
		return fail(DG_ERR_INVALID, "null argument");
	info->rank = hf->rank;
	info->nranks = hf->nranks;
	info->device = hf->device;
	info->registered = hf->registered ? 1 : 0;
	info->n_doubles = hf->n_doubles;
	return DG_OK;
}

dg_status dg_sdf_sample_to_host_field(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_host_field* hf, int pieces,
									  const float* const plane_cost[4], double* d_field, void* stream)
{
	if (!mesh || !grid || !hf || !d_field)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (dg_grid_n_nodes(grid) != hf->n_doubles)
		return fail(DG_ERR_INVALID, "the grid has %llu nodes, the shared vector %llu", (unsigned long long)dg_grid_n_nodes(grid),
					(unsigned long long)hf->n_doubles);
	if (mesh->device != hf->device || !hf->copy)
		return fail(DG_ERR_INVALID, "mesh lives on device %d, the shared vector was opened with device %d current", mesh->device, hf->device);
	DG_ON_DEVICE_OF(mesh);
	hipStream_t st = static_cast<hipStream_t>(stream);
	const int N = hf->nranks;
	pieces = std::max(1, std::min(pieces, dg::kMaxRanks / N));
	const int V = pieces * N;
	uint32_t cuts[4][dg::kMaxRanks + 1];
	dg::chunk_planes(grid->resolution, V, plane_cost, cuts);
	// this rank's view of the cuts, for the peers to compare after the barrier (FNV-1a as dg_capi_comm.cpp)
	uint64_t h = 1469598103934665603ull;
	for (int c = 0; c < 4; ++c)
		for (int v = 0; v <= V; ++v)
		{
			h ^= cuts[c][v];
			h *= 1099511628211ull;
		}
	const int parity = (int)(hf->calls & 1u);
	++hf->calls;
	hf->hdr->cuts_hash[parity][hf->rank] = h;
	dg::ClassGeom cg[4];
	dg::class_geometry(grid->resolution, cg);
	while ((int)hf->sampled.size() < pieces)
	{
		hipEvent_t a = nullptr;
		DG_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
		hf->sampled.push_back(a);
		DG_HIP(hipEventCreate(&a));
		hf->t_begin.push_back(a);
		DG_HIP(hipEventCreate(&a));
		hf->t_end.push_back(a);
	}
	for (int p = 0; p < pieces; ++p)
	{
		const int v = p * N + hf->rank;
		uint32_t qb[4], qe[4];
		for (int c = 0; c < 4; ++c)
		{
			qb[c] = cuts[c][v];
			qe[c] = cuts[c][v + 1];
		}
		DG_HIP(hipEventRecord(hf->t_begin[(size_t)p], st));
		const dg_status s = dg_sdf_sample_planes_device(mesh, grid, invert, qb, qe, d_field, st);
		if (s != DG_OK)
			return s;
		DG_HIP(hipEventRecord(hf->t_end[(size_t)p], st));
		DG_HIP(hipEventRecord(hf->sampled[(size_t)p], st));
		// piece p goes to the host while piece p + 1 is sampled
		DG_HIP(hipStreamWaitEvent(hf->copy, hf->sampled[(size_t)p], 0));
		for (int c = 0; c < 4; ++c)
		{
			const uint64_t off = cg[c].off + (uint64_t)cuts[c][v] * cg[c].D[0] * cg[c].D[1];
			const size_t len = (size_t)(cuts[c][v + 1] - cuts[c][v]) * cg[c].D[0] * cg[c].D[1] * sizeof(double);
			if (len)
				DG_HIP(hipMemcpyAsync(hf->data + off, d_field + off, len, hipMemcpyDeviceToHost, hf->copy));
		}
	}
	hf->last_pieces = pieces;
	DG_HIP(hipStreamSynchronize(hf->copy)); // this rank's chunks are in the shared vector ...
	const dg_status bs = shm_barrier(hf);      // ... and so are everybody else's
	if (bs != DG_OK)
		return bs;
	for (int r = 0; r < N; ++r)
		if (hf->hdr->cuts_hash[parity][r] != hf->hdr->cuts_hash[parity][0])
			return fail(DG_ERR_INVALID, "rank %d cut the lattice differently from rank 0: plane_cost must hold the same values on every rank (this is rank %d)",
						r, hf->rank);
	return DG_OK;
}

dg_status dg_host_field_last_chunk_ms(dg_host_field* hf, float* ms, int* n_pieces)
{
	if (!hf || !ms || !n_pieces)
		return fail(DG_ERR_INVALID, "null argument");
	DG_ON_DEVICE_OF(hf);
	const int n = std::min(*n_pieces, hf->last_pieces);
	for (int p = 0; p < n; ++p)
	{
		DG_HIP(hipEventSynchronize(hf->t_end[(size_t)p]));
		DG_HIP(hipEventElapsedTime(&ms[p], hf->t_begin[(size_t)p], hf->t_end[(size_t)p]));
	}
	*n_pieces = n;
	return DG_OK;
}

} // extern "C"
