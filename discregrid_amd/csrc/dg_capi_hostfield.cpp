// dg_capi_hostfield.cpp -- the exchange step of the multi-GPU path that needs NEITHER collective kernels NOR device IPC:
// the coefficient vector is assembled in a POSIX shared-memory segment every rank of the node maps, each rank copies the
// chunks it sampled into their places with its own copy engine (hipMemcpyAsync D2H over its own PCIe link), and a barrier
// that lives in the segment itself says when the vector is whole.  This is SURVEY 8(e)'s "if only the host copy is needed,
// N independent D2H copies into disjoint ranges of m_nodes[field] avoid the collective entirely": what the reference's
// addFunction leaves behind is exactly that host vector (cubic_lagrange_discrete_grid.cpp:806-831).  Nothing here touches
// RCCL, HIP IPC or the virtual-memory API, so it is the form that still works when those do not.
// No reference counterpart (the reference is one process).
#include "dg_capi_shm.h"

namespace
{
constexpr uint32_t kKindHostField = 1;
// the owner's words of the segment header: [call parity][rank] the cuts every rank derived for the call (a rank is at most one call ahead)
inline uint64_t& cuts_hash(dgshm::Segment& seg, int parity, int rank) { return seg.hdr->user[(size_t)parity * dg::kMaxRanks + (size_t)rank]; }
static_assert(2 * dg::kMaxRanks <= 496, "header words");
} // namespace

struct dg_host_field
{
	dgshm::Segment seg;
	double* data = nullptr;
	uint64_t n_doubles = 0;
	int rank = 0, nranks = 1, device = -1;
	bool registered = false; // hipHostRegister took the mapping: the copies are direct DMA
	uint64_t calls = 0;
	hipStream_t copy = nullptr;
	std::vector<hipEvent_t> sampled, t_begin, t_end;
	int last_pieces = 0;
};

static dg_status shm_barrier(dg_host_field* hf) { return dgshm::barrier(hf->seg); }

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
	hf->rank = rank;
	hf->nranks = nranks;
	hf->n_doubles = n_doubles;
	const dg_status os = dgshm::open(hf->seg, name, (size_t)n_doubles * sizeof(double), kKindHostField, rank, nranks);
	if (os != DG_OK)
	{
		delete hf;
		return os;
	}
	hf->data = reinterpret_cast<double*>(hf->seg.payload);
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
	dgshm::close(hf->seg);
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
	cuts_hash(hf->seg, parity, hf->rank) = h;
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
		// Before this call's FIRST copy into the shared vector: every rank has entered the call, i.e. is done reading the result of
		// the previous one (a fast rank would otherwise overwrite what a slower rank is still reading).  The first piece is being
		// sampled meanwhile: the wait hides behind it.
		if (p == 0 && N > 1)
		{
			const dg_status es = shm_barrier(hf);
			if (es != DG_OK)
				return es;
		}
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
		if (cuts_hash(hf->seg, parity, r) != cuts_hash(hf->seg, parity, 0))
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
