// dg_capi_comm.cpp -- the exchange step of the multi-GPU path behind the C ABI: an RCCL communicator
// handle (one process per GPU) and dg_sdf_sample_allgather_device, which samples this rank's shards,
// all-gathers them over xGMI and restores reference node order so that every GPU ends up holding the
// whole coefficient vector.  No reference counterpart (the reference is one OpenMP process).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first dg_comm_* call): single-GPU users of
// this library never load it, and a process that already carries an RCCL (PyTorch ships one) keeps
// exactly that one.
#include "dg_capi_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace
{
struct Rccl
{
	void* lib = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string error;
};
Rccl* rccl()
{
	static Rccl R;
	static std::once_flag once;
	std::call_once(once, []() {
		const char* names[] = {"librccl.so.1", "librccl.so"};
		for (const char* n : names)
			if ((R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr)
				break;
		if (!R.lib)
		{
			R.error = std::string("cannot load librccl.so.1: ") + dlerror();
			return;
		}
		R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.lib, "ncclGetUniqueId"));
		R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.lib, "ncclCommInitRank"));
		R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
		R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.lib, "ncclAllGather"));
		R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
		if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.GetErrorString)
			R.error = "librccl.so.1 lacks an expected symbol";
	});
	return R.error.empty() ? &R : nullptr;
}
dg_status rccl_unavailable()
{
	static Rccl dummy;
	(void)dummy;
	return fail(DG_ERR_HIP, "RCCL unavailable");
}
} // namespace

struct dg_comm
{
	ncclComm_t comm = nullptr;
	bool owned = false;
	int rank = 0, nranks = 1, device = -1;
	hipStream_t gather = nullptr, unpack = nullptr; // the exchange and the unpack run beside the caller's stream
	std::vector<hipEvent_t> sampled, gathered;      // per piece
	hipEvent_t done = nullptr;
	void* d_mine = nullptr;     // this rank's packed pieces
	void* d_gathered = nullptr; // all ranks' pieces, the buffer one (pieces * nranks)-rank all-gather would produce
	size_t mine_bytes = 0, gathered_bytes = 0;
};

static_assert(DG_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C"
{

dg_status dg_comm_unique_id(uint8_t id[DG_UNIQUE_ID_BYTES])
{
	if (!id)
		return fail(DG_ERR_INVALID, "null argument");
	Rccl* R = rccl();
	if (!R)
		return rccl_unavailable();
	ncclUniqueId u;
	const ncclResult_t r = R->GetUniqueId(&u);
	if (r != ncclSuccess)
		return fail(DG_ERR_HIP, "ncclGetUniqueId: %s", R->GetErrorString(r));
	std::memcpy(id, u.internal, DG_UNIQUE_ID_BYTES);
	return DG_OK;
}

static dg_status comm_finish_setup(dg_comm* c)
{
	hipError_t e = hipGetDevice(&c->device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->gather, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->unpack, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
	if (e != hipSuccess)
		return fail(DG_ERR_HIP, "dg_comm: %s", hipGetErrorString(e));
	return DG_OK;
}

dg_status dg_comm_create(const uint8_t id[DG_UNIQUE_ID_BYTES], int rank, int nranks, dg_comm** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!id || nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "rank %d / nranks %d out of range (max %d ranks)", rank, nranks, dg::kMaxRanks);
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	Rccl* R = rccl();
	if (!R)
		return rccl_unavailable();
	dg_comm* c = new (std::nothrow) dg_comm;
	if (!c)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	ncclUniqueId u;
	std::memcpy(u.internal, id, DG_UNIQUE_ID_BYTES);
	const ncclResult_t r = R->CommInitRank(&c->comm, nranks, u, rank);
	if (r != ncclSuccess)
	{
		delete c;
		return fail(DG_ERR_HIP, "ncclCommInitRank: %s", R->GetErrorString(r));
	}
	c->owned = true;
	c->rank = rank;
	c->nranks = nranks;
	s = comm_finish_setup(c);
	if (s != DG_OK)
	{
		dg_comm_destroy(c);
		return s;
	}
	*out = c;
	return DG_OK;
}

dg_status dg_comm_adopt(void* nccl_comm, int rank, int nranks, dg_comm** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!nccl_comm || nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "bad communicator, rank or nranks");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	if (!rccl())
		return rccl_unavailable();
	dg_comm* c = new (std::nothrow) dg_comm;
	if (!c)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	c->comm = static_cast<ncclComm_t>(nccl_comm);
	c->rank = rank;
	c->nranks = nranks;
	s = comm_finish_setup(c);
	if (s != DG_OK)
	{
		dg_comm_destroy(c);
		return s;
	}
	*out = c;
	return DG_OK;
}

void dg_comm_destroy(dg_comm* c)
{
	if (!c)
		return;
	DeviceGuard guard(c->device);
	if (c->gather) (void)hipStreamSynchronize(c->gather);
	if (c->unpack) (void)hipStreamSynchronize(c->unpack);
	if (c->owned && c->comm)
		if (Rccl* R = rccl())
			(void)R->CommDestroy(c->comm);
	for (hipEvent_t e : c->sampled) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->gathered) (void)hipEventDestroy(e);
	if (c->done) (void)hipEventDestroy(c->done);
	if (c->gather) (void)hipStreamDestroy(c->gather);
	if (c->unpack) (void)hipStreamDestroy(c->unpack);
	if (c->d_mine) (void)hipFree(c->d_mine);
	if (c->d_gathered) (void)hipFree(c->d_gathered);
	delete c;
}

dg_status dg_sdf_sample_allgather_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_comm* comm, int pieces,
										 double* d_field, void* stream)
{
	if (!mesh || !grid || !comm || !d_field)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	Rccl* R = rccl();
	if (!R)
		return rccl_unavailable();
	if (mesh->device != comm->device)
		return fail(DG_ERR_INVALID, "mesh lives on device %d, the communicator on device %d", mesh->device, comm->device);
	const int N = comm->nranks;
	pieces = std::max(1, std::min(pieces, dg::kMaxRanks / N)); // the shard layout handles kMaxRanks virtual ranks
	const int vworld = pieces * N;
	DG_ON_DEVICE_OF(mesh);
	hipStream_t st = static_cast<hipStream_t>(stream);

	// Piece p of this rank = the shard of virtual rank p * N + rank in a (pieces * N)-way deal of the 4-plane
	// slabs: all virtual ranks share one slot size, so the `pieces` gathered pieces, laid end to end, are
	// exactly the buffer ONE all-gather among pieces * N ranks would produce and the unpack is unchanged.
	dg_shard_info si;
	dg_status s = dg_shard_layout(grid, comm->rank, vworld, &si);
	if (s != DG_OK)
		return s;
	const uint64_t stride = si.stride;
	const size_t mine_bytes = (size_t)pieces * stride * sizeof(double);
	const size_t gathered_bytes = (size_t)vworld * stride * sizeof(double);
	if (mine_bytes > comm->mine_bytes || gathered_bytes > comm->gathered_bytes)
	{
		DG_HIP(hipStreamSynchronize(comm->gather));
		DG_HIP(hipStreamSynchronize(comm->unpack));
		if (comm->d_mine) (void)hipFree(comm->d_mine);
		if (comm->d_gathered) (void)hipFree(comm->d_gathered);
		comm->d_mine = comm->d_gathered = nullptr;
		comm->mine_bytes = comm->gathered_bytes = 0;
		DG_HIP(hipMalloc(&comm->d_mine, mine_bytes));
		DG_HIP(hipMalloc(&comm->d_gathered, gathered_bytes));
		DG_HIP(hipMemsetAsync(comm->d_mine, 0, mine_bytes, st)); // slot padding is exchanged as well: keep it defined
		comm->mine_bytes = mine_bytes;
		comm->gathered_bytes = gathered_bytes;
	}
	while ((int)comm->sampled.size() < pieces)
	{
		hipEvent_t a = nullptr, b = nullptr;
		DG_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
		comm->sampled.push_back(a);
		DG_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
		comm->gathered.push_back(b);
	}
	double* mine = static_cast<double*>(comm->d_mine);
	double* gathered = static_cast<double*>(comm->d_gathered);
	// Three streams form a pipeline: the caller's stream samples piece p + 1 while RCCL's stream gathers
	// piece p over xGMI and the unpack stream scatters piece p - 1 into reference node order.  (Buffers of
	// an earlier call are free again: the caller's stream waited for that call's last unpack.)
	for (int p = 0; p < pieces; ++p)
	{
		double* mp = mine + (size_t)p * stride;
		s = dg_sdf_sample_shard_device(mesh, grid, invert, p * N + comm->rank, vworld, mp, st);
		if (s != DG_OK)
			return s;
		DG_HIP(hipEventRecord(comm->sampled[(size_t)p], st));
		DG_HIP(hipStreamWaitEvent(comm->gather, comm->sampled[(size_t)p], 0));
		const ncclResult_t r = R->AllGather(mp, gathered + (size_t)p * N * stride, (size_t)stride, ncclDouble, comm->comm, comm->gather);
		if (r != ncclSuccess)
			return fail(DG_ERR_HIP, "ncclAllGather: %s", R->GetErrorString(r));
		DG_HIP(hipEventRecord(comm->gathered[(size_t)p], comm->gather));
		DG_HIP(hipStreamWaitEvent(comm->unpack, comm->gathered[(size_t)p], 0));
		s = dg_unpack_shard_range_device(grid, vworld, gathered, stride, p * N, (p + 1) * N, d_field, comm->unpack);
		if (s != DG_OK)
			return s;
	}
	DG_HIP(hipEventRecord(comm->done, comm->unpack));
	DG_HIP(hipStreamWaitEvent(st, comm->done, 0)); // the field is complete in the order of the caller's stream
	return DG_OK;
}

} // extern "C"
