// dg_capi_comm.cpp -- the exchange step of the multi-GPU path behind the C ABI: an RCCL communicator
// handle (one process per GPU) and dg_sdf_sample_allgather_device, which samples this rank's shards,
// all-gathers them over xGMI and restores reference node order so that every GPU ends up holding the
// whole coefficient vector.  No reference counterpart (the reference is one OpenMP process).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first dg_comm_* call): single-GPU users of
// this library never load it, and a process that already carries an RCCL (PyTorch ships one) keeps
// exactly that one.
#include "dg_capi_internal.h"

#include <condition_variable>
#include <dlfcn.h>
#include <memory>
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
	decltype(&ncclBroadcast) Broadcast = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
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
		R.Broadcast = reinterpret_cast<decltype(R.Broadcast)>(dlsym(R.lib, "ncclBroadcast"));
		R.Send = reinterpret_cast<decltype(R.Send)>(dlsym(R.lib, "ncclSend"));
		R.Recv = reinterpret_cast<decltype(R.Recv)>(dlsym(R.lib, "ncclRecv"));
		R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(dlsym(R.lib, "ncclGroupStart"));
		R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(dlsym(R.lib, "ncclGroupEnd"));
		R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
		if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.GetErrorString || !R.Broadcast || !R.Send || !R.Recv ||
			!R.GroupStart || !R.GroupEnd)
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
	std::vector<hipEvent_t> t_begin, t_end;         // per piece: around this rank's sampling launch (timing enabled)
	int last_pieces = 0;
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
	// Communicator set-up is a collective: a rank that never arrives (a process that died, a wrong id) would block the
	// others forever.  The blocking ncclCommInitRank therefore runs on a helper thread and this call waits for it against
	// a deadline (DG_COMM_TIMEOUT_S seconds, default 180; 0: wait without limit); on expiry the call FAILS -- the helper
	// stays behind in the library call it cannot leave (the process is expected to give up) -- instead of hanging.  (A
	// non-blocking communicator would make every later collective non-blocking as well; the data path wants them blocking.)
	struct InitState
	{
		std::mutex m;
		std::condition_variable cv;
		bool done = false;
		ncclResult_t result = ncclSuccess;
		ncclComm_t comm = nullptr;
	};
	auto state = std::make_shared<InitState>();
	int device = 0;
	(void)hipGetDevice(&device);
	std::thread([state, R, nranks, u, rank, device]() {
		(void)hipSetDevice(device);
		ncclComm_t comm = nullptr;
		const ncclResult_t res = R->CommInitRank(&comm, nranks, u, rank);
		std::lock_guard<std::mutex> lock(state->m);
		state->result = res;
		state->comm = comm;
		state->done = true;
		state->cv.notify_all();
	}).detach();
	ncclResult_t r = ncclSuccess;
	{
		const int deadline_s = env_int("DG_COMM_TIMEOUT_S", 180, 0, 86400);
		std::unique_lock<std::mutex> lock(state->m);
		if (deadline_s > 0)
		{
			if (!state->cv.wait_for(lock, std::chrono::seconds(deadline_s), [&]() { return state->done; }))
			{
				delete c;
				return fail(DG_ERR_HIP, "communicator set-up did not complete within %d s (rank %d of %d): is every rank running?", deadline_s,
							rank, nranks);
			}
		}
		else
			state->cv.wait(lock, [&]() { return state->done; });
		r = state->result;
		c->comm = state->comm;
	}
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
	for (hipEvent_t e : c->t_begin) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->t_end) (void)hipEventDestroy(e);
	if (c->done) (void)hipEventDestroy(c->done);
	if (c->gather) (void)hipStreamDestroy(c->gather);
	if (c->unpack) (void)hipStreamDestroy(c->unpack);
	if (c->d_mine) (void)hipFree(c->d_mine);
	if (c->d_gathered) (void)hipFree(c->d_gathered);
	delete c;
}

static hipError_t piece_events(dg_comm* comm, int pieces)
{
	while ((int)comm->sampled.size() < pieces)
	{
		hipEvent_t a = nullptr;
		hipError_t e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
		if (e != hipSuccess) return e;
		comm->sampled.push_back(a);
		e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
		if (e != hipSuccess) return e;
		comm->gathered.push_back(a);
		e = hipEventCreate(&a);
		if (e != hipSuccess) return e;
		comm->t_begin.push_back(a);
		e = hipEventCreate(&a);
		if (e != hipSuccess) return e;
		comm->t_end.push_back(a);
	}
	return hipSuccess;
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
	DG_HIP(piece_events(comm, pieces));
	double* mine = static_cast<double*>(comm->d_mine);
	double* gathered = static_cast<double*>(comm->d_gathered);
	// Three streams form a pipeline: the caller's stream samples piece p + 1 while RCCL's stream gathers
	// piece p over xGMI and the unpack stream scatters piece p - 1 into reference node order.  (Buffers of
	// an earlier call are free again: the caller's stream waited for that call's last unpack.)
	for (int p = 0; p < pieces; ++p)
	{
		double* mp = mine + (size_t)p * stride;
		DG_HIP(hipEventRecord(comm->t_begin[(size_t)p], st));
		s = dg_sdf_sample_shard_device(mesh, grid, invert, p * N + comm->rank, vworld, mp, st);
		if (s != DG_OK)
			return s;
		DG_HIP(hipEventRecord(comm->t_end[(size_t)p], st));
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
	comm->last_pieces = pieces;
	return DG_OK;
}

dg_status dg_sdf_sample_exchange_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_comm* comm, int pieces,
										int flags, int root, const float* const plane_cost[4], double* d_field, void* stream)
{
	if ((flags & DG_EXCHANGE_INPLACE) == 0)
	{
		if (flags != 0)
			return fail(DG_ERR_INVALID, "DG_EXCHANGE_P2P / DG_EXCHANGE_TO_ROOT need DG_EXCHANGE_INPLACE");
		return dg_sdf_sample_allgather_device(mesh, grid, invert, comm, pieces, d_field, stream);
	}
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
	const bool to_root = (flags & DG_EXCHANGE_TO_ROOT) != 0, p2p = (flags & DG_EXCHANGE_P2P) != 0 || to_root;
	if (to_root && (root < 0 || root >= N))
		return fail(DG_ERR_INVALID, "root %d outside [0, %d)", root, N);
	pieces = std::max(1, std::min(pieces, dg::kMaxRanks / N));
	const int V = pieces * N;
	DG_ON_DEVICE_OF(mesh);
	hipStream_t st = static_cast<hipStream_t>(stream);
	uint32_t cuts[4][dg::kMaxRanks + 1];
	dg::chunk_planes(grid->resolution, V, plane_cost, cuts);
	dg::ClassGeom cg[4];
	dg::class_geometry(grid->resolution, cg);
	DG_HIP(piece_events(comm, pieces));
	// chunk v of class c: planes [cuts[c][v], cuts[c][v + 1]) = one contiguous run of d_field; owner v % N, piece v / N
	auto chunk_ptr = [&](int c, int v) { return d_field + cg[c].off + (uint64_t)cuts[c][v] * cg[c].D[0] * cg[c].D[1]; };
	auto chunk_len = [&](int c, int v) { return (size_t)(cuts[c][v + 1] - cuts[c][v]) * cg[c].D[0] * cg[c].D[1]; };
	for (int p = 0; p < pieces; ++p)
	{
		const int mine = p * N + comm->rank;
		uint32_t qb[4], qe[4];
		for (int c = 0; c < 4; ++c)
		{
			qb[c] = cuts[c][mine];
			qe[c] = cuts[c][mine + 1];
		}
		DG_HIP(hipEventRecord(comm->t_begin[(size_t)p], st));
		dg_status s = dg_sdf_sample_planes_device(mesh, grid, invert, qb, qe, d_field, st);
		if (s != DG_OK)
			return s;
		DG_HIP(hipEventRecord(comm->t_end[(size_t)p], st));
		DG_HIP(hipEventRecord(comm->sampled[(size_t)p], st));
		DG_HIP(hipStreamWaitEvent(comm->gather, comm->sampled[(size_t)p], 0));
		// piece p's exchange: grouped, on the communicator's stream, while this rank samples piece p + 1
		ncclResult_t r = R->GroupStart();
		for (int o = 0; o < N && r == ncclSuccess; ++o) // o: owner of the chunks that travel
		{
			const int v = p * N + o;
			for (int c = 0; c < 4 && r == ncclSuccess; ++c)
			{
				const size_t len = chunk_len(c, v);
				if (len == 0)
					continue;
				if (!p2p)
					r = R->Broadcast(chunk_ptr(c, v), chunk_ptr(c, v), len, ncclDouble, o, comm->comm, comm->gather);
				else if (o == comm->rank)
				{
					for (int dst = 0; dst < N && r == ncclSuccess; ++dst)
						if (dst != o && (!to_root || dst == root))
							r = R->Send(chunk_ptr(c, v), len, ncclDouble, dst, comm->comm, comm->gather);
				}
				else if (!to_root || comm->rank == root)
					r = R->Recv(chunk_ptr(c, v), len, ncclDouble, o, comm->comm, comm->gather);
			}
		}
		const ncclResult_t rg = R->GroupEnd();
		if (r == ncclSuccess)
			r = rg;
		if (r != ncclSuccess)
			return fail(DG_ERR_HIP, "in-place exchange of piece %d: %s", p, R->GetErrorString(r));
		DG_HIP(hipEventRecord(comm->gathered[(size_t)p], comm->gather));
	}
	DG_HIP(hipStreamWaitEvent(st, comm->gathered[(size_t)pieces - 1], 0)); // (one stream: the last piece's event covers all)
	comm->last_pieces = pieces;
	return DG_OK;
}

dg_status dg_comm_last_chunk_ms(dg_comm* comm, float* ms, int* n_pieces)
{
	if (!comm || !ms || !n_pieces)
		return fail(DG_ERR_INVALID, "null argument");
	DG_ON_DEVICE_OF(comm);
	const int n = std::min(*n_pieces, comm->last_pieces);
	for (int p = 0; p < n; ++p)
	{
		DG_HIP(hipEventSynchronize(comm->t_end[(size_t)p]));
		DG_HIP(hipEventElapsedTime(&ms[p], comm->t_begin[(size_t)p], comm->t_end[(size_t)p]));
	}
	*n_pieces = n;
	return DG_OK;
}

} // extern "C"
