// dg_capi_comm.cpp -- the exchange step of the multi-GPU path behind the C ABI: an RCCL communicator
// handle (one process per GPU) and dg_sdf_sample_allgather_device, which samples this rank's shards,
// all-gathers them over xGMI and restores reference node order so that every GPU ends up holding the
// whole coefficient vector.  No reference counterpart (the reference is one OpenMP process).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first dg_comm_* call): single-GPU users of
// this library never load it, and a process that already carries an RCCL (PyTorch ships one) keeps
// exactly that one.
#include "dg_capi_internal.h"
#include "dg_capi_shm.h"
#include "dg_capi_vmm.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <dlfcn.h>
#include <map>
#include <memory>
#include <rccl/rccl.h>
#include <unistd.h>

namespace
{
struct Rccl
{
	void* lib = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclCommCount) CommCount = nullptr;
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
		R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(dlsym(R.lib, "ncclAllReduce"));
		R.CommCount = reinterpret_cast<decltype(R.CommCount)>(dlsym(R.lib, "ncclCommCount"));
		R.Send = reinterpret_cast<decltype(R.Send)>(dlsym(R.lib, "ncclSend"));
		R.Recv = reinterpret_cast<decltype(R.Recv)>(dlsym(R.lib, "ncclRecv"));
		R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(dlsym(R.lib, "ncclGroupStart"));
		R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(dlsym(R.lib, "ncclGroupEnd"));
		R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
		if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.GetErrorString || !R.Broadcast || !R.Send || !R.Recv ||
			!R.GroupStart || !R.GroupEnd || !R.AllReduce || !R.CommCount)
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

// one peer's view of a registered field: the allocation its process exported, opened here
struct PeerField
{
	std::vector<char*> base;   // [nranks]: peer r's d_field in THIS process' address space (own rank: the local pointer)
	std::vector<void*> opened; // what hipIpcOpenMemHandle returned (closed with the communicator)
	uint64_t bytes = 0;
};
struct IpcRecord // what travels through the control plane when a field is registered
{
	hipIpcMemHandle_t handle; // kind 0: the whole allocation through HIP IPC
	uint64_t offset; // of d_field inside the exported allocation / mapped range
	uint64_t bytes;
	uint64_t alloc_bytes; // size of that allocation / range
	uint64_t chunk;       // kind 1: size of every chunk but the last
	uint64_t serial;      // kind 1: names the unix socket that serves the chunks' descriptors ("dg_vmm_<pid>_<64 random bits>")
	int32_t device;
	int32_t pid;
	int32_t kind;   // 0: hipIpcMemHandle_t, 1: an array of dg_comm_field_alloc (hipMemCreate chunks, one descriptor each)
	int32_t status; // 0: exported; otherwise the rank could not export its field and EVERY rank refuses
	uint32_t n_chunks;
	uint32_t pad;
};
enum
{
	kExportOk = 0,
	kExportNoRange = 1,  // hipMemGetAddressRange failed (not a hipMalloc allocation?)
	kExportNoHandle = 2, // hipIpcGetMemHandle failed
	kExportNoFds = 3,    // hipMemExportToShareableHandle failed
	kExportNoSocket = 4, // the descriptor server could not be started
};
struct dg_comm
{
	ncclComm_t comm = nullptr;
	bool owned = false;
	// control plane without RCCL (dg_comm_create_external): small host-blocking collectives supplied by the caller
	dg_comm_allgather_fn ext_allgather = nullptr;
	dg_comm_barrier_fn ext_barrier = nullptr;
	void* ext_user = nullptr;
	dgshm::Segment* shm = nullptr; // dg_comm_create_shm: the control plane lives in a shared-memory segment (ext_* point at it)
	// DG_EXCHANGE_COPY: one copy stream per peer (the chunks are pushed by the copy engines), the fields whose peers
	// are known, a device word for the stream-ordered barriers of the RCCL control plane
	std::vector<hipStream_t> copy_streams;
	std::vector<hipEvent_t> copy_done;
	std::map<const void*, PeerField> peer_fields;
	int* d_token = nullptr; // device words of the control plane: [0, 256) barrier tokens, [256, kCtrlBytes) small all-gathers
	hipEvent_t entered = nullptr;
	std::map<char*, dgvmm::Array> vmm_owned;  // dg_comm_field_alloc: base -> array
	std::vector<dgvmm::Array> vmm_imported;   // peers' arrays mapped here
	hipEvent_t t_last_sampled = nullptr, t_complete = nullptr; // timing: end of this rank's last sampling launch / field complete
	bool wait_timed = false;
	std::map<std::string, void*> opened_handles; // (pid, handle) -> mapping: an allocation is opened once however many fields live in it
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
constexpr size_t kCtrlBytes = 16384;

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
	if (e == hipSuccess) e = hipEventCreateWithFlags(&c->entered, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreate(&c->t_last_sampled);
	if (e == hipSuccess) e = hipEventCreate(&c->t_complete);
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_token), kCtrlBytes);
	if (e == hipSuccess) e = hipMemset(c->d_token, 0, kCtrlBytes);
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

dg_status dg_comm_create_external(int rank, int nranks, dg_comm_allgather_fn allgather, dg_comm_barrier_fn barrier, void* user,
								  dg_comm** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!allgather || !barrier || nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "null callback, or rank %d / nranks %d out of range (max %d ranks)", rank, nranks, dg::kMaxRanks);
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	dg_comm* c = new (std::nothrow) dg_comm;
	if (!c)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	c->rank = rank;
	c->nranks = nranks;
	c->ext_allgather = allgather;
	c->ext_barrier = barrier;
	c->ext_user = user;
	s = comm_finish_setup(c);
	if (s != DG_OK)
	{
		dg_comm_destroy(c);
		return s;
	}
	*out = c;
	return DG_OK;
}

// ---- a control plane in shared memory: the copy form without RCCL and without callbacks -------------------------------------
namespace
{
constexpr uint32_t kKindCtrl = 2;
constexpr size_t kCtrlSlot = 256; // bytes per rank and message (the registration record is the largest: 136 bytes)
int shm_ctrl_allgather(const void* mine, void* all, size_t bytes, void* user)
{
	dgshm::Segment* seg = static_cast<dgshm::Segment*>(user);
	if (bytes > kCtrlSlot)
		return 1;
	std::memcpy(seg->payload + (size_t)seg->rank * kCtrlSlot, mine, bytes);
	if (dgshm::barrier(*seg) != DG_OK) // everybody has written
		return 1;
	for (int r = 0; r < seg->nranks; ++r)
		std::memcpy(static_cast<char*>(all) + (size_t)r * bytes, seg->payload + (size_t)r * kCtrlSlot, bytes);
	return dgshm::barrier(*seg) == DG_OK ? 0 : 1; // everybody has read: the slots may be written again
}
int shm_ctrl_barrier(void* user) { return dgshm::barrier(*static_cast<dgshm::Segment*>(user)) == DG_OK ? 0 : 1; }
} // namespace

dg_status dg_comm_create_shm(const char* name, int rank, int nranks, dg_comm** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!name || !*name || nranks < 1 || nranks > dg::kMaxRanks || rank < 0 || rank >= nranks)
		return fail(DG_ERR_INVALID, "bad name, or rank %d / nranks %d out of range (max %d ranks)", rank, nranks, dg::kMaxRanks);
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	dgshm::Segment* seg = new (std::nothrow) dgshm::Segment;
	if (!seg)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	s = dgshm::open(*seg, name, (size_t)dg::kMaxRanks * kCtrlSlot, kKindCtrl, rank, nranks);
	if (s != DG_OK)
	{
		delete seg;
		return s;
	}
	s = dg_comm_create_external(rank, nranks, shm_ctrl_allgather, shm_ctrl_barrier, seg, out);
	if (s != DG_OK)
	{
		dgshm::close(*seg);
		delete seg;
		return s;
	}
	(*out)->shm = seg;
	return DG_OK;
}

dg_status dg_comm_get_info(dg_comm* comm, dg_comm_info* info)
{
	if (!comm || !info)
		return fail(DG_ERR_INVALID, "null argument");
	info->rank = comm->rank;
	info->nranks = comm->nranks;
	info->device = comm->device;
	info->rccl_nranks = -1;
	info->registered_fields = (int32_t)comm->peer_fields.size();
	if (comm->comm)
	{
		Rccl* R = rccl();
		int n = 0;
		if (R && R->CommCount(comm->comm, &n) == ncclSuccess)
			info->rccl_nranks = n;
	}
	return DG_OK;
}

void dg_comm_destroy(dg_comm* c)
{
	if (!c)
		return;
	DeviceGuard guard(c->device); // (device < 0 -- set-up failed before a device was known -- leaves the current device alone)
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
	for (hipStream_t cs : c->copy_streams)
		if (cs)
		{
			(void)hipStreamSynchronize(cs);
			(void)hipStreamDestroy(cs);
		}
	for (hipEvent_t e : c->copy_done)
		if (e) (void)hipEventDestroy(e);
	for (auto& kv : c->opened_handles)
		(void)hipIpcCloseMemHandle(kv.second);
	if (c->shm)
	{
		dgshm::close(*c->shm);
		delete c->shm;
	}
	for (dgvmm::Array& a : c->vmm_imported)
		dgvmm::destroy(a);
	for (auto& kv : c->vmm_owned)
		dgvmm::destroy(kv.second);
	if (c->entered) (void)hipEventDestroy(c->entered);
	if (c->t_last_sampled) (void)hipEventDestroy(c->t_last_sampled);
	if (c->t_complete) (void)hipEventDestroy(c->t_complete);
	if (c->d_token) (void)hipFree(c->d_token);
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
	if (!comm->comm)
		return fail(DG_ERR_INVALID, "a communicator with an external control plane runs DG_EXCHANGE_INPLACE | DG_EXCHANGE_COPY only");
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
	DG_HIP(hipEventRecord(comm->t_last_sampled, st));
	DG_HIP(hipEventRecord(comm->done, comm->unpack));
	DG_HIP(hipStreamWaitEvent(st, comm->done, 0)); // the field is complete in the order of the caller's stream
	DG_HIP(hipEventRecord(comm->t_complete, st));
	comm->wait_timed = true;
	comm->last_pieces = pieces;
	return DG_OK;
}

// ---- the control plane (small, host-blocking collectives) ---------------------------------------------------------------
static dg_status ctrl_allgather(dg_comm* c, const void* mine, void* all, size_t bytes)
{
	if (c->ext_allgather)
	{
		if (c->ext_allgather(mine, all, bytes, c->ext_user) != 0)
			return fail(DG_ERR_HIP, "the caller's allgather failed");
		return DG_OK;
	}
	Rccl* R = rccl();
	if (!R || !c->comm)
		return rccl_unavailable();
	// small messages (the hash of the cuts on every exchange call) go through the communicator's own device words
	const size_t need = ((bytes + 255) / 256 * 256) + bytes * (size_t)c->nranks;
	void* d = nullptr;
	const bool own = need > kCtrlBytes - 256;
	if (own)
		DG_HIP(hipMalloc(&d, bytes * (size_t)(c->nranks + 1) + 256));
	else
		d = reinterpret_cast<char*>(c->d_token) + 256;
	char* dall = static_cast<char*>(d) + (bytes + 255) / 256 * 256;
	hipError_t e = hipMemcpyAsync(d, mine, bytes, hipMemcpyHostToDevice, c->gather);
	ncclResult_t r = ncclSuccess;
	if (e == hipSuccess)
		r = R->AllGather(d, dall, bytes, ncclUint8, c->comm, c->gather);
	if (e == hipSuccess && r == ncclSuccess)
		e = hipMemcpyAsync(all, dall, bytes * (size_t)c->nranks, hipMemcpyDeviceToHost, c->gather);
	if (e == hipSuccess)
		e = hipStreamSynchronize(c->gather);
	if (own)
		(void)hipFree(d);
	if (r != ncclSuccess)
		return fail(DG_ERR_HIP, "ncclAllGather (control plane): %s", R->GetErrorString(r));
	DG_HIP(e);
	return DG_OK;
}
// one word all-reduced on the communicator's stream: stream-ordered, a few microseconds of one CU
static dg_status rccl_stream_barrier(dg_comm* c)
{
	Rccl* R = rccl();
	if (!R || !c->comm)
		return rccl_unavailable();
	const ncclResult_t r = R->AllReduce(c->d_token, c->d_token + 16, 1, ncclInt, ncclSum, c->comm, c->gather);
	if (r != ncclSuccess)
		return fail(DG_ERR_HIP, "ncclAllReduce (barrier): %s", R->GetErrorString(r));
	return DG_OK;
}
// DG_COMM_DEBUG=1: one line per protocol stage on stderr (where does a rank wait?)
static void comm_trace(const dg_comm* c, const char* what, int p = -1)
{
	static const int on = env_int("DG_COMM_DEBUG", 0, 0, 1);
	if (!on)
		return;
	const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	std::fprintf(stderr, "[dg_comm %.6f rank %d/%d] %s%s%s\n", t, c->rank, c->nranks, what, p >= 0 ? " piece " : "", p >= 0 ? std::to_string(p).c_str() : "");
	std::fflush(stderr);
}
// host-blocking barrier of the control plane
static dg_status ctrl_barrier_host(dg_comm* c)
{
	if (c->ext_barrier)
	{
		if (c->ext_barrier(c->ext_user) != 0)
			return fail(DG_ERR_HIP, "the caller's barrier failed");
		return DG_OK;
	}
	dg_status s = rccl_stream_barrier(c);
	if (s != DG_OK)
		return s;
	DG_HIP(hipStreamSynchronize(c->gather));
	return DG_OK;
}

static dgvmm::Array* owned_array_of(dg_comm* c, const void* p)
{
	for (auto& kv : c->vmm_owned)
		if (static_cast<const char*>(p) >= kv.second.base && static_cast<const char*>(p) < kv.second.base + kv.second.bytes)
			return &kv.second;
	return nullptr;
}
static const char* export_failure(int status)
{
	switch (status)
	{
	case kExportNoRange: return "hipMemGetAddressRange failed for its field (not a hipMalloc allocation? allocate the field with dg_comm_field_alloc)";
	case kExportNoHandle: return "hipIpcGetMemHandle failed for its field (memory from a pool or a virtual-memory range? allocate the field with dg_comm_field_alloc)";
	case kExportNoFds: return "hipMemExportToShareableHandle failed for the chunks of its field";
	case kExportNoSocket: return "the unix socket that serves the descriptors of its field could not be opened";
	default: return "unknown export failure";
	}
}

// The peers' views of d_field (collective and host-blocking on first use of the pointer).  Every decision to give up is taken
// from data EVERY rank holds (the gathered records, then the gathered open statuses), so the ranks fail together: nobody
// returns early and leaves its peers in a collective.
static dg_status register_field(dg_comm* c, double* d_field, uint64_t bytes, PeerField** out)
{
	auto it = c->peer_fields.find(d_field);
	if (it != c->peer_fields.end() && it->second.bytes >= bytes)
	{
		*out = &it->second;
		return DG_OK;
	}
	const int N = c->nranks;
	IpcRecord mine;
	std::memset(&mine, 0, sizeof(mine));
	PeerField pf;
	pf.base.assign((size_t)N, nullptr);
	pf.base[(size_t)c->rank] = reinterpret_cast<char*>(d_field);
	pf.bytes = bytes;
	if (N > 1)
	{
		// -- step 1: export (local; a failure travels in the record instead of ending the call)
		dgvmm::FdServer server;
		mine.bytes = bytes;
		mine.device = c->device;
		mine.pid = (int32_t)getpid();
		if (dgvmm::Array* arr = owned_array_of(c, d_field))
		{
			mine.kind = 1;
			mine.offset = (uint64_t)(reinterpret_cast<char*>(d_field) - arr->base);
			mine.alloc_bytes = arr->bytes;
			mine.chunk = arr->chunk;
			mine.n_chunks = (uint32_t)arr->handles.size();
			mine.serial = dgvmm::random_token(); // names the socket: unguessable, travels through the control plane only
			std::vector<int> fds;
			if (dgvmm::export_fds(*arr, fds) != hipSuccess)
			{
				(void)hipGetLastError();
				mine.status = kExportNoFds;
			}
			else if (!server.start("dg_vmm_" + std::to_string(mine.pid) + "_" + std::to_string(mine.serial), fds, N - 1))
			{
				for (int f : fds)
					(void)close(f);
				mine.status = kExportNoSocket;
			}
			comm_trace(c, "  chunks exported");
		}
		else
		{
			void* base = nullptr;
			size_t size = 0;
			if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, d_field) != hipSuccess)
			{
				(void)hipGetLastError();
				mine.status = kExportNoRange;
			}
			else if (hipIpcGetMemHandle(&mine.handle, base) != hipSuccess)
			{
				(void)hipGetLastError();
				mine.status = kExportNoHandle;
			}
			else
			{
				mine.offset = (uint64_t)(reinterpret_cast<char*>(d_field) - static_cast<char*>(base));
				mine.alloc_bytes = size;
			}
			comm_trace(c, "  handle exported");
		}
		// -- step 2: every rank sees every record
		std::vector<IpcRecord> all((size_t)N);
		dg_status s = ctrl_allgather(c, &mine, all.data(), sizeof(IpcRecord));
		if (s != DG_OK)
			return s;
		comm_trace(c, "  records gathered");
		{
			// only now does the descriptor server accept: connections from the processes of this communicator and nobody else
			std::vector<int32_t> pids;
			for (int r = 0; r < N; ++r)
				if (r != c->rank)
					pids.push_back(all[(size_t)r].pid);
			server.allow(pids);
		}
		// -- step 3: refuse together
		// Measured on the MI355X boxes of rounds 4 and 5 (ROCm 7.2, dmabuf IPC, several processes on one device):
		// hipIpcOpenMemHandle never returns for an allocation above 2 GiB (1.9 GB: fine with 2, 3 and 4 processes; 3.8 GB: hangs
		// with 2).  Larger fields come from dg_comm_field_alloc (chunks of 512 MiB, kind 1); a plain allocation above the limit
		// is refused (DG_IPC_MAX_MB raises the limit where the platform is known to cope).
		const uint64_t max_alloc = (uint64_t)env_int("DG_IPC_MAX_MB", 2047, 1, 1 << 30) << 20;
		for (int r = 0; r < N; ++r)
		{
			const IpcRecord& rec = all[(size_t)r];
			if (rec.status != kExportOk)
				return fail(DG_ERR_HIP, "DG_EXCHANGE_COPY: rank %d: %s", r, export_failure(rec.status));
			if (rec.bytes != bytes)
				return fail(DG_ERR_INVALID, "rank %d registered %llu bytes for this field, this rank %llu", r, (unsigned long long)rec.bytes,
							(unsigned long long)bytes);
			if (rec.kind == 0 && rec.alloc_bytes > max_alloc)
				return fail(DG_ERR_INVALID, "DG_EXCHANGE_COPY: the field of rank %d lives in an allocation of %.2f GB; opening allocations above %llu MB "
											"through HIP IPC hung on the platform this was developed on: allocate the field with dg_comm_field_alloc "
											"(or set DG_IPC_MAX_MB to try)", r, (double)rec.alloc_bytes * 1e-9, (unsigned long long)(max_alloc >> 20));
			if (rec.kind == 1 && (rec.n_chunks == 0 || rec.chunk == 0 || rec.offset + rec.bytes > rec.alloc_bytes))
				return fail(DG_ERR_INVALID, "rank %d sent an inconsistent record for its field", r);
			for (int q = 0; q < r; ++q)
				if (all[(size_t)q].pid == rec.pid)
					return fail(DG_ERR_INVALID, "ranks %d and %d live in one process: DG_EXCHANGE_COPY is one process per rank", q, r);
		}
		// -- step 4: the ranks map each other's fields ONE RANK AT A TIME (DG_FORCE=ipc_stagger=0: all at once): four processes that
		// opened each other's handles simultaneously never returned from hipIpcOpenMemHandle on the box this was developed
		// on (two did), and the set-up happens once per field.  A rank whose open fails keeps taking part in the barriers.
		const bool stagger = force_int("ipc_stagger", 1, 0, 1) != 0;
		uint64_t my_open_status = 0; // 0: every peer mapped; else 1 + the rank whose field could not be mapped
		std::string my_open_error;
		dg_status barrier_failure = DG_OK;
		for (int turn = 0; turn < (stagger ? N : 1); ++turn)
		{
			if (stagger)
			{
				s = ctrl_barrier_host(c);
				if (s != DG_OK)
				{
					barrier_failure = s; // (the control plane itself broke: nothing collective can follow)
					break;
				}
			}
			if (stagger && turn != c->rank)
				continue;
			for (int r = 0; r < N && my_open_status == 0; ++r)
			{
				if (r == c->rank)
					continue;
				const IpcRecord& rec = all[(size_t)r];
				if (rec.kind == 1)
				{
					std::vector<int> fds;
					std::vector<size_t> sizes;
					for (uint32_t i = 0; i < rec.n_chunks; ++i)
						sizes.push_back((size_t)std::min<uint64_t>(rec.chunk, rec.alloc_bytes - (uint64_t)i * rec.chunk));
					dgvmm::Array imp;
					hipError_t e = hipSuccess;
					if (!dgvmm::fetch_fds("dg_vmm_" + std::to_string(rec.pid) + "_" + std::to_string(rec.serial), (int)rec.n_chunks, fds, 30000))
					{
						my_open_status = 1 + (uint64_t)r;
						my_open_error = "could not fetch the descriptors of the field of rank " + std::to_string(r);
						break;
					}
					e = dgvmm::import_fds(imp, fds, sizes, c->device);
					for (int f : fds)
						(void)close(f);
					if (e != hipSuccess)
					{
						(void)hipGetLastError();
						my_open_status = 1 + (uint64_t)r;
						my_open_error = std::string("mapping the chunks of the field of rank ") + std::to_string(r) + ": " + hipGetErrorString(e);
						break;
					}
					pf.base[(size_t)r] = imp.base + rec.offset;
					c->vmm_imported.push_back(std::move(imp));
					comm_trace(c, "  mapped the chunks of rank", r);
					continue;
				}
				std::string key(reinterpret_cast<const char*>(&rec.pid), sizeof(rec.pid));
				key.append(reinterpret_cast<const char*>(&rec.handle), sizeof(rec.handle));
				auto oh = c->opened_handles.find(key);
				void* p = nullptr;
				if (oh != c->opened_handles.end())
					p = oh->second;
				else
				{
					const hipError_t e = hipIpcOpenMemHandle(&p, rec.handle, hipIpcMemLazyEnablePeerAccess);
					if (e != hipSuccess)
					{
						(void)hipGetLastError();
						my_open_status = 1 + (uint64_t)r;
						my_open_error = std::string("hipIpcOpenMemHandle (field of rank ") + std::to_string(r) + ", device " + std::to_string(rec.device) +
										"): " + hipGetErrorString(e);
						break;
					}
					c->opened_handles.emplace(key, p);
					comm_trace(c, "  opened the field of rank", r);
				}
				pf.base[(size_t)r] = static_cast<char*>(p) + rec.offset;
			}
		}
		if (barrier_failure != DG_OK)
			return barrier_failure;
		// -- step 5: did everybody map everybody?  (also the barrier that ends the set-up: the descriptor servers may stop)
		std::vector<uint64_t> statuses((size_t)N);
		s = ctrl_allgather(c, &my_open_status, statuses.data(), sizeof(uint64_t));
		server.finish();
		if (s != DG_OK)
			return s;
		if (my_open_status != 0)
			return fail(DG_ERR_HIP, "DG_EXCHANGE_COPY: %s", my_open_error.c_str());
		for (int r = 0; r < N; ++r)
			if (statuses[(size_t)r] != 0)
				return fail(DG_ERR_HIP, "DG_EXCHANGE_COPY: rank %d could not map the field of rank %d", r, (int)(statuses[(size_t)r] - 1));
	}
	PeerField& slot = c->peer_fields[d_field];
	slot = std::move(pf);
	*out = &slot;
	return DG_OK;
}
static uint64_t hash_cuts(const uint32_t cuts[4][dg::kMaxRanks + 1], int V)
{
	uint64_t h = 1469598103934665603ull;
	for (int c = 0; c < 4; ++c)
		for (int v = 0; v <= V; ++v)
		{
			h ^= cuts[c][v];
			h *= 1099511628211ull;
		}
	return h;
}
// Every rank derives the cuts from its own copy of plane_cost; ranks that disagree would post transfers that do not match (a
// hang or a corrupted field).  The hash of the cuts therefore goes round on EVERY exchange call -- a collective every rank
// enters whatever its arguments look like -- and a mismatch fails the call on all of them.  Host-blocking: the call returns
// from here once every rank has entered it (8 bytes per rank; the in-place forms serialise their steps on the field anyway).
static dg_status agree_on_cuts(dg_comm* comm, const uint32_t cuts[4][dg::kMaxRanks + 1], int V)
{
	const int N = comm->nranks;
	if (N <= 1)
		return DG_OK;
	const uint64_t h = hash_cuts(cuts, V);
	std::vector<uint64_t> all((size_t)N);
	const dg_status s = ctrl_allgather(comm, &h, all.data(), sizeof(h));
	if (s != DG_OK)
		return s;
	for (int r = 0; r < N; ++r)
		if (all[(size_t)r] != all[0])
			return fail(DG_ERR_INVALID, "rank %d cut the lattice differently from rank 0: plane_cost must hold the same values on every rank (this is rank %d)",
						r, comm->rank);
	return DG_OK;
}

static dg_status exchange_copy(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_comm* comm, int pieces,
							   const float* const plane_cost[4], double* d_field, hipStream_t st)
{
	const int N = comm->nranks;
	const bool ext = comm->ext_allgather != nullptr;
	pieces = std::max(1, std::min(pieces, dg::kMaxRanks / N));
	const int V = pieces * N;
	while ((int)comm->copy_streams.size() < N)
	{
		hipStream_t cs = nullptr;
		hipEvent_t ce = nullptr;
		// (highest priority: a queue of their own, so that the pushes are not serialised behind the sampling launches)
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
		DG_HIP(hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, hi));
		comm->copy_streams.push_back(cs);
		DG_HIP(hipEventCreateWithFlags(&ce, hipEventDisableTiming));
		comm->copy_done.push_back(ce);
	}
	PeerField* pf = nullptr;
	comm_trace(comm, "register_field");
	dg_status s = register_field(comm, d_field, dg_grid_n_nodes(grid) * sizeof(double), &pf);
	if (s != DG_OK)
		return s;
	comm_trace(comm, "registered");
	uint32_t cuts[4][dg::kMaxRanks + 1];
	dg::chunk_planes(grid->resolution, V, plane_cost, cuts);
	dg::ClassGeom cg[4];
	dg::class_geometry(grid->resolution, cg);
	DG_HIP(piece_events(comm, pieces));
	auto chunk_off = [&](int c, int v) { return (size_t)(cg[c].off + (uint64_t)cuts[c][v] * cg[c].D[0] * cg[c].D[1]) * sizeof(double); };
	auto chunk_bytes = [&](int c, int v) { return (size_t)(cuts[c][v + 1] - cuts[c][v]) * cg[c].D[0] * cg[c].D[1] * sizeof(double); };
	// barrier 1: every rank's stream has reached this call, i.e. nothing reads its field any more -- it may be written.  The
	// all-gather of the cuts' hash IS that barrier (nobody gets the result before everybody has contributed), and it is the check
	// that the ranks cut the lattice the same way.
	if (N > 1)
	{
		if (ext)
		{
			comm_trace(comm, "barrier 1: stream sync");
			DG_HIP(hipStreamSynchronize(st));
		}
		else
		{
			DG_HIP(hipEventRecord(comm->entered, st));
			DG_HIP(hipStreamWaitEvent(comm->gather, comm->entered, 0));
		}
		comm_trace(comm, "barrier 1: hash all-gather");
		s = agree_on_cuts(comm, cuts, V);
		if (s != DG_OK)
			return s;
		comm_trace(comm, "barrier 1: left");
	}
	for (int p = 0; p < pieces; ++p)
	{
		const int v = p * N + comm->rank;
		uint32_t qb[4], qe[4];
		for (int c = 0; c < 4; ++c)
		{
			qb[c] = cuts[c][v];
			qe[c] = cuts[c][v + 1];
		}
		DG_HIP(hipEventRecord(comm->t_begin[(size_t)p], st));
		s = dg_sdf_sample_planes_device(mesh, grid, invert, qb, qe, d_field, st);
		if (s != DG_OK)
			return s;
		DG_HIP(hipEventRecord(comm->t_end[(size_t)p], st));
		DG_HIP(hipEventRecord(comm->sampled[(size_t)p], st));
		comm_trace(comm, "sampling enqueued", p);
		// this rank's chunks of piece p, pushed into every peer's field by the copy engines while piece p + 1 is sampled
		for (int k = 1; k < N; ++k)
		{
			const int d = (comm->rank + k) % N; // (every rank starts with its right neighbour: no peer is everybody's first target)
			hipStream_t cs = comm->copy_streams[(size_t)d];
			DG_HIP(hipStreamWaitEvent(cs, comm->sampled[(size_t)p], 0));
			for (int c = 0; c < 4; ++c)
			{
				const size_t len = chunk_bytes(c, v);
				if (len)
					DG_HIP(hipMemcpyAsync(pf->base[(size_t)d] + chunk_off(c, v), reinterpret_cast<char*>(d_field) + chunk_off(c, v), len,
										  hipMemcpyDeviceToDevice, cs));
			}
		}
	}
	DG_HIP(hipEventRecord(comm->t_last_sampled, st));
	// barrier 2: this rank's pushes are complete -> every rank's pushes are complete -> the field is whole
	if (N > 1)
	{
		for (int d = 0; d < N; ++d)
			if (d != comm->rank)
				DG_HIP(hipEventRecord(comm->copy_done[(size_t)d], comm->copy_streams[(size_t)d]));
		if (ext)
		{
			comm_trace(comm, "barrier 2: waiting for the pushes");
			for (int d = 0; d < N; ++d)
				if (d != comm->rank)
					DG_HIP(hipEventSynchronize(comm->copy_done[(size_t)d]));
			comm_trace(comm, "barrier 2: enter");
			if (comm->ext_barrier(comm->ext_user) != 0)
				return fail(DG_ERR_HIP, "the caller's barrier failed");
			comm_trace(comm, "barrier 2: left");
		}
		else
		{
			for (int d = 0; d < N; ++d)
				if (d != comm->rank)
					DG_HIP(hipStreamWaitEvent(comm->gather, comm->copy_done[(size_t)d], 0));
			s = rccl_stream_barrier(comm);
			if (s != DG_OK)
				return s;
			DG_HIP(hipEventRecord(comm->done, comm->gather));
			DG_HIP(hipStreamWaitEvent(st, comm->done, 0));
		}
	}
	DG_HIP(hipEventRecord(comm->t_complete, st));
	comm->wait_timed = true;
	comm->last_pieces = pieces;
	return DG_OK;
}

dg_status dg_sdf_sample_exchange_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_comm* comm, int pieces,
										int flags, int root, const float* const plane_cost[4], double* d_field, void* stream)
{
	if ((flags & DG_EXCHANGE_COPY) != 0)
	{
		if (flags != (DG_EXCHANGE_COPY | DG_EXCHANGE_INPLACE))
			return fail(DG_ERR_INVALID, "DG_EXCHANGE_COPY goes with DG_EXCHANGE_INPLACE and nothing else");
		if (!mesh || !grid || !comm || !d_field)
			return fail(DG_ERR_INVALID, "null argument");
		if (!valid_grid(grid))
			return fail(DG_ERR_INVALID, "invalid grid");
		if (mesh->device != comm->device)
			return fail(DG_ERR_INVALID, "mesh lives on device %d, the communicator on device %d", mesh->device, comm->device);
		DG_ON_DEVICE_OF(mesh);
		return exchange_copy(mesh, grid, invert, comm, pieces, plane_cost, d_field, static_cast<hipStream_t>(stream));
	}
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
	if (!comm->comm)
		return fail(DG_ERR_INVALID, "a communicator with an external control plane runs DG_EXCHANGE_INPLACE | DG_EXCHANGE_COPY only");
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
	{
		const dg_status hs = agree_on_cuts(comm, cuts, V);
		if (hs != DG_OK)
			return hs;
	}
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
	DG_HIP(hipEventRecord(comm->t_last_sampled, st));
	DG_HIP(hipStreamWaitEvent(st, comm->gathered[(size_t)pieces - 1], 0)); // (one stream: the last piece's event covers all)
	DG_HIP(hipEventRecord(comm->t_complete, st));
	comm->wait_timed = true;
	comm->last_pieces = pieces;
	return DG_OK;
}

dg_status dg_comm_field_alloc(dg_comm* comm, uint64_t n_doubles, double** d_field)
{
	if (!comm || !d_field)
		return fail(DG_ERR_INVALID, "null argument");
	*d_field = nullptr;
	if (n_doubles == 0)
		return fail(DG_ERR_INVALID, "empty field");
	DG_ON_DEVICE_OF(comm);
	dgvmm::Array a;
	const hipError_t e = dgvmm::create(a, (size_t)n_doubles * sizeof(double), comm->device);
	if (e != hipSuccess)
	{
		(void)hipGetLastError();
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "dg_comm_field_alloc: %.2f GB as chunks of %zu MiB: %s", (double)n_doubles * 8e-9,
					dgvmm::kVmmChunkBytes >> 20, hipGetErrorString(e));
	}
	*d_field = reinterpret_cast<double*>(a.base);
	char* key = a.base;
	comm->vmm_owned.emplace(key, std::move(a));
	return DG_OK;
}

dg_status dg_comm_field_free(dg_comm* comm, double* d_field)
{
	if (!comm)
		return fail(DG_ERR_INVALID, "null argument");
	if (!d_field)
		return DG_OK;
	auto it = comm->vmm_owned.find(reinterpret_cast<char*>(d_field));
	if (it == comm->vmm_owned.end())
		return fail(DG_ERR_INVALID, "not an array of dg_comm_field_alloc on this communicator");
	// registered with the peers -- under its base pointer or under any pointer INSIDE the array (register_field accepts those):
	// they hold mappings of it
	if (comm->nranks > 1)
		for (const auto& kv : comm->peer_fields)
		{
			const char* p = reinterpret_cast<const char*>(kv.first);
			if (p >= it->second.base && p < it->second.base + it->second.bytes)
				return fail(DG_ERR_INVALID, "the array is registered with the peers (they hold mappings of it): it lives until dg_comm_destroy");
		}
	DG_ON_DEVICE_OF(comm);
	DG_HIP(hipDeviceSynchronize());
	for (auto pf = comm->peer_fields.begin(); pf != comm->peer_fields.end();) // (one rank: entries of pointers inside the array go with it)
	{
		const char* p = reinterpret_cast<const char*>(pf->first);
		if (p >= it->second.base && p < it->second.base + it->second.bytes)
			pf = comm->peer_fields.erase(pf);
		else
			++pf;
	}
	dgvmm::destroy(it->second);
	comm->vmm_owned.erase(it);
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

dg_status dg_comm_last_exchange_wait_ms(dg_comm* comm, float* ms)
{
	if (!comm || !ms)
		return fail(DG_ERR_INVALID, "null argument");
	if (!comm->wait_timed)
		return fail(DG_ERR_INVALID, "no exchange call on this communicator yet");
	DG_ON_DEVICE_OF(comm);
	DG_HIP(hipEventSynchronize(comm->t_complete));
	DG_HIP(hipEventElapsedTime(ms, comm->t_last_sampled, comm->t_complete));
	return DG_OK;
}

} // extern "C"
