/* discregrid_hip.h -- C ABI of the MI355X (gfx950) SDF-discretisation hot path.
 *
 * Drop-in boundary for Discregrid's node-sampling / interpolation path.  Every entry
 * point names the reference interface it replaces (paths relative to the Discregrid
 * source tree, InteractiveComputerGraphics/Discregrid @ v1):
 *
 *   dg_mesh_create            TriangleMeshDistance(TriangleMesh const&)  +  _construct()
 *                             discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:227-230, 336-441
 *   dg_sdf_sample_nodes[_device]
 *                             the `omp for` node loop of CubicLagrangeDiscreteGrid::addFunction with
 *                             func = +-TriangleMeshDistance::signed_distance(x).distance
 *                             discregrid/src/cubic_lagrange_discrete_grid.cpp:806-831, :604-665
 *                             TriangleMeshDistance.h:269-308, 514-562, 564-820; cmd/generate_sdf/main.cpp:95-105
 *   dg_signed_distance[_device]
 *                             TriangleMeshDistance::signed_distance for a batch of points
 *                             TriangleMeshDistance.h:269-314
 *   dg_signed_distance_point  the same for one point, on the calling host thread (:188-208, 269-328)
 *   dg_field_create / dg_field_attach_device
 *                             the per-field storage m_nodes / m_cells / m_cell_map
 *                             discregrid/include/Discregrid/cubic_lagrange_discrete_grid.hpp:69-71
 *   dg_sdf_sample_field / dg_density_map_field / dg_field_host_wait
 *                             addFunction as a whole (:780-899): the new field's storage is created by the call,
 *                             device-resident, with the host vector filled asynchronously
 *   dg_interpolate_batch[_device]
 *                             CubicLagrangeDiscreteGrid::interpolate(field_id, x, gradient*)
 *                             discregrid/src/cubic_lagrange_discrete_grid.cpp:977-1063 (shape functions :339-580)
 *   dg_reduce_field           CubicLagrangeDiscreteGrid::reduceField for value predicates
 *                             discregrid/src/cubic_lagrange_discrete_grid.cpp:1065-1174, zValue :583-601
 *   dg_shard_* / dg_unpack_shards_device, dg_comm_*, dg_sdf_sample_allgather_device
 *                             no counterpart (the reference is single-process OpenMP): lattice sharding
 *                             for one-process-per-GPU runs; the exchange is one logical RCCL all-gather,
 *                             issued either by the caller (dg_sdf_sample_shard_device + dg_unpack_*) or
 *                             by the library itself (dg_sdf_sample_allgather_device).
 *
 * Conventions: plain pointers and sizes only; every function returns a dg_status (0 = ok)
 * and never throws or exits; dg_last_error() gives a thread-local message.  Functions
 * suffixed _device take DEVICE pointers and a hipStream_t (passed as void*; NULL = the
 * default stream) and are asynchronous; the unsuffixed forms take HOST pointers, stage
 * through device memory and synchronise before returning.  All work runs on the HIP
 * device that is current on the calling thread (hipSetDevice / dg_set_device).
 *
 * There is NO CPU fallback behind this ABI: without a HIP device every compute entry point returns
 * DG_ERR_NO_DEVICE.  (What does work without one: the grid helpers, dg_mesh_create as a host-only handle and the
 * per-point query dg_signed_distance_point -- see dg_mesh_device().)
 *
 * Sentinel: DG_NO_VALUE == std::numeric_limits<double>::max() marks "no value" exactly
 * like the reference (cubic_lagrange_discrete_grid.cpp:817, 982, 994, 1017).
 */
#ifndef DISCREGRID_HIP_H
#define DISCREGRID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points below are ALL it exports (tests/test_abi.py checks that
 * `nm -D --defined-only` lists no other text / data symbol), so a host application's own fail(), TraceRange ... never
 * interpose its internals. */
#define DG_API __attribute__((visibility("default")))

#define DG_NO_VALUE 1.7976931348623157e308

typedef enum dg_status {
	DG_OK = 0,
	DG_ERR_INVALID = 1,   /* bad argument (null pointer, empty mesh, range out of bounds, ...) */
	DG_ERR_NO_DEVICE = 2, /* no usable HIP device / wrong architecture */
	DG_ERR_HIP = 3,       /* a HIP runtime call failed; see dg_last_error() */
	DG_ERR_ALLOC = 4      /* host or device allocation failed */
} dg_status;

/* The serialised members of Discregrid::DiscreteGrid (discrete_grid.hpp:91-96).  Pass the
 * SAME doubles the host grid holds: parity at 1e-10 depends on it. */
typedef struct dg_grid_desc {
	double domain_min[3];
	double domain_max[3];
	uint32_t resolution[3];
	uint32_t reserved_;
	double cell_size[3];
	double inv_cell_size[3];
} dg_grid_desc;

typedef struct dg_mesh dg_mesh;   /* device-resident flattened BVH + triangle packets + pseudonormals */
typedef struct dg_field dg_field; /* device-resident coefficient vector (+ optional cell table / map) */

typedef struct dg_mesh_info {
	uint64_t n_vertices;
	uint64_t n_triangles;
	uint64_t n_bvh_nodes;
	uint32_t bvh_depth;
	uint32_t not_watertight;  /* bit0: an edge with one face, bit1: an edge with >2 faces (TriangleMeshDistance.h:422-438) */
	uint64_t device_bytes;
	double build_seconds;     /* host BVH + pseudonormal construction */
} dg_mesh_info;

typedef struct dg_shard_info {
	uint64_t count;       /* nodes owned by (rank, nranks) */
	uint64_t stride;      /* max count over ranks, rounded up to 64: per-rank slot size for the all-gather */
} dg_shard_info;

/* ---- runtime ------------------------------------------------------------------------- */
DG_API const char* dg_version(void);
DG_API const char* dg_last_error(void);
DG_API dg_status dg_device_count(int* count);
DG_API dg_status dg_set_device(int device);
DG_API dg_status dg_current_device(int* device); /* the calling thread's current device (hipGetDevice) */

/* ---- grid helpers (host arithmetic of discrete_grid.hpp:22-29, no device work) ---------- */
DG_API dg_status dg_grid_desc_init(const double domain_min[3], const double domain_max[3], const uint32_t resolution[3],
							dg_grid_desc* out);
/* Default sampling domain of the reference's GenerateSDF tool (cmd/generate_sdf/main.cpp:83-91): the
 * bounding box of the vertices, max grown by 1e-3*|diagonal| first, then min by 1e-3*|diagonal of
 * the already grown box| (the asymmetry is the reference's).  out = {min xyz, max xyz}. */
DG_API dg_status dg_default_domain(const double* verts, uint64_t n_vertices, double out_min_max[6]);
DG_API uint64_t dg_grid_n_nodes(const dg_grid_desc* grid); /* (N+1)^3 + 6N(N+1)^2 generalised, :790-796 */
DG_API uint64_t dg_grid_n_cells(const dg_grid_desc* grid);

/* ---- mesh / BVH handle ----------------------------------------------------------------- */
/* verts: 3*n_vertices doubles (xyzxyz...), tris: 3*n_triangles vertex indices.  Builds the
 * pseudonormals exactly as the reference does and a flattened BVH of this library's own design
 * (oriented-box bounds, sibling-pair records) on the host, then uploads everything to the
 * current device. */
DG_API dg_status dg_mesh_create(const double* verts, size_t n_vertices, const uint32_t* tris, size_t n_triangles,
						 dg_mesh** out);
DG_API dg_status dg_mesh_get_info(const dg_mesh* mesh, dg_mesh_info* info);
/* The device the handle's arrays live on, or -1 for a HOST-ONLY handle: dg_mesh_create on a machine without a HIP device (or
 * under DG_FORCE_CPU=1) still builds the BVH and the pseudonormals and keeps them in host memory, so that
 * dg_signed_distance_point() -- the reference's per-point TriangleMeshDistance::signed_distance,
 * TriangleMeshDistance.h:252-267, 269-328 -- works wherever the reference works; every entry point that would launch a
 * kernel returns DG_ERR_NO_DEVICE for such a handle. */
DG_API int dg_mesh_device(const dg_mesh* mesh);
DG_API void dg_mesh_destroy(dg_mesh* mesh);

/* ---- K1: SDF node sampling --------------------------------------------------------------- */
/* out[l - node_begin] = (invert ? -1 : 1) * signed_distance(indexToNodePosition(l)) for
 * l in [node_begin, node_end); nodes whose pred_mask byte (indexed l - node_begin, nullable)
 * is 0 receive DG_NO_VALUE, mirroring the SamplePredicate branch (:814-817). */
/* (`out` is scratch from the moment of the call: the direct form touches and pins it before the first result arrives,
 * so a call that returns an error may leave it partly overwritten.) */
DG_API dg_status dg_sdf_sample_nodes(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, uint64_t node_begin,
							  uint64_t node_end, const uint8_t* pred_mask, double* out);
DG_API dg_status dg_sdf_sample_nodes_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, uint64_t node_begin,
									 uint64_t node_end, const uint8_t* d_pred_mask, double* d_out, void* stream);
/* Several devices, one process (host results): meshes[i] is the same mesh created once per device
 * (dg_set_device + dg_mesh_create; the same device may appear more than once).  The node range is cut
 * into chunks of whole 4-plane slabs that are dealt round-robin to the meshes; every mesh runs its
 * own kernel / copy pipeline from its own host thread straight into `out`, so no collective is
 * needed (the all-gather of the sharded device path exists to leave the field on every GPU). */
DG_API dg_status dg_sdf_sample_nodes_multi(const dg_mesh* const* meshes, int n_meshes, const dg_grid_desc* grid, int invert,
									uint64_t node_begin, uint64_t node_end, const uint8_t* pred_mask, double* out);

/* Signed distance at arbitrary points (3*n doubles) -> dist[n]; optional nearest triangle id
 * (original index), nearest entity (0..6 = V0,V1,V2,E01,E12,E02,F) and nearest point
 * (TriangleMeshDistance::signed_distance, TriangleMeshDistance.h:269-314).  Batches of >= 4096
 * points that arrive in no spatial order are processed tile by tile (decided on the device, the
 * results land at the caller's positions); where several triangles are exactly equidistant the
 * id of any of them may be returned, as in the reference the first one met wins. */
DG_API dg_status dg_signed_distance(const dg_mesh* mesh, const double* xyz, uint64_t n, double* dist, int32_t* tri,
							 int32_t* entity, double* nearest);
DG_API dg_status dg_signed_distance_device(const dg_mesh* mesh, const double* d_xyz, uint64_t n, double* d_dist,
									int32_t* d_tri, int32_t* d_entity, double* d_nearest, void* stream);

/* ONE point, evaluated on the calling thread (no launch): TriangleMeshDistance::signed_distance(point)
 * as user code calls it per node or per particle (TriangleMeshDistance.h:188-208, 269-328;
 * cmd/generate_sdf/main.cpp:97-101).  Walks the host copy of the very arrays dg_mesh_create uploaded with
 * the same per-triangle arithmetic as the kernels, so `dist` has the bits dg_signed_distance returns for
 * the point (of exactly equidistant triangles either may be named).  Const and lock-free: any number of
 * threads may call it on one mesh.  tri / entity / nearest may be NULL.  Batches belong on the GPU
 * (dg_signed_distance, dg_sdf_sample_nodes): this is the per-point evaluator, not a CPU path for them. */
DG_API dg_status dg_signed_distance_point(const dg_mesh* mesh, const double xyz[3], double* dist, int32_t* tri,
								   int32_t* entity, double* nearest);

/* ---- multi-GPU sharding of the node lattice (one process per GPU) -------------------------- */
/* The lattice of each of the four node classes [V | X | Y | Z] is cut into slabs of 4 planes
 * along its slowest-varying index (k, k, i, j -- i.e. Z-slabs for V and X) and the slabs are
 * dealt round-robin to the ranks.  A rank writes its nodes into a packed buffer of
 * `count` doubles; after ONE all-gather of `stride` doubles per rank,
 * dg_unpack_shards_device() scatters the gathered buffer into reference node order.
 * `nranks` may exceed the number of GPUs (up to 64): a process that takes the "virtual ranks"
 * p*G + g, p = 0..C-1, of a C*G-way deal can all-gather piece p among the G GPUs while it samples
 * piece p+1; the C gathered pieces laid end to end are the buffer a single C*G-rank all-gather
 * would produce, so the same unpack call (nranks = C*G) applies (bench.py --pieces). */
DG_API dg_status dg_shard_layout(const dg_grid_desc* grid, int rank, int nranks, dg_shard_info* out);
DG_API dg_status dg_sdf_sample_shard_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, int rank, int nranks,
									 double* d_packed, void* stream);
DG_API dg_status dg_unpack_shards_device(const dg_grid_desc* grid, int nranks, const double* d_gathered, uint64_t stride,
								  double* d_field, void* stream);
/* Same for the slots [rank_begin, rank_end) of the gathered buffer only (d_gathered is still the
 * base of the whole buffer): lets a pieced gather unpack piece p while piece p+1 is in flight. */
DG_API dg_status dg_unpack_shard_range_device(const dg_grid_desc* grid, int nranks, const double* d_gathered, uint64_t stride,
									   int rank_begin, int rank_end, double* d_field, void* stream);

/* The exchange step behind the ABI.  A dg_comm wraps an RCCL communicator of one-process-per-GPU ranks
 * (RCCL is loaded on the first dg_comm_* call; single-GPU users never load it):
 *   rank 0 calls dg_comm_unique_id() and hands the 128 bytes to the other ranks out of band (a file, MPI,
 *   a TCP store); every rank then calls dg_comm_create() with its device current (collective:
 *   ncclCommInitRank).  dg_comm_adopt() wraps an existing ncclComm_t of the same RCCL instance instead.
 * dg_sdf_sample_allgather_device() is the whole multi-GPU node-sampling step (collective, asynchronous):
 * this rank samples its shards of the lattice in `pieces` pieces, each piece is all-gathered over xGMI
 * while the next one is sampled and unpacked into reference node order while the one after is gathered;
 * when the work enqueued on `stream` has run, d_field (dg_grid_n_nodes doubles, device memory) holds the
 * WHOLE coefficient vector on every rank -- bit for bit what dg_sdf_sample_nodes_device writes on one GPU.
 * pieces is clamped to 64 / nranks; 4 is a good value.  No reference counterpart (single process). */
#define DG_UNIQUE_ID_BYTES 128
typedef struct dg_comm dg_comm;
DG_API dg_status dg_comm_unique_id(uint8_t id[DG_UNIQUE_ID_BYTES]);
DG_API dg_status dg_comm_create(const uint8_t id[DG_UNIQUE_ID_BYTES], int rank, int nranks, dg_comm** out);
DG_API dg_status dg_comm_adopt(void* nccl_comm, int rank, int nranks, dg_comm** out); /* does not take ownership */
DG_API void dg_comm_destroy(dg_comm* comm);
DG_API dg_status dg_sdf_sample_allgather_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_comm* comm,
										 int pieces, double* d_field, void* stream);

/* The same step with the lattice cut into CONTIGUOUS chunks, exchanged in place (DG_EXCHANGE_INPLACE): every class's
 * planes are cut into pieces * nranks runs of whole planes (dg_chunk_layout; chunk v belongs to rank v % nranks and to
 * piece v / nranks), a rank samples its chunks straight into their places in d_field, and the chunks travel with one
 * grouped set of ncclBroadcast per piece (DG_EXCHANGE_P2P: ncclSend / ncclRecv pairs, every pair of GPUs over its own
 * xGMI link) -- no packed buffers, no unpack pass over the whole field, no scratch beyond d_field itself.  Interleaved
 * thin slabs balance the uneven cost per node by construction; contiguous chunks need the cuts to follow the cost:
 * plane_cost[c] (nullable) holds a relative cost for each plane of class c along its slowest index (k, k, i, j) --
 * e.g. the times dg_comm_last_chunk_ms() reported for the previous call, spread over the planes of each chunk.
 * plane_cost must hold the SAME values on every rank (share the measured times first): each rank derives the cuts from it
 * by itself, and ranks that disagree about the cuts post transfers that do not match -- a hang or a corrupted field.
 * DG_EXCHANGE_TO_ROOT: only rank `root` ends up with the whole field (the others keep their own chunks): every other
 * GPU then SENDS its 1 / nranks of the field instead of receiving (nranks - 1) / nranks of it.  flags == 0 is
 * dg_sdf_sample_allgather_device (interleaved slabs, packed buffers, all-gather, unpack).  No reference counterpart. */
#define DG_EXCHANGE_INPLACE 1
#define DG_EXCHANGE_P2P 2
#define DG_EXCHANGE_TO_ROOT 4
/* DG_EXCHANGE_COPY (with DG_EXCHANGE_INPLACE): the chunks are PUSHED into the peers' fields by the copy engines -- one
 * hipMemcpyAsync per chunk and peer on a copy stream per peer, into the peer's d_field opened through HIP IPC -- so that
 * no collective kernel competes with the sampling kernel for the compute units (K1 fills 0.95 of the VALU issue slots of
 * all 256 CUs; the RCCL forms run their copy kernels beside it).  The first call with a given d_field is collective and
 * host-blocking (the ranks exchange IPC handles of the allocations behind their d_field; the pointer must then stay
 * allocated until dg_comm_destroy); every call brackets the pushes with two barriers ("every rank's stream has reached
 * this call: its field may be written", "every rank's pushes have landed") -- one-word all-reduces on the communicator's
 * stream with RCCL, host-blocking calls of the caller's barrier with an external control plane (below).  plane_cost must
 * be the SAME array on every rank (true for all in-place forms: the cuts decide who sends what; every in-place form
 * all-gathers a hash of the cuts on EVERY call -- host-blocking until every rank has entered the call -- and all ranks fail
 * with DG_ERR_INVALID on a mismatch).  Limit found on the development platform
 * (ROCm 7.2, dmabuf IPC): hipIpcOpenMemHandle does not return for allocations above 2 GiB, so fields living in larger
 * hipMalloc allocations are REFUSED (every rank gets DG_ERR_INVALID; DG_IPC_MAX_MB overrides): allocate such fields with
 * dg_comm_field_alloc below.  A rank that cannot export or map a field makes EVERY rank fail (the statuses are gathered).
 * UNVERIFIED on more than one device. */
#define DG_EXCHANGE_COPY 8
/* Device arrays for DG_EXCHANGE_COPY beyond that limit: n_doubles of device memory on the communicator's device as
 * hipMemCreate chunks of 512 MiB behind ONE contiguous address range.  When such an array is registered (first exchange call
 * with it) every chunk is exported as a POSIX descriptor (hipMemExportToShareableHandle), the descriptors travel to the peers
 * over a unix-domain socket (SCM_RIGHTS) and each peer maps them side by side -- no allocation above 2 GiB is ever opened as a
 * whole.  To kernels and copies the array is ordinary device memory.  dg_comm_field_free releases an array no peer maps;
 * a registered one lives until dg_comm_destroy.  No reference counterpart (single process). */
DG_API dg_status dg_comm_field_alloc(dg_comm* comm, uint64_t n_doubles, double** d_field);
DG_API dg_status dg_comm_field_free(dg_comm* comm, double* d_field);
/* cuts[c * (nchunks + 1) + v], v = 0..nchunks: first plane of chunk v of class c (class order V, X, Y, Z); host only */
DG_API dg_status dg_chunk_layout(const dg_grid_desc* grid, int nchunks, const float* const plane_cost[4], uint32_t* cuts);
/* planes [plane_begin[c], plane_end[c]) of every class, sampled into their places in d_field (the WHOLE vector) */
DG_API dg_status dg_sdf_sample_planes_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, const uint32_t plane_begin[4],
									  const uint32_t plane_end[4], double* d_field, void* stream);
DG_API dg_status dg_sdf_sample_exchange_device(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_comm* comm, int pieces,
										int flags, int root, const float* const plane_cost[4], double* d_field, void* stream);
/* A communicator WITHOUT RCCL for DG_EXCHANGE_COPY (whose data path is peer copies): the two small host-side collectives
 * it needs come from the caller -- MPI, a TCP store, torch.distributed's gloo backend (tests/test_gpu_multirank.py drives
 * several ranks on ONE GPU this way, which RCCL refuses).  allgather(mine, all, bytes, user): all[r * bytes ...] = rank r's
 * `mine`, returns 0; barrier(user): returns 0 once every rank has called it.  Both are called from the thread that calls
 * dg_sdf_sample_exchange_device, in the same order on every rank.  Only DG_EXCHANGE_INPLACE | DG_EXCHANGE_COPY runs on
 * such a communicator, and the call blocks the host until the exchange is complete. */
typedef int (*dg_comm_allgather_fn)(const void* mine, void* all, size_t bytes, void* user);
typedef int (*dg_comm_barrier_fn)(void* user);
DG_API dg_status dg_comm_create_external(int rank, int nranks, dg_comm_allgather_fn allgather, dg_comm_barrier_fn barrier, void* user,
								  dg_comm** out);
/* The same without callbacks: the control plane lives in a POSIX shared-memory segment the ranks of ONE node map (rank 0
 * creates "/name", the others wait for it up to DG_COMM_TIMEOUT_S seconds; `name` must be unique to the job).  With fields from
 * dg_comm_field_alloc this is the exchange that leaves the whole field on every GPU with NO collective library at all: copy
 * engines for the data, two barriers in shared memory per step.  Collective; no reference counterpart. */
DG_API dg_status dg_comm_create_shm(const char* name, int rank, int nranks, dg_comm** out);
typedef struct dg_comm_info {
	int32_t rank, nranks, device;
	int32_t rccl_nranks; /* ncclCommCount() of the wrapped communicator; -1: external control plane */
	int32_t registered_fields; /* d_field pointers whose peers are mapped (DG_EXCHANGE_COPY) */
} dg_comm_info;
DG_API dg_status dg_comm_get_info(dg_comm* comm, dg_comm_info* info);
/* device time of this rank's sampling launches of the most recent dg_sdf_sample_exchange_device / _allgather_device
 * call on `comm`, one value per piece (waits for that call); *n_pieces in: capacity of ms, out: pieces */
DG_API dg_status dg_comm_last_chunk_ms(dg_comm* comm, float* ms, int* n_pieces);
/* how long the caller's stream had to wait, after this rank's last sampling launch of the most recent exchange call, until
 * the field was complete (the part of the exchange the sampling did not hide); waits for that call */
DG_API dg_status dg_comm_last_exchange_wait_ms(dg_comm* comm, float* ms);

/* The exchange step WITHOUT collective kernels and WITHOUT device IPC: the coefficient vector is assembled in a POSIX
 * shared-memory segment that every rank (one process per GPU, one node) maps; a rank samples its chunks (the contiguous
 * cost-weighted chunks of dg_chunk_layout, as the in-place forms) into their places in its own d_field and copies them into
 * the shared vector with its own copy engine over its own PCIe link while it samples the next piece; a barrier that lives in
 * the segment says when the vector is whole.  What results is what the reference's addFunction leaves behind -- the HOST
 * vector m_nodes[field] (cubic_lagrange_discrete_grid.cpp:806-831), here shared by all ranks -- not a device-resident field
 * on every GPU: use the dg_comm forms for that.  Neither RCCL, HIP IPC nor the virtual-memory API is touched.
 *   dg_host_field_open: collective over the ranks of the node (rank 0 creates the segment, the others wait for it up to
 *   DG_COMM_TIMEOUT_S seconds; the name is unlinked once everybody has mapped it); with a HIP device current the mapping is
 *   registered as a DMA target.  `name` must be unique to the job (e.g. contain the launcher's pid).
 *   dg_sdf_sample_to_host_field: collective and HOST-BLOCKING; on return the shared vector holds every node, bit for bit
 *   what dg_sdf_sample_nodes writes on one GPU; d_field (dg_grid_n_nodes doubles, this rank's device memory) holds this
 *   rank's chunks only.  plane_cost as for dg_sdf_sample_exchange_device: the ranks compare a hash of their cuts behind
 *   the barrier and all fail with DG_ERR_INVALID when they differ.
 * No reference counterpart (single process). */
typedef struct dg_host_field dg_host_field;
typedef struct dg_host_field_info {
	int32_t rank, nranks, device;
	int32_t registered; /* 1: hipHostRegister took the mapping (direct DMA); 0: pageable copies */
	uint64_t n_doubles;
} dg_host_field_info;
DG_API dg_status dg_host_field_open(const char* name, uint64_t n_doubles, int rank, int nranks, dg_host_field** out);
DG_API double* dg_host_field_data(dg_host_field* hf);
DG_API dg_status dg_host_field_barrier(dg_host_field* hf); /* every rank; fails after DG_COMM_TIMEOUT_S seconds instead of hanging */
DG_API dg_status dg_host_field_get_info(dg_host_field* hf, dg_host_field_info* info);
DG_API void dg_host_field_close(dg_host_field* hf);
DG_API dg_status dg_sdf_sample_to_host_field(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, dg_host_field* hf, int pieces,
									  const float* const plane_cost[4], double* d_field, void* stream);
/* device time of this rank's sampling launches of the most recent dg_sdf_sample_to_host_field call, one value per piece */
DG_API dg_status dg_host_field_last_chunk_ms(dg_host_field* hf, float* ms, int* n_pieces);

/* ---- field handle + K2: batched interpolate ------------------------------------------------ */
/* cells (32 uint32 per row, n_cell_rows rows) and cell_map (one uint32 per grid cell) may both
 * be NULL for an unreduced field: the kernel then uses the closed-form node indices the
 * reference's own table holds right after addFunction (:833-891). */
DG_API dg_status dg_field_create(const dg_grid_desc* grid, const double* coeffs, uint64_t n_coeffs, const uint32_t* cells,
						  uint64_t n_cell_rows, const uint32_t* cell_map, dg_field** out);
/* Non-owning: wraps device arrays that stay valid for the lifetime of the handle. */
DG_API dg_status dg_field_attach_device(const dg_grid_desc* grid, const double* d_coeffs, uint64_t n_coeffs,
								 const uint32_t* d_cells, uint64_t n_cell_rows, const uint32_t* d_cell_map,
								 dg_field** out);
DG_API void dg_field_destroy(dg_field* field);

typedef struct dg_field_info {
	uint64_t n_coeffs;
	uint64_t n_cell_rows;      /* rows of the cell table (= grid cells for an unreduced field) */
	uint64_t device_bytes;     /* device memory the handle owns (coefficients it owns, tables, cell-/tile-major copies) */
	const double* d_coeffs;    /* the coefficient vector on the device (m_nodes[field], cubic_lagrange_discrete_grid.hpp:69) */
	int32_t device;
	int32_t owns_coefficients; /* dg_field_create / dg_sdf_sample_field / dg_density_map_field / dg_reduction_to_field */
	int32_t has_cell_major, has_tile_major;
	int32_t immutable;         /* dg_field_set_immutable */
	int32_t host_copy_pending; /* the asynchronous copy into the caller's host array has not been collected yet */
	uint64_t band_rows;        /* rows of the band-limited cell-major copy (0: none), dg_field_build_cell_major_band */
} dg_field_info;
DG_API dg_status dg_field_get_info(const dg_field* field, dg_field_info* info);
/* An ATTACHED device array (dg_field_attach_device) may change between calls, so batched queries never build the
 * cell-major copy for it by themselves.  immutable != 0 promises that it will not change while the handle lives:
 * the handle then behaves like one made by dg_field_create (copy built on the first batch of >= 2^18 queries). */
DG_API dg_status dg_field_set_immutable(dg_field* field, int immutable);

/* ---- fields produced ON the device (one resident copy per field) ------------------------------------------
 * The reference keeps ONE coefficient vector per field (m_nodes, cubic_lagrange_discrete_grid.hpp:69) that
 * addFunction fills (:806-831) and interpolate reads (:977-1063).  Here that vector lives on the device: the
 * producing call returns a field handle that OWNS the device array its kernel wrote, and K2 (dg_interpolate_batch*),
 * K3 (dg_density_map_field / dg_density_map_nodes*), dg_reduce_field_device and the cell-/tile-major builders read
 * that very array -- nothing is uploaded again.  If host_out is not NULL the coefficients are ALSO copied into the
 * caller's host array (n doubles), asynchronously: the call returns as soon as everything is enqueued (node
 * sampling is cut into chunks whose copies overlap the sampling of the next chunk; the array is made a DMA target
 * piece by piece from a worker thread), consumers on the device start when the last kernel ends, and
 * dg_field_host_wait() blocks until host_out is complete (dg_field_destroy waits too).  host_out must stay
 * allocated until then and is scratch from the moment of the call.  pred_mask (host, nullable, one byte per node)
 * works as in dg_sdf_sample_nodes.  host_first is a hint without effect since round 4: ONE chunk profile serves callers
 * that read on the device next and callers that wait for the host vector -- seven chunks that shrink towards the end
 * [MI355X, 256^3: field complete on the device after 18.5 ms, in the host vector after 21.7 ms; one launch followed by
 * the copy: 16.3 / 34.6 ms]; DG_FORCE="field_fractions=f0,f1,..." overrides, DG_FIELD_ONE_LAUNCH=1 with host_first == 0 is
 * round 3's single launch, DG_FORCE=field_streams=2 alternates the chunks between two streams (measured: no gain). */
DG_API dg_status dg_sdf_sample_field(const dg_mesh* mesh, const dg_grid_desc* grid, int invert, const uint8_t* pred_mask,
							  double* host_out, int host_first, dg_field** out);
/* K3 over the whole lattice of `sdf` into a NEW device-resident field on the same grid (the density map the
 * reference's GenerateDensityMap adds as field 1, cmd/generate_density_map/main.cpp:83-133); arguments as
 * dg_density_map_nodes, host_out / asynchrony as dg_sdf_sample_field. */
DG_API dg_status dg_density_map_field(dg_field* sdf, double support_radius, double rho0, int band_predicate,
							   const uint8_t* pred_mask, double* host_out, dg_field** out);
/* The coefficient arrays of destroyed produced fields are kept for the next field of the same size (a 1 GB
 * hipMalloc costs ~3 ms), up to DG_FIELD_CACHE_MB megabytes per process (default 2048, 0 = keep nothing);
 * this releases them now. */
DG_API void dg_field_cache_trim(void);
/* Blocks until the host copy a producing call started is complete (DG_OK at once if there is none or it was
 * collected before); returns the status of that copy. */
DG_API dg_status dg_field_host_wait(dg_field* field);

/* Optional: builds (once, asynchronously on `stream`) a cell-major device copy of the field --
 * 32 doubles = 256 contiguous bytes per cell row -- that dg_interpolate_batch* then reads
 * instead of gathering 16 scattered 16-byte segments per query.  Costs 256 bytes per cell of
 * device memory (4.3 GB at 256^3); results are bit-identical.  No reference counterpart: the
 * reference gathers through m_cells (cubic_lagrange_discrete_grid.cpp:1005-1019).  The
 * coefficient array must not change afterwards (drop and rebuild if it does).
 * With the copy, batches run through a kernel that fetches a wavefront's 64 rows cooperatively: no binning,
 * the order of the queries does not matter (10 M unordered queries on a 256^3 field: 18 instead of 8 G/s).
 * For a field made by dg_field_create (the library owns the coefficients) dg_interpolate_batch* builds the copy
 * itself on the first batch of >= 2^18 queries, up to DG_K2_AUTO_CELL_MAJOR_MB megabytes (default 16384; 0: never)
 * and a quarter of the free device memory -- unless dg_field_drop_cell_major was called on the field. */
DG_API dg_status dg_field_build_cell_major(dg_field* field, void* stream);
DG_API dg_status dg_field_drop_cell_major(dg_field* field); /* (drops the band-limited copy below as well) */
/* The same rows for a VALUE BAND only: a cell row gets its 256 contiguous bytes if the values its 32 coefficients span reach
 * into [lo, hi] (min <= hi and max >= lo), every other cell stays where it is; one bit per cell row (+ a running count per 64) tells which.
 * dg_interpolate_batch* then serves queries into mapped cells from their rows (fetched cooperatively like the full copy's)
 * and, in the same launch, all others by the plain gather -- queries in any order, no sort.  SPH boundary handling and
 * GenerateDensityMap (cmd/generate_density_map/main.cpp:99,125-132) query the shell |phi| < 2h around the surface: for
 * [-(2h + cell diagonal), 2h + cell diagonal] the copy holds 10-20 % of the cells -- less than 1 x the field instead of 4.6 x.
 * One-off and host-blocking (the row count has to reach the host); *rows (nullable): rows in the copy.  A field that also
 * has the full or the tile-major copy uses that.  The coefficients must not change afterwards.  No reference counterpart. */
DG_API dg_status dg_field_build_cell_major_band(dg_field* field, double lo, double hi, void* stream, uint64_t* rows);
/* Optional, for UNREDUCED fields: builds (once, asynchronously on `stream`) a tile-major device copy --
 * for every tile of 4x4x4 cells all the nodes its cells reference, 736 contiguous doubles -- that
 * dg_interpolate_batch* and dg_density_map_nodes* then read.  The reference layout [V|X|Y|Z] puts a cell's
 * 32 coefficients into 16 different cache lines; with this copy the queries of one wavefront (one or two
 * tiles once a batch is in tile order) share a few dozen lines.  Costs 1.64 x the field's memory (1.5 GB at
 * 256^3; the cell-major copy: 4.3 GB) and one pass over the field; results are bit-identical.  Takes
 * precedence over the cell-major copy when both exist.  No reference counterpart.  The coefficient
 * array must not change afterwards (drop and rebuild if it does). */
DG_API dg_status dg_field_build_tile_major(dg_field* field, void* stream);
DG_API dg_status dg_field_drop_tile_major(dg_field* field);

/* phi[q] = interpolate(field, x_q [, &grad_q]); DG_NO_VALUE outside the domain, in removed
 * cells or when a coefficient is DG_NO_VALUE (grad_q is then zero).  grad may be NULL. */
DG_API dg_status dg_interpolate_batch(const dg_field* field, const double* xyz, uint64_t n, double* phi, double* grad);
DG_API dg_status dg_interpolate_batch_device(const dg_field* field, const double* d_xyz, uint64_t n, double* d_phi,
									  double* d_grad, void* stream);

/* ---- reduceField (sparsification of a field) on the device ------------------------------------------ */
/* CubicLagrangeDiscreteGrid::reduceField (cubic_lagrange_discrete_grid.cpp:1065-1174) for an UNREDUCED field
 * (n_coeffs == dg_grid_n_nodes) and a predicate on the node value of one of the two forms the reference's
 * GenerateDensityMap uses (cmd/generate_density_map/main.cpp:138-145):
 *     closed == 0:  lo < v + offset && v - offset < hi        closed != 0:  lo <= v && v <= hi
 * (a node whose value is DG_NO_VALUE never passes, :1090).  On the device: node flags, surviving cells (any of
 * their 32 nodes passes), the nodes those cells reference, their Morton keys with the reference's own
 * arithmetic (zValue :583-601), a radix sort, renumbering of the cell rows.  The reference orders nodes
 * with an unstable std::sort; the result is unique -- and this function reproduces it -- unless two
 * surviving nodes share a key (possible on strongly anisotropic lattices only): then *tied_keys is set, no
 * result can be fetched, and the caller must run the host algorithm (the C++ class does).
 * dg_reduction_fetch copies: coeffs[n_coeffs_out], cells[32 * n_cell_rows], cell_map[dg_grid_n_cells]. */
typedef struct dg_reduction dg_reduction;
DG_API dg_status dg_reduce_field(const dg_grid_desc* grid, const double* coeffs, uint64_t n_coeffs, int closed, double lo,
						  double hi, double offset, dg_reduction** out);
/* The same on the coefficients a field handle already holds on the device (no upload). */
DG_API dg_status dg_reduce_field_device(const dg_field* field, int closed, double lo, double hi, double offset,
								 dg_reduction** out);
/* The reduced field as a handle of its own (table mode): the reduction's device arrays change owner, nothing is
 * copied or uploaded; call dg_reduction_fetch BEFORE this if the host needs the arrays too. */
DG_API dg_status dg_reduction_to_field(dg_reduction* r, dg_field** out);
DG_API dg_status dg_reduction_info(const dg_reduction* r, uint64_t* n_coeffs_out, uint64_t* n_cell_rows, int* tied_keys);
DG_API dg_status dg_reduction_fetch(const dg_reduction* r, double* coeffs, uint32_t* cells, uint32_t* cell_map);
DG_API void dg_reduction_destroy(dg_reduction* r);

/* ---- K3: SPH boundary density map (next row of the path: GenerateDensityMap) ------------------ */
/* out[l - node_begin] = density_func(indexToNodePosition(l)) of cmd/generate_density_map/main.cpp:96-112
 * evaluated on the lattice of `sdf`'s own grid: rho0 * integral over [-h,h]^3 of gamma(x + xi) W(xi),
 * gamma from the SDF field (:86-93), W the cubic spline kernel (sph_kernel.hpp:22-42), 16^3-point
 * Gauss-Legendre rule summed in the reference's order (gauss_quadrature.cpp:5927-5960), 0 where the
 * node is farther than 2h from the surface.  band_predicate != 0 additionally applies the node
 * predicate of main.cpp:119-133 (DG_NO_VALUE outside the band -6h < phi + cell_diag, phi - cell_diag < 2h);
 * pred_mask (nullable, indexed l - node_begin) works as in dg_sdf_sample_nodes. */
DG_API dg_status dg_density_map_nodes(dg_field* sdf, double support_radius, double rho0, int band_predicate,
							   uint64_t node_begin, uint64_t node_end, const uint8_t* pred_mask, double* out);
DG_API dg_status dg_density_map_nodes_device(dg_field* sdf, double support_radius, double rho0, int band_predicate,
									  uint64_t node_begin, uint64_t node_end, const uint8_t* d_pred_mask,
									  double* d_out, void* stream);

/* Progress of the HOST-pointer node-sampling calls issued by the calling thread (dg_sdf_sample_nodes,
 * dg_sdf_sample_nodes_multi): `cb(done, total, user)` is called on that thread as chunks of nodes
 * arrive in the caller's array, at most once per second and once at the end -- what
 * addFunction(verbose) prints (cubic_lagrange_discrete_grid.cpp:819-829).  NULL switches it off. */
typedef void (*dg_progress_fn)(uint64_t done, uint64_t total, void* user);
DG_API void dg_set_progress_callback(dg_progress_fn cb, void* user);

/* ---- instrumentation ------------------------------------------------------------------------ */
/* Device time (HIP events on the launch stream) of the most recent K1 / K2 kernel launch issued
 * by this thread through the HOST entry points, in milliseconds; <0 if none. */
DG_API double dg_last_kernel_ms(void);
/* Number of 4x4x4-node bricks of the most recent node-sampling launch on `mesh` that exhausted
 * their work budget and asked for the split path ("heavy bricks"); at most
 * `*split` of them (the slot count) were actually split, the others ran on in their own wave.
 * Waits for that launch to finish.  Both zero if that launch ran without the split path. */
DG_API dg_status dg_mesh_last_heavy_bricks(const dg_mesh* mesh, uint32_t* heavy, uint32_t* split);
/* Test hook of the filtered sampling kernel's epilogue ("the per-lane double tests, pooled"): with DG_FORCE=pool_stats=1
 * set at launch time, the number of wavefronts of the most recent launch on `mesh` that tested the tails of their lanes' candidate
 * lists POOLED (64 owner / triangle pairs per round) and the number whose tails did not fit the pool and ran lane by lane
 * (DG_FORCE=pool_cap=<pairs> lowers the pool's capacity so that a test reaches that branch).  Both zero without pool_stats.  Waits
 * for the launch.  Nothing to replace in the reference: its per-node loop (TriangleMeshDistance.h:514-562) has no such stage. */
DG_API dg_status dg_mesh_last_epilogue_stats(const dg_mesh* mesh, uint32_t* pooled, uint32_t* lane_by_lane);

#ifdef __cplusplus
}
#endif
#endif /* DISCREGRID_HIP_H */
