// dg_kernels.h -- launch interface between the C ABI (dg_capi*.cpp) and the gfx950 kernels
// (dg_kernels_k1.hip / _k2.hip / _k3.hip / _aux.hip).  Plain C++ structs, no HIP types in the signatures except the stream.
#pragma once
#include <cstdint>
#include <hip/hip_runtime_api.h>
#include "dg_density.h"

namespace dg
{

static const int kMaxRanks = 64;   // shard table size
static const int kSlabPlanes = 4;  // planes per slab == brick depth
static const uint32_t kXcdChunk = 1024; // logical blocks per XCD chunk (see logical_block())
#ifndef DG_WAVES_PER_BLOCK
#define DG_WAVES_PER_BLOCK 1
#endif
static const int kWavesPerBlock = DG_WAVES_PER_BLOCK; // K1: one brick per wave

// Heavy bricks.  A brick whose 64 nodes are (nearly) equidistant from large parts of the surface --
// the centre of a sphere-like mesh is the extreme -- needs the exact test for thousands of
// triangles, serially in ONE wave, while the rest of the chip idles at the end of the launch.
// K1 therefore gives a brick a work budget; a wave that exhausts it parks its running bests in an
// overflow slot and exits, a second kernel walks each of the kSubtrees top-level subtrees of the
// BVH for every parked brick in its own wave (starting from the parked bests), and a third kernel
// takes the per-lane minimum over the subtrees and writes the node values.  min is exact, so the
// result is the one the single wave would have produced.
#ifndef DG_OVERFLOW_SLOTS
#define DG_OVERFLOW_SLOTS 2048
#endif
static const int kOverflowSlots = DG_OVERFLOW_SLOTS; // most bricks one launch can park (12 B x 64 lanes x kSubtrees of scratch each); further heavy bricks simply run on
static const int kHeavyWork = 1600;     // traversal steps + exact triangle tests before a brick counts as heavy (~0.7 ms of one wave)
// Budget of a brick for a mesh of n_positions triangle slots.  Cutting a brick's search over the
// top-level subtrees pays only when it needs a sizeable part of the WHOLE tree; a brick whose 64
// nodes merely face a few thousand triangles of a finely tessellated mesh is local work.
inline int heavy_work_for(int32_t n_positions)
{
	const int scaled = n_positions / 64;
	return scaled > kHeavyWork ? scaled : kHeavyWork;
}
// slots a launch of `bricks` bricks gets: heavy bricks are a small, slowly growing fraction of a launch
inline uint32_t overflow_slots_for(uint64_t bricks)
{
	uint64_t want = bricks / 2048;
	want = want < 1024 ? 1024 : want;   // (197 KB of scratch per slot)
	want = want > bricks ? bricks : want;
	return (uint32_t)(want > (uint64_t)kOverflowSlots ? (uint64_t)kOverflowSlots : want);
}

struct OverflowBuf // device scratch of one K1 launch (null count: splitting disabled)
{
	uint32_t* count;     // slots claimed so far (may run past `slots`)
	uint32_t* brick;     // [slots] brick id parked in the slot
	double* saved_d2;    // [slots][64] running best of every lane when the wave parked
	int32_t* saved_tri;  // [slots][64]
	double* cand_d2;     // [slots][kSubtrees][64] best of every lane within one subtree
	int32_t* cand_tri;   // [slots][kSubtrees][64]
	uint32_t slots;      // <= kOverflowSlots
	int32_t heavy_work;  // work budget of a brick
	// test hooks of the filtered kernel's pooled epilogue (DG_FORCE=pool_stats=1 / pool_cap=<n>): waves that pooled their tail / waves whose
	// tail did not fit the pool and ran lane by lane (stats[0], stats[1]; null: not counted -- 1.85 M atomics on two words would cost
	// more than the kernel), and a cap on the pool below its LDS capacity so that a test can reach the second branch
	uint32_t* stats;
	uint32_t pool_cap;
};
static const int kFastListCap = 10; // candidate triangles a lane can hold (a lane that fills its list gets the exact traversal)
// The filtered traversal counts node steps + triangle PAIRS (the exact one: node steps + exact tests): about the
// same cost per unit, but ~1.7 x as many bricks pass the budget (632 against 381 at 256^3 on the icosphere).
// A factor 2 on the budget brings the count back down and was measured SLOWER on small launches (a 64^3
// lattice ends with the un-parked near-heavy waves running alone: 2.8 against 1.5 ms), so the budget stays and
// the launches get more slots instead (overflow_slots_for).
static const int kFastWorkFactor = 1;
// meshes below this triangle count run the exact kernel: with a few hundred triangles a brick's exact tests
// are fewer than its filter tests + per-lane candidates (box 256^3: 2.6 vs 4.6 ms, 57 600-triangle torus 18.4 vs 17.2)
static const uint64_t kFastMinTriangles = 8192;
// lattices: filtered kernel if 3 x (geometric mean cell edge) >= this x mean triangle edge.  Re-measured at 512^3 with the pooled
// epilogue (round 5; ratio: filtered / exact ms): icosphere 0.693: 87.9 / 90.0, bunny 0.604: 116.2 / 111.5 -- the crossover
// lies near 0.66 (0.8 until round 4, when the filtered kernel lost by 2.8 % at 0.693)
static const double kFastMinBrickRatio = 0.66;
static const int32_t kSeedOnly = -2; // OverflowBuf::saved_tri: saved_d2 is an upper bound of the lane's d^2, no triangle yet
inline size_t overflow_bytes(uint32_t slots, size_t off[6])
{
	size_t o = 0;
	auto take = [&](size_t n) { const size_t at = o; o += (n + 255) & ~(size_t)255; return at; };
	off[0] = take(sizeof(uint32_t));
	off[1] = take(sizeof(uint32_t) * slots);
	off[2] = take(sizeof(double) * 64 * slots);
	off[3] = take(sizeof(int32_t) * 64 * slots);
	off[4] = take(sizeof(double) * 64 * kSubtrees * slots);
	off[5] = take(sizeof(int32_t) * 64 * kSubtrees * slots);
	return o;
}

struct MeshDev
{
	const PairRec* pairs;     // node pairs (dg_geom.h)
	const PairRec* tri_pairs; // triangle bound pairs, position t -> [t / 2] side t & 1
	const TriApproxPair* tri_approx; // float filter data of the triangles, same indexing
	const TriPacket* tris;    // one per position
	const double* pn;         // pseudonormals, kPnSlots x 3 per position
	int32_t root_info;
	int32_t n_positions;
	int32_t stack_levels; // tree depth + 1 (<= kStackDepth): LDS bound-stack levels a traversal can need
	int32_t n_sub;        // entries of sub_roots (1..kSubtrees)
	double origin[3];
	float mesh_l1;
	float pad_;
	int32_t sub_roots[kSubtrees]; // info words of the subtrees the tree is cut into (see "Heavy bricks")
};
static const int kStackDepth = 32; // >= tree depth; 2^32 leaves of >= 1 triangle is beyond the 2^27 triangle limit

// K1p: caller-supplied points instead of lattice nodes (xyz == nullptr: lattice mode).
struct PointsDesc
{
	const double* xyz;        // 3 doubles per point
	uint64_t n;
	const uint32_t* bin_flag; // nullable; *bin_flag != 0: thread t takes point perm[t] (points binned into tiles)
	const uint32_t* perm;
	double* dist;             // outputs, indexed like xyz; tri / entity / nearest nullable
	int32_t* tri;
	int32_t* entity;
	double* nearest;
};

// Division of wave-uniform 32-bit integers by a launch constant.  The GPU has no scalar divide: `n / d` with a
// run-time d becomes a float reciprocal on the VECTOR unit (5 VALU + fix-ups per division, ~60 VALU per brick for
// the brick map of a VALU-bound kernel).  With m = udiv_magic(d) from the host, floor(n m / 2^32) is floor(n / d)
// or one less (n (2^32/d - m) / 2^32 < 1), so one multiply-high and one fix-up -- all scalar -- give the exact
// quotient and remainder for every n < 2^32, d >= 1.
DG_HD uint32_t udiv_magic(uint32_t d)
{
	return d <= 1u ? 0xffffffffu : (uint32_t)(0x100000000ull / d);
}
DG_HD uint32_t udiv_by(uint32_t n, uint32_t d, uint32_t m, uint32_t* rem)
{
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t q = __umulhi(n, m);
#else
	uint32_t q = (uint32_t)(((uint64_t)n * m) >> 32);
#endif
	uint32_t r = n - q * d;
	if (r >= d)
	{
		++q;
		r -= d;
	}
	*rem = r;
	return q;
}

// One of the four node classes of the lattice as the K1 kernel sees it (see dg_geom.h
// node_position() for the (a, b, s) coordinates).  The kernel walks "packed planes"
// q in [q_begin, q_end); plane q is lattice plane s = q (whole-grid / range mode) or the
// q-th plane owned by this rank (shard mode).
struct ClassDesc
{
	uint32_t D0, D1, D2;      // lattice extents (fastest, middle, slowest)
	uint32_t nb0, nb1, nbq;   // bricks of 4x4x4 nodes along a, b, q
	uint32_t rcp_nb0, rcp_nb01; // udiv_magic(nb0), udiv_magic(nb0 * nb1)
	uint32_t q_begin, q_end;
	uint64_t l_begin, l_end;  // valid class-local flat node range (range mode; full range otherwise)
	int64_t out_base;         // out index = out_base + (q*D1 + b)*D0 + a
	uint64_t brick_prefix;    // first brick id of this class in the launch
};

// K3's brick order (SampleParams::brick_blocking): the 128 waves an XCD has in flight integrate over a region of
// +-h around their bricks in lockstep; as a compact block of bricks they sweep (nearly) the same tiles of the field
// at the same time, as 128 bricks of two lattice rows they sweep 64 different tile planes.
#ifndef DG_BLK0
#define DG_BLK0 4
#define DG_BLK1 4
#define DG_BLKQ 8
#endif
static const uint32_t kBlk0 = DG_BLK0, kBlk1 = DG_BLK1, kBlkQ = DG_BLKQ;
struct SampleParams
{
	MeshDev mesh;
	double dmin[3];
	double cell[3];
	ClassDesc cls[4];
	uint64_t total_bricks;
	uint32_t n_blocks;       // ceil(total_bricks / kWavesPerBlock)
	uint32_t blocks_per_xcd; // blocks launched per XCD (multiple of xcd_chunk); grid = 8 * blocks_per_xcd
	uint32_t xcd_chunk;      // consecutive logical blocks that stay on one XCD
	uint32_t rcp_xcd_chunk;  // udiv_magic(xcd_chunk)
	int32_t shard_rank, shard_n; // shard_n == 1: identity plane map
	int32_t invert;
	const uint8_t* mask;     // indexed like out; nullable
	double* out;
	OverflowBuf ovf;
	int32_t filtered;        // K1 / K1p: 1 = the filtered kernel (k_sample_fast), 0 = the exact kernel only
	int32_t brick_blocking;  // 0: bricks of a class in row-major order; 1: in blocks of kBlk0 x kBlk1 x kBlkQ bricks (K3)
	PointsDesc pts;          // K1p: "brick" b = the 64 points (in processing order) b*64 .. b*64+63
};

// Which lattice node does `lane` of brick `brick` own?  Shared by the kernel and by the host-side
// wave emulator used in the CPU tests.
struct LaneNode
{
	int cls;
	uint32_t a, b, s;  // clamped class-lattice coordinates (always a valid node)
	bool valid;        // lane owns a node of the requested range
	int64_t out_idx;
};
// The wave-uniform half of the map: class and brick coordinates of a brick id.  Kernels that need the lane map
// twice (before and after a long traversal) keep this in scalar registers instead of repeating its divisions.
struct BrickMap
{
	int cls;
	uint32_t b0, b1, bq;
};
template <bool BLOCKED>
DG_HD BrickMap map_brick_order(const SampleParams& P, uint64_t brick)
{
	int c = 0;
	if (brick >= P.cls[1].brick_prefix) c = 1;
	if (brick >= P.cls[2].brick_prefix) c = 2;
	if (brick >= P.cls[3].brick_prefix) c = 3;
	const ClassDesc& C = P.cls[c];
	const uint32_t local = (uint32_t)(brick - C.brick_prefix);
	uint32_t b0, b1, bq;
	if (!BLOCKED)
	{
		uint32_t r;
		bq = udiv_by(local, C.nb0 * C.nb1, C.rcp_nb01, &r);
		b1 = udiv_by(r, C.nb0, C.rcp_nb0, &b0);
	}
	else
	{
		// blocked order (K3): consecutive brick ids fill blocks of kBlk0 x kBlk1 x kBlkQ bricks (truncated at the
		// upper faces), blocks in row-major order -- a bijection on [0, nb0 nb1 nbq), all in wave-uniform integers
		const uint32_t slab = C.nb0 * C.nb1 * kBlkQ;
		const uint32_t iq = local / slab, r = local - iq * slab;
		const uint32_t sq = (C.nbq - iq * kBlkQ) < kBlkQ ? (C.nbq - iq * kBlkQ) : kBlkQ;
		const uint32_t row = C.nb0 * kBlk1 * sq;
		const uint32_t i1 = r / row, r2 = r - i1 * row;
		const uint32_t s1 = (C.nb1 - i1 * kBlk1) < kBlk1 ? (C.nb1 - i1 * kBlk1) : kBlk1;
		const uint32_t blk = kBlk0 * s1 * sq;
		const uint32_t i0 = r2 / blk, r3 = r2 - i0 * blk;
		const uint32_t s0 = (C.nb0 - i0 * kBlk0) < kBlk0 ? (C.nb0 - i0 * kBlk0) : kBlk0;
		b0 = i0 * kBlk0 + r3 % s0;
		b1 = i1 * kBlk1 + (r3 / s0) % s1;
		bq = iq * kBlkQ + r3 / (s0 * s1);
	}
	BrickMap m;
	m.cls = c;
	m.b0 = b0;
	m.b1 = b1;
	m.bq = bq;
	return m;
}
// (K1 launches never use the blocked order and instantiate map_brick_order<false> directly: with a run-time
// choice the compiler hoists the blocked order's reciprocals in front of the branch)
DG_HD BrickMap map_brick(const SampleParams& P, uint64_t brick)
{
	return P.brick_blocking == 0 ? map_brick_order<false>(P, brick) : map_brick_order<true>(P, brick);
}
DG_HD LaneNode map_lane(const SampleParams& P, const BrickMap& m, int lane)
{
	LaneNode n;
	const int c = m.cls;
	const ClassDesc& C = P.cls[c];
	const uint32_t a = m.b0 * 4u + (uint32_t)(lane & 3);
	const uint32_t b = m.b1 * 4u + (uint32_t)((lane >> 2) & 3);
	const uint32_t qp = C.q_begin + m.bq * 4u + (uint32_t)(lane >> 4);
	// plane map: identity, or the qp-th plane owned by this rank (slabs of 4 dealt round-robin)
	uint32_t s = qp;
	if (P.shard_n > 1)
		s = ((qp / (uint32_t)kSlabPlanes) * (uint32_t)P.shard_n + (uint32_t)P.shard_rank) * (uint32_t)kSlabPlanes +
			(qp % (uint32_t)kSlabPlanes);
	bool valid = (a < C.D0) && (b < C.D1) && (qp < C.q_end) && (s < C.D2);
	const uint64_t l_class = ((uint64_t)s * C.D1 + b) * C.D0 + a;
	valid = valid && (l_class >= C.l_begin) && (l_class < C.l_end);
	n.cls = c;
	n.a = a < C.D0 ? a : C.D0 - 1;
	n.b = b < C.D1 ? b : C.D1 - 1;
	n.s = s < C.D2 ? s : C.D2 - 1;
	n.valid = valid;
	n.out_idx = C.out_base + (int64_t)(((uint64_t)qp * C.D1 + b) * C.D0 + a);
	return n;
}
DG_HD LaneNode map_lane(const SampleParams& P, uint64_t brick, int lane)
{
	return map_lane(P, map_brick(P, brick), lane);
}

// XCD-aware remap: hardware deals blockIdx round-robin over the 8 XCDs (each with its own L2).
// Logical blocks are cut into chunks of xcd_chunk consecutive blocks and the chunks are dealt
// round-robin to the XCDs: neighbouring bricks -- which walk the same BVH subtrees -- share an
// L2, while every XCD still sees every region of the lattice (expensive regions, e.g. the
// inside of the mesh, do not pile up on one XCD).  Returns false for padding blocks.
DG_HD bool logical_block(const SampleParams& P, uint32_t block_idx, uint32_t* blk)
{
	const uint32_t xcd = block_idx & 7u;
	const uint32_t within = block_idx >> 3;
	uint32_t in_chunk;
	const uint32_t group = udiv_by(within, P.xcd_chunk, P.rcp_xcd_chunk, &in_chunk);
	// the chunk an XCD takes rotates from group to group, so that a group period close to a
	// row/plane period of the lattice cannot pin one XCD to one region
	const uint32_t b = (group * 8u + ((xcd + group) & 7u)) * P.xcd_chunk + in_chunk;
	*blk = b;
	return within < P.blocks_per_xcd && b < P.n_blocks;
}

struct UnpackParams
{
	uint32_t D0[4], D1[4], D2[4];
	uint64_t class_off[5];            // global node offset of each class (+ total)
	uint64_t pack_off[4][kMaxRanks];  // offset of class c inside rank r's packed buffer
	uint64_t count[kMaxRanks];        // nodes of rank r (its packed buffer holds count[r] <= stride values)
	int32_t nranks;
	int32_t rank_begin, rank_end;     // k_unpack_ranks: the slots [rank_begin, rank_end) of `gathered`
	uint64_t stride;
	const double* gathered;
	double* field;
};


// the inverse map k_unpack_ranks uses: global node index of element `off` of rank r's packed buffer
// (off < U.count[r])
DG_HD uint64_t unpack_dest(const UnpackParams& U, uint32_t r, uint64_t off)
{
	int c = 0;
	if (off >= U.pack_off[1][r]) c = 1;
	if (off >= U.pack_off[2][r]) c = 2;
	if (off >= U.pack_off[3][r]) c = 3;
	const uint64_t local = off - U.pack_off[c][r];
	const uint64_t plane = (uint64_t)U.D0[c] * U.D1[c];
	const uint32_t q = (uint32_t)(local / plane);
	const uint64_t inplane = local - (uint64_t)q * plane;
	const uint32_t s = ((q / kSlabPlanes) * (uint32_t)U.nranks + r) * kSlabPlanes + (q % kSlabPlanes);
	return U.class_off[c] + (uint64_t)s * plane + inplane;
}

// K1 (+ the two heavy-brick kernels when p.ovf.count is set; the caller zeroes *p.ovf.count first)
hipError_t launch_sample_nodes(const SampleParams& p, hipStream_t stream);
hipError_t launch_unpack(const UnpackParams& p, hipStream_t stream);
hipError_t launch_unpack_ranks(const UnpackParams& p, hipStream_t stream);
// K3; with p.skip_mode == 2 a check of the n_coeffs coefficients (NaN / Inf / huge values -> *p.unsafe) runs first
hipError_t launch_density_bricks(const SampleParams& layout, const FieldDev& f, uint64_t n_coeffs, const DensityParams& p,
								 hipStream_t stream);
hipError_t launch_expand_cells(const FieldDev& f, uint64_t n_rows, double* d_out, hipStream_t stream);
// the x-major copy of the Y and Z edge classes (dg_lattice.h): xmajor_doubles(f.res) doubles
hipError_t launch_xmajor_copy(const FieldDev& f, double* d_out, hipStream_t stream);
// f.ntile must be set; d_out: n_tiles * kTmNodes doubles
hipError_t launch_expand_tiles(const FieldDev& f, uint64_t n_tiles, double* d_out, hipStream_t stream);
// K2 over the cell-major copy of a field (f.cell_major set), queries in any order, no binning
hipError_t launch_interpolate_rows(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, hipStream_t stream);
hipError_t launch_interpolate(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad,
							  hipStream_t stream);

// K2 with query binning.  Queries in arbitrary order touch 16 scattered cache lines of a GB-sized
// array each; processing them tile by tile (8x8x8 cells, 28 KB of coefficients) lets a wave's
// gathers hit lines its neighbours just fetched.  The decision is taken on the device (a probe of
// the first queries counts how often consecutive queries change tile), so the call stays
// asynchronous: for inputs that are already spatially ordered the binning kernels return at once
// and K2 runs in input order.
struct TileGrid // uniform grid of tiles the points are binned into (points outside are clamped to it)
{
	double origin[3];
	double inv_size[3]; // 1 / tile edge
	uint32_t dims[3];
};
inline uint32_t tile_count(const TileGrid& g) { return g.dims[0] * g.dims[1] * g.dims[2]; }
// Sort key of a tile.  DG_SORT_MORTON=1 (default): the bits of the three tile coordinates interleaved (each axis
// contributes as many bits as its extent needs), so that consecutive keys are spatial neighbours at every
// scale -- tiles that follow each other in the processing order share coefficient lines in y and z as well
// as in x; 0: row-major (x fastest), the first design.
#ifndef DG_SORT_MORTON
#define DG_SORT_MORTON 1
#endif
inline uint32_t axis_bits(uint32_t dim)
{
	uint32_t b = 0;
	while (b < 31 && (1u << b) < dim)
		++b;
	return b;
}
// number of significant key bits (what the radix sort has to look at)
inline uint32_t tile_key_bits(const TileGrid& g)
{
#if DG_SORT_MORTON
	const uint32_t b = axis_bits(g.dims[0]) + axis_bits(g.dims[1]) + axis_bits(g.dims[2]);
#else
	const uint32_t b = axis_bits(tile_count(g));
#endif
	return b < 1 ? 1 : (b > 32 ? 32 : b);
}
DG_HD uint32_t tile_key(const uint32_t dims[3], const uint32_t t[3])
{
#if DG_SORT_MORTON
	uint32_t key = 0, pos = 0;
	for (uint32_t b = 0; b < 11; ++b) // at most 2^11 tiles per axis fit 32 key bits with three full axes; fewer bits on short axes
		for (int d = 0; d < 3; ++d)
			if ((dims[d] - 1u) >> b) // axis d still has bit b
			{
				key |= ((t[d] >> b) & 1u) << (pos & 31u);
				++pos;
			}
	return key;
#else
	return (t[2] * dims[1] + t[1]) * dims[0] + t[0];
#endif
}
// The points are ordered by a radix sort of (tile, index) pairs (rocPRIM).  Two decisions are
// involved: whether the batch is unordered at all -- taken exactly, on the device, by a probe of the
// first points (flag) -- and whether the sort is launched, which only the host can decide and which it
// bases on the probe result of the handle's PREVIOUS batch (flag_host, pinned memory the probe also
// writes; a stale or wrong value costs time, never correctness: the kernels take point perm[t] only if
// the sort was launched AND this batch's flag says unordered).
struct BinScratch
{
	uint32_t* flag;       // [1] device: 1 = this batch is unordered
	uint32_t* flag_host;  // pinned host copy of the flag, read by the NEXT call as its prediction
	uint32_t* keys;       // [n] tile of every point
	uint32_t* keys_out;   // [n]
	uint32_t* vals;       // [n] 0..n-1
	uint32_t* perm;       // [n] point indices in tile order
	void* sort_tmp;       // rocPRIM's temporary storage
	size_t sort_tmp_bytes;
	int sort_launched;    // host decision for this batch
};
// inverse of tile_key(): the tile coordinates of a key
DG_HD void tile_from_key(const uint32_t dims[3], uint32_t key, uint32_t t[3])
{
	t[0] = t[1] = t[2] = 0;
#if DG_SORT_MORTON
	uint32_t pos = 0;
	for (uint32_t b = 0; b < 11; ++b)
		for (int d = 0; d < 3; ++d)
			if ((dims[d] - 1u) >> b)
			{
				t[d] |= ((key >> (pos & 31u)) & 1u) << b;
				++pos;
			}
#else
	t[0] = key % dims[0];
	t[1] = (key / dims[0]) % dims[1];
	t[2] = key / (dims[0] * dims[1]);
#endif
}
// ---- K2 on the plain layout, staged (round 6): counting sort by tile of 8^3 cells + a gather that serves a tile's queries from LDS ----
// Unordered queries on an unreduced field WITHOUT any copy of it.  Rounds 1-5 sorted the queries by single cell (rocPRIM radix
// sort, three passes) and let every lane gather its 16 coefficient pairs from the field: 16 load instructions per wave, each
// touching up to 64 different lines -- the texture-data path's roof (TA / TD busy 0.94 / 0.95, 123 cycles per load instruction),
// 0.72 ms per 10 M queries behind 0.46 ms of binning.  Now:
//  (1) a sort on the 15-bit key of the query's TILE of kStageCells^3 cells (256^3: 32 768 tiles, 305 queries each): two radix
//      passes instead of the three of the 24-bit cell key (rocPRIM onesweep; a counting sort through global atomics -- one
//      returning atomic per query for its rank, a scattered 4-byte store per query -- measured 0.58 + 0.23 ms per 10 M against
//      0.21: its LDS-local reordering is what makes a radix pass cheap), then k_tile_bounds (where each tile's run begins and
//      ends) and k_tile_items (one block: the list of work items -- a tile's queries in chunks of kStageChunk);
//  (2) k_interpolate_tiles: one block per work item copies the tile's part of the field -- 9^3 vertex nodes and 3 x 8 x 9 x 9 edges
//      of two nodes = 4617 doubles (37 KB), whole rows, 16 bytes per lane side by side -- into LDS ONCE and every lane then
//      takes its 32 coefficients from LDS: each coefficient crosses the texture path once per tile visit instead of once per
//      query that uses it (121 instead of 256 bytes per query, in row-sized pieces instead of 16-byte ones).
// Same locate_query / evaluate_cell statements as every other K2 path: same bits.  The key is computed from the query's CELL
// exactly as locate_query computes it (a tile key of its own rounding could put a query next to a tile face into the
// neighbour tile).  Batches too thin for staging to pay (fewer than kStageMinPerTile queries per tile on average) keep the
// per-lane gather behind the radix sort by cell.
static const uint32_t kStageChunk = 1024;   // queries per work item (a tile with more gets several items)
// batches with fewer queries per tile on average take the per-lane gather behind the radix sort by cell.  Measured at 512^3 (262 144 tiles,
// profiles/r06_k2_tile_density.txt; ms staged / per lane): 38 queries per tile 3.18 / 2.80 (value), 3.27 / 3.24 (gradient); 114 per tile
// 4.77 / 5.07, 5.08 / 6.40; at 256^3 (305 per tile) 0.91 / 1.17; at 128^3 (2 441) 0.72 / 0.86
static const uint32_t kStageMinPerTile = 64, kStageMinPerTileGrad = 32;
static const uint32_t kStageMaxBits = 21;      // tile tables up to 2 M entries (16 MB); beyond: the per-lane path
// the tile shape (cells per axis as powers of two) the gather is instantiated for
static const int kStageShapes = 1;
static const uint32_t kStageLog[kStageShapes][3] = {{3, 3, 3}};
size_t bin_sort_tmp_bytes(uint64_t n, uint32_t n_tiles); // rocPRIM's requirement for n pairs (dg_kernels_k2.hip)
struct StageItem // 16 bytes: one load
{
	uint32_t key;    // tile
	uint32_t q0, q1; // its queries [q0, q1) in tile order
	uint32_t tijk;   // tile coordinates, 10 bits each
};
struct TileBin
{
	uint32_t* flag;        // [1] this batch is unordered (k_bin_probe; informational on this path)
	uint32_t* flag_host;   // pinned: the prediction for the handle's next batch
	uint32_t* keys;        // [n] tile key of every query
	uint32_t* keys_out;    // [n] sorted
	uint32_t* perm;        // [n] query indices in tile order
	uint32_t* begin;       // [key_space] first query of the tile in tile order
	uint32_t* end;         // [key_space] one past its last (begin == end == 0: none)
	StageItem* items;      // [max_items]
	uint32_t* n_items;     // [1]
	uint32_t* row_items;   // [key_space / 1024] items per row of 1024 tiles
	double* packed;        // [4 n] value + gradient per query, query order (gradient batches; null: results go straight to the caller's arrays)
	void* sort_tmp;        // rocPRIM's temporary storage
	size_t sort_tmp_bytes;
	uint32_t key_space, key_bits, max_items, n_queries;
	uint32_t tdims[3];     // tiles per axis
	uint32_t tlog[3];      // log2 of the cells per tile and axis
	int shape;             // index into kStageLog
	int sort_launched;
};
inline uint32_t stage_key_bits(const uint32_t res[3], int shape, uint32_t tdims[3], uint32_t tlog[3])
{
	TileGrid g;
	for (int d = 0; d < 3; ++d)
	{
		tlog[d] = kStageLog[shape][d];
		g.dims[d] = tdims[d] = (res[d] + (1u << tlog[d]) - 1) >> tlog[d];
	}
	return tile_key_bits(g);
}
inline uint32_t stage_max_items(uint32_t key_space, uint64_t n)
{
	const uint64_t m = (n < key_space ? n : key_space) + n / kStageChunk + 1;
	return (uint32_t)(m < 0xffffffffull ? m : 0xffffffffull);
}
inline size_t tile_bin_bytes(uint32_t key_space, uint64_t n, size_t off[8], bool packed_results = false)
{
	size_t o = 0;
	auto take = [&](size_t b) { const size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
	off[0] = take(8 + 4 * ((size_t)key_space / 1024 + 1)); // flag, n_items, row_items
	off[1] = take((size_t)n * 4);                       // keys
	off[2] = take((size_t)n * 4);                       // keys_out
	off[3] = take((size_t)n * 4);                       // perm
	off[4] = take((size_t)key_space * 8);               // begin, end (cleared together)
	off[5] = take((size_t)stage_max_items(key_space, n) * sizeof(StageItem));
	off[6] = take(bin_sort_tmp_bytes(n, key_space));
	off[7] = take(packed_results ? (size_t)n * 32 : 0);
	return o;
}
inline void tile_bin_assign(TileBin& B, void* mem, const size_t off[8], uint32_t key_bits, uint64_t n)
{
	char* base = static_cast<char*>(mem);
	const uint32_t key_space = 1u << key_bits;
	B.flag = reinterpret_cast<uint32_t*>(base + off[0]);
	B.n_items = B.flag + 1;
	B.row_items = B.flag + 2;
	B.keys = reinterpret_cast<uint32_t*>(base + off[1]);
	B.keys_out = reinterpret_cast<uint32_t*>(base + off[2]);
	B.perm = reinterpret_cast<uint32_t*>(base + off[3]);
	B.begin = reinterpret_cast<uint32_t*>(base + off[4]);
	B.end = B.begin + key_space;
	B.items = reinterpret_cast<StageItem*>(base + off[5]);
	B.sort_tmp = base + off[6];
	B.packed = nullptr; // (the caller points it at off[7] for gradient batches)
	B.sort_tmp_bytes = bin_sort_tmp_bytes(n, key_space);
	B.key_space = key_space;
	B.key_bits = key_bits;
	B.max_items = stage_max_items(key_space, n);
	B.n_queries = (uint32_t)n;
}
// probe (always; the prediction for the next batch) and, if B.sort_launched, the counting sort + the staged gather; otherwise
// the caller runs the queries in the order they came (launch_interpolate).  xcd_chunk: consecutive work items per XCD (0: off)
hipError_t launch_interpolate_tiles(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, const TileBin& B,
									uint32_t xcd_chunk, hipStream_t stream);
#ifndef DG_TILE_CELLS
#define DG_TILE_CELLS 8
#endif
static const uint32_t kTileCells = DG_TILE_CELLS;
#ifndef DG_SORT_CELLS
#define DG_SORT_CELLS 1
#endif
// K2 sorts by tiles of kSortCells^3 cells (the probe looks at kTileCells^3).  With Morton keys the finest key costs
// no extra radix pass up to 256^3 (24 bits = 3 passes, as 18): 10 M uniform queries 7.5 (4^3 tiles) -> 7.9 Gq/s
// (single cells), shell 8.7 -> 9.4; row-major 4^3 tiles were the round-1 choice (profiles/r02_k2_sort_layout_ab.txt).
static const uint32_t kSortCells = DG_SORT_CELLS;
// K2: tiles of cells^3 grid cells
inline TileGrid field_tiles(const FieldDev& f, uint32_t cells = kTileCells)
{
	TileGrid g;
	for (int d = 0; d < 3; ++d)
	{
		g.origin[d] = f.dmin[d];
		g.inv_size[d] = f.inv_cell[d] / (double)cells;
		g.dims[d] = (f.res[d] + cells - 1) / cells;
	}
	return g;
}
// K1p: about 64 points per tile if the n points fill the box [lo, hi] evenly
inline TileGrid point_tiles(const double lo[3], const double hi[3], uint64_t n)
{
	TileGrid g;
	double per_axis = 1.0;
	while (per_axis * per_axis * per_axis * 64.0 < (double)n && per_axis < 64.0)
		per_axis += 1.0;
	for (int d = 0; d < 3; ++d)
	{
		const double ext = hi[d] > lo[d] ? hi[d] - lo[d] : 1.0;
		g.origin[d] = lo[d];
		g.inv_size[d] = per_axis / ext;
		g.dims[d] = (uint32_t)per_axis;
	}
	return g;
}
// ---- reduceField on the device (cubic_lagrange_discrete_grid.cpp:1065-1174) for value predicates ------------
// keep(v) = closed ? (lo <= v && v <= hi) : (lo < v + offset && v - offset < hi), and v != DBL_MAX: the two
// predicates of the reference's GenerateDensityMap (cmd/generate_density_map/main.cpp:138-145).
struct ReducePredicate
{
	double lo, hi, offset;
	int closed;
};
struct ReduceResult // device arrays owned by the caller of reduce_field_device(); host counts
{
	uint64_t n_nodes_out = 0; // m: nodes referenced by a surviving cell
	uint64_t n_rows = 0;      // surviving cells
	int tied_keys = 0;        // two survivors share a Morton key: the reference's order is then libstdc++'s business
	void* d_coeffs = nullptr;   // [m] f64, Morton order
	void* d_cells = nullptr;    // [n_rows][32] u32, renumbered
	void* d_cell_map = nullptr; // [n_cells] u32, row or 0xffffffff
};
// Unreduced field (closed-form cell rows) with n = all lattice nodes.  Synchronises with `stream`.
hipError_t reduce_field_device(const uint32_t res[3], const double dmin[3], const double cell[3], const double inv_cell[3],
							   const double* d_coeffs, uint64_t n, const ReducePredicate& pred, ReduceResult& out, hipStream_t stream);

size_t bin_sort_tmp_bytes(uint64_t n, uint32_t n_tiles); // rocPRIM's requirement for n pairs
// the binning passes shared by K2 and K1p (dg_kernels_k2.hip): probe (always) and -- if the host predicts an unordered batch
// (S.sort_launched) -- tile keys + radix sort, which leaves the processing order in S.perm
hipError_t launch_binning(const TileGrid& probe_tiles, const TileGrid& tiles, const double* d_xyz, uint64_t n, const BinScratch& S, uint32_t one_in,
						  hipStream_t stream);
inline size_t bin_scratch_bytes(uint32_t n_tiles, uint64_t n, size_t off[6])
{
	size_t o = 0;
	auto take = [&](size_t b) { const size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
	off[0] = take(4);
	off[1] = take((size_t)n * 4);
	off[2] = take((size_t)n * 4);
	off[3] = take((size_t)n * 4);
	off[4] = take((size_t)n * 4);
	off[5] = take(bin_sort_tmp_bytes(n, n_tiles));
	return o;
}
// fills the pointers of S from one scratch allocation laid out by bin_scratch_bytes()
inline void bin_scratch_assign(BinScratch& S, void* mem, const size_t off[6], uint32_t n_tiles, uint64_t n)
{
	char* base = static_cast<char*>(mem);
	S.flag = reinterpret_cast<uint32_t*>(base + off[0]);
	S.keys = reinterpret_cast<uint32_t*>(base + off[1]);
	S.keys_out = reinterpret_cast<uint32_t*>(base + off[2]);
	S.vals = reinterpret_cast<uint32_t*>(base + off[3]);
	S.perm = reinterpret_cast<uint32_t*>(base + off[4]);
	S.sort_tmp = base + off[5];
	S.sort_tmp_bytes = bin_sort_tmp_bytes(n, n_tiles);
}
// K2 through the band-limited cell-major copy (FieldDev::band_rows / band_map) and the copy's builders
hipError_t launch_interpolate_band(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, hipStream_t stream);
// host_counts (pinned, 2 words): sampled queries with a cell / of those with a row in the band copy -- the routing prediction
hipError_t launch_band_probe(const FieldDev& f, const double* d_xyz, uint64_t n, uint32_t* host_counts, hipStream_t stream);
hipError_t launch_band_flags(const FieldDev& f, uint64_t n_rows, double lo, double hi, uint32_t* d_flag, hipStream_t stream);
hipError_t band_scan(const uint32_t* d_flag, uint32_t* d_pos, uint64_t n_rows, void* d_tmp, size_t* tmp_bytes, hipStream_t stream);
hipError_t launch_band_expand(const FieldDev& f, uint64_t n_rows, const uint32_t* d_flag, const uint32_t* d_pos, uint64_t* d_bits, uint32_t* d_rank,
							  double* d_rows, hipStream_t stream);
hipError_t launch_interpolate_binned(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad,
									 const BinScratch& scratch, hipStream_t stream);
// K1p: p.pts describes the points and outputs, p.total_bricks etc. come from layout_points().  The
// packet traversal of a wave costs the UNION of what its 64 points need, so with tiles != nullptr
// points that arrive in arbitrary order (7 x slower than lattice order) are first grouped into
// compact tiles (scratch = the BinScratch behind p.pts.bin_flag / perm).  Heavy "bricks" (64 points
// around the centre of a sphere-like mesh) are split like K1's when p.ovf is set.
hipError_t launch_signed_distance(const SampleParams& p, const TileGrid* tiles, const BinScratch* scratch, hipStream_t stream);


} // namespace dg

#include "dg_density_cells.h" // K3, one lane per lattice point (needs SampleParams)
