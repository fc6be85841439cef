// dg_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the SDF-discretisation hot path.
//
//   K1  k_sample_nodes     lattice node -> signed distance to the mesh   (addFunction node loop,
//                          discregrid/src/cubic_lagrange_discrete_grid.cpp:806-831 +
//                          TriangleMeshDistance.h:269-308, 514-562, 564-820)
//   K1p k_sample_nodes<1>  same kernels for caller-supplied points (TriangleMeshDistance.h:269-314)
//   K2  k_interpolate      batched CubicLagrangeDiscreteGrid::interpolate (:977-1063), with
//       k_bin_*            device-side binning of unordered query batches into 8^3-cell tiles
//   K3  k_density_bricks   SPH boundary density map (cmd/generate_density_map/main.cpp:86-133)
//   U   k_unpack_shards    packed all-gather buffer -> reference node order (multi-GPU),
//       k_unpack_ranks     the same for a range of rank slots (pieced gather)
//       k_expand_cells     cell-major copy of a field (optional K2 layout)
//
// Design of K1 (wave64, no MFMA: this is point-vs-BVH, not a contraction):
//   * ONE WAVEFRONT = ONE 4x4x4 BRICK of lattice nodes.  The 64 query points are spatially
//     compact, so the wave walks the BVH as a packet: control flow is wave-uniform, every bound
//     record (two siblings, 128 B) and triangle packet (128 B) is fetched ONCE per wave through
//     the scalar unit (s_load_dwordx16 into SGPRs) and broadcast to all lanes for free; lanes only
//     differ in their query point and running best.  No per-lane stack, no divergent gathers in
//     the loop.
//   * Near-first traversal with one wave-shared stack (subtree ids in one VGPR, per-lane bounds
//     parked in LDS); bounds = oriented boxes (normal + principal tangents for flat patches),
//     stored as sibling pairs and evaluated two at a time with packed float math (the kernel is
//     VALU-issue bound: 96 % of the issue slots are busy).
//   * Heavy bricks (work budget exhausted) are parked and finished by k_heavy_subtrees /
//     k_heavy_finish, one wave per top-level subtree (dg_kernels.h).
//   * Bound tests in conservative float (they only prune); triangle tests in double with the
//     reference's exact operation order (no FMA contraction) so d^2, the winning feature and
//     the sign reproduce the reference bit for bit.
//   * Positions are computed from the lattice index (nothing is read from HBM but the mesh);
//     the only compulsory HBM traffic is the 8-byte result per node.
//   * blockIdx is remapped so that chunks of 1024 consecutive bricks stay on one XCD (the BVH
//     subtrees they touch share its private 4 MiB L2) while the chunks rotate over the XCDs.
//
// Compile with -ffp-contract=off (parity) -- see discregrid_amd/build.py.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <stdint.h>
#include <algorithm>
#include "dg_kernels.h"
#include "dg_traverse.h"

namespace dg
{
namespace
{

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define DG_CONST_AS __attribute__((address_space(4)))
#ifndef DG_K1_MIN_WAVES
#define DG_K1_MIN_WAVES 8 // K1 is issue bound and hides its scalar-load latency with waves: cap it at 64 VGPRs
#endif
#ifndef DG_K3_WAVES
#define DG_K3_WAVES 1 // min waves per SIMD requested for K3 (register budget)
#endif

// Wave-uniform loads through the scalar data cache.  The address must be uniform across the
// wave (callers pass indices that went through readfirstlane); the data is immutable for the
// lifetime of the kernel.
__device__ __forceinline__ v8i sload8(const void* p)
{
	return *(const DG_CONST_AS v8i*)(uintptr_t)p;
}
__device__ __forceinline__ v16i sload16(const void* p)
{
	return *(const DG_CONST_AS v16i*)(uintptr_t)p;
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double pack_double(int lo, int hi)
{
	return __hiloint2double(hi, lo);
}

// ---- the device's wave context of the packet traversal (dg_traverse.h) -------------------------------------------------------
// One lane per thread; everything wave-uniform goes through the SCALAR unit: a bound record (two siblings, 128 B), a
// triangle packet (128 B) or a filter record (192 B) is fetched ONCE per wave with s_load_dwordx16 into SGPRs and is an
// operand of every lane's vector instruction for free.  The shared stack: the info word of entry i lives in lane i of one
// VGPR (pushed with a lane select, popped with v_readlane under the wave-uniform stack pointer), the lanes' bounds for the
// entry are parked in LDS -- as floats or, in the filtered kernel, whose LDS also holds the candidate lists, as the upper
// 16 bits of the float (truncation = a lower bound of a non-negative value, relative loss < 2^-7): LDS per wave decides
// how many waves a CU holds.
struct SPair // a pair record in SGPRs: the interleaved bound floats + the two info words
{
	float r[kPairFloats];
	int info0, info1;
};
struct SApprox // a filter record in SGPRs
{
	float r[kApproxFloats];
	int valid0, valid1;
};
__device__ __forceinline__ void park_bound(float* p, int i, float lb) { p[i] = lb; }
__device__ __forceinline__ float parked_bound(const float* p, int i) { return p[i]; }
__device__ __forceinline__ void park_bound(uint16_t* p, int i, float lb) { p[i] = (uint16_t)(__float_as_uint(lb) >> 16); }
__device__ __forceinline__ float parked_bound(const uint16_t* p, int i) { return __uint_as_float((uint32_t)p[i] << 16); }
typedef __attribute__((address_space(3))) int lds_int_t;

template <class StackT>
struct DevWave
{
	static constexpr int kLanes = 1;
	typedef SPair Pair;
	typedef SApprox Approx;
	StackT* lds_lb; // [M.stack_levels][64] of this wave
	int lane_id;
	int stackv = 0; // info words: lane i holds entry i
	__device__ __forceinline__ DevWave(StackT* lds, int lane) : lds_lb(lds), lane_id(lane) {}
	template <class F>
	__device__ __forceinline__ void lanes(F f) const { f(0); }
	template <class P>
	__device__ __forceinline__ unsigned long long ballot(P p) const { return __ballot(p(0)); }
	__device__ __forceinline__ SPair load_pair(const PairRec* base, int idx) const
	{
		const char* p = (const char*)(base + idx);
		SPair s;
		const v16i a = sload16(p);
		const v16i b = sload16(p + 64);
#pragma unroll
		for (int i = 0; i < 16; ++i)
			s.r[i] = __int_as_float(a[i]);
#pragma unroll
		for (int i = 0; i < 14; ++i)
			s.r[16 + i] = __int_as_float(b[i]);
		s.info0 = b[14];
		s.info1 = b[15];
		return s;
	}
	__device__ __forceinline__ TriRegs load_tri(const TriPacket* tris, int t) const
	{
		const char* base = (const char*)(tris + t);
		const v16i a = sload16(base);
		const v16i b = sload16(base + 64);
		TriRegs T;
		T.v0x = pack_double(a[0], a[1]), T.v0y = pack_double(a[2], a[3]), T.v0z = pack_double(a[4], a[5]);
		T.e0x = pack_double(a[6], a[7]), T.e0y = pack_double(a[8], a[9]), T.e0z = pack_double(a[10], a[11]);
		T.e1x = pack_double(a[12], a[13]), T.e1y = pack_double(a[14], a[15]), T.e1z = pack_double(b[0], b[1]);
		T.a00 = pack_double(b[2], b[3]), T.a01 = pack_double(b[4], b[5]), T.a11 = pack_double(b[6], b[7]);
		T.det = pack_double(b[8], b[9]), T.inv_det = pack_double(b[10], b[11]);
		T.denom = pack_double(b[12], b[13]);
		return T;
	}
	__device__ __forceinline__ SApprox load_approx(const TriApproxPair* recs, int idx) const
	{
		const char* base = (const char*)(recs + idx);
		const v16i a = sload16(base);
		const v16i b = sload16(base + 64);
		const v16i c = sload16(base + 128);
		SApprox s;
#pragma unroll
		for (int i = 0; i < 16; ++i)
		{
			s.r[i] = __int_as_float(a[i]);
			s.r[16 + i] = __int_as_float(b[i]);
		}
#pragma unroll
		for (int i = 0; i < kApproxFloats - 32; ++i)
			s.r[32 + i] = __int_as_float(c[i]);
		s.valid0 = c[14];
		s.valid1 = c[15];
		return s;
	}
	template <class B>
	__device__ __forceinline__ void push(int sp, int info, B lb)
	{
		stackv = (lane_id == sp) ? info : stackv;
		park_bound(lds_lb, sp * 64 + lane_id, lb(0));
	}
	__device__ __forceinline__ float parked(int sp, int) const { return parked_bound(lds_lb, sp * 64 + lane_id); }
	__device__ __forceinline__ int info(int sp) const { return __builtin_amdgcn_readlane(stackv, sp); }
	__device__ __forceinline__ uint32_t claim(uint32_t* counter) const
	{
		int slot = 0;
		if (lane_id == 0)
			slot = (int)atomicAdd(counter, 1u);
		return (uint32_t)uniform(slot);
	}
	__device__ __forceinline__ void list_store(uint32_t slot, int v) const { *(lds_int_t*)(uintptr_t)slot = v; }
	// (the emulator's counters)
	__device__ __forceinline__ void note_pair_step(const MeshDev&, int) const {}
	__device__ __forceinline__ void note_leaf(int, int) const {}
	__device__ __forceinline__ void note_leaf_pair() const {}
	__device__ __forceinline__ void note_tri_test(int, bool) const {}
	__device__ __forceinline__ void note_pop() const {}
	__device__ __forceinline__ void note_stale_pop() const {}
	__device__ __forceinline__ void note_filter_pair() const {}
	__device__ __forceinline__ void note_filter_rest() const {}
	__device__ __forceinline__ void note_append(bool) const {}
};

// the exact traversal of the subtree `start` by this wave (dg_traverse.h: packet_walk + ExactWalk); returns the heavy slot
// the wave claimed, or -1 when the subtree was searched to the end
template <class StackT>
__device__ __forceinline__ int traverse(const MeshDev& M, LaneQuery& q, StackT* lds_lb, int start, uint32_t* ovf_count, uint32_t ovf_slots,
										int heavy_work)
{
	DevWave<StackT> w(lds_lb, (int)__lane_id());
	auto lane_query = [&](int) -> LaneQuery& { return q; };
	ExactWalk<DevWave<StackT>, decltype(lane_query)> pol(lane_query);
	return packet_walk(w, pol, M, start, ovf_count, ovf_slots, heavy_work);
}

struct DeviceSqrt
{
	__device__ __forceinline__ double operator()(double x) const { return sqrt(x); } // correctly rounded (OCML)
};
__device__ __forceinline__ LaneResult finish(const MeshDev& M, const LaneQuery& q)
{
	return finish_query(M.tris, M.pn, q, DeviceSqrt());
}

// What one lane of a K1 wave works on: a lattice node of the wave's brick (POINTS = false) or one of
// 64 consecutive caller-supplied points (K1p, POINTS = true; "consecutive" in processing order, i.e.
// through SampleParams::pts.perm when the points were binned).
struct LaneTask
{
	bool valid;     // the lane owns a result
	bool sample;    // ... and has to compute it (not masked off)
	int64_t out_idx;
	double x0, x1, x2;
};
// (`bm`: the brick's wave-uniform map for lattice launches -- map_brick(P, brick); unused for points)
template <bool POINTS>
__device__ __forceinline__ LaneTask lane_task(const SampleParams& P, uint64_t brick, const BrickMap& bm, int lane)
{
	LaneTask t;
	if (POINTS)
	{
		const uint64_t slot = brick * 64u + (uint64_t)lane;
		t.valid = slot < P.pts.n;
		uint64_t i = t.valid ? slot : P.pts.n - 1;
		if (P.pts.perm != nullptr && P.pts.bin_flag[0] != 0u) // perm is set only if this batch was sorted
			i = P.pts.perm[i];
		t.sample = t.valid;
		t.out_idx = (int64_t)i;
		t.x0 = P.pts.xyz[3 * i];
		t.x1 = P.pts.xyz[3 * i + 1];
		t.x2 = P.pts.xyz[3 * i + 2];
	}
	else
	{
		const LaneNode ln = map_lane(P, bm, lane);
		t.valid = ln.valid;
		t.out_idx = ln.out_idx;
		t.sample = ln.valid;
		if (ln.valid && P.mask != nullptr)
			t.sample = P.mask[ln.out_idx] != 0;
		// masked-off lanes still carry a sane (clamped) point; they never hit anything
		double x[3];
		node_position(ln.cls, ln.a, ln.b, ln.s, P.dmin, P.cell, x);
		t.x0 = x[0];
		t.x1 = x[1];
		t.x2 = x[2];
	}
	return t;
}
template <bool POINTS>
__device__ __forceinline__ LaneTask lane_task(const SampleParams& P, uint64_t brick, int lane)
{
	BrickMap bm = {};
	if (!POINTS)
		bm = map_brick_order<false>(P, brick); // (K1 launches: row-major brick order, launch_k1 rejects the blocked one)
	return lane_task<POINTS>(P, brick, bm, lane);
}
// epilogue: the lane's result(s)
template <bool POINTS>
__device__ __forceinline__ void write_result(const SampleParams& P, const LaneTask& t, const LaneQuery& q)
{
	if (!t.valid)
		return;
	const bool hit = t.sample && q.best_tri >= 0;
	if (POINTS)
	{
		const int64_t i = t.out_idx;
		if (!hit)
		{
			P.pts.dist[i] = 1.7976931348623157e308;
			if (P.pts.tri) P.pts.tri[i] = -1;
			if (P.pts.entity) P.pts.entity[i] = -1;
			return;
		}
		const LaneResult r = finish(P.mesh, q);
		P.pts.dist[i] = r.signed_dist;
		if (P.pts.tri) P.pts.tri[i] = r.tri_id;
		if (P.pts.entity) P.pts.entity[i] = r.entity;
		if (P.pts.nearest)
		{
			P.pts.nearest[3 * i] = r.nearest[0];
			P.pts.nearest[3 * i + 1] = r.nearest[1];
			P.pts.nearest[3 * i + 2] = r.nearest[2];
		}
	}
	else
	{
		double v = 1.7976931348623157e308; // predicate-rejected node (:817)
		if (hit)
		{
			const LaneResult r = finish(P.mesh, q);
			v = P.invert ? -1.0 * r.signed_dist : r.signed_dist;
		}
		P.out[t.out_idx] = v;
	}
}

// ------------------------------------------------------------------------------------------------
// K1 / K1p: one wave per 4x4x4 brick of one node class, or per 64 points.
// ------------------------------------------------------------------------------------------------
// the exact traversal of one brick by one wave (double test on every triangle some lane may need)
template <bool POINTS>
__device__ __forceinline__ void sample_brick_exact(const SampleParams& P, uint64_t brick, int lane, float* lds_lb)
{
	const LaneTask t = lane_task<POINTS>(P, brick, lane);
	LaneQuery q;
	init_query(P.mesh.origin, P.mesh.mesh_l1, t.sample, t.x0, t.x1, t.x2, q);
	if (__ballot(t.sample) != 0ull)
	{
		const int slot = traverse(P.mesh, q, lds_lb, P.mesh.root_info, P.ovf.count, P.ovf.slots, P.ovf.heavy_work);
		if (slot >= 0) // heavy brick: park the running bests, k_heavy_subtrees / k_heavy_finish take over
		{
			if (lane == 0)
				P.ovf.brick[slot] = (uint32_t)brick;
			P.ovf.saved_d2[slot * 64 + lane] = q.best_d2;
			P.ovf.saved_tri[slot * 64 + lane] = q.best_tri;
			return;
		}
	}
	write_result<POINTS>(P, t, q);
}

template <bool POINTS>
__global__ __launch_bounds__(64 * kWavesPerBlock, DG_K1_MIN_WAVES) void k_sample_nodes(const SampleParams P)
{
	uint32_t blk;
	if (!logical_block(P, blockIdx.x, &blk)) // XCD-aware remap, dg_kernels.h
		return;
	const int wave = uniform((int)(threadIdx.x >> 6));
	const int lane = (int)(threadIdx.x & 63u);
	const uint64_t brick = (uint64_t)blk * (uint64_t)kWavesPerBlock + (uint64_t)wave;
	if (brick >= P.total_bricks)
		return;
	extern __shared__ __attribute__((aligned(16))) float lds_lb[]; // [waves][stack_levels][64]
	sample_brick_exact<POINTS>(P, brick, lane, lds_lb + wave * (P.mesh.stack_levels * 64));
}

// ------------------------------------------------------------------------------------------------
// K1 / K1p, filtered: the same packet traversal, but a visited leaf's triangles go through the FLOAT
// filter (dg_geom.h: tri_approx_frame / tri_approx_rest, two triangles per record with packed math) instead of a bound
// test plus the double test.  Every lane keeps an upper bound U of its minimum d^2 (what the traversal
// prunes with) and, in LDS, the list of triangles whose interval [q - err, q + err] reaches below U:
// the only ones that can attain the lane's minimum.  After the traversal each lane runs the double
// test on ITS OWN candidates (typically 1-2, six around a mesh vertex) -- instead of the whole wave
// running it on every triangle any of its lanes was interested in (29 of 36 per brick improved some
// lane, hardly ever more than a few lanes each).  Bit-exactness: the triangle with the smallest double
// d^2 is always among the lane's candidates (error analysis in dg_geom.h), and the winner among the
// candidates is found with the double test in list order (strict <), as before.
// Lanes the filter cannot serve (list full, coordinates outside the filter's range, a degenerate triangle
// met) get the exact traversal in the same wave, pruned from the start by their upper bounds; a brick whose
// work budget runs out is parked as a heavy brick with the upper bounds as seeds.
// ------------------------------------------------------------------------------------------------
// returns -1 (searched to the end), -2 (searched to the end, but a degenerate triangle was met: the lists are
// incomplete) or the heavy slot the wave claimed
__device__ __forceinline__ int traverse_fast(const MeshDev& M, FastLane& f, uint16_t* lds_lb, uint32_t list_base /* f.slot of an empty list */,
											 uint32_t* ovf_count, uint32_t ovf_slots, int heavy_work)
{
	DevWave<uint16_t> w(lds_lb, (int)__lane_id());
	auto lane_state = [&](int) -> FastLane& { return f; };
	auto lane_list = [&](int) { return list_base; };
	FastWalk<DevWave<uint16_t>, decltype(lane_state), decltype(lane_list)> pol(lane_state, lane_list);
	const int parked = packet_walk(w, pol, M, M.root_info, ovf_count, ovf_slots, kFastWorkFactor * heavy_work);
	if (parked >= 0)
		return parked;
	return pol.degenerate ? -2 : -1;
}

template <bool POINTS>
__global__ __launch_bounds__(64, DG_K1_MIN_WAVES) void k_sample_fast(const SampleParams P)
{
	uint32_t blk;
	if (!logical_block(P, blockIdx.x, &blk))
		return;
	const int lane = (int)threadIdx.x;
	const uint64_t brick = (uint64_t)blk;
	if (brick >= P.total_bricks)
		return;
	extern __shared__ __attribute__((aligned(16))) float lds_lb[]; // (one extern array per module: the same name in every kernel)
	uint16_t* lds_lb16 = (uint16_t*)lds_lb;                     // [stack_levels][64] bounds (16 bit), then [kFastListCap + 1][64] candidates
	int* lds_list = (int*)(lds_lb16 + P.mesh.stack_levels * 64);
	FastLane f;
	bool sample;
	BrickMap bm = {}; // wave-uniform, kept in scalar registers across the traversal for the second lane_task below
	if (!POINTS)
		bm = map_brick_order<false>(P, brick);
	{
		const LaneTask t = lane_task<POINTS>(P, brick, bm, lane);
		sample = t.sample;
		f.a = make_approx_lane(t.x0 - P.mesh.origin[0], t.x1 - P.mesh.origin[1], t.x2 - P.mesh.origin[2], P.mesh.mesh_l1);
	}
	// `exact`: lanes the filter cannot serve -- outside its range (or NaN) from the start, later those
	// whose list filled up -- get the exact traversal below, in this wave, with only them active
	bool exact = sample && !(f.a.E < __builtin_inff());
	const uint32_t list_base = (uint32_t)(uintptr_t)(lds_int_t*)(lds_list + lane); // LDS byte address of the lane's entry 0
	init_fast_lane(f, sample && !exact, list_base);
	if (__ballot(sample && !exact) != 0ull)
	{
		const int slot = traverse_fast(P.mesh, f, lds_lb16, list_base, P.ovf.count, P.ovf.slots, P.ovf.heavy_work);
		if (slot >= 0) // heavy brick: park the lanes' upper bounds as seeds, k_heavy_subtrees / k_heavy_finish take over
		{
			if (lane == 0)
				P.ovf.brick[slot] = (uint32_t)brick;
			P.ovf.saved_d2[slot * 64 + lane] = exact ? 1.7976931348623157e308 : (double)f.U;
			P.ovf.saved_tri[slot * 64 + lane] = kSeedOnly;
			return;
		}
		// a degenerate triangle was met: nobody's list is complete
		exact = exact || (sample && (slot == -2 || f.slot >= list_base + 256u * (uint32_t)kFastListCap));
	}
	const LaneTask t = lane_task<POINTS>(P, brick, bm, lane);
	LaneQuery q;
	double ex_d2 = 1.7976931348623157e308; // what the exact traversal found (lanes with `exact` only)
	int ex_tri = -1;
	if (__ballot(exact) != 0ull)
	{
		// Exact traversal for the few lanes that need it, pruned from the start by their upper bounds (so it
		// only meets what lies within those lanes' distance); the fast traversal's bound stack is free by now.
		init_query(P.mesh.origin, P.mesh.mesh_l1, exact, t.x0, t.x1, t.x2, q);
		if (exact && f.U > 0.0f) // (U > 0 always for a lane that went through the filter; -inf if it did not)
			q.bestf = best_as_float((double)f.U);
		const int slot = traverse(P.mesh, q, lds_lb16, P.mesh.root_info, P.ovf.count, P.ovf.slots, P.ovf.heavy_work);
		if (slot >= 0) // over budget after all: the heavy-brick kernels redo every lane, seeded with what is known
		{
			if (lane == 0)
				P.ovf.brick[slot] = (uint32_t)brick;
			const bool have = exact && q.best_tri >= 0;
			P.ovf.saved_d2[slot * 64 + lane] = have ? q.best_d2 : (exact ? 1.7976931348623157e308 : (double)f.U);
			P.ovf.saved_tri[slot * 64 + lane] = have ? q.best_tri : kSeedOnly;
			return;
		}
		ex_d2 = q.best_d2;
		ex_tri = q.best_tri;
	}
	// each of the other lanes: the double test on its own candidates, in list (= traversal) order
	init_query(P.mesh.origin, P.mesh.mesh_l1, t.sample, t.x0, t.x1, t.x2, q);
	const int n_cand = (sample && !exact) ? (int)((f.slot - list_base) >> 8) : 0;
	// wave totals: T candidates in all, the longest list; `before` = candidates of the lanes below this one
	uint32_t before = 0, T = 0;
	int longest = 0;
	unsigned long long holders = ~0ull; // lanes whose count agrees with `longest` in the bits seen so far
#pragma unroll
	for (int b = 3; b >= 0; --b) // n_cand <= kFastListCap < 16
	{
		const unsigned long long m = __ballot(((n_cand >> b) & 1) != 0);
		before += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
		T += (uint32_t)__popcll(m) << b;
		if ((m & holders) != 0ull)
		{
			longest |= 1 << b;
			holders &= m;
		}
	}
	const uint32_t item_cap = (uint32_t)P.mesh.stack_levels * 32u, res_cap = (uint32_t)(kFastListCap + 1) * 32u;
	const uint32_t rounds = (T + 63u) >> 6;
#ifndef DG_EPILOGUE_COMPACT
#define DG_EPILOGUE_COMPACT 0 // measured 1.1 % SLOWER than the lane-by-lane loop on the judged workload (same box A/B): kept as a variant
#endif
	if (DG_EPILOGUE_COMPACT && rounds + 1u < (uint32_t)longest && T <= item_cap && T <= res_cap)
	{
		// Transposed form.  Per-lane lists are short but uneven (a lane next to a mesh vertex holds six candidates,
		// its neighbours one or two): lane by lane the wave would run `longest` rounds of the double test, most of
		// them for a handful of lanes.  Instead the (lane, triangle) pairs are laid out contiguously (the bound stack's
		// LDS is free by now), the 64 lanes test 64 pairs per round -- every lane fetches the point of the pair's
		// owner with ds_bpermute --, the values go back through LDS (over the lists, which are no longer needed)
		// and every owner takes the minimum of its own pairs in list order: the same values, the same winner.
		uint32_t* items = (uint32_t*)lds_lb16;
		double* res = (double*)lds_list;
		for (int k = 0; __ballot(k < n_cand) != 0ull; ++k)
			if (k < n_cand)
				items[before + (uint32_t)k] = ((uint32_t)lane << 26) | (uint32_t)lds_list[k * 64 + lane];
		__syncthreads();
		for (uint32_t base = 0; base < T; base += 64u)
		{
			const uint32_t i = base + (uint32_t)lane;
			const bool active = i < T;
			const uint32_t item = items[active ? i : 0u];
			const int owner = (int)(item >> 26);
			const double opx = __shfl(q.px, owner), opy = __shfl(q.py, owner), opz = __shfl(q.pz, owner);
			if (active)
				res[i] = tri_closest<false>(P.mesh.tris[item & 0x3ffffffu], opx, opy, opz).d2;
		}
		__syncthreads();
		for (int k = 0; __ballot(k < n_cand) != 0ull; ++k)
			if (k < n_cand)
				offer(q, res[before + (uint32_t)k], (int)(items[before + (uint32_t)k] & 0x3ffffffu));
	}
	else
	{
		for (int k = 0; __ballot(k < n_cand) != 0ull; ++k)
		{
			if (k < n_cand)
			{
				const int tri = lds_list[k * 64 + lane];
				const Hit h = tri_closest<false>(P.mesh.tris[tri], q.px, q.py, q.pz);
				offer(q, h.d2, tri);
			}
		}
	}
	if (exact)
	{
		q.best_d2 = ex_d2;
		q.best_tri = ex_tri;
	}
	write_result<POINTS>(P, t, q);
}

// Heavy bricks, step 2: job (slot, s) searches subtree s of the BVH for the brick parked in `slot`,
// starting from the parked bests.  One wave per block, jobs dealt grid-stride (the number of parked
// bricks is only known on the device).
template <bool POINTS>
__global__ __launch_bounds__(64) void k_heavy_subtrees(const SampleParams P)
{
	const uint32_t n_sub = (uint32_t)P.mesh.n_sub;
	const uint32_t parked = min(*P.ovf.count, P.ovf.slots);
	const int lane = (int)threadIdx.x;
	extern __shared__ __attribute__((aligned(16))) float lds_lb[];
	for (uint32_t job = blockIdx.x; job < parked * n_sub; job += gridDim.x)
	{
		const uint32_t slot = job / n_sub;
		const uint32_t s = job - slot * n_sub;
		const LaneTask t = lane_task<POINTS>(P, (uint64_t)P.ovf.brick[slot], lane);
		LaneQuery q;
		init_query(P.mesh.origin, P.mesh.mesh_l1, t.sample, t.x0, t.x1, t.x2, q);
		const int tri = P.ovf.saved_tri[slot * 64 + lane];
		if (t.sample && tri >= 0)
			offer(q, P.ovf.saved_d2[slot * 64 + lane], tri);
		else if (t.sample && tri == kSeedOnly) // parked by the filtered kernel: an upper bound, no triangle yet
			q.bestf = fmin2(q.bestf, best_as_float(P.ovf.saved_d2[slot * 64 + lane]));
		traverse(P.mesh, q, lds_lb, P.mesh.sub_roots[s], nullptr, 0u, 0);
		const size_t at = ((size_t)slot * kSubtrees + s) * 64 + (size_t)lane;
		P.ovf.cand_d2[at] = q.best_d2;
		P.ovf.cand_tri[at] = q.best_tri;
	}
}

// Heavy bricks, step 3: per lane the minimum over the subtrees (the parked best is part of every
// candidate; of exactly tied candidates the lowest subtree wins), then K1's epilogue.
template <bool POINTS>
__global__ __launch_bounds__(64) void k_heavy_finish(const SampleParams P)
{
	const uint32_t slot = blockIdx.x;
	if (slot >= min(*P.ovf.count, P.ovf.slots))
		return;
	const int lane = (int)threadIdx.x;
	const LaneTask t = lane_task<POINTS>(P, (uint64_t)P.ovf.brick[slot], lane);
	LaneQuery q;
	init_query(P.mesh.origin, P.mesh.mesh_l1, t.sample, t.x0, t.x1, t.x2, q);
	if (t.sample)
		for (int s = 0; s < P.mesh.n_sub; ++s)
		{
			const size_t at = ((size_t)slot * kSubtrees + (size_t)s) * 64 + (size_t)lane;
			const int tri = P.ovf.cand_tri[at];
			if (tri >= 0)
				offer(q, P.ovf.cand_d2[at], tri);
		}
	write_result<POINTS>(P, t, q);
}

// ------------------------------------------------------------------------------------------------
// U: gathered packed shards -> reference node order.  One thread per node, coalesced stores,
// reads are contiguous runs of one plane row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unpack_shards(const UnpackParams P)
{
	const uint64_t total = P.class_off[4];
	for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < total;
		 l += (uint64_t)gridDim.x * blockDim.x)
	{
		int c = 0;
		if (l >= P.class_off[1]) c = 1;
		if (l >= P.class_off[2]) c = 2;
		if (l >= P.class_off[3]) c = 3;
		const uint64_t lc = l - P.class_off[c];
		const uint64_t plane = (uint64_t)P.D0[c] * P.D1[c];
		const uint32_t s = (uint32_t)(lc / plane);
		const uint64_t inplane = lc - (uint64_t)s * plane;
		const uint32_t slab = s / kSlabPlanes;
		const uint32_t r = slab % (uint32_t)P.nranks;
		const uint32_t q = (slab / (uint32_t)P.nranks) * kSlabPlanes + (s % kSlabPlanes);
		P.field[l] = P.gathered[(uint64_t)r * P.stride + P.pack_off[c][r] + (uint64_t)q * plane + inplane];
	}
}

// ------------------------------------------------------------------------------------------------
// K2: one thread per query point.  The 32-term sum must run in j order for parity, so the
// evaluation is per-lane; the 32 coefficients are fetched as 16 x 16-byte pairs (closed-form
// rows) with all loads issued before the first use.
// ------------------------------------------------------------------------------------------------
// XCD-aware block order for K2: the hardware deals consecutive workgroups to the 8 XCDs in turn, so eight
// neighbouring blocks of queries -- which, in tile order, gather from the same coefficient lines -- would
// each pull those lines into a different L2.  Chunks of kK2XcdChunk consecutive LOGICAL blocks (one
// chunk = what an XCD holds in flight) go to one XCD instead.  Returns false for padding blocks.
#ifndef DG_K2_XCD_CHUNK
#define DG_K2_XCD_CHUNK 256
#endif
static const uint32_t kK2XcdChunk = DG_K2_XCD_CHUNK;
__device__ __forceinline__ bool k2_logical_block(uint32_t block_idx, uint32_t n_blocks, uint32_t* blk)
{
	if (kK2XcdChunk == 0)
	{
		*blk = block_idx;
		return block_idx < n_blocks;
	}
	const uint32_t xcd = block_idx & 7u, within = block_idx >> 3;
	const uint32_t b = ((within / kK2XcdChunk) * 8u + xcd) * kK2XcdChunk + within % kK2XcdChunk;
	*blk = b;
	return b < n_blocks;
}
static uint32_t k2_grid(uint64_t n)
{
	const uint32_t blocks = (uint32_t)((n + 255) / 256);
	if (kK2XcdChunk == 0)
		return blocks;
	const uint32_t round = 8u * kK2XcdChunk;
	return (blocks + round - 1) / round * round;
}

// Occupancy matters more than anything else for this gather-latency bound kernel: left alone the compiler keeps all
// 32 coefficients AND all 32 shape functions in registers (132 / 154 VGPRs, 3 waves per SIMD); asked for more
// waves it forms the shape functions where they are consumed.
#ifndef DG_K2_WAVES
#define DG_K2_WAVES 3
#endif
template <bool GRAD, int MODE>
__global__ __launch_bounds__(256, DG_K2_WAVES) void k_interpolate(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
													  double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	uint32_t blk;
	if (!k2_logical_block(blockIdx.x, (uint32_t)((n + 255) / 256), &blk))
		return;
	const uint64_t gid = (uint64_t)blk * blockDim.x + threadIdx.x;
	if (gid >= n)
		return;
	const double x[3] = {xyz[3 * gid], xyz[3 * gid + 1], xyz[3 * gid + 2]};
	double g[3];
	phi_out[gid] = interpolate_point_mode<GRAD, MODE>(F, x, g);
	if (GRAD)
	{
		grad_out[3 * gid] = g[0];
		grad_out[3 * gid + 1] = g[1];
		grad_out[3 * gid + 2] = g[2];
	}
}

// ---- K2 query binning (dg_kernels.h: BinScratch) ---------------------------------------------------------
__device__ __forceinline__ uint32_t tile_of(const TileGrid& G, const double* __restrict__ xyz, uint64_t i)
{
	uint32_t t[3];
#pragma unroll
	for (int d = 0; d < 3; ++d)
	{
		const double u = (xyz[3 * i + d] - G.origin[d]) * G.inv_size[d];
		uint32_t c = u > 0.0 ? (uint32_t)(u < 4.0e9 ? u : 4.0e9) : 0u; // NaN -> 0
		t[d] = c < G.dims[d] ? c : G.dims[d] - 1;
	}
	return tile_key(G.dims, t);
}
// one block: how often do consecutive queries (among the first 4096) change tile?
__global__ __launch_bounds__(256) void k_bin_probe(const TileGrid F, const double* __restrict__ xyz, uint64_t n, BinScratch S, uint32_t one_in)
{
	__shared__ uint32_t changes;
	if (threadIdx.x == 0)
		changes = 0;
	__syncthreads();
	const uint64_t m = n < 4096 ? n : 4096;
	uint32_t mine = 0;
	for (uint64_t i = threadIdx.x; i + 1 < m; i += blockDim.x)
		mine += tile_of(F, xyz, i) != tile_of(F, xyz, i + 1);
	atomicAdd(&changes, mine);
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const uint32_t unordered = ((uint64_t)one_in * changes > m) ? 1u : 0u; // more than one change of tile in `one_in` steps
		S.flag[0] = unordered;
		*(volatile uint32_t*)S.flag_host = unordered; // prediction for the handle's next batch
	}
}
// sort keys: tile of every point, values: the point indices
__global__ __launch_bounds__(256) void k_bin_keys(const TileGrid F, const double* __restrict__ xyz, uint64_t n, BinScratch S)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
	{
		S.keys[i] = tile_of(F, xyz, i);
		S.vals[i] = (uint32_t)i;
	}
}
template <bool GRAD, int MODE>
__global__ __launch_bounds__(256, DG_K2_WAVES) void k_interpolate_binned(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
															 double* __restrict__ phi_out, double* __restrict__ grad_out, BinScratch S)
{
	uint32_t blk;
	if (!k2_logical_block(blockIdx.x, (uint32_t)((n + 255) / 256), &blk))
		return;
	uint64_t gid = (uint64_t)blk * blockDim.x + threadIdx.x;
	if (gid >= n)
		return;
	if (S.sort_launched != 0 && S.flag[0] != 0)
		gid = S.perm[gid];
	const double x[3] = {xyz[3 * gid], xyz[3 * gid + 1], xyz[3 * gid + 2]};
	double g[3];
	phi_out[gid] = interpolate_point_mode<GRAD, MODE>(F, x, g);
	if (GRAD)
	{
		grad_out[3 * gid] = g[0];
		grad_out[3 * gid + 1] = g[1];
		grad_out[3 * gid + 2] = g[2];
	}
}

// K2 over a field with a CELL-MAJOR copy, queries in ANY order, no binning: one wave = 64 queries.  A query's 32
// coefficients are one contiguous 256-byte row of the copy; left to itself every lane would read its own row with
// sixteen 16-byte loads, 64 different lines per instruction.  Here the wave fetches the 64 rows TOGETHER -- sixteen
// loads, each covering four whole rows (lane = (row in the group, 16-byte piece)), i.e. four fully used 256-byte
// segments per instruction --, hands them to their owners through LDS (rows padded to 33 doubles: the owners' reads are
// conflict-free) and every lane then evaluates its own query exactly as interpolate_point_mode does (locate_query /
// evaluate_cell: the same statements).  HBM moves 256 B + 24 B + 8 B per query whatever the order of the queries.
static const int kRowStride = 33; // doubles per staged row
template <bool GRAD>
__global__ __launch_bounds__(64) void k_interpolate_rows(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
														   double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	__shared__ double rows[64 * kRowStride];
	const int lane = (int)threadIdx.x;
	const int sub = lane & 15, grp = lane >> 4;
	for (uint64_t base = (uint64_t)blockIdx.x * 64u; base < n; base += (uint64_t)gridDim.x * 64u)
	{
		const uint64_t gid = base + (uint64_t)lane;
		const bool have = gid < n;
		double x[3] = {0.0, 0.0, 0.0};
		if (have)
		{
			x[0] = xyz[3 * gid];
			x[1] = xyz[3 * gid + 1];
			x[2] = xyz[3 * gid + 2];
		}
		CellQuery q = locate_query(F, x);
		q.valid = q.valid && have;
		const uint32_t my_row = q.valid ? q.row : 0u; // (row 0 exists: a field has at least one cell row)
		__syncthreads(); // the previous round's rows have been read
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			const int owner = 4 * k + grp;
			const uint32_t r = (uint32_t)__shfl((int)my_row, owner);
			const double2 v = *reinterpret_cast<const double2*>(F.cell_major + 32 * (size_t)r + 2 * sub);
			rows[owner * kRowStride + 2 * sub] = v.x;
			rows[owner * kRowStride + 2 * sub + 1] = v.y;
		}
		__syncthreads();
		double cf[32];
#pragma unroll
		for (int j = 0; j < 32; ++j)
			cf[j] = rows[lane * kRowStride + j];
		double g[3] = {0.0, 0.0, 0.0};
		double phi = 1.7976931348623157e308;
		if (q.valid)
			phi = evaluate_cell<GRAD>(cf, q.xi, q.c0, g);
		if (have)
		{
			phi_out[gid] = phi;
			if (GRAD)
			{
				grad_out[3 * gid] = g[0];
				grad_out[3 * gid + 1] = g[1];
				grad_out[3 * gid + 2] = g[2];
			}
		}
	}
}

// K2 over a field with a BAND-LIMITED cell-major copy (FieldDev::band_rows / band_map): k_interpolate_rows for the queries
// whose cell has a row in the copy -- the wave fetches those rows together, four whole rows per load instruction, owners
// read them from LDS --, and in the same launch the plain gather (closed-form indices or the cell table) for the lanes
// whose cell has none.  A load group whose owner has no row (or no query) is switched off, so a batch that lives in the
// band moves 256 B per query and a batch far from it moves what the plain kernel moves plus 4 B of map.  Same
// locate_query / evaluate_cell statements as every other K2 path: same bits.
// (one lane's query of a round: located, and looked up in the band copy)
struct BandQuery
{
	CellQuery q;
	uint32_t row; // in the band copy, 0xffffffff: none (or no query)
	bool have;
};
__device__ __forceinline__ BandQuery band_locate(const FieldDev& F, const double* __restrict__ xyz, uint64_t gid, uint64_t n)
{
	BandQuery b;
	b.have = gid < n;
	double x[3] = {0.0, 0.0, 0.0};
	if (b.have)
	{
		x[0] = xyz[3 * gid];
		x[1] = xyz[3 * gid + 1];
		x[2] = xyz[3 * gid + 2];
	}
	b.q = locate_query(F, x);
	b.q.valid = b.q.valid && b.have;
	b.row = b.q.valid ? band_row_of(F, b.q.row) : 0xffffffffu;
	return b;
}
template <bool GRAD, int MODE>
__global__ __launch_bounds__(64) void k_interpolate_band(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
														   double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	__shared__ double rows[64 * kRowStride];
	const int lane = (int)threadIdx.x;
	const int sub = lane & 15, grp = lane >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * 64u;
	uint64_t base = (uint64_t)blockIdx.x * 64u;
	if (base >= n)
		return;
	// The look-up (query -> cell -> bit / rank words -> row) is a chain of two dependent memory round trips in front of the
	// row fetch; the wave therefore locates the queries of its NEXT round while the rows of the current one are in flight.
	BandQuery cur = band_locate(F, xyz, base + (uint64_t)lane, n);
	for (; base < n; base += stride)
	{
		const uint64_t gid = base + (uint64_t)lane;
		const bool mapped = cur.row != 0xffffffffu;
		__syncthreads(); // the previous round's rows have been read
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			const int owner = 4 * k + grp;
			// (owners without a row read row 0 -- it exists, and it is the same cached line for all of them --: unconditional
			// loads let the sixteen fetches be in flight together; behind a branch each they ran one after the other, 11 instead
			// of 18 Gq/s on a batch that lives in the band)
			uint32_t r = (uint32_t)__shfl((int)cur.row, owner);
			r = r != 0xffffffffu ? r : 0u;
			const double2 v = *reinterpret_cast<const double2*>(F.band_rows + 32 * (size_t)r + 2 * sub);
			rows[owner * kRowStride + 2 * sub] = v.x;
			rows[owner * kRowStride + 2 * sub + 1] = v.y;
		}
		BandQuery nxt;
		nxt.have = false;
		nxt.row = 0xffffffffu;
		nxt.q.valid = false;
		if (base + stride < n) // (wave-uniform)
			nxt = band_locate(F, xyz, base + stride + (uint64_t)lane, n);
		__syncthreads();
		double cf[32];
		if (mapped)
		{
#pragma unroll
			for (int j = 0; j < 32; ++j)
				cf[j] = rows[lane * kRowStride + j];
		}
		else if (cur.q.valid)
			fetch_cell<MODE>(F, cur.q.mi[0], cur.q.mi[1], cur.q.mi[2], cur.q.row, cf);
		double g[3] = {0.0, 0.0, 0.0};
		double phi = 1.7976931348623157e308;
		if (cur.q.valid)
			phi = evaluate_cell<GRAD>(cf, cur.q.xi, cur.q.c0, g);
		if (cur.have)
		{
			phi_out[gid] = phi;
			if (GRAD)
			{
				grad_out[3 * gid] = g[0];
				grad_out[3 * gid + 1] = g[1];
				grad_out[3 * gid + 2] = g[2];
			}
		}
		cur = nxt;
	}
}
// the band copy's builders: (1) per cell row, does any value the cell's 32 coefficients span reach into [lo, hi]?
// (min <= hi and max >= lo: a cell that straddles a thin band counts); (2) after a scan of the flags: rows and map
__device__ __forceinline__ void band_cell_indices(const FieldDev& F, uint64_t row, uint32_t idx[32])
{
	if (F.cells)
	{
#pragma unroll
		for (int j = 0; j < 32; ++j)
			idx[j] = F.cells[32 * row + j];
	}
	else
	{
		const uint32_t n01 = F.res[0] * F.res[1];
		const uint32_t k = (uint32_t)(row / n01), r = (uint32_t)(row % n01);
		cell_node_indices(r % F.res[0], r / F.res[0], k, F.res, idx);
	}
}
__global__ __launch_bounds__(256) void k_band_flags(const FieldDev F, uint64_t n_rows, double lo, double hi, uint32_t* __restrict__ flag)
{
	const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= n_rows)
		return;
	uint32_t idx[32];
	band_cell_indices(F, row, idx);
	double mn = F.coeffs[idx[0]], mx = mn;
#pragma unroll
	for (int j = 1; j < 32; ++j)
	{
		const double v = F.coeffs[idx[j]];
		mn = v < mn ? v : mn;
		mx = v > mx ? v : mx;
	}
	flag[row] = (mn <= hi && mx >= lo) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_band_expand(const FieldDev F, uint64_t n_rows, const uint32_t* __restrict__ flag,
													   const uint32_t* __restrict__ pos, uint64_t* __restrict__ bits, uint32_t* __restrict__ rank,
													   double* __restrict__ out)
{
	const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // (64 consecutive rows per wave: one word of bits)
	const bool keep = row < n_rows && flag[row] != 0u;
	const unsigned long long word = __ballot(keep);
	if ((threadIdx.x & 63u) == 0u && row < n_rows)
	{
		bits[row >> 6] = word;
		rank[row >> 6] = pos[row]; // rows of the copy before this word (exclusive scan of the flags)
	}
	if (!keep)
		return;
	const uint32_t r = pos[row];
	uint32_t idx[32];
	band_cell_indices(F, row, idx);
	double* o = out + 32 * (size_t)r;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		o[j] = F.coeffs[idx[j]];
}

// Builds the cell-major copy of a field (FieldDev::cell_major): one thread per cell row.
__global__ __launch_bounds__(256) void k_expand_cells(const FieldDev F, uint64_t n_rows, double* __restrict__ out)
{
	const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= n_rows)
		return;
	uint32_t idx[32];
	if (F.cells)
	{
#pragma unroll
		for (int j = 0; j < 32; ++j)
			idx[j] = F.cells[32 * row + j];
	}
	else
	{
		const uint32_t n01 = F.res[0] * F.res[1];
		const uint32_t k = (uint32_t)(row / n01), r = (uint32_t)(row % n01);
		cell_node_indices(r % F.res[0], r / F.res[0], k, F.res, idx);
	}
	double* o = out + 32 * row;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		o[j] = F.coeffs[idx[j]];
}

// Builds the tile-major copy of an unreduced field (dg_lattice.h): one thread per slot, one block row per tile;
// reads are gathers from the reference layout (each node is read by at most 8 tiles), writes are contiguous.
__global__ __launch_bounds__(256) void k_expand_tiles(const FieldDev F, uint64_t n_tiles, double* __restrict__ out)
{
	const uint64_t total = n_tiles * kTmNodes;
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x)
	{
		const uint64_t tile = e / kTmNodes;
		const uint32_t slot = (uint32_t)(e - tile * kTmNodes);
		const uint32_t ti = (uint32_t)(tile % F.ntile[0]);
		const uint32_t tj = (uint32_t)((tile / F.ntile[0]) % F.ntile[1]);
		const uint32_t tk = (uint32_t)(tile / ((uint64_t)F.ntile[0] * F.ntile[1]));
		const uint32_t node = tile_slot_node(slot, ti, tj, tk, F.res);
		out[e] = node == 0xffffffffu ? 0.0 : F.coeffs[node];
	}
}
// second pass: the "no value" flags of the 64 cells of every tile (dg_lattice.h: kTmFlags); one wave per tile
__global__ __launch_bounds__(256) void k_tile_flags(uint64_t n_tiles, double* __restrict__ tiles)
{
	const uint64_t tile = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
	if (tile >= n_tiles)
		return;
	const uint32_t c = threadIdx.x & 63u;
	uint32_t slots[32];
	tile_node_slots(c & 3u, (c >> 2) & 3u, c >> 4, slots);
	const double* t = tiles + tile * kTmNodes;
	bool nov = false;
#pragma unroll
	for (int q = 0; q < 32; ++q)
		nov = nov || (t[slots[q]] == 1.7976931348623157e308);
	const unsigned long long flags = __ballot(nov);
	if (c == 0)
		*(unsigned long long*)(tiles + tile * kTmNodes + kTmFlags) = flags;
}

// ------------------------------------------------------------------------------------------------
// K3: SPH boundary density map (GenerateDensityMap).  One wave = one 4x4x4 brick of lattice nodes
// (K1's decomposition), so the 64 lanes evaluate the SDF in a 3x3x3-cell neighbourhood at every
// quadrature step and the 256-byte coefficient rows they read stay in L1.  Lanes whose node is
// rejected or beyond 2h idle; waves without an active lane exit at once.
template <bool STAGED, int MODE>
__global__ __launch_bounds__(64, DG_K3_WAVES) void k_density_bricks(const SampleParams L, const FieldDev F, const DensityParams P)
{
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	const uint64_t brick = (uint64_t)blk; // one wave per block
	if (brick >= L.total_bricks)
		return;
	const LaneNode ln = map_lane(L, brick, (int)(threadIdx.x & 63u));
	if (!ln.valid)
		return;
	double v = 1.7976931348623157e308;
	if (L.mask == nullptr || L.mask[ln.out_idx] != 0)
	{
		double x[3];
		node_position(ln.cls, ln.a, ln.b, ln.s, L.dmin, L.cell, x);
		if (density_prefilter(F, P, x, &v))
			v = density_integral_t<STAGED, MODE>(F, P, x);
	}
	L.out[ln.out_idx] = v;
}

__device__ __forceinline__ int k3_cell_guess(const FieldDev& F, int d, double y)
{
	double t = (y - F.dmin[d]) * F.inv_cell[d];
	t = fmin(fmax(t, -8.0), (double)F.res[d] + 8.0);
	return (int)floor(t);
}

// ---- K3, two nodes per lane ---------------------------------------------------------------------------------------------
// Counters of k_density_bricks (profiles/r03_k3_pmc.txt): texture-address and texture-data units busy 0.94 of the
// launch, VALU 0.67 -- the kernel sits on the vector-memory roof: at every quadrature point every lane pulls the 256
// bytes of its cell through the L1 path (sixteen 16-byte loads, 20.8 TD cycles per wave instruction).  Six of seven
// lattice nodes are edge nodes, two per cell edge, a third of a cell apart; shifted by the same quadrature offset the
// two land in the SAME cell two times out of three.  Here a lane owns such a pair (a wave: a "double brick" of 8 x 4 x 4
// edge nodes, or a plain 4 x 4 x 4 brick of vertex nodes), fetches the cell's 32 coefficients once and evaluates both
// points from the same registers; only when a cell face separates the two does it fetch again.  The axes the two nodes
// share are evaluated once.  Per node and point: 0.67 of the loads, 0.86 of the arithmetic -- the same operations on
// the same values in the same order as density_integral_t<true, .>, so the bits do not change.
template <int MODE>
__device__ __forceinline__ double k3_cell_value(const FieldDev& F, const double cf[32], bool ok, const Axis1D& ax, const Axis1D& ay,
												 const Axis1D& az)
{
	const double NOVAL = 1.7976931348623157e308;
	const double mxmy = ax.m * ay.m, mxpy = ax.m * ay.p, pxmy = ax.p * ay.m, pxpy = ax.p * ay.p;
	const double x2y2 = ax.t2 + ay.t2;
	const double mz = az.m, pz = az.p;
	const double fac = 1.0 / 64.0 * (9.0 * (x2y2 + az.t2) - 19.0);
	double phi = 0.0;
#define DG_ACC(q, n)                                   \
	if (MODE != kFieldTileMajor && MODE != kFieldXMajor) \
		ok = ok && (cf[q] != NOVAL);                   \
	phi += cf[q] * (n);
	DG_ACC(0, fac * mxmy * mz)
	DG_ACC(1, fac * pxmy * mz)
	DG_ACC(2, fac * mxpy * mz)
	DG_ACC(3, fac * pxpy * mz)
	DG_ACC(4, fac * mxmy * pz)
	DG_ACC(5, fac * pxmy * pz)
	DG_ACC(6, fac * mxpy * pz)
	DG_ACC(7, fac * pxpy * pz)
	{
		const double mymz = ay.m * mz, mypz = ay.m * pz, pymz = ay.p * mz, pypz = ay.p * pz;
		DG_ACC(8, ax.fm3 * mymz)
		DG_ACC(9, ax.fp3 * mymz)
		DG_ACC(10, ax.fm3 * mypz)
		DG_ACC(11, ax.fp3 * mypz)
		DG_ACC(12, ax.fm3 * pymz)
		DG_ACC(13, ax.fp3 * pymz)
		DG_ACC(14, ax.fm3 * pypz)
		DG_ACC(15, ax.fp3 * pypz)
	}
	{
		const double mxmz = ax.m * mz, mxpz = ax.m * pz, pxmz = ax.p * mz, pxpz = ax.p * pz;
		DG_ACC(16, ay.fm3 * mxmz)
		DG_ACC(17, ay.fp3 * mxmz)
		DG_ACC(18, ay.fm3 * pxmz)
		DG_ACC(19, ay.fp3 * pxmz)
		DG_ACC(20, ay.fm3 * mxpz)
		DG_ACC(21, ay.fp3 * mxpz)
		DG_ACC(22, ay.fm3 * pxpz)
		DG_ACC(23, ay.fp3 * pxpz)
	}
	DG_ACC(24, az.fm3 * mxmy)
	DG_ACC(25, az.fp3 * mxmy)
	DG_ACC(26, az.fm3 * mxpy)
	DG_ACC(27, az.fp3 * mxpy)
	DG_ACC(28, az.fm3 * pxmy)
	DG_ACC(29, az.fp3 * pxmy)
	DG_ACC(30, az.fm3 * pxpy)
	DG_ACC(31, az.fp3 * pxpy)
#undef DG_ACC
	return ok ? phi : NOVAL;
}

// axis_eval(F, 2, y) with the cell's (c0, c1) -- the two divisions of the affine map -- read from a table the wave
// filled once for the z cells its points can fall into (zt == nullptr: no table, plain axis_eval).  Same expressions
// on the same inputs, same bits.
__device__ __forceinline__ Axis1D k3_axis_z(const FieldDev& F, double y, const double2* zt, int zlo)
{
	if (zt == nullptr)
		return axis_eval(F, 2, y);
	Axis1D a;
	a.inside = (F.dmin[2] <= y) && (y <= F.dmax[2]);
	uint32_t mi = (uint32_t)((y - F.dmin[2]) * F.inv_cell[2]);
	if (mi >= F.res[2])
		mi = F.res[2] - 1;
	if (!a.inside)
		mi = 0;
	a.mi = mi;
	const double2 cc = zt[a.inside ? (int)mi - zlo : 0];
	a.t = cc.x * y - cc.y;
	a.t2 = a.t * a.t;
	a.m = 1.0 - a.t;
	a.p = 1.0 + a.t;
	const double fac = 9.0 / 64.0 * (1.0 - a.t2);
	a.fm3 = fac * (1.0 - 3.0 * a.t);
	a.fp3 = fac * (1.0 + 3.0 * a.t);
	return a;
}

// the quadrature for the pair (A, B) of one lane; E: the axis along which the two nodes differ (-1: no node B)
template <int MODE, int E>
__device__ __forceinline__ void k3_pair_integral(const FieldDev& F, const DensityParams& P, bool has_noval, bool skip, const double2* zt, int zlo,
												  const double xa[3], double xb_e, bool need_a, bool need_b, double* out_a, double* out_b)
{
	const double NOVAL = 1.7976931348623157e308;
	double res_a = 0.0, res_b = 0.0;
	DG_NOUNROLL
	for (int i = 0; i < 16; ++i)
	{
		const double wi = P.w[i];
		const Axis1D ax = axis_eval(F, 0, xa[0] + P.xi[i]);
		Axis1D bx = ax;
		if (E == 0)
			bx = axis_eval(F, 0, xb_e + P.xi[i]);
		DG_NOUNROLL
		for (int j = 0; j < 16; ++j)
		{
			const uint32_t kmask = skip ? (uint32_t)P.kmask[i * 16 + j] : 0xffffu;
			if (kmask == 0u)
				continue; // the whole column lies outside the kernel's support
			const double wij = wi * P.w[j];
			const Axis1D ay = axis_eval(F, 1, xa[1] + P.xi[j]);
			Axis1D by = ay;
			if (E == 1)
				by = axis_eval(F, 1, xb_e + P.xi[j]);
			DG_NOUNROLL
			for (int k = 0; k < 16; ++k)
			{
				if (((kmask >> k) & 1u) == 0u)
					continue;
				const double wijk = wij * P.w[k];
				const Axis1D az = k3_axis_z(F, xa[2] + P.xi[k], zt, zlo);
				Axis1D bz = az;
				if (E == 2)
					bz = k3_axis_z(F, xb_e + P.xi[k], zt, zlo);
				const bool want_a = need_a && ax.inside && ay.inside && az.inside;
				const bool want_b = E >= 0 && need_b && bx.inside && by.inside && bz.inside;
				const uint32_t be = E == 0 ? bx.mi : (E == 1 ? by.mi : bz.mi), ae = E == 0 ? ax.mi : (E == 1 ? ay.mi : az.mi);
				const bool shared = want_a && want_b && be == ae; // the two points lie in one cell
				double cf[32];
				bool ok = true;
				double da = NOVAL, db = NOVAL;
				if (want_a)
				{
					fetch_cell<MODE>(F, ax.mi, ay.mi, az.mi, F.res[1] * F.res[0] * az.mi + F.res[0] * ay.mi + ax.mi, cf);
					if (MODE == kFieldTileMajor && has_noval)
						ok = !tile_cell_has_novalue(F.tile_major, F.ntile, ax.mi, ay.mi, az.mi);
					if (MODE == kFieldXMajor && has_noval)
						ok = !xmajor_cell_has_novalue(F, ax.mi, ay.mi, az.mi);
					da = k3_cell_value<MODE>(F, cf, ok, ax, ay, az);
				}
				if (want_b && !shared)
				{
					fetch_cell<MODE>(F, bx.mi, by.mi, bz.mi, F.res[1] * F.res[0] * bz.mi + F.res[0] * by.mi + bx.mi, cf);
					ok = true;
					if (MODE == kFieldTileMajor && has_noval)
						ok = !tile_cell_has_novalue(F.tile_major, F.ntile, bx.mi, by.mi, bz.mi);
					if (MODE == kFieldXMajor && has_noval)
						ok = !xmajor_cell_has_novalue(F, bx.mi, by.mi, bz.mi);
				}
				if (want_b)
					db = k3_cell_value<MODE>(F, cf, ok, bx, by, bz);
				const double wv = P.wtab[(i * 16 + j) * 16 + k];
				if (need_a)
				{
					const double gamma = (da > P.h) ? 0.0 : 1.0 - da / P.h;
					res_a += wijk * (gamma * wv);
				}
				if (E >= 0 && need_b)
				{
					const double gamma = (db > P.h) ? 0.0 : 1.0 - db / P.h;
					res_b += wijk * (gamma * wv);
				}
			}
		}
	}
	res_a *= P.c0prod;
	*out_a = P.rho0 * res_a;
	res_b *= P.c0prod;
	*out_b = P.rho0 * res_b;
}

template <int MODE>
__device__ __forceinline__ void k3_pair_wave(const SampleParams& L, const FieldDev& F, const DensityParams& P, int cls, int lane, bool valid_a,
											   bool valid_b, int64_t out_a, int64_t out_b, const double xa[3], const double xb[3]);
template <int MODE, int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_density_pairs(const SampleParams L, const FieldDev F, const DensityParams P)
{
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	const uint64_t brick = (uint64_t)blk; // one wave per block
	if (brick >= L.total_bricks)
		return;
	const int lane = (int)(threadIdx.x & 63u);
	BrickMap m = map_brick(L, brick);
	const int cls = m.cls; // wave-uniform
	// vertex class: the lane's node of the 4x4x4 brick.  Edge classes: L counts DOUBLE bricks along a (pair_bricks());
	// the lane takes the two nodes a = 8 t + 2 p, + 1 of edge p of the row -- nodes (2 p) & 3, + 1 of brick 2 t + (p >> 1)
	int la = lane;
	if (cls != 0)
	{
		const int p = lane & 3;
		m.b0 = 2u * m.b0 + (uint32_t)(p >> 1);
		la = (lane & ~3) | ((2 * p) & 3);
	}
	const LaneNode na = map_lane(L, m, la);
	LaneNode nb = na;
	nb.valid = false;
	if (cls != 0)
		nb = map_lane(L, m, la | 1);
	double xa[3], xb[3];
	node_position(na.cls, na.a, na.b, na.s, L.dmin, L.cell, xa);
	node_position(nb.cls, nb.a, nb.b, nb.s, L.dmin, L.cell, xb);
	k3_pair_wave<MODE>(L, F, P, cls, lane, na.valid, nb.valid, na.out_idx, nb.out_idx, xa, xb);
}

// the work of one wave of K3's two-nodes-per-lane kernels once every lane knows its nodes: cls (wave-uniform) names the
// axis along which A and B differ; lane 0 must hold the lowest, lane 63 the highest z of the wave
template <int MODE>
__device__ __forceinline__ void k3_pair_wave(const SampleParams& L, const FieldDev& F, const DensityParams& P, int cls, int lane, bool valid_a,
											   bool valid_b, int64_t out_a, int64_t out_b, const double xa[3], const double xb[3])
{
	const double NOVAL = 1.7976931348623157e308;
	double va = NOVAL, vb = NOVAL;
	bool need_a = false, need_b = false;
	if (valid_a && (L.mask == nullptr || L.mask[out_a] != 0))
		need_a = density_prefilter(F, P, xa, &va);
	if (valid_b && (L.mask == nullptr || L.mask[out_b] != 0))
		need_b = density_prefilter(F, P, xb, &vb);
	if (__ballot(need_a || need_b) != 0ull)
	{
		const uint32_t flags = P.unsafe ? P.unsafe[0] : 2u; // bit 0: NaN / Inf / huge values, bit 1: "no value" coefficients
		const bool has_noval = (flags & 2u) != 0u;
		const bool skip = P.skip_mode == 1 || (P.skip_mode == 2 && (flags & 1u) == 0u);
		// (c0, c1) of the z cells the wave can meet: from lane 0's lowest point to lane 63's highest (positions and the
		// cell computation are monotone); a window of more than 64 cells (fine lattices, large h) goes without the table
		__shared__ double2 sT[64];
		const int zlo = max(0, __builtin_amdgcn_readlane(k3_cell_guess(F, 2, xa[2] + P.xi[0]), 0));
		const int zhi = min((int)F.res[2] - 1, __builtin_amdgcn_readlane(k3_cell_guess(F, 2, fmax(xa[2], xb[2]) + P.xi[15]), 63));
		const double2* zt = nullptr;
		if (zhi - zlo < 64)
		{
			if (zlo + lane <= zhi)
			{
				const double lo = F.dmin[2] + (double)(uint32_t)(zlo + lane) * F.cell[2];
				const double hi = lo + F.cell[2];
				const double den = hi - lo;
				double2 e;
				e.x = 2.0 / den;
				e.y = (hi + lo) / den;
				sT[lane] = e;
			}
			zt = sT;
			__syncthreads();
		}
		double ra = 0.0, rb = 0.0;
		// the axis along which the pair's nodes differ: x for the X class, y for Y, z for Z (dg_geom.h node_position())
		if (cls == 0)
			k3_pair_integral<MODE, -1>(F, P, has_noval, skip, zt, zlo, xa, 0.0, need_a, false, &ra, &rb);
		else if (cls == 1)
			k3_pair_integral<MODE, 0>(F, P, has_noval, skip, zt, zlo, xa, xb[0], need_a, need_b, &ra, &rb);
		else if (cls == 2)
			k3_pair_integral<MODE, 1>(F, P, has_noval, skip, zt, zlo, xa, xb[1], need_a, need_b, &ra, &rb);
		else
			k3_pair_integral<MODE, 2>(F, P, has_noval, skip, zt, zlo, xa, xb[2], need_a, need_b, &ra, &rb);
		if (need_a)
			va = ra;
		if (need_b)
			vb = rb;
	}
	if (valid_a)
		L.out[out_a] = va;
	if (valid_b)
		L.out[out_b] = vb;
}

// ---- K3 in row blocks (unreduced fields, whole lattice) --------------------------------------------------------------
// k_density_pairs sits on the texture path (profiles/r03_k3_vmem_pmc.txt: TD busy 0.96, VALU 0.55): its waves are cubes
// of 4 x 4 x 4 cells, so every 64-lane load is sixteen runs of four lanes = 64 bytes that straddle a sector (or a tile)
// boundary three times out of four -- 31 sectors of 64 bytes touched per instruction where 16 hold the data.  The texture
// path's work is the sector count.  Here a wave is a ROW block: LX cells along x (16), LY x LZ = 2 x 2 rows, every lane
// again the node (vertex class) or the two nodes of the cell edge (edge classes) at its cell -- whichever axis the edge
// runs along, the lanes of a row sit a cell apart along x.  With the x-major copy of the Y and Z classes (dg_lattice.h)
// all sixteen pair loads of a cell are 256-byte runs along x: 19 sectors per instruction.  Same per-lane arithmetic as
// k_density_pairs (k3_pair_wave), same bits.
template <int MODE, int LX, int LY, int LZ, int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_density_rows(const SampleParams L, const FieldDev F, const DensityParams P)
{
	static_assert(LX * LY * LZ == 64, "one wave");
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	if (blk >= P.row_prefix[4])
		return;
	const RowWave m = row_wave_map(P, blk);
	const int lane = (int)(threadIdx.x & 63u);
	const int cls = m.cls; // wave-uniform
	const RowItem it = row_lane_item(m, lane, (uint32_t)LX, (uint32_t)LY, (uint32_t)LZ, F.res);
	double xa[3], xb[3];
	node_position(cls, it.a, it.b, it.s, L.dmin, L.cell, xa);
	node_position(cls, cls != 0 ? it.a + 1u : it.a, it.b, it.s, L.dmin, L.cell, xb);
	// (a launch over a node range: the waves cover the whole lattice, lanes whose node lies outside the range idle)
	const bool in_a = it.valid && it.node >= P.row_node_begin && it.node < P.row_node_end;
	const bool in_b = it.valid && cls != 0 && it.node + 1 >= P.row_node_begin && it.node + 1 < P.row_node_end;
	k3_pair_wave<MODE>(L, F, P, cls, lane, in_a, in_b, (int64_t)(it.node - P.row_node_begin), (int64_t)(it.node + 1 - P.row_node_begin), xa, xb);
}


// ---- K3, one lane per lattice point (dg_density_cells.h) ----------------------------------------------------------------
// The wave context of k3c_lane() on the device: ballots, and the LDS tables of the axis states of the shifted coordinates.
// Producers: the X states of (variant v, lane x) come from the lane (x, y = v, z = 0), the Y states of (v, lane y) from
// the lane (x = v, y, z = 0), the Z states from lanes 0..5 for one k after the other -- each the pure function
// k3_axis_entry() of a coordinate that depends on the lattice index along that axis only, i.e. the value the consuming
// lane would compute itself.  One wave per workgroup: the barriers cost nothing and fence the compiler.
struct K3WaveDev
{
	K3Axis* sX; // [2][16]    (variant, lane x)
	K3Axis* sY; // [2][2]     (variant, lane y)
	K3Axis* sZ; // [16][3][2] (k, variant: lattice point / node A / node B, lane z)
	uint32_t* sZb; // [16][2] (k, lane z): k3c_axis_bits() of the z axis
	double* sR; // [7][64]    the seven sums of every lane (registers are what this kernel is short of)
	int lane, ix, iy, iz;
	__device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
	__device__ __forceinline__ uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
	__device__ __forceinline__ void acc_init()
	{
#pragma unroll
		for (int n = 0; n < 7; ++n)
			sR[n * 64 + lane] = 0.0;
	}
	__device__ __forceinline__ void acc_add(int n, double t) { sR[n * 64 + lane] += t; }
	__device__ __forceinline__ double acc_get(int n) const { return sR[n * 64 + lane]; }
	__device__ __forceinline__ void set_x(const FieldDev& F, double a, double b)
	{
		const K3Axis e = k3_axis_entry(F, 0, iy == 0 ? a : b);
		__syncthreads();
		if (iz == 0)
			sX[iy * 16 + ix] = e;
		__syncthreads();
	}
	__device__ __forceinline__ void set_y(const FieldDev& F, double a, double b)
	{
		const K3Axis e = k3_axis_entry(F, 1, ix == 0 ? a : b);
		__syncthreads();
		if (iz == 0 && ix < 2)
			sY[ix * 2 + iy] = e;
		__syncthreads();
	}
	__device__ __forceinline__ void set_z(const FieldDev& F, const DensityParams& P, const SampleParams& L, const RowWave& m, double, double, double)
	{
		const int v = lane >> 1, s = lane & 1; // lanes 0..5
		uint32_t kz = m.w[2] * (uint32_t)kK3cLz + (uint32_t)s;
		kz = kz <= F.res[2] ? kz : F.res[2];
		double z = L.dmin[2] + L.cell[2] * (double)kz;
		if (v == 1)
			z = z + 1.0 / 3.0 * L.cell[2];
		else if (v == 2)
			z = z + 2.0 / 3.0 * L.cell[2];
		DG_NOUNROLL
		for (int k = 0; k < 16; ++k)
		{
			const K3Axis e = k3_axis_entry(F, 2, z + P.xi[k]);
			if (lane < 6)
				sZ[k * 6 + lane] = e; // (k * 3 + v) * 2 + s
		}
		__syncthreads();
		// the z axis' verdict on the seven nodes (k3c_axis_bits()) for every k and lane z, once
		if (lane < 32)
		{
			const int k = lane >> 1, sl = lane & 1;
			const K3Axis* z = sZ + k * 6 + sl;
			sZb[lane] = k3c_axis_bits(kK3cZ0, kK3cZA, kK3cZB, z[0].mi, z[0].inside != 0u, z[2].mi, z[2].inside != 0u, z[4].mi, z[4].inside != 0u);
		}
		__syncthreads();
	}
	__device__ __forceinline__ uint32_t z_bits(const FieldDev&, const DensityParams&, int k) const { return sZb[k * 2 + iz]; }
	__device__ __forceinline__ K3Axis x_var(const FieldDev&, int v) const { return sX[v * 16 + ix]; }
	__device__ __forceinline__ K3Axis y_var(const FieldDev&, int v) const { return sY[v * 2 + iy]; }
	__device__ __forceinline__ K3Axis z_var(const FieldDev&, const DensityParams&, int k, int v) const { return sZ[(k * 3 + v) * 2 + iz]; }
	__device__ __forceinline__ void x_id(const FieldDev&, int v, uint32_t* mi, bool* in) const
	{
		*mi = sX[v * 16 + ix].mi;
		*in = sX[v * 16 + ix].inside != 0u;
	}
	__device__ __forceinline__ void y_id(const FieldDev&, int v, uint32_t* mi, bool* in) const
	{
		*mi = sY[v * 2 + iy].mi;
		*in = sY[v * 2 + iy].inside != 0u;
	}
	__device__ __forceinline__ void z_id(const FieldDev&, const DensityParams&, int k, int v, uint32_t* mi, bool* in) const
	{
		*mi = sZ[(k * 3 + v) * 2 + iz].mi;
		*in = sZ[(k * 3 + v) * 2 + iz].inside != 0u;
	}
};
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_density_cells(const SampleParams L, const FieldDev F, const DensityParams P, const K3CellsGeom G)
{
	__shared__ K3Axis sX[2 * 16], sY[2 * 2], sZ[16 * 3 * 2];
	__shared__ double sR[7 * 64];
	__shared__ uint32_t sZb[16 * 2];
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	if (blk >= P.row_prefix[4])
		return;
	const RowWave m = row_wave_map(P, blk);
	K3WaveDev w;
	w.sX = sX;
	w.sY = sY;
	w.sZ = sZ;
	w.sR = sR;
	w.sZb = sZb;
	w.lane = (int)(threadIdx.x & 63u);
	w.ix = w.lane & 15;
	w.iy = (w.lane >> 4) & 1;
	w.iz = w.lane >> 5;
	k3c_lane(w, L, F, P, G, m, w.lane);
}

// the x-major copy (dg_lattice.h): one thread per pair of the copy, contiguous 16-byte writes, reads a plane apart
__global__ __launch_bounds__(256) void k_xmajor_copy(const FieldDev F, uint32_t n_pairs, double* __restrict__ out)
{
	for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n_pairs; e += gridDim.x * 256u)
	{
		const double* src = F.coeffs + xmajor_pair_node(e, F.res);
		double2 v;
		v.x = src[0];
		v.y = src[1];
		*(double2*)(out + 2 * (size_t)e) = v;
	}
}
// the "no value" bit of every cell, one wave per 64 cells of a row; only if k_field_check found such a value at all
// (flag bit 1) -- k_density_rows does not read the bits otherwise
__global__ __launch_bounds__(256) void k_xmajor_flags(const FieldDev F, const uint32_t* __restrict__ field_flags, uint64_t* __restrict__ out)
{
	if ((field_flags[0] & 2u) == 0u)
		return;
	const uint32_t words = xmajor_flag_words(F.res);
	const uint64_t total = (uint64_t)F.res[2] * F.res[1] * words;
	const int lane = (int)(threadIdx.x & 63u);
	for (uint64_t w = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); w < total; w += (uint64_t)gridDim.x * 4u)
	{
		const uint32_t wi = (uint32_t)(w % words), row = (uint32_t)(w / words);
		const uint32_t j = row % F.res[1], k = row / F.res[1];
		const uint32_t i = wi * 64u + (uint32_t)lane;
		bool nov = false;
		if (i < F.res[0])
		{
			double cf[32];
			fetch_cell<kFieldXMajor>(F, i, j, k, 0u, cf);
#pragma unroll
			for (int q = 0; q < 32; ++q)
				nov = nov || (cf[q] == 1.7976931348623157e308);
		}
		const unsigned long long bits = __ballot(nov);
		if (lane == 0)
			out[w] = bits;
	}
}

// ---- K3 with the coefficients staged through LDS (unreduced fields) ---------------------------------------------------
// Counters of k_density_bricks on the tile copy (profiles/r03_k3_pmc.txt): the texture-address and texture-data units
// are busy 0.94 of the launch, VALU 0.67 -- every lane pulls the 256 bytes of its cell through the vector memory
// pipeline at each quadrature point (sixteen 16-byte loads, 20 TA cycles per wave instruction), although the 64 lanes
// of a brick sit in neighbouring cells and share most of those nodes: the 4x4x4 cells of a brick reference 5^3
// vertices + 3 x 100 edges = 725 doubles = 5.8 KB, against 16 KB fetched.  Here the wave fetches the box ONCE per
// quadrature point, straight into LDS (global_load_lds_dwordx4: no registers, no ds_write pass), as 4 x 100 entries of
// 16 bytes -- the x-adjacent vertex pair and the x edge of box position (a, b, c), the y edge of (c, a, b), the z edge
// of (b, c, a), a < 4, b, c < 5: for every class the fastest digit runs along the axis the class is contiguous in
// ([V | X | Y | Z]: x, x, y, z), so that four consecutive lanes read 64 consecutive bytes, and the reference layout is
// linear in the cell coordinates, so that a lane's address is (origin of the box, scalar) + (lane constant).  Every
// lane then reads its 32 coefficients from LDS right where the sum consumes them: sixteen ds_read_b128, no 64
// coefficient registers.  The loads of point p + 1 are in flight while point p is evaluated (two LDS buffers, counted
// s_waitcnt: the wave keeps its own memory latency covered instead of relying on other waves, which -- running the same
// loop in lockstep -- want the same unit at the same time).  Same values, same arithmetic, same order: bit-identical.
//
// The box origin is the cell of lane 0's evaluation point (the brick's minimum corner: node positions and the cell
// computation are monotone).  A lane whose cell lies outside [origin, origin + 3]^3 -- possible only when rounding
// splits two lanes by a fifth cell, or for a point ON the domain's upper face -- evaluates its point the plain way.
// "No value" coefficients: k_field_check tells whether the field holds any (bit 1); only then are the per-cell flag
// bits of the tile copy consulted.
typedef __attribute__((address_space(3))) void* k3_lds_ptr;
__device__ __forceinline__ double k3_lane0(double v)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 0), hi = __builtin_amdgcn_readlane(__double2hiint(v), 0);
	return __hiloint2double(hi, lo);
}
// 16 bytes per active lane from `gsrc` (per lane) to LDS byte address lds_dst + 16 * lane (lds_dst wave-uniform).  Not
// tracked by the compiler's s_waitcnt bookkeeping: the caller counts (k3 kernels: exactly eight per point).
__device__ __forceinline__ void k3_glds16(const void* gsrc, uint32_t lds_dst)
{
	unsigned keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
				 : "=&s"(keep)
				 : "v"(gsrc), "s"(lds_dst)
				 : "memory");
}
struct K3Entry // what lane e (two rounds: e = lane, lane + 64; live below 100) brings of a box
{
	uint32_t a, b, c;                // e = a + 4 b + 20 c
	uint32_t offV, offX, offY, offZ; // bytes from the box origin's node of each class
	bool live;
};
__device__ __forceinline__ K3Entry k3_entry(const FieldDev& F, int e)
{
	K3Entry k;
	const uint32_t nx = F.res[0], ny = F.res[1], nz = F.res[2];
	k.live = e < 100;
	k.c = (uint32_t)e / 20u;
	k.b = ((uint32_t)e - 20u * k.c) / 4u;
	k.a = (uint32_t)e & 3u;
	k.offV = 8u * (k.a + k.b * (nx + 1) + k.c * (nx + 1) * (ny + 1));
	k.offX = 16u * (k.a + k.b * nx + k.c * nx * (ny + 1));
	k.offY = 16u * (k.a + k.b * ny + k.c * ny * (nz + 1));
	k.offZ = 16u * (k.a + k.b * nz + k.c * nz * (nx + 1));
	return k;
}
// digit + origin within [0, n] (le) / [0, n) (lt) along one axis; negative sums wrap to huge values
__device__ __forceinline__ bool k3_le(int o, uint32_t digit, uint32_t n) { return (uint32_t)o + digit <= n; }
__device__ __forceinline__ bool k3_lt(int o, uint32_t digit, uint32_t n) { return (uint32_t)o + digit < n; }

template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_density_bricks_lds(const SampleParams L, const FieldDev F, const DensityParams P)
{
	__shared__ double2 sE[2][4][100]; // [buffer][V pairs, X, Y, Z edges][entry]
	__shared__ double2 sT[48];        // (c0, c1) of the affine map of the z cells this brick's quadrature points can fall into
	const double NOVAL = 1.7976931348623157e308;
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	const uint64_t brick = (uint64_t)blk; // one wave per block
	if (brick >= L.total_bricks)
		return;
	const int lane = (int)(threadIdx.x & 63u);
	const LaneNode ln = map_lane(L, brick, lane);
	double x[3];
	node_position(ln.cls, ln.a, ln.b, ln.s, L.dmin, L.cell, x); // (lanes beyond the lattice carry a clamped, valid node)
	double v = NOVAL;
	bool need = false;
	if (ln.valid && (L.mask == nullptr || L.mask[ln.out_idx] != 0))
		need = density_prefilter(F, P, x, &v);
	if (__ballot(need) == 0ull)
	{
		if (ln.valid)
			L.out[ln.out_idx] = v;
		return;
	}
	const uint32_t nx = F.res[0], ny = F.res[1], nz = F.res[2];
	const uint32_t flags = P.unsafe ? P.unsafe[0] : 2u; // bit 0: NaN / Inf / huge values, bit 1: "no value" coefficients
	const bool has_noval = (flags & 2u) != 0u;
	const bool skip = P.skip_mode == 1 || (P.skip_mode == 2 && (flags & 1u) == 0u);
	const K3Entry e0 = k3_entry(F, lane), e1 = k3_entry(F, lane + 64);
	const int64_t nv = (int64_t)(nx + 1) * (ny + 1) * (nz + 1), nex = (int64_t)nx * (ny + 1) * (nz + 1), ney = (int64_t)(nx + 1) * ny * (nz + 1);
	// lane l < 16 holds the box origin (cell of lane 0's point) for quadrature index l along each axis
	const double xq = P.xi[lane & 15];
	const int vgx = k3_cell_guess(F, 0, k3_lane0(x[0]) + xq), vgy = k3_cell_guess(F, 1, k3_lane0(x[1]) + xq),
			  vgz = k3_cell_guess(F, 2, k3_lane0(x[2]) + xq);
	// The z cells the brick can meet: from lane 0's lowest point to lane 63's highest (positions and the cell
	// computation are monotone).  Their (c0, c1) -- two divisions per cell -- are computed once here instead of at
	// every quadrature point: same expressions on the same inputs as axis_eval(), same bits.
	const int zlo = max(0, __builtin_amdgcn_readlane(vgz, 0));
	const int zhi = min((int)nz - 1, __builtin_amdgcn_readlane(k3_cell_guess(F, 2, x[2] + P.xi[15]), 63));
	const bool ztab = zhi - zlo < 48;
	if (ztab && zlo + lane <= zhi)
	{
		const double lo = F.dmin[2] + (double)(uint32_t)(zlo + lane) * F.cell[2];
		const double hi = lo + F.cell[2];
		const double den = hi - lo;
		double2 e;
		e.x = 2.0 / den;
		e.y = (hi + lo) / den;
		sT[lane] = e;
	}
	__syncthreads();
	const uint32_t lds0 = (uint32_t)(uintptr_t)(k3_lds_ptr)&sE[0][0][0];

	// the active quadrature points in the reference's order (i, j, k); wave-uniform
	auto kmask_of = [&](int i, int j) -> uint32_t { return skip ? (uint32_t)P.kmask[i * 16 + j] : 0xffffu; };
	auto advance = [&](int& i, int& j, int& k) -> bool { // to the next active point after (i, j, k); k = -1: from the column's start
		++k;
		while (true)
		{
			const uint32_t m = k < 16 ? (kmask_of(i, j) >> k) : 0u;
			if (m != 0u)
			{
				k += __builtin_ctz(m);
				return true;
			}
			k = 0;
			if (++j == 16)
			{
				j = 0;
				if (++i == 16)
					return false;
			}
		}
	};
	// the eight loads of a point's box into buffer `buf`
	auto issue = [&](int i, int j, int k, int buf) {
		const int ox = __builtin_amdgcn_readlane(vgx, i), oy = __builtin_amdgcn_readlane(vgy, j), oz = __builtin_amdgcn_readlane(vgz, k);
		// origin nodes of the four classes (element indices; may lie outside the lattice: such lanes are masked)
		const int64_t pl = (int64_t)oz * (ny + 1) + oy;
		const char* gV = (const char*)(F.coeffs + (pl * (nx + 1) + ox));
		const char* gX = (const char*)(F.coeffs + (nv + 2 * (pl * nx + ox)));
		const char* gY = (const char*)(F.coeffs + (nv + 2 * nex + 2 * (((int64_t)ox * (nz + 1) + oz) * ny + oy)));
		const char* gZ = (const char*)(F.coeffs + (nv + 2 * nex + 2 * ney + 2 * (((int64_t)oy * (nx + 1) + ox) * nz + oz)));
		const char* safe = (const char*)F.coeffs;
		const uint32_t base = lds0 + (uint32_t)buf * 6400u;
#pragma unroll
		for (int r = 0; r < 2; ++r)
		{
			const K3Entry& q = r == 0 ? e0 : e1;
			// (lane 0 always takes part, with a harmless address, so that every one of the eight instructions is issued
			// whatever the masks: the waits below count them)
			const bool mx = q.live && k3_lt(ox, q.a, nx) && k3_le(oy, q.b, ny) && k3_le(oz, q.c, nz);
			const bool my = q.live && k3_lt(oy, q.a, ny) && k3_le(oz, q.b, nz) && k3_le(ox, q.c, nx);
			const bool mz = q.live && k3_lt(oz, q.a, nz) && k3_le(ox, q.b, nx) && k3_le(oy, q.c, ny);
			if (mx || lane == 0)
			{
				k3_glds16(mx ? gV + q.offV : safe, base + 0u * 1600u + 1024u * (uint32_t)r);
				k3_glds16(mx ? gX + q.offX : safe, base + 1u * 1600u + 1024u * (uint32_t)r);
			}
			if (my || lane == 0)
				k3_glds16(my ? gY + q.offY : safe, base + 2u * 1600u + 1024u * (uint32_t)r);
			if (mz || lane == 0)
				k3_glds16(mz ? gZ + q.offZ : safe, base + 3u * 1600u + 1024u * (uint32_t)r);
		}
	};

	double res = 0.0;
	int ci = 0, cj = 0, ck = -1;
	bool have = advance(ci, cj, ck);
	int buf = 0;
	if (have)
		issue(ci, cj, ck, buf);
	int li = -1, lj = -1;
	Axis1D ax = {}, ay = {};
	double wi = 0.0, wij = 0.0, mxmy = 0.0, mxpy = 0.0, pxmy = 0.0, pxpy = 0.0, x2y2 = 0.0;
	while (have)
	{
		int ni = ci, nj = cj, nk = ck;
		const bool more = advance(ni, nj, nk);
		if (more)
		{
			issue(ni, nj, nk, buf ^ 1);
			asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // the eight loads of THIS point's box have landed
		}
		else
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (ci != li)
		{
			wi = P.w[ci];
			ax = axis_eval(F, 0, x[0] + P.xi[ci]);
			li = ci;
			lj = -1;
		}
		if (cj != lj)
		{
			wij = wi * P.w[cj];
			ay = axis_eval(F, 1, x[1] + P.xi[cj]);
			mxmy = ax.m * ay.m;
			mxpy = ax.m * ay.p;
			pxmy = ax.p * ay.m;
			pxpy = ax.p * ay.p;
			x2y2 = ax.t2 + ay.t2;
			lj = cj;
		}
		const double wijk = wij * P.w[ck];
		const double yz = x[2] + P.xi[ck];
		Axis1D az;
		if (ztab)
		{
			// axis_eval(F, 2, yz) with (c0, c1) from the table
			az.inside = (F.dmin[2] <= yz) && (yz <= F.dmax[2]);
			uint32_t mi = (uint32_t)((yz - F.dmin[2]) * F.inv_cell[2]);
			if (mi >= nz)
				mi = nz - 1;
			if (!az.inside)
				mi = 0;
			az.mi = mi;
			const double2 cc = sT[az.inside ? (int)mi - zlo : 0];
			az.t = cc.x * yz - cc.y;
			az.t2 = az.t * az.t;
			az.m = 1.0 - az.t;
			az.p = 1.0 + az.t;
			const double f9 = 9.0 / 64.0 * (1.0 - az.t2);
			az.fm3 = f9 * (1.0 - 3.0 * az.t);
			az.fp3 = f9 * (1.0 + 3.0 * az.t);
		}
		else
			az = axis_eval(F, 2, yz);
		const bool inside = ax.inside && ay.inside && az.inside;
		double d = NOVAL;
		const int ox = __builtin_amdgcn_readlane(vgx, ci), oy = __builtin_amdgcn_readlane(vgy, cj), oz = __builtin_amdgcn_readlane(vgz, ck);
		const uint32_t lx = ax.mi - (uint32_t)ox, ly = ay.mi - (uint32_t)oy, lz = az.mi - (uint32_t)oz;
		const bool fits = lx < 4u && ly < 4u && lz < 4u;
		if (need && inside && fits)
		{
			const double2* eV = &sE[buf][0][lx + 4u * ly + 20u * lz];
			const double2* eX = &sE[buf][1][lx + 4u * ly + 20u * lz];
			const double2* eY = &sE[buf][2][ly + 4u * lz + 20u * lx];
			const double2* eZ = &sE[buf][3][lz + 4u * lx + 20u * ly];
			const double mz = az.m, pz = az.p;
			const double fac = 1.0 / 64.0 * (9.0 * (x2y2 + az.t2) - 19.0);
			bool ok = true;
			if (has_noval)
				ok = !tile_cell_has_novalue(F.tile_major, F.ntile, ax.mi, ay.mi, az.mi);
			double phi = 0.0;
			{
				const double2 c0 = eV[0], c1 = eV[4], c2 = eV[20], c3 = eV[24];
				phi += c0.x * (fac * mxmy * mz);
				phi += c0.y * (fac * pxmy * mz);
				phi += c1.x * (fac * mxpy * mz);
				phi += c1.y * (fac * pxpy * mz);
				phi += c2.x * (fac * mxmy * pz);
				phi += c2.y * (fac * pxmy * pz);
				phi += c3.x * (fac * mxpy * pz);
				phi += c3.y * (fac * pxpy * pz);
			}
			{
				const double mymz = ay.m * mz, mypz = ay.m * pz, pymz = ay.p * mz, pypz = ay.p * pz;
				const double2 c0 = eX[0], c1 = eX[20], c2 = eX[4], c3 = eX[24];
				phi += c0.x * (ax.fm3 * mymz);
				phi += c0.y * (ax.fp3 * mymz);
				phi += c1.x * (ax.fm3 * mypz);
				phi += c1.y * (ax.fp3 * mypz);
				phi += c2.x * (ax.fm3 * pymz);
				phi += c2.y * (ax.fp3 * pymz);
				phi += c3.x * (ax.fm3 * pypz);
				phi += c3.y * (ax.fp3 * pypz);
			}
			{
				const double mxmz = ax.m * mz, mxpz = ax.m * pz, pxmz = ax.p * mz, pxpz = ax.p * pz;
				const double2 c0 = eY[0], c1 = eY[20], c2 = eY[4], c3 = eY[24];
				phi += c0.x * (ay.fm3 * mxmz);
				phi += c0.y * (ay.fp3 * mxmz);
				phi += c1.x * (ay.fm3 * pxmz);
				phi += c1.y * (ay.fp3 * pxmz);
				phi += c2.x * (ay.fm3 * mxpz);
				phi += c2.y * (ay.fp3 * mxpz);
				phi += c3.x * (ay.fm3 * pxpz);
				phi += c3.y * (ay.fp3 * pxpz);
			}
			{
				const double2 c0 = eZ[0], c1 = eZ[20], c2 = eZ[4], c3 = eZ[24];
				phi += c0.x * (az.fm3 * mxmy);
				phi += c0.y * (az.fp3 * mxmy);
				phi += c1.x * (az.fm3 * mxpy);
				phi += c1.y * (az.fp3 * mxpy);
				phi += c2.x * (az.fm3 * pxmy);
				phi += c2.y * (az.fp3 * pxmy);
				phi += c3.x * (az.fm3 * pxpy);
				phi += c3.y * (az.fp3 * pxpy);
			}
			d = ok ? phi : NOVAL;
		}
		else if (need && inside)
		{
			// a cell the box does not cover: this lane's point the plain way (same bits: tests/test_density_map.py)
			const double y[3] = {x[0] + P.xi[ci], x[1] + P.xi[cj], yz};
			double g[3];
			d = interpolate_point_mode<false, kFieldClosed>(F, y, g);
		}
		if (need)
		{
			const double gamma = (d > P.h) ? 0.0 : 1.0 - d / P.h;
			// (through the scalar cache: the inline-asm loads above make the compiler treat ordinary memory as changing)
			const double wv = *(const DG_CONST_AS double*)(uintptr_t)(P.wtab + ((ci * 16 + cj) * 16 + ck));
			res += wijk * (gamma * wv);
		}
		ci = ni;
		cj = nj;
		ck = nk;
		have = more;
		buf ^= 1;
	}
	if (need)
	{
		res *= P.c0prod;
		v = P.rho0 * res;
	}
	if (ln.valid)
		L.out[ln.out_idx] = v;
}

// U, one piece: the slots [rank_begin, rank_end) of the gathered buffer -> reference node order.
// One thread per gathered value: contiguous reads, writes in contiguous runs of one plane.
__global__ __launch_bounds__(256) void k_unpack_ranks(const UnpackParams P)
{
	const uint64_t total = (uint64_t)(P.rank_end - P.rank_begin) * P.stride;
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x)
	{
		const uint32_t r = (uint32_t)P.rank_begin + (uint32_t)(e / P.stride);
		const uint64_t off = e % P.stride;
		if (off >= P.count[r])
			continue; // padding of the slot
		P.field[unpack_dest(P, r, off)] = P.gathered[(uint64_t)r * P.stride + off];
	}
}

// K3 pre-pass: does the field hold values for which skipping the zero-weight quadrature points would
// change the result (NaN, Inf, |c| >= 1e290)?  DBL_MAX is the regular "no value" marker.
__global__ __launch_bounds__(256) void k_field_check(const double* __restrict__ coeffs, uint64_t n, uint32_t* __restrict__ unsafe)
{
	bool bad = false, nov = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
	{
		const double c = coeffs[i];
		nov = nov || c == 1.7976931348623157e308;
		bad = bad || (c != 1.7976931348623157e308 && !(fabs(c) < 1.0e290));
	}
	// bit 0: values that forbid skipping the zero-weight points; bit 1: the field holds "no value" coefficients at all
	const uint32_t bits = (__ballot(bad) != 0ull ? 1u : 0u) | (__ballot(nov) != 0ull ? 2u : 0u);
	if (bits != 0u && (threadIdx.x & 63u) == 0u)
		atomicOr(unsafe, bits);
}

// ---- reduceField (dg_kernels.h: reduce_field_device) --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_keep(const double* __restrict__ v, uint64_t n, const ReducePredicate P, uint8_t* __restrict__ keep,
													  uint8_t* __restrict__ used)
{
	for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n; l += (uint64_t)gridDim.x * blockDim.x)
	{
		const double x = v[l];
		const bool ok = P.closed ? (P.lo <= x && x <= P.hi) : (P.lo < x + P.offset && x - P.offset < P.hi);
		keep[l] = (ok && x != 1.7976931348623157e308) ? 1 : 0;
		used[l] = 0;
	}
}
// a cell survives if any of its 32 nodes is kept (:1091-1098)
__global__ __launch_bounds__(256) void k_reduce_cell_flags(const uint32_t rx, const uint32_t ry, const uint32_t rz, const uint8_t* __restrict__ keep,
															uint32_t* __restrict__ flag)
{
	const uint64_t n_cells = (uint64_t)rx * ry * rz;
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	const uint32_t res[3] = {rx, ry, rz};
	const uint32_t n01 = rx * ry;
	const uint32_t k = (uint32_t)(c / n01), r = (uint32_t)(c % n01);
	uint32_t idx[32];
	cell_node_indices(r % rx, r / rx, k, res, idx);
	bool any = false;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		any = any || keep[idx[j]] != 0;
	flag[c] = any ? 1u : 0u;
}
// cell map + the nodes the surviving cells reference (:1099-1128)
__global__ __launch_bounds__(256) void k_reduce_cell_map(const uint32_t rx, const uint32_t ry, const uint32_t rz, const uint32_t* __restrict__ flag,
														  const uint32_t* __restrict__ row, uint32_t* __restrict__ cell_map, uint8_t* __restrict__ used)
{
	const uint64_t n_cells = (uint64_t)rx * ry * rz;
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	if (flag[c] == 0u)
	{
		cell_map[c] = 0xffffffffu;
		return;
	}
	cell_map[c] = row[c];
	const uint32_t res[3] = {rx, ry, rz};
	const uint32_t n01 = rx * ry;
	const uint32_t k = (uint32_t)(c / n01), r = (uint32_t)(c % n01);
	uint32_t idx[32];
	cell_node_indices(r % rx, r / rx, k, res, idx);
#pragma unroll
	for (int j = 0; j < 32; ++j)
		used[idx[j]] = 1; // same value from every writer
}
struct ReduceGeom
{
	uint32_t res[3];
	double dmin[3], cell[3];
	double zscale;
};
// Morton keys of the surviving nodes, with the reference's arithmetic (:1110-1115)
__global__ __launch_bounds__(256) void k_reduce_keys(const ReduceGeom G, const uint32_t* __restrict__ node, uint64_t m, uint64_t* __restrict__ keys)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
	{
		double x[3];
		node_position_flat(node[i], G.res, G.dmin, G.cell, x);
		keys[i] = reference_z_value(x, G.zscale);
	}
}
// after the sort: new numbering, coefficients in the new order, and whether two survivors share a key
__global__ __launch_bounds__(256) void k_reduce_renumber(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ node, uint64_t m,
														  const double* __restrict__ v, uint32_t* __restrict__ new_id, double* __restrict__ out,
														  uint32_t* __restrict__ tied)
{
	bool tie = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
	{
		const uint32_t l = node[i];
		new_id[l] = (uint32_t)i;
		out[i] = v[l];
		tie = tie || (i + 1 < m && keys[i] == keys[i + 1]);
	}
	if (__ballot(tie) != 0ull && (threadIdx.x & 63u) == 0u)
		atomicOr(tied, 1u);
}
// rows of the surviving cells with the new node numbers (:1161-1173)
__global__ __launch_bounds__(256) void k_reduce_rows(const uint32_t rx, const uint32_t ry, const uint32_t rz, const uint32_t* __restrict__ cell_map,
													  const uint32_t* __restrict__ new_id, uint32_t* __restrict__ rows)
{
	const uint64_t n_cells = (uint64_t)rx * ry * rz;
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	const uint32_t row = cell_map[c];
	if (row == 0xffffffffu)
		return;
	const uint32_t res[3] = {rx, ry, rz};
	const uint32_t n01 = rx * ry;
	const uint32_t k = (uint32_t)(c / n01), r = (uint32_t)(c % n01);
	uint32_t idx[32];
	cell_node_indices(r % rx, r / rx, k, res, idx);
	uint32_t* o = rows + 32 * (size_t)row;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		o[j] = new_id[idx[j]];
}

} // namespace

hipError_t reduce_field_device(const uint32_t res[3], const double dmin[3], const double cell[3], const double inv_cell[3],
							   const double* d_coeffs, uint64_t n, const ReducePredicate& pred, ReduceResult& out, hipStream_t stream)
{
	const uint64_t n_cells = (uint64_t)res[0] * res[1] * res[2];
	struct Scratch
	{
		std::vector<void*> p;
		~Scratch()
		{
			for (void* q : p)
				(void)hipFree(q);
		}
		hipError_t get(void** q, size_t bytes)
		{
			const hipError_t e = hipMalloc(q, bytes ? bytes : 8);
			if (e == hipSuccess)
				p.push_back(*q);
			return e;
		}
	} S;
#define DG_TRY(x)                \
	do                           \
	{                            \
		const hipError_t e_ = (x); \
		if (e_ != hipSuccess)    \
			return e_;           \
	} while (0)
	uint8_t *keep = nullptr, *used = nullptr;
	uint32_t *flag = nullptr, *row = nullptr, *cell_map = nullptr, *node = nullptr, *node_sorted = nullptr, *new_id = nullptr, *counts = nullptr;
	DG_TRY(S.get((void**)&keep, n));
	DG_TRY(S.get((void**)&used, n));
	DG_TRY(S.get((void**)&flag, n_cells * 4));
	DG_TRY(S.get((void**)&row, n_cells * 4));
	DG_TRY(S.get((void**)&counts, 16));
	DG_TRY(hipMalloc((void**)&cell_map, n_cells * 4));
	out.d_cell_map = cell_map; // results are freed by the caller (also on failure: it owns `out`)
	DG_TRY(hipMemsetAsync(counts, 0, 16, stream));
	const uint32_t wide = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256ull * 64ull);
	const uint32_t cgrid = (uint32_t)((n_cells + 255) / 256);
	hipLaunchKernelGGL(k_reduce_keep, dim3(wide), dim3(256), 0, stream, d_coeffs, n, pred, keep, used);
	hipLaunchKernelGGL(k_reduce_cell_flags, dim3(cgrid), dim3(256), 0, stream, res[0], res[1], res[2], keep, flag);
	size_t tmp_bytes = 0, b2 = 0;
	DG_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, flag, row, 0u, (size_t)n_cells, rocprim::plus<uint32_t>(), stream));
	DG_TRY(rocprim::select(nullptr, b2, rocprim::counting_iterator<uint32_t>(0u), used, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, stream));
	tmp_bytes = std::max(tmp_bytes, b2);
	void* tmp = nullptr;
	DG_TRY(S.get(&tmp, tmp_bytes));
	size_t tb = tmp_bytes;
	DG_TRY(rocprim::exclusive_scan(tmp, tb, flag, row, 0u, (size_t)n_cells, rocprim::plus<uint32_t>(), stream));
	hipLaunchKernelGGL(k_reduce_cell_map, dim3(cgrid), dim3(256), 0, stream, res[0], res[1], res[2], flag, row, cell_map, used);
	DG_TRY(S.get((void**)&node, n * 4)); // survivors in node order (at most n)
	tb = tmp_bytes;
	DG_TRY(rocprim::select(tmp, tb, rocprim::counting_iterator<uint32_t>(0u), used, node, counts, (size_t)n, stream));
	DG_TRY(hipGetLastError());
	// sizes: surviving nodes, surviving cells
	uint32_t h_m = 0, h_last_row = 0, h_last_flag = 0;
	DG_TRY(hipMemcpyAsync(&h_m, counts, 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipMemcpyAsync(&h_last_row, row + (n_cells - 1), 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipMemcpyAsync(&h_last_flag, flag + (n_cells - 1), 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipStreamSynchronize(stream));
	const uint64_t m = h_m, rows = (uint64_t)h_last_row + h_last_flag;
	out.n_nodes_out = m;
	out.n_rows = rows;
	DG_TRY(hipMalloc(&out.d_coeffs, std::max<uint64_t>(m, 1) * sizeof(double)));
	DG_TRY(hipMalloc(&out.d_cells, std::max<uint64_t>(rows, 1) * 32 * sizeof(uint32_t)));
	if (m == 0)
		return hipSuccess;
	uint64_t *keys = nullptr, *keys_sorted = nullptr;
	DG_TRY(S.get((void**)&keys, m * 8));
	DG_TRY(S.get((void**)&keys_sorted, m * 8));
	DG_TRY(S.get((void**)&node_sorted, m * 4));
	DG_TRY(S.get((void**)&new_id, n * 4));
	ReduceGeom G;
	for (int d = 0; d < 3; ++d)
	{
		G.res[d] = res[d];
		G.dmin[d] = dmin[d];
		G.cell[d] = cell[d];
	}
	G.zscale = 4.0 * std::min(std::min(inv_cell[0], inv_cell[1]), inv_cell[2]); // :1112
	const uint32_t mwide = (uint32_t)std::min<uint64_t>((m + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_reduce_keys, dim3(mwide), dim3(256), 0, stream, G, node, m, keys);
	size_t sort_bytes = 0;
	DG_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys_sorted, node, node_sorted, (size_t)m, 0u, (unsigned)kReferenceZBits, stream));
	void* sort_tmp = nullptr;
	DG_TRY(S.get(&sort_tmp, sort_bytes));
	DG_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys_sorted, node, node_sorted, (size_t)m, 0u, (unsigned)kReferenceZBits, stream));
	hipLaunchKernelGGL(k_reduce_renumber, dim3(mwide), dim3(256), 0, stream, keys_sorted, node_sorted, m, d_coeffs, new_id,
					   static_cast<double*>(out.d_coeffs), counts + 1);
	hipLaunchKernelGGL(k_reduce_rows, dim3(cgrid), dim3(256), 0, stream, res[0], res[1], res[2], cell_map, new_id, static_cast<uint32_t*>(out.d_cells));
	DG_TRY(hipGetLastError());
	uint32_t h_tied = 0;
	DG_TRY(hipMemcpyAsync(&h_tied, counts + 1, 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipStreamSynchronize(stream));
	out.tied_keys = (int)h_tied;
#undef DG_TRY
	return hipSuccess;
}

namespace
{
} // namespace

hipError_t launch_density_bricks(const SampleParams& layout, const FieldDev& f, uint64_t n_coeffs, const DensityParams& p,
								 hipStream_t stream)
{
	if (layout.total_bricks == 0)
		return hipSuccess;
	if (p.unsafe != nullptr)
	{
		const hipError_t e = hipMemsetAsync(const_cast<uint32_t*>(p.unsafe), 0, sizeof(uint32_t), stream);
		if (e != hipSuccess)
			return e;
		const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_coeffs + 255) / 256, 256ull * 16ull);
		hipLaunchKernelGGL(k_field_check, dim3(blocks), dim3(256), 0, stream, f.coeffs, n_coeffs, const_cast<uint32_t*>(p.unsafe));
	}
	static_assert(kWavesPerBlock == 1, "k_density_bricks assumes one brick per block");
	const dim3 grid(layout.blocks_per_xcd * 8u), block(64);
	const bool unreduced = f.cells == nullptr && f.cell_map == nullptr; // staged evaluator
	switch (field_mode(f))
	{
	case kFieldXMajor:
	{
		// row blocks over the whole lattice (dg_layout.h layout_density_rows()); the per-cell "no value" bits first
		if (p.row_shape == 0 || !unreduced)
			return hipErrorInvalidValue;
		if (f.xmajor_flags != nullptr && p.unsafe != nullptr)
			hipLaunchKernelGGL(k_xmajor_flags, dim3(2048), dim3(256), 0, stream, f, p.unsafe, const_cast<uint64_t*>(f.xmajor_flags));
		if (p.row_shape == kRowShapeCells) // one lane per lattice point, all seven nodes of the point (dg_density_cells.h)
		{
			if (f.xmajor_flags == nullptr || !k3c_geometry_fits(f.res))
				return hipErrorInvalidValue;
			const K3CellsGeom g = k3c_geometry(f);
			if (p.row_waves3)
				hipLaunchKernelGGL((k_density_cells<3>), grid, block, 0, stream, layout, f, p, g);
			else
				hipLaunchKernelGGL((k_density_cells<2>), grid, block, 0, stream, layout, f, p, g);
			break;
		}
#define DG_K3_ROWS(LX, LY, LZ)                                                                                          \
	if (p.row_waves3)                                                                                                   \
		hipLaunchKernelGGL((k_density_rows<kFieldXMajor, LX, LY, LZ, 3>), grid, block, 0, stream, layout, f, p);       \
	else                                                                                                                \
		hipLaunchKernelGGL((k_density_rows<kFieldXMajor, LX, LY, LZ, 2>), grid, block, 0, stream, layout, f, p)
		switch (p.row_shape)
		{
		case 4: DG_K3_ROWS(8, 4, 2); break;
		default: DG_K3_ROWS(16, 2, 2); break;
		}
#undef DG_K3_ROWS
		break;
	}
	case kFieldTileMajor:
		// pair_nodes: two edge nodes per lane (the layout counts double bricks: dg_layout.h pair_bricks()); lds_waves > 0:
		// the experiment that stages the coefficients through LDS (DG_K3_LDS)
		if (layout.pair_nodes && layout.pair_nodes >= 3) hipLaunchKernelGGL((k_density_pairs<kFieldTileMajor, 3>), grid, block, 0, stream, layout, f, p);
		else if (layout.pair_nodes) hipLaunchKernelGGL((k_density_pairs<kFieldTileMajor, 2>), grid, block, 0, stream, layout, f, p);
		else if (p.lds_waves >= 3) hipLaunchKernelGGL((k_density_bricks_lds<3>), grid, block, 0, stream, layout, f, p);
		else if (p.lds_waves > 0) hipLaunchKernelGGL((k_density_bricks_lds<2>), grid, block, 0, stream, layout, f, p);
		else hipLaunchKernelGGL((k_density_bricks<true, kFieldTileMajor>), grid, block, 0, stream, layout, f, p);
		break;
	case kFieldCellMajor:
		if (unreduced) hipLaunchKernelGGL((k_density_bricks<true, kFieldCellMajor>), grid, block, 0, stream, layout, f, p);
		else hipLaunchKernelGGL((k_density_bricks<false, kFieldCellMajor>), grid, block, 0, stream, layout, f, p);
		break;
	case kFieldTable: hipLaunchKernelGGL((k_density_bricks<false, kFieldTable>), grid, block, 0, stream, layout, f, p); break;
	default:
		if (unreduced && layout.pair_nodes) hipLaunchKernelGGL((k_density_pairs<kFieldClosed, 2>), grid, block, 0, stream, layout, f, p);
		else if (unreduced) hipLaunchKernelGGL((k_density_bricks<true, kFieldClosed>), grid, block, 0, stream, layout, f, p);
		else hipLaunchKernelGGL((k_density_bricks<false, kFieldClosed>), grid, block, 0, stream, layout, f, p);
	}
	return hipGetLastError();
}

hipError_t launch_expand_tiles(const FieldDev& f, uint64_t n_tiles, double* d_out, hipStream_t stream)
{
	if (n_tiles == 0)
		return hipSuccess;
	const uint64_t total = n_tiles * kTmNodes;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_expand_tiles, dim3(blocks), dim3(256), 0, stream, f, n_tiles, d_out);
	hipLaunchKernelGGL(k_tile_flags, dim3((uint32_t)((n_tiles + 3) / 4)), dim3(256), 0, stream, n_tiles, d_out);
	return hipGetLastError();
}

hipError_t launch_xmajor_copy(const FieldDev& f, double* d_out, hipStream_t stream)
{
	const uint64_t n_pairs = xmajor_doubles(f.res) / 2;
	if (n_pairs == 0 || n_pairs > 0xffffffffull)
		return n_pairs == 0 ? hipSuccess : hipErrorInvalidValue;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_pairs + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_xmajor_copy, dim3(blocks), dim3(256), 0, stream, f, (uint32_t)n_pairs, d_out);
	return hipGetLastError();
}

hipError_t launch_interpolate_band(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 63) / 64, 256ull * 64ull);
	const bool table = f.cells != nullptr;
#define DG_K2_BAND(G, M) hipLaunchKernelGGL((k_interpolate_band<G, M>), dim3(blocks), dim3(64), 0, stream, f, d_xyz, n, d_phi, d_grad)
	if (d_grad && table) DG_K2_BAND(true, kFieldTable);
	else if (d_grad) DG_K2_BAND(true, kFieldClosed);
	else if (table) DG_K2_BAND(false, kFieldTable);
	else DG_K2_BAND(false, kFieldClosed);
#undef DG_K2_BAND
	return hipGetLastError();
}
hipError_t launch_band_flags(const FieldDev& f, uint64_t n_rows, double lo, double hi, uint32_t* d_flag, hipStream_t stream)
{
	if (n_rows == 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_band_flags, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, stream, f, n_rows, lo, hi, d_flag);
	return hipGetLastError();
}
// exclusive scan of the flags (rocPRIM); tmp: scratch of *tmp_bytes (query with d_tmp == nullptr)
hipError_t band_scan(const uint32_t* d_flag, uint32_t* d_pos, uint64_t n_rows, void* d_tmp, size_t* tmp_bytes, hipStream_t stream)
{
	return rocprim::exclusive_scan(d_tmp, *tmp_bytes, d_flag, d_pos, 0u, (size_t)n_rows, rocprim::plus<uint32_t>(), stream);
}
hipError_t launch_band_expand(const FieldDev& f, uint64_t n_rows, const uint32_t* d_flag, const uint32_t* d_pos, uint64_t* d_bits, uint32_t* d_rank,
							  double* d_rows, hipStream_t stream)
{
	if (n_rows == 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_band_expand, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, stream, f, n_rows, d_flag, d_pos, d_bits, d_rank, d_rows);
	return hipGetLastError();
}

hipError_t launch_expand_cells(const FieldDev& f, uint64_t n_rows, double* d_out, hipStream_t stream)
{
	if (n_rows == 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_expand_cells, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, stream, f, n_rows, d_out);
	return hipGetLastError();
}

template <bool POINTS>
static hipError_t launch_k1(const SampleParams& p, hipStream_t stream)
{
	if (p.total_bricks == 0)
		return hipSuccess;
	if (p.brick_blocking != 0) // the K1 kernels only know the row-major brick order (map_brick_order<false>)
		return hipErrorInvalidValue;
	const uint32_t grid = p.blocks_per_xcd * 8u;
	const size_t lds = (size_t)kWavesPerBlock * p.mesh.stack_levels * 64 * sizeof(float);
	if (p.filtered != 0)
	{
		static_assert(kWavesPerBlock == 1, "k_sample_fast assumes one brick per block");
		const size_t lds_fast = (size_t)p.mesh.stack_levels * 64 * sizeof(uint16_t) + (size_t)(kFastListCap + 1) * 64 * sizeof(int);
		hipLaunchKernelGGL(k_sample_fast<POINTS>, dim3(grid), dim3(64), lds_fast, stream, p);
	}
	else
		hipLaunchKernelGGL(k_sample_nodes<POINTS>, dim3(grid), dim3(64 * kWavesPerBlock), lds, stream, p);
	if (p.ovf.count != nullptr)
	{
		const size_t lds1 = (size_t)p.mesh.stack_levels * 64 * sizeof(float);
		const uint32_t jobs = p.ovf.slots * (uint32_t)p.mesh.n_sub;
		hipLaunchKernelGGL(k_heavy_subtrees<POINTS>, dim3(jobs < 32768u ? jobs : 32768u), dim3(64), lds1, stream, p);
		hipLaunchKernelGGL(k_heavy_finish<POINTS>, dim3(p.ovf.slots), dim3(64), 0, stream, p);
	}
	return hipGetLastError();
}

hipError_t launch_sample_nodes(const SampleParams& p, hipStream_t stream) { return launch_k1<false>(p, stream); }

// the binning passes shared by K2 and K1p: S.flag / S.perm describe the order to process the points in
static uint32_t key_bits(uint32_t n_tiles)
{
	uint32_t bits = 1;
	while (bits < 32 && (1u << bits) < n_tiles)
		++bits;
	return bits;
}
size_t bin_sort_tmp_bytes(uint64_t n, uint32_t n_tiles)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
									(uint32_t*)nullptr, (size_t)n, 0u, 32u, (hipStream_t) nullptr); // (all 32 bits: an upper bound for any key width)
	return bytes;
}
// the binning passes shared by K2 and K1p: probe (always), and -- if the host predicts an unordered batch
// (S.sort_launched) -- tile keys + radix sort, which leaves the processing order in S.perm
static hipError_t launch_binning(const TileGrid& probe_tiles, const TileGrid& tiles, const double* d_xyz, uint64_t n, const BinScratch& S, uint32_t one_in, hipStream_t stream)
{
	hipLaunchKernelGGL(k_bin_probe, dim3(1), dim3(256), 0, stream, probe_tiles, d_xyz, n, S, one_in);
	if (S.sort_launched == 0)
		return hipGetLastError();
	const uint32_t wide = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_bin_keys, dim3(wide), dim3(256), 0, stream, tiles, d_xyz, n, S);
	size_t bytes = S.sort_tmp_bytes;
	const hipError_t e = rocprim::radix_sort_pairs(S.sort_tmp, bytes, (const uint32_t*)S.keys, S.keys_out, (const uint32_t*)S.vals, S.perm,
												  (size_t)n, 0u, tile_key_bits(tiles), stream);
	return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_signed_distance(const SampleParams& p, const TileGrid* tiles, const BinScratch* scratch, hipStream_t stream)
{
	if (p.pts.n == 0)
		return hipSuccess;
	if (tiles != nullptr && scratch != nullptr)
	{
		// K1p gains from 3-D compactness even for row-ordered input: bin unless consecutive points share a tile 15 times out of 16
		const hipError_t e = launch_binning(*tiles, *tiles, p.pts.xyz, p.pts.n, *scratch, 16u, stream);
		if (e != hipSuccess)
			return e;
	}
	return launch_k1<true>(p, stream);
}

hipError_t launch_unpack(const UnpackParams& p, hipStream_t stream)
{
	const uint64_t total = p.class_off[4];
	if (total == 0)
		return hipSuccess;
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 256ull * 32ull)
		blocks = 256ull * 32ull;
	hipLaunchKernelGGL(k_unpack_shards, dim3((uint32_t)blocks), dim3(256), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_unpack_ranks(const UnpackParams& p, hipStream_t stream)
{
	const uint64_t total = (uint64_t)(p.rank_end - p.rank_begin) * p.stride;
	if (total == 0)
		return hipSuccess;
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 256ull * 32ull)
		blocks = 256ull * 32ull;
	hipLaunchKernelGGL(k_unpack_ranks, dim3((uint32_t)blocks), dim3(256), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_interpolate(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad,
							  hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	const uint32_t grid = k2_grid(n);
#define DG_K2_LAUNCH(MODE)                                                                                                  \
	if (d_grad)                                                                                                             \
		hipLaunchKernelGGL((k_interpolate<true, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad);      \
	else                                                                                                                    \
		hipLaunchKernelGGL((k_interpolate<false, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad);
	switch (field_mode(f))
	{
	case kFieldTileMajor: DG_K2_LAUNCH(kFieldTileMajor) break;
	case kFieldCellMajor: DG_K2_LAUNCH(kFieldCellMajor) break;
	case kFieldTable: DG_K2_LAUNCH(kFieldTable) break;
	default: DG_K2_LAUNCH(kFieldClosed)
	}
#undef DG_K2_LAUNCH
	return hipGetLastError();
}

hipError_t launch_interpolate_rows(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	if (f.cell_major == nullptr)
		return hipErrorInvalidValue;
	const uint64_t waves = (n + 63) / 64;
	const uint32_t grid = (uint32_t)std::min<uint64_t>(waves, 256ull * 64ull); // grid-stride beyond 64 waves per CU
	if (d_grad)
		hipLaunchKernelGGL((k_interpolate_rows<true>), dim3(grid), dim3(64), 0, stream, f, d_xyz, n, d_phi, d_grad);
	else
		hipLaunchKernelGGL((k_interpolate_rows<false>), dim3(grid), dim3(64), 0, stream, f, d_xyz, n, d_phi, d_grad);
	return hipGetLastError();
}

hipError_t launch_interpolate_binned(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad,
									 const BinScratch& S, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	// K2: row-ordered queries are coherent enough; bin only if more than a quarter of the steps change tile
	const hipError_t e = launch_binning(field_tiles(f), field_tiles(f, kSortCells), d_xyz, n, S, 4u, stream);
	if (e != hipSuccess)
		return e;
	const uint32_t grid = k2_grid(n);
#define DG_K2_LAUNCH(MODE)                                                                                                        \
	if (d_grad)                                                                                                                   \
		hipLaunchKernelGGL((k_interpolate_binned<true, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad, S);  \
	else                                                                                                                          \
		hipLaunchKernelGGL((k_interpolate_binned<false, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad, S);
	switch (field_mode(f))
	{
	case kFieldTileMajor: DG_K2_LAUNCH(kFieldTileMajor) break;
	case kFieldCellMajor: DG_K2_LAUNCH(kFieldCellMajor) break;
	case kFieldTable: DG_K2_LAUNCH(kFieldTable) break;
	default: DG_K2_LAUNCH(kFieldClosed)
	}
#undef DG_K2_LAUNCH
	return hipGetLastError();
}

} // namespace dg
