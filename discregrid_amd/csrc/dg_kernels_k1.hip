// dg_kernels_k1.hip -- K1 / K1p: the hand-written CDNA4 (gfx950) SDF sampling kernels.
//
//   K1  k_sample_nodes     lattice node -> signed distance to the mesh   (addFunction node loop,
//                          discregrid/src/cubic_lagrange_discrete_grid.cpp:806-831 +
//                          TriangleMeshDistance.h:269-308, 514-562, 564-820)
//   K1p k_sample_nodes<1>  same kernels for caller-supplied points (TriangleMeshDistance.h:269-314)
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
#include <stdint.h>
#include <algorithm>
#include "dg_kernels.h"
#include "dg_device.h"
#include "dg_traverse.h"

namespace dg
{
namespace
{

#ifndef DG_EPILOGUE_HEAD
#define DG_EPILOGUE_HEAD 1 // rounds of the per-lane double tests before the rest is pooled (k_sample_fast's epilogue; 0: lane by lane to the end)
#endif
#ifndef DG_HEAVY_WAVES_PER_BLOCK
#define DG_HEAVY_WAVES_PER_BLOCK 2
#endif
static const int kHeavyWavesPerBlock = DG_HEAVY_WAVES_PER_BLOCK; // k_heavy_subtrees: jobs (waves) per block
#ifndef DG_K1_MIN_WAVES
#define DG_K1_MIN_WAVES 8 // K1 is issue bound and hides its scalar-load latency with waves: cap it at 64 VGPRs
#endif

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
	__device__ __forceinline__ int uniform(int v) const { return dg::uniform(v); }
	__device__ __forceinline__ void pick(LaneVar<float, 1>& out, const LaneVar<f2, 1>& lb, bool first) const { out[0] = first ? lb[0].x : lb[0].y; }
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
	__device__ __forceinline__ void push(int sp, int info, const LaneVar<f2, 1>& lb, bool second)
	{
		// entry sp lives in lane sp: ONE v_writelane_b32 (a lane select costs a compare, a move of the info word and the select);
		// the lane index goes through M0 (two different SGPR operands would exceed the constant-bus limit of the encoding)
		__asm__("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stackv) : "s"(info), "s"(sp) : "m0");
		park_bound(lds_lb, sp * 64 + lane_id, second ? lb[0].y : lb[0].x);
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
	__device__ __forceinline__ void note_dead() const {}
	__device__ __forceinline__ void note_push() const {}
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
	auto lane_query = [&](int) DG_LANE -> LaneQuery& { return q; };
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
	auto lane_state = [&](int) DG_LANE -> FastLane& { return f; };
	auto lane_list = [&](int) DG_LANE { return list_base; };
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
	// Lists are short but uneven: on the judged workload 34 % of the lanes hold one candidate, 46 % two, 11 % six (the lanes next
	// to a mesh vertex) -- lane by lane the wave runs as many rounds as its LONGEST list (5.5 on average) at 40 % lane
	// utilisation.  So every lane runs the first DG_EPILOGUE_HEAD round(s) on its own candidates, and what is left (1.2 candidates
	// per lane after one round) is POOLED: the (owner, triangle) pairs laid out contiguously (the bound stack's LDS is free by
	// now), 64 pairs tested per round with the owner's point fetched by ds_bpermute, the values handed back through LDS (over the
	// lists, which are no longer needed), and every owner offers its pooled values in list order -- the same values in the same
	// order, hence the same winner.  Same-box A/B at 256^3 (icosphere / bunny / dragon, ms): lane by lane 14.84 / 17.30 / 16.57,
	// head 1 14.44 / 17.27 / 16.57, head 2 14.52 / 17.23 / 16.53, head 3 14.67 / 17.28 / 16.56 (profiles/r05_k1_epilogue_ab.txt).
	// (Pooling EVERYTHING was measured 1.1 % slower in round 3: the per-lane loops that build and read back the pool then run as
	// many rounds as the longest list again.)
	int k = 0;
	for (; (DG_EPILOGUE_HEAD <= 0 || k < DG_EPILOGUE_HEAD) && __ballot(k < n_cand) != 0ull; ++k)
	{
		if (k < n_cand)
		{
			const int tri = lds_list[k * 64 + lane];
			const Hit h = tri_closest<false>(P.mesh.tris[tri], q.px, q.py, q.pz);
			offer(q, h.d2, tri);
		}
	}
	const int tail_n = n_cand > k ? n_cand - k : 0; // (k is wave-uniform)
	if (__ballot(tail_n > 0) != 0ull)
	{
		// `before`: pooled pairs of the lanes below this one; T: pairs in all (tail_n <= kFastListCap < 16: four bit planes)
		uint32_t before = 0, T = 0;
#pragma unroll
		for (int b = 3; b >= 0; --b)
		{
			const unsigned long long m = __ballot(((tail_n >> b) & 1) != 0);
			before += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
			T += (uint32_t)__popcll(m) << b;
		}
		const uint32_t item_cap = (uint32_t)P.mesh.stack_levels * 32u, res_cap = (uint32_t)(kFastListCap + 1) * 32u;
		const bool pooled = T <= item_cap && T <= res_cap && T <= P.ovf.pool_cap;
		if (P.ovf.stats != nullptr && lane == 0) // (test hook, wave-uniform pointer: null in production)
			atomicAdd(&P.ovf.stats[pooled ? 0 : 1], 1u);
		if (pooled)
		{
			uint32_t* items = (uint32_t*)lds_lb16;
			double* res = (double*)lds_list;
			for (int j = 0; __ballot(j < tail_n) != 0ull; ++j)
				if (j < tail_n)
					items[before + (uint32_t)j] = ((uint32_t)lane << 26) | (uint32_t)lds_list[(k + j) * 64 + lane];
			__syncthreads(); // (one wave per workgroup: the pool is complete, the lists have been read)
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
			for (int j = 0; __ballot(j < tail_n) != 0ull; ++j)
				if (j < tail_n)
					offer(q, res[before + (uint32_t)j], (int)(items[before + (uint32_t)j] & 0x3ffffffu));
		}
		else
		{
			for (; __ballot(k < n_cand) != 0ull; ++k)
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
// starting from the parked bests.  ONE job per wave, kHeavyWavesPerBlock waves per block, a block for every job a full
// set of slots could bring (the number of parked bricks is only known on the device; waves without a job leave at once).
// Four fifths of the jobs end at their subtree's root pair, a few -- the bricks next to the centre of a sphere-like mesh --
// test every triangle of the subtree (tests/perf/emu_heavy_study.py): dealt several to a wave (a capped grid, grid-stride)
// the launch lasts as long as the unluckiest wave's jobs in a row -- icosphere 256^3: 0.58 ms with 32 768 waves, 0.47 with
// 131 071, 0.41 one job each.  (Drawing the jobs from a counter is no way out: 161 000 atomic adds on one address take 2 ms.)
// Two waves per block: a launch without a parked brick pays for its empty blocks -- bunny 256^3: 55 us with one wave per
// block, 28 with two, 15 with four (8 before), against 0.41 / 0.44 / 0.46 ms on the icosphere.
template <bool POINTS>
__global__ __launch_bounds__(64 * kHeavyWavesPerBlock) void k_heavy_subtrees(const SampleParams P)
{
	const uint32_t n_sub = (uint32_t)P.mesh.n_sub;
	const uint32_t parked = min(*P.ovf.count, P.ovf.slots);
	const int wave = uniform((int)(threadIdx.x >> 6));
	const int lane = (int)(threadIdx.x & 63u);
	extern __shared__ __attribute__((aligned(16))) float lds_lb[]; // [waves][stack_levels][64]
	const uint32_t job = blockIdx.x * (uint32_t)kHeavyWavesPerBlock + (uint32_t)wave;
	if (job >= parked * n_sub)
		return;
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
	traverse(P.mesh, q, lds_lb + wave * (P.mesh.stack_levels * 64), P.mesh.sub_roots[s], nullptr, 0u, 0);
	const size_t at = ((size_t)slot * kSubtrees + s) * 64 + (size_t)lane;
	P.ovf.cand_d2[at] = q.best_d2;
	P.ovf.cand_tri[at] = q.best_tri;
}

// Heavy bricks, step 3: per lane the minimum over the subtrees (the parked best is part of every
// candidate; of exactly tied candidates the lowest subtree wins), then K1's epilogue.
template <bool POINTS>
__global__ __launch_bounds__(64) void k_heavy_finish(const SampleParams P)
{
	const uint32_t slot = blockIdx.x;
	if (slot >= min(P.ovf.count[0], P.ovf.slots))
		return;
	const int lane = (int)threadIdx.x;
	const LaneTask t = lane_task<POINTS>(P, (uint64_t)P.ovf.brick[slot], lane);
	LaneQuery q;
	init_query(P.mesh.origin, P.mesh.mesh_l1, t.sample, t.x0, t.x1, t.x2, q);
	// (eight subtrees' candidates are fetched before the first is looked at: one subtree at a time the loop is a chain of 256
	// memory round trips)
	if (t.sample)
		for (int s0 = 0; s0 < P.mesh.n_sub; s0 += 8)
		{
			int tri[8];
			double d2[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
			{
				const int s = s0 + k < P.mesh.n_sub ? s0 + k : P.mesh.n_sub - 1;
				const size_t at = ((size_t)slot * kSubtrees + (size_t)s) * 64 + (size_t)lane;
				tri[k] = P.ovf.cand_tri[at];
				d2[k] = P.ovf.cand_d2[at];
			}
#pragma unroll
			for (int k = 0; k < 8; ++k)
				if (s0 + k < P.mesh.n_sub && tri[k] >= 0)
					offer(q, d2[k], tri[k]);
		}
	write_result<POINTS>(P, t, q);
}

} // namespace

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
		hipLaunchKernelGGL(k_heavy_subtrees<POINTS>, dim3((jobs + kHeavyWavesPerBlock - 1) / kHeavyWavesPerBlock), dim3(64 * kHeavyWavesPerBlock),
						   lds1 * kHeavyWavesPerBlock, stream, p);
		hipLaunchKernelGGL(k_heavy_finish<POINTS>, dim3(p.ovf.slots), dim3(64), 0, stream, p);
	}
	return hipGetLastError();
}

hipError_t launch_sample_nodes(const SampleParams& p, hipStream_t stream) { return launch_k1<false>(p, stream); }

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

} // namespace dg
