// dg_host_query.h -- the packet traversal of K1 (dg_traverse.h) on the HOST: a wave context whose lanes are array elements,
// and TriangleMeshDistance::signed_distance for ONE point on the calling thread as its one-lane instantiation (see
// dg_host_query.cpp for the ABI entry point and the rationale).  Header-only: the CPU test emulator instantiates the same
// context with 64 lanes and its counters -- it has no traversal of its own.
#pragma once
#include <cmath>
#include <cstring>
#include "dg_build.h"
#include "dg_kernels.h"
#include "dg_traverse.h"

namespace dg
{
namespace host
{
struct HostSqrt
{
	double operator()(double x) const { return std::sqrt(x); } // IEEE, correctly rounded
};

// the host arrays of a built mesh as the kernels' MeshDev (pointers into B: B must outlive the view)
inline void mesh_view(const MeshBuild& B, MeshDev& M)
{
	M.pairs = B.pairs.data();
	M.tri_pairs = B.tri_pairs.data();
	M.tri_approx = B.tri_approx.data();
	M.tris = B.tris.data();
	M.pn = B.pn.data();
	M.root_info = B.root_info;
	M.n_positions = (int32_t)B.tris.size();
	M.stack_levels = (int32_t)(B.depth + 1 < (uint32_t)kStackDepth ? B.depth + 1 : (uint32_t)kStackDepth);
	M.n_sub = (int32_t)B.sub_roots.size();
	for (int d = 0; d < 3; ++d)
		M.origin[d] = B.origin[d];
	M.mesh_l1 = B.mesh_l1;
	M.pad_ = 0.0f;
	for (size_t i = 0; i < (size_t)kSubtrees; ++i)
		M.sub_roots[i] = i < B.sub_roots.size() ? B.sub_roots[i] : B.root_info;
}

struct NoStats // the product's instantiation counts nothing
{
	void pair_step(const MeshDev&, int) {}
	void leaf(int, int) {}
	void leaf_pair() {}
	void tri_test(int, bool) {}
	void dead() {}
	void push() {}
	void pop() {}
	void stale_pop() {}
	void filter_pair() {}
	void filter_rest() {}
	void append(bool) {}
};

// N lanes carried by one host thread.  The shared stack is two arrays; STACK16 parks the bounds as the filtered kernel
// does (upper 16 bits of the float); the candidate lists of the filtered traversal live in `lists` with the kernel's
// layout (entry k of lane l at byte 256 k + 4 l, i.e. lists[64 k + l] for a full wave).
template <int N, class S = NoStats, bool STACK16 = false>
struct HostWave
{
	static constexpr int kLanes = N;
	struct Pair
	{
		const float* r;
		int info0, info1;
	};
	struct Approx
	{
		const float* r;
		int valid0, valid1;
	};
	S* stats = nullptr;
	int* lists = nullptr;
	int infos[kStackDepth];
	float bounds[kStackDepth][N];

	template <class F>
	void lanes(F f) const
	{
		for (int l = 0; l < N; ++l)
			f(l);
	}
	int uniform(int v) const { return v; }
	void pick(LaneVar<float, N>& out, const LaneVar<f2, N>& lb, bool first) const
	{
		for (int l = 0; l < N; ++l)
			out[l] = first ? lb[l].x : lb[l].y;
	}
	template <class P>
	unsigned long long ballot(P p) const
	{
		unsigned long long m = 0ull;
		for (int l = 0; l < N; ++l)
			if (p(l))
				m |= 1ull << l;
		return m;
	}
	Pair load_pair(const PairRec* base, int idx) const { return Pair{&base[idx].f[0][0], base[idx].info[0], base[idx].info[1]}; }
	Approx load_approx(const TriApproxPair* base, int idx) const { return Approx{&base[idx].f[0][0], base[idx].valid[0], base[idx].valid[1]}; }
	TriRegs load_tri(const TriPacket* tris, int t) const
	{
		const TriPacket& P = tris[t];
		return TriRegs{P.v0[0], P.v0[1], P.v0[2], P.e0[0], P.e0[1], P.e0[2], P.e1[0], P.e1[1], P.e1[2], P.a00, P.a01, P.a11, P.det, P.inv_det, P.denom};
	}
	static float truncate16(float v)
	{
		uint32_t bits;
		std::memcpy(&bits, &v, 4);
		bits &= 0xffff0000u;
		std::memcpy(&v, &bits, 4);
		return v;
	}
	void push(int sp, int info, const LaneVar<f2, N>& lb, bool second)
	{
		infos[sp] = info;
		for (int l = 0; l < N; ++l)
		{
			const float v = second ? lb[l].y : lb[l].x;
			bounds[sp][l] = STACK16 ? truncate16(v) : v;
		}
	}
	float parked(int sp, int l) const { return bounds[sp][l]; }
	int info(int sp) const { return infos[sp]; }
	uint32_t claim(uint32_t* counter) const { return __atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED); }
	void list_store(uint32_t slot, int v) const { lists[slot >> 2] = v; }
	void note_pair_step(const MeshDev& M, int cur) const { if (stats) stats->pair_step(M, cur); }
	void note_leaf(int first, int cnt) const { if (stats) stats->leaf(first, cnt); }
	void note_leaf_pair() const { if (stats) stats->leaf_pair(); }
	void note_tri_test(int interested, bool useful) const { if (stats) stats->tri_test(interested, useful); }
	void note_dead() const { if (stats) stats->dead(); }
	void note_push() const { if (stats) stats->push(); }
	void note_pop() const { if (stats) stats->pop(); }
	void note_stale_pop() const { if (stats) stats->stale_pop(); }
	void note_filter_pair() const { if (stats) stats->filter_pair(); }
	void note_filter_rest() const { if (stats) stats->filter_rest(); }
	void note_append(bool reset) const { if (stats) stats->append(reset); }
};

// One point = a wave of one lane walking the tree with the kernels' own traversal: the nearer child (by the distance to
// the box centre) first, the other postponed with its bound and re-tested against the by then tighter best when popped.
inline bool signed_distance_point(const MeshDev& M, double px, double py, double pz, LaneResult& out)
{
	LaneQuery q;
	init_query(M.origin, M.mesh_l1, true, px, py, pz, q);
	HostWave<1> w;
	auto lane_query = [&](int) -> LaneQuery& { return q; };
	ExactWalk<HostWave<1>, decltype(lane_query)> pol(lane_query);
	(void)packet_walk(w, pol, M, M.root_info, nullptr, 0u, 0);
	if (q.best_tri < 0)
		return false; // a NaN point: no comparison ever succeeds (the kernels write "no value" as well)
	out = finish_query(M.tris, M.pn, q, HostSqrt());
	return true;
}
} // namespace host
} // namespace dg
