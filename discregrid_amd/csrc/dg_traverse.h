// dg_traverse.h -- THE packet traversal of K1 (TriangleMeshDistance::_query, TriangleMeshDistance.h:514-562, re-designed for a
// wavefront): ONE template that
//   * the device kernels instantiate with a wave context of one lane per thread (ballots, v_readlane, LDS, scalar loads:
//     dg_kernels_k1.hip),
//   * the host point query instantiates with a wave of ONE lane (dg_host_query.h), and
//   * the CPU test emulator instantiates with 64 lanes carried as arrays,
// so that a change to the node format, the visiting order or a leaf test is made once, and the CPU tests run the code the
// device runs.  (The same pattern as k3c_lane<W> in dg_density_cells.h.)
//
// A wave context W provides
//   static constexpr int kLanes                     lanes one thread of this instantiation carries (device 1, emulator 64)
//   void lanes(F f)                                 f(l) for each of them
//   void pick(out, lb, bool first)                  out[l] = first ? lb[l].x : lb[l].y  (a wave-uniform flag the lanes select by goes
//                                                   in as an ARGUMENT of a context member: captured in a callable it takes a detour
//                                                   through a vector register on the device)
//   unsigned long long ballot(P p)                  bit l = p(l) over the lanes of the WAVE
//   Pair   load_pair(const PairRec*, int)           .r -> kPairFloats floats, .info0, .info1   (device: SGPRs)
//   Tri    load_tri(const TriPacket*, int)          the 15 doubles of a packet                  (device: SGPRs)
//   Approx load_approx(const TriApproxPair*, int)   .r -> kApproxFloats floats, .valid0, .valid1 (device: SGPRs)
//   void push(int sp, int info, lb, bool second)    stack entry sp: the info word, and per lane lb[l].y (second) or lb[l].x parked
//   float parked(int sp, int l); int info(int sp)   ... and back
//   int uniform(int v)                              v, known to be the same in every lane (device: v_readfirstlane)
//   uint32_t claim(uint32_t* counter)               atomic fetch-add 1 (heavy-brick slots)
//   void list_store(uint32_t slot, int v)           a word of the lanes' candidate lists (LDS on the device)
//   note_pair_step / note_leaf / note_*             counters of the emulator's design studies; empty on the device
// and a policy (below: ExactWalk, FastWalk) provides what differs between the exact and the filtered traversal: the bound
// of a child pair, the pruning test and the leaf.
//
// The walk.  One shared stack: the info word of a postponed subtree lives with the wave, each lane's own lower bound for
// that subtree is parked beside it, so a popped entry is re-tested against the by then tighter running bests with one
// compare -- no reload, no recomputation.  At an inner node ONE record gives the bounds of both children; a child is
// entered if ANY lane may still improve there, the child most lanes are closer to (by the distance to the box centre)
// first, the other one is pushed.  `start` is the info word of the subtree to search.  With `ovf_count` set the wave
// counts its work; when the count passes `budget0` it claims an overflow slot when the next subtree is finished and
// returns the slot number at once -- the caller parks the lanes' running bests there (dg_kernels.h, "Heavy bricks").
// If all slots are taken the wave carries on.  Returns -1 when the subtree was searched to the end.
#pragma once
#include "dg_geom.h"
#include "dg_kernels.h"

// Per-lane callables are always inlined.  Device lore learnt the hard way (same-box A/B, +3.7 %): a wave-uniform variable that a
// per-lane callable captures BY REFERENCE and that changes in a loop (a counter, a flag) is kept in a vector register by the
// device compiler -- loop counters become v_add + vcc, selects by a flag become v_cndmask 0/1 + v_cmp.  Such values go in by
// value (ints) or as arguments of a context member (flags: W::pick, W::push).
#define DG_LANE __attribute__((always_inline))

namespace dg
{
// a value per lane, for a thread that carries N lanes
template <class T, int N>
struct LaneVar
{
	T v[N];
	DG_HD T& operator[](int l) { return v[N == 1 ? 0 : l]; }
	DG_HD const T& operator[](int l) const { return v[N == 1 ? 0 : l]; }
};

struct TriRegs // a triangle packet's numbers by value (the device holds them in scalar registers)
{
	double v0x, v0y, v0z, e0x, e0y, e0z, e1x, e1y, e1z, a00, a01, a11, det, inv_det, denom;
};

template <class W, class P>
DG_HD int packet_walk(W& w, P& pol, const MeshDev& M, int start, uint32_t* ovf_count, uint32_t ovf_slots, int budget0)
{
	int sp = 0; // wave-uniform
	int cur = start;
	LaneVar<float, W::kLanes> lbcur; // every lane's lower bound for `cur`
	w.lanes([&](int l) DG_LANE { lbcur[l] = 0.0f; });
	int work = 0; // wave-uniform
	int budget = ovf_count ? budget0 : 0x7fffffff;
	int parked = -1;
	// (Shape of the loop: an inner loop for the way down and the budget test after a pop keep every wave-uniform
	// variable defined on every path -- with one loop and a `continue` the device compiler carries undefined values
	// for cur/sp/work across the leaf branch and materialises them with VALU moves on every step.)
	while (true)
	{
		// down the tree while some lane needs a child; `dead`: the node's children are out of every lane's reach
		bool dead = false;
		while (cur >= 0)
		{
			++work;
			w.note_pair_step(M, cur);
			const typename W::Pair pr = w.load_pair(M.pairs, cur);
			LaneVar<f2, W::kLanes> lb, cd;
			w.lanes([&](int l) DG_LANE { lb[l] = pol.bounds(l, pr.r, &cd[l]); });
			const unsigned long long bl = w.ballot([&](int l) DG_LANE { return pol.reach(l, lb[l].x); });
			const unsigned long long br = w.ballot([&](int l) DG_LANE { return pol.reach(l, lb[l].y); });
			if ((bl | br) == 0ull)
			{
				w.note_dead();
				dead = true;
				break;
			}
			bool left = bl != 0ull;
			if (bl != 0ull && br != 0ull)
			{
				// both children are needed: the one most lanes are closer to -- by the distance to the box CENTRE --
				// first, the other is postponed (its info word and every lane's bound for it go on the stack)
				const unsigned long long pref = w.ballot([&](int l) DG_LANE { return cd[l].x <= cd[l].y; }) & (bl | br);
				left = 2 * __builtin_popcountll(pref) >= __builtin_popcountll(bl | br);
				if (sp < M.stack_levels) // always true: one push per tree level at most
				{
					// (the flag goes in as an argument: captured in a callable it takes a detour through a vector register on the device)
					w.push(sp, left ? pr.info1 : pr.info0, lb, left);
					w.note_push();
					++sp;
				}
			}
			cur = left ? pr.info0 : pr.info1;
			w.pick(lbcur, lb, left);
		}
		if (!dead)
		{
			const unsigned code = ~(unsigned)cur;
			work += pol.leaf(w, M, (int)(code >> kLeafBits), (int)(code & (unsigned)(kMaxLeaf - 1)) + 1, lbcur);
		}
		// pop the next postponed subtree that some lane still needs
		bool found = false;
		while (sp > 0)
		{
			--sp;
			w.lanes([&](int l) DG_LANE { lbcur[l] = w.parked(sp, l); });
			w.note_pop();
			if (w.ballot([&](int l) DG_LANE { return pol.reach(l, lbcur[l]); }) != 0ull)
			{
				cur = w.info(sp);
				found = true;
				break;
			}
			w.note_stale_pop();
		}
		if (!found)
			break;
		// the work budget is looked at when a subtree is finished, not on every step (the traversal is the same either
		// way, only the moment a brick is declared heavy moves by a few steps)
		if (work > budget)
		{
			const int slot = (int)w.claim(ovf_count);
			if ((unsigned)slot < ovf_slots)
			{
				parked = slot;
				break;
			}
			budget = 0x7fffffff;
		}
	}
	return parked;
}

// ---- the exact traversal: double-precision test on every triangle some lane may need ------------------------------------
// Q: callable, Q(l) -> LaneQuery& of lane l.  On return every active lane holds the minimum squared distance over all
// triangles of the subtree (best_d2) and the position attaining it.
template <class W, class Q>
struct ExactWalk
{
	Q q;
	DG_HD explicit ExactWalk(Q q_) : q(q_) {}
	DG_HD f2 bounds(int l, const float* r, f2* cd) const { return pair_lb2(r, q(l).fp, cd); }
	DG_HD bool reach(int l, float lb) const { return lb < q(l).bestf; }
	// Leaf: `cnt` triangle positions (even, <= 16) starting at the even position `first` (wave-uniform), handled pair by
	// pair.  A triangle gets the full double-precision test only if some lane's float lower bound -- the larger of the
	// leaf's bound and the triangle's own box bound -- is below that lane's running best.  Returns 1 + the tests made.
	DG_HD int leaf(W& w, const MeshDev& M, int first, int cnt, const LaneVar<float, W::kLanes>& leaf_lb2) const
	{
		int tests = 0; // wave-uniform
		w.note_leaf(first, cnt);
		for (int g = 0; g < cnt; g += 2)
		{
			const typename W::Pair pr = w.load_pair(M.tri_pairs, (first + g) >> 1);
			LaneVar<f2, W::kLanes> lb;
			w.lanes([&](int l) DG_LANE { lb[l] = pair_lb2(pr.r, q(l).fp); });
			const unsigned long long m0 = w.ballot([&](int l) DG_LANE { return fmax2(lb[l].x, leaf_lb2[l]) < q(l).bestf; });
			const unsigned long long m1 = w.ballot([&](int l) DG_LANE { return fmax2(lb[l].y, leaf_lb2[l]) < q(l).bestf; });
			w.note_leaf_pair();
#pragma unroll
			for (int side = 0; side < 2; ++side)
			{
				const unsigned long long m = side == 0 ? m0 : m1;
				if (m == 0ull)
					continue;
				++tests;
				const int t = first + g + side;
				const TriRegs T = w.load_tri(M.tris, t);
				bool useful = false; // (emulator statistics; dead code on the device)
				w.lanes([&](int l) DG_LANE {
					const Hit h = tri_closest<false>(T.v0x, T.v0y, T.v0z, T.e0x, T.e0y, T.e0z, T.e1x, T.e1y, T.e1z, T.a00, T.a01, T.a11, T.det,
													 T.inv_det, T.denom, q(l).px, q(l).py, q(l).pz);
					useful = useful || h.d2 < q(l).best_d2;
					offer(q(l), h.d2, t);
				});
				w.note_tri_test(__builtin_popcountll(m), useful);
			}
		}
		return 1 + tests;
	}
};

// ---- the filtered traversal -------------------------------------------------------------------------------------------------
// The same walk, but a visited leaf's triangles go through the FLOAT filter (dg_geom.h: tri_approx_frame / tri_approx_rest,
// two triangles per record with packed math) instead of a bound test plus the double test.  Every lane keeps an upper bound U
// of its minimum d^2 (what the walk prunes with) and the list of triangles whose interval [q - err, q + err] reaches below U:
// the only ones that can attain the lane's minimum.  After the walk each lane runs the double test on ITS OWN candidates
// (typically 1-2, six around a mesh vertex).  Bit-exactness: the triangle with the smallest double d^2 is always among the
// lane's candidates (error analysis in dg_geom.h), and the winner among the candidates is found with the double test in list
// order (strict <).
struct FastLane
{
	ApproxLane a;
	float U;      // upper bound of the lane's minimum d^2; -inf: lane inactive
	float Uprune; // what bound tests compare with: U (1 + theta) + kappa
	float Lmin;   // smallest lower value among the listed candidates
	uint32_t slot; // address (context-defined: LDS bytes on the device) of the lane's next list entry: base + 256 * listed
	               // candidates, capped at base + 256 * kFastListCap (a list that reaches the cap may have overflowed)
};
DG_HD void init_fast_lane(FastLane& f, bool serve, uint32_t list_base)
{
	f.U = serve ? __builtin_inff() : -__builtin_inff();
	f.Uprune = f.U;
	f.Lmin = __builtin_inff();
	f.slot = list_base;
}
#ifndef DG_TRI_PREFILTER
#define DG_TRI_PREFILTER 1 // 0: A/B variant without the early-out of step 1 (tests/perf)
#endif
// F: callable, F(l) -> FastLane& of lane l; B: callable, B(l) -> address of lane l's list entry 0
template <class W, class F, class B>
struct FastWalk
{
	F f;
	B list_base;
	bool degenerate = false; // wave-uniform: a degenerate triangle was met (nobody's list is complete)
	DG_HD FastWalk(F f_, B b_) : f(f_), list_base(b_) {}
	DG_HD f2 bounds(int l, const float* r, f2* cd) const { return pair_lb2_fast(r, f(l).a.x, cd); }
	DG_HD bool reach(int l, float lb) const { return lb <= f(l).Uprune; }
	// returns the work of the leaf: 1 + the triangle pairs looked at
	DG_HD int leaf(W& w, const MeshDev& M, int first, int cnt, const LaneVar<float, W::kLanes>& lbcur)
	{
		int work = 1;
		cnt = w.uniform(cnt); // (the loop below is a scalar loop: tell the device compiler so)
		w.note_leaf(first, cnt);
		// error terms for this leaf's triangles around the lane's current distance estimate (its upper bound, or the
		// leaf's own bound while no triangle has been seen); they are valid for any estimate
		LaneVar<float, W::kLanes> theta, kappa;
		w.lanes([&](int l) DG_LANE { approx_err_terms(f(l).a.E, f(l).U < __builtin_inff() ? f(l).U : lbcur[l], &theta[l], &kappa[l]); });
		for (int g = 0; g < cnt; g += 2)
		{
			const typename W::Approx rec = w.load_approx(M.tri_approx, (first + g) >> 1);
			const int valid0 = rec.valid0, valid1 = rec.valid1; // 1: triangle, 0: padding slot of an odd leaf, 2: degenerate triangle
			degenerate = degenerate || valid0 == 2 || valid1 == 2;
			// step 1: frame coordinates + rectangle bound; most pairs of a visited leaf end here
			LaneVar<TriFrame, W::kLanes> fr;
			LaneVar<f2, W::kLanes> lo_lb;
			w.lanes([&](int l) DG_LANE {
				const f2 qlb = tri_approx_frame(rec.r, f(l).a, &fr[l]);
				lo_lb[l] = qlb - f2_fma(qlb, f2_splat(theta[l]), f2_splat(kappa[l]));
			});
			++work;
			w.note_filter_pair();
			// (two ballots combined with the wave-uniform validity on the scalar side: written as ONE per-lane predicate the
			// uniform flags take a detour through vector registers on the device)
			const unsigned long long reach0 = w.ballot([&](int l) DG_LANE { return lo_lb[l].x <= f(l).U; });
			const unsigned long long reach1 = w.ballot([&](int l) DG_LANE { return lo_lb[l].y <= f(l).U; });
			if (DG_TRI_PREFILTER && ((valid0 == 1 ? reach0 : 0ull) | (valid1 == 1 ? reach1 : 0ull)) == 0ull)
				continue;
			w.note_filter_rest();
			const int position = first + g; // (loop-varying wave-uniform scalars go into the per-lane code BY VALUE: see DG_LANE)
			w.lanes([&, valid0, valid1, position](int l) DG_LANE {
				FastLane& fl = f(l);
				const f2 q = tri_approx_rest(rec.r, fl.a, fr[l]);
				const f2 err = f2_fma(q, f2_splat(theta[l]), f2_splat(kappa[l]));
				const f2 up = q + err, lo = q - err;
				const uint32_t base = list_base(l), limit = base + 256u * (uint32_t)kFastListCap;
#pragma unroll
				for (int side = 0; side < 2; ++side)
				{
					if ((side == 0 ? valid0 : valid1) != 1) // wave-uniform
						continue;
					const float lo_s = side == 0 ? lo.x : lo.y, up_s = side == 0 ? up.x : up.y;
					if (lo_s <= fl.U)
					{
						// a candidate; if even its upper value is below every listed lower value, the list is obsolete
						const bool reset = up_s < fl.Lmin;
						w.note_append(reset && fl.slot != base);
						fl.slot = reset ? base : fl.slot;
						w.list_store(fl.slot, position + side);
						fl.slot = fl.slot + 256u < limit ? fl.slot + 256u : limit;
						fl.Lmin = fmin_sel(reset ? __builtin_inff() : fl.Lmin, lo_s);
					}
					fl.U = fmin_sel(fl.U, up_s);
				}
			});
		}
		// the threshold the bound tests compare with (dg_geom.h: approx_err_terms)
		w.lanes([&](int l) DG_LANE { f(l).Uprune = __builtin_fmaf(f(l).U, 1.0f + theta[l], kappa[l]); });
		return work;
	}
};

} // namespace dg
