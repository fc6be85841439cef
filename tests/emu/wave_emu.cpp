// tests/emu/wave_emu.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host-side lock-step emulation of the K1 / K1p / K2 / unpack kernels of
// discregrid_amd/csrc/dg_kernels_k*.hip.  It runs the PRODUCT's own data structures (dg_build:
// flattened BVH, triangle packets, pseudonormals), the product's own lattice decomposition
// (dg_layout.h, map_lane) and the product's own per-lane arithmetic (dg_geom.h) -- only the
// wave-level control flow (ballots over 64 lanes) is re-expressed as loops over lane arrays.
// It exists because the build container has no GPU: the algorithmic logic (does the packet
// traversal with conservative float boxes find the reference's distances bit for bit? does
// the brick decomposition cover every node exactly once?) is checked here on CPU before GPU
// minutes are spent.  The product never links this file and has no CPU path.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../discregrid_amd/csrc/dg_build.h"
#include "../../discregrid_amd/csrc/dg_kernels.h"
#include "../../discregrid_amd/csrc/dg_layout.h"
#include "../../discregrid_amd/csrc/dg_host_query.h"



using namespace dg;

namespace
{

struct Stats
{
	uint64_t bricks = 0, node_visits = 0, leaf_visits = 0, tri_tests = 0, descent_nodes = 0, slab_tests = 0, lane_interest = 0, useful_tests = 0, leaf_groups = 0, pops = 0, stale_pops = 0, heavy_bricks = 0;
};

struct HostSqrt
{
	double operator()(double x) const { return std::sqrt(x); }
};

struct Wave
{
	LaneQuery q[64];
};

// ---- the traversals: the PRODUCT's template (dg_traverse.h: packet_walk + ExactWalk / FastWalk) instantiated with the host
// wave context of dg_host_query.h carrying 64 lanes; what is written here is only where the counters of the design studies go
struct FastStats;
struct EmuCounters
{
	Stats* st = nullptr;     // the exact traversal counts into this ...
	FastStats* fs = nullptr; // ... the filtered one into this
	void pair_step(const MeshDev& M, int cur);
	void leaf(int first, int cnt);
	void leaf_pair();
	void tri_test(int interested, bool useful);
	void dead();
	void push();
	void pop();
	void stale_pop();
	void filter_pair();
	void filter_rest();
	void append(bool reset);
	// design study "a bound per leaf half" (round 6): which pairs of the leaf being visited went on to step 2
	int leaf_pairs = 0;
	unsigned leaf_mask = 0;
	void flush_leaf();
};
typedef dg::host::HostWave<64, EmuCounters, false> EmuWave;   // bounds parked as floats (k_sample_nodes, k_heavy_subtrees)
typedef dg::host::HostWave<64, EmuCounters, true> EmuWave16;  // ... in 16 bits (k_sample_fast: both of its traversals)

// the exact traversal of subtree `start` by the 64 lanes of w (k_sample_nodes / k_heavy_subtrees; stack16: as the second pass
// of k_sample_fast runs it, bounds parked in 16 bits); returns the heavy slot claimed or -1
template <class EW>
int walk_exact_with(const MeshDev& M, Wave& w, Stats& st, int start, const OverflowBuf* ovf)
{
	EmuCounters c;
	c.st = &st;
	EW ew;
	ew.stats = &c;
	auto lane_query = [&](int l) -> LaneQuery& { return w.q[l]; };
	ExactWalk<EW, decltype(lane_query)> pol(lane_query);
	return packet_walk(ew, pol, M, start, ovf ? ovf->count : nullptr, ovf ? ovf->slots : 0u, ovf ? ovf->heavy_work : 0);
}
int walk_exact(const MeshDev& M, Wave& w, Stats& st, int start, const OverflowBuf* ovf, bool stack16 = false)
{
	return stack16 ? walk_exact_with<EmuWave16>(M, w, st, start, ovf) : walk_exact_with<EmuWave>(M, w, st, start, ovf);
}


// ---- filtered kernel: the wave-level bookkeeping of k_sample_fast around the product's traversal ----------------------
static unsigned long long g_need_sum = 0, g_need_max = 0;
extern "C" void emu_need(unsigned long long* o) { o[0] = g_need_sum; o[1] = g_need_max; }
struct FastStats
{
	uint64_t bricks = 0, pair_steps = 0, leaf_visits = 0, tri_pairs = 0, appends = 0, resets = 0, redo_bricks = 0,
			 sum_max_list = 0, sum_list = 0, lanes = 0, parked = 0, pooled_waves = 0, unpooled_waves = 0, hist[17] = {0};
	void add(const FastStats& o)
	{
		bricks += o.bricks; pair_steps += o.pair_steps; leaf_visits += o.leaf_visits; tri_pairs += o.tri_pairs;
		appends += o.appends; resets += o.resets; redo_bricks += o.redo_bricks; sum_max_list += o.sum_max_list;
		sum_list += o.sum_list; lanes += o.lanes; parked += o.parked; pooled_waves += o.pooled_waves; unpooled_waves += o.unpooled_waves;
		for (int k = 0; k < 17; ++k) hist[k] += o.hist[k];
	}
};
int g_fast = 1;     // 0: emulate a launch without the filtered kernel (DG_FORCE=k1_fast=0)
uint32_t g_pool_cap = 0x7fffffffu; // DG_FORCE=pool_cap: pairs the epilogue's pool holds at most (tests: below its LDS capacity)
int g_seed_study = 0; // 1: the filtered traversal starts from an oracle-tight upper bound per lane (design study)
int g_brick_blocking = 0; // 1: the blocked brick order K3 launches use (dg_kernels.h: map_lane)
FastStats g_fs;

// the candidate lists of one wave with the kernel's LDS layout: entry k of lane l at byte 256 k + 4 l (dg::FastLane::slot)
struct FastLists
{
	int v[(kFastListCap + 1) * 64];
	static uint32_t base(int l) { return 4u * (uint32_t)l; }
	static int count(const FastLane& f, int l) { return (int)((f.slot - base(l)) >> 8); }
	int at(int l, int k) const { return v[k * 64 + l]; }
};
// design study (EMU_DEPTH_HIST=1): pair steps of the filtered traversal by depth of the node in the tree
int g_depth_hist_on = getenv("EMU_DEPTH_HIST") != nullptr;
static std::vector<int> g_pair_depth;
static uint64_t g_depth_hist[64];
static void depth_hist_note(const MeshDev& M, int cur)
{
	static std::mutex mu;
	std::lock_guard<std::mutex> lock(mu);
	if (g_pair_depth.empty())
	{
		g_pair_depth.assign((size_t)M.n_positions + 8, -1); // (more than enough: pair records < positions)
		std::vector<std::pair<int, int>> open(1, std::make_pair(M.root_info, 0));
		while (!open.empty())
		{
			const std::pair<int, int> t = open.back();
			open.pop_back();
			if (t.first < 0)
				continue;
			if ((size_t)t.first >= g_pair_depth.size())
				g_pair_depth.resize((size_t)t.first + 1, -1);
			g_pair_depth[(size_t)t.first] = t.second;
			open.push_back(std::make_pair(M.pairs[t.first].info[0], t.second + 1));
			open.push_back(std::make_pair(M.pairs[t.first].info[1], t.second + 1));
		}
	}
	const int d = g_pair_depth[(size_t)cur];
	g_depth_hist[d < 0 ? 63 : (d > 62 ? 62 : d)]++;
}
extern "C" void emu_depth_hist(uint64_t* out /*64*/) { for (int i = 0; i < 64; ++i) out[i] = g_depth_hist[i]; }
thread_local std::vector<std::pair<int,int>> g_leaf_log; // (first, cnt) of the leaves a traversal visited (design studies)
// the filtered traversal (k_sample_fast's first pass); returns -1, -2 (a degenerate triangle was met) or the heavy slot claimed
template <class EW>
int walk_fast_with(const MeshDev& M, FastLane* fl, FastLists& lists, FastStats& st, const OverflowBuf* ovf, int start)
{
	g_leaf_log.clear();
	EmuCounters c;
	c.fs = &st;
	EW ew;
	ew.stats = &c;
	ew.lists = lists.v;
	auto lane_state = [&](int l) -> FastLane& { return fl[l]; };
	auto lane_list = [](int l) { return FastLists::base(l); };
	FastWalk<EW, decltype(lane_state), decltype(lane_list)> pol(lane_state, lane_list);
	const bool budgeted = ovf && ovf->count;
	const int parked = packet_walk(ew, pol, M, start, budgeted ? ovf->count : nullptr, budgeted ? ovf->slots : 0u,
								   budgeted ? kFastWorkFactor * ovf->heavy_work : 0);
	c.flush_leaf();
	if (parked >= 0)
		return parked;
	return pol.degenerate ? -2 : -1;
}
int walk_fast(const MeshDev& M, FastLane* fl, FastLists& lists, FastStats& st, const OverflowBuf* ovf, int start)
{
	static const bool stack32 = getenv("EMU_STACK32") != nullptr; // (design study: bounds parked as full floats)
	return stack32 ? walk_fast_with<EmuWave>(M, fl, lists, st, ovf, start) : walk_fast_with<EmuWave16>(M, fl, lists, st, ovf, start);
}
int walk_fast(const MeshDev& M, FastLane* fl, FastLists& lists, FastStats& st, const OverflowBuf* ovf)
{
	return walk_fast(M, fl, lists, st, ovf, M.root_info);
}
// design study (emu_set_heavy_study): what the jobs of k_heavy_subtrees cost as they are (exact walk) and what they would cost with
// the filtered walk seeded by the parked upper bounds (exact walk only for the jobs in which some lane's list fills up)
int g_heavy_study = 0;
uint64_t g_hs_hist[8], g_hs_brick_max[4];
uint64_t g_hs[16]; // 0 jobs, 1 exact: pair steps, 2 leaf groups, 3 tri tests; 4 fast: pair steps, 5 leaf visits, 6 step-1 pairs, 7 step-2 pairs,
                   // 8 candidates, 9 epilogue rounds (pooled: ceil(candidates / 64)), 10 jobs that fall back, 11..13 exact counts of those jobs,
                   // 14 jobs whose subtree is pruned at its root pair (exact walk: one pair step)
void EmuCounters::pair_step(const MeshDev& M, int cur)
{
	if (st) st->node_visits += 2;
	if (fs)
	{
		fs->pair_steps++;
		if (g_depth_hist_on)
			depth_hist_note(M, cur);
	}
}
// Leaf-half study: [0] leaves, [1] pairs at step 1, [2] pairs at step 2, [3] leaves none of whose pairs reached step 2, [4] their
// pairs, [5] groups of two consecutive pairs (4 triangles: (0,1), (2,3) ...), [6] groups neither of whose pairs reached step 2,
// [7] leaves by number of pairs 1..8 -> [8..15]
static uint64_t g_half[16];
extern "C" void emu_leaf_half_stats(uint64_t* out, int reset)
{
	for (int i = 0; i < 16; ++i)
	{
		out[i] = g_half[i];
		if (reset) g_half[i] = 0;
	}
}
void EmuCounters::flush_leaf()
{
	if (!fs || leaf_pairs == 0)
		return;
	uint64_t add[16] = {0};
	add[0] = 1;
	add[1] = (uint64_t)leaf_pairs;
	add[2] = (uint64_t)__builtin_popcount(leaf_mask);
	if (leaf_mask == 0u)
	{
		add[3] = 1;
		add[4] = (uint64_t)leaf_pairs;
	}
	for (int g = 0; g + 1 < leaf_pairs; g += 2)
	{
		add[5]++;
		add[6] += ((leaf_mask >> g) & 3u) == 0u;
	}
	add[7 + (leaf_pairs < 8 ? leaf_pairs : 8)]++;
	for (int i = 0; i < 16; ++i)
		if (add[i])
			__atomic_fetch_add(&g_half[i], add[i], __ATOMIC_RELAXED);
	leaf_pairs = 0;
	leaf_mask = 0;
}
void EmuCounters::leaf(int first, int cnt)
{
	flush_leaf();
	if (st) st->leaf_visits++;
	if (fs)
	{
		fs->leaf_visits++;
		g_leaf_log.push_back(std::make_pair(first, cnt));
	}
}
void EmuCounters::leaf_pair()
{
	if (st)
	{
		st->leaf_groups++;
		st->slab_tests += 2;
	}
}
void EmuCounters::tri_test(int interested, bool useful)
{
	if (st)
	{
		st->tri_tests++;
		st->lane_interest += (uint64_t)interested;
		st->useful_tests += useful;
	}
}
static uint64_t g_fast_events[4]; // filtered traversal: dead steps, pushes, pops, stale pops (design studies)
extern "C" void emu_fast_events(uint64_t* out, int reset)
{
	for (int i = 0; i < 4; ++i)
	{
		out[i] = g_fast_events[i];
		if (reset) g_fast_events[i] = 0;
	}
}
void EmuCounters::dead() { if (fs) __atomic_fetch_add(&g_fast_events[0], 1, __ATOMIC_RELAXED); }
void EmuCounters::push() { if (fs) __atomic_fetch_add(&g_fast_events[1], 1, __ATOMIC_RELAXED); }
void EmuCounters::pop()
{
	if (st) st->pops++;
	if (fs) __atomic_fetch_add(&g_fast_events[2], 1, __ATOMIC_RELAXED);
}
void EmuCounters::stale_pop()
{
	if (st) st->stale_pops++;
	if (fs) __atomic_fetch_add(&g_fast_events[3], 1, __ATOMIC_RELAXED);
}
void EmuCounters::filter_pair() // (pairs that reached step 1)
{
	if (fs)
	{
		fs->hist[16] += 1;
		++leaf_pairs;
	}
}
void EmuCounters::filter_rest()
{
	if (fs)
	{
		fs->tri_pairs++;
		leaf_mask |= 1u << (leaf_pairs - 1);
	}
}
void EmuCounters::append(bool reset)
{
	if (fs)
	{
		fs->appends++;
		fs->resets += reset;
	}
}


// k_sample_fast for one wave without heavy-brick parking (K1p emulation): w.q[l] end up with the lanes'
// winners exactly as the kernel's epilogue leaves them
void fast_wave_unparked(const MeshDev& M, Wave& w, const bool* sample, FastStats& fs, Stats& ls)
{
	FastLane fl[64];
	FastLists lists;
	bool exact[64];
	bool any_fast = false, any_exact = false;
	for (int l = 0; l < 64; ++l)
	{
		fl[l].a = make_approx_lane(w.q[l].px - M.origin[0], w.q[l].py - M.origin[1], w.q[l].pz - M.origin[2], M.mesh_l1);
		exact[l] = sample[l] && !(fl[l].a.E < __builtin_inff());
		init_fast_lane(fl[l], sample[l] && !exact[l], FastLists::base(l));
		any_fast = any_fast || (sample[l] && !exact[l]);
	}
	fs.bricks++;
	if (any_fast)
	{
		const int r = walk_fast(M, fl, lists, fs, nullptr);
		for (int l = 0; l < 64; ++l)
			exact[l] = exact[l] || (sample[l] && (r == -2 || FastLists::count(fl[l], l) >= kFastListCap));
	}
	for (int l = 0; l < 64; ++l)
		any_exact = any_exact || exact[l];
	Wave wx;
	if (any_exact)
	{
		fs.redo_bricks++;
		for (int l = 0; l < 64; ++l)
		{
			init_query(M.origin, M.mesh_l1, exact[l], w.q[l].px, w.q[l].py, w.q[l].pz, wx.q[l]);
			if (exact[l] && fl[l].U > 0.0f)
				wx.q[l].bestf = best_as_float((double)fl[l].U);
		}
		walk_exact(M, wx, ls, M.root_info, nullptr, true);
	}
	for (int l = 0; l < 64; ++l)
	{
		if (!sample[l])
			continue;
		if (exact[l])
		{
			w.q[l].best_d2 = wx.q[l].best_d2;
			w.q[l].best_tri = wx.q[l].best_tri;
			continue;
		}
		for (int k = 0; k < FastLists::count(fl[l], l); ++k)
		{
			const int t = lists.at(l, k);
			const Hit h = tri_closest<false>(M.tris[t], w.q[l].px, w.q[l].py, w.q[l].pz);
			offer(w.q[l], h.d2, t);
		}
	}
}

// heavy-brick settings of the emulated launches (defaults = the product's)
const uint32_t kAutoSlots = 0xffffffffu; // slots chosen per launch as dg_capi.cpp does
uint32_t g_heavy_slots = kAutoSlots;
int g_heavy_work = 0; // 0: the product's rule (heavy_work_for)

struct HostMesh
{
	MeshBuild B;
	MeshDev dev;
};

} // namespace

extern "C"
{

void* emu_mesh_create(const double* verts, size_t nv, const uint32_t* tris, size_t nt, int max_leaf)
{
	auto m = new HostMesh;
	if (!build_mesh(verts, nv, tris, nt, max_leaf, m->B))
	{
		delete m;
		return nullptr;
	}
	dg::host::mesh_view(m->B, m->dev);
	return m;
}
void emu_mesh_free(void* h) { delete static_cast<HostMesh*>(h); }
// slots = 0: no splitting; work: budget of a brick (node steps + exact tests)
void emu_set_heavy(uint32_t slots, int work)
{
	g_heavy_slots = slots == kAutoSlots ? kAutoSlots : (slots < (uint32_t)kOverflowSlots ? slots : (uint32_t)kOverflowSlots);
	g_heavy_work = work;
}
void emu_set_fast(int on)
{
	g_fast = on;
	g_fs = FastStats();
}
void emu_set_pool_cap(uint32_t cap) { g_pool_cap = cap; }
void emu_pool_stats(uint64_t* out /*2*/)
{
	out[0] = g_fs.pooled_waves;
	out[1] = g_fs.unpooled_waves;
}
void emu_fast_stats(uint64_t* out /*28*/)
{
	out[0] = g_fs.bricks; out[1] = g_fs.pair_steps; out[2] = g_fs.leaf_visits; out[3] = g_fs.tri_pairs;
	out[4] = g_fs.appends; out[5] = g_fs.resets; out[6] = g_fs.redo_bricks; out[7] = g_fs.sum_max_list;
	out[8] = g_fs.sum_list; out[9] = g_fs.lanes; out[10] = g_fs.parked;
	for (int k = 0; k < 17; ++k) out[11 + k] = g_fs.hist[k];
}
void emu_set_brick_blocking(int on) { g_brick_blocking = on; }
void emu_set_seed_study(int on) { g_seed_study = on; }
void emu_set_heavy_study(int on)
{
	g_heavy_study = on;
	for (uint64_t& v : g_hs) v = 0;
}
void emu_heavy_study(uint64_t* out /*16*/) { for (int i = 0; i < 16; ++i) out[i] = g_hs[i]; }
void emu_heavy_study_hist(uint64_t* out /*8*/) { for (int i = 0; i < 8; ++i) { out[i] = g_hs_hist[i]; g_hs_hist[i] = 0; } }
// udiv_by / udiv_magic of dg_kernels.h against the host's division; returns the number of mismatches
uint64_t emu_udiv_check(const uint32_t* n, const uint32_t* d, uint64_t count)
{
	uint64_t bad = 0;
	for (uint64_t i = 0; i < count; ++i)
	{
		uint32_t r;
		const uint32_t q = udiv_by(n[i], d[i], udiv_magic(d[i]), &r);
		bad += (q != n[i] / d[i]) || (r != n[i] % d[i]);
	}
	return bad;
}
int emu_n_subtrees(void* h) { return static_cast<HostMesh*>(h)->dev.n_sub; }
// number of triangles reachable from the subtree roots; -1 if a triangle is reachable twice
long long emu_subtree_triangles(void* h)
{
	auto m = static_cast<HostMesh*>(h);
	std::vector<uint8_t> seen(m->B.tris.size(), 0);
	long long n = 0;
	std::vector<int32_t> todo(m->B.sub_roots.begin(), m->B.sub_roots.end());
	while (!todo.empty())
	{
		const int32_t info = todo.back();
		todo.pop_back();
		if (info >= 0)
		{
			todo.push_back(m->B.pairs[(size_t)info].info[0]);
			todo.push_back(m->B.pairs[(size_t)info].info[1]);
			continue;
		}
		const unsigned code = ~(unsigned)info;
		const int first = (int)(code >> kLeafBits), cnt = (int)(code & (unsigned)(kMaxLeaf - 1)) + 1;
		for (int t = first; t < first + cnt; ++t)
		{
			if (seen[(size_t)t])
				return -1;
			seen[(size_t)t] = 1;
			n += m->B.tris[(size_t)t].tri_id >= 0;
		}
	}
	return n;
}
void emu_mesh_info(void* h, uint64_t* n_nodes, uint32_t* depth, uint32_t* flags)
{
	auto m = static_cast<HostMesh*>(h);
	*n_nodes = 2 * m->B.pairs.size() + 1;
	*depth = m->B.depth;
	*flags = m->B.not_watertight;
}
// pseudonormals in the CALLER's triangle order: pn[t][slot][3] (slot 7 unused)
void emu_mesh_pseudonormals(void* h, double* pn)
{
	auto m = static_cast<HostMesh*>(h);
	for (size_t k = 0; k < m->B.tris.size(); ++k)
		if (m->B.tris[k].tri_id >= 0)
			std::memcpy(pn + (size_t)m->B.tris[k].tri_id * kPnSlots * 3, m->B.pn.data() + k * kPnSlots * 3,
						kPnSlots * 3 * sizeof(double));
}
// structural self-check of the pair BVH: every triangle in exactly one leaf slot, every bound of
// every ancestor contains the triangle (float rounding included), padding slots unreachable
static int check_subtree(const HostMesh* m, int32_t info, const double* verts, const uint32_t* tris,
						 std::vector<int>& seen, std::vector<const float*>& anc /* ancestor bounds: rec ptr + side */,
						 std::vector<int>& anc_side, int depth)
{
	if (depth > kStackDepth)
		return 9;
	if (info < 0)
	{
		const unsigned code = ~(unsigned)info;
		const unsigned first = code >> kLeafBits, cnt = (code & (unsigned)(kMaxLeaf - 1)) + 1;
		if ((first & 1u) || (cnt & 1u) || first + cnt > m->B.tris.size())
			return 5;
		for (unsigned t = first; t < first + cnt; ++t)
		{
			const int id = m->B.tris[t].tri_id;
			const PairRec& tp = m->B.tri_pairs[t >> 1];
			if (id < 0)
			{
				if (!(tp.f[12][t & 1] < 0.0f)) // padding must have empty slabs
					return 10;
				continue;
			}
			seen[id]++;
			std::vector<const float*> all = anc;
			std::vector<int> sides = anc_side;
			all.push_back(&tp.f[0][0]);
			sides.push_back((int)(t & 1));
			for (size_t a = 0; a < all.size(); ++a)
			{
				const float* r = all[a];
				const int sd = sides[a];
				// every vertex inside the three slabs; Gram matrix of the directions: no eigenvalue above 1
				double U[3][3];
				for (int x = 0; x < 3; ++x)
					for (int d = 0; d < 3; ++d)
						U[x][d] = (double)r[2 * (3 + 3 * x + d) + sd];
				for (int x = 0; x < 3; ++x)
				{
					double row = 0;
					for (int y = 0; y < 3; ++y)
						row += std::fabs(U[x][0] * U[y][0] + U[x][1] * U[y][1] + U[x][2] * U[y][2]);
					if (row > 1.0)
						return 11;
				}
				for (int k = 0; k < 3; ++k)
					for (int x = 0; x < 3; ++x)
					{
						double pr = 0;
						for (int d = 0; d < 3; ++d)
							pr += U[x][d] * (verts[3 * tris[3 * id + k] + d] - m->B.origin[d] - (double)r[2 * d + sd]);
						if (!(std::fabs(pr) <= (double)r[2 * (12 + x) + sd]))
							return 12;
					}
			}
		}
		return 0;
	}
	if ((size_t)info >= m->B.pairs.size())
		return 2;
	const PairRec& pr = m->B.pairs[info];
	for (int sd = 0; sd < 2; ++sd)
	{
		anc.push_back(&pr.f[0][0]);
		anc_side.push_back(sd);
		const int e = check_subtree(m, pr.info[sd], verts, tris, seen, anc, anc_side, depth + 1);
		anc.pop_back();
		anc_side.pop_back();
		if (e)
			return e;
	}
	return 0;
}
int emu_mesh_check(void* h, const double* verts, const uint32_t* tris)
{
	auto m = static_cast<HostMesh*>(h);
	std::vector<int> seen(m->B.n_triangles, 0);
	std::vector<const float*> anc;
	std::vector<int> anc_side;
	const int e = check_subtree(m, m->B.root_info, verts, tris, seen, anc, anc_side, 0);
	if (e)
		return e;
	for (int s : seen)
		if (s != 1)
			return 7;
	return 0;
}

// mode 0: flat range [a0, a1) -> out[l - a0];  mode 1: shard (rank = a0, nranks = a1) -> packed
int emu_sample_nodes(void* h, const double dmin[3], const double cell[3], const uint32_t res[3], int invert, int mode,
					 uint64_t a0, uint64_t a1, const uint8_t* mask, double* out, uint8_t* written /*nullable*/,
					 uint64_t* stats /*12, nullable*/)
{
	auto m = static_cast<HostMesh*>(h);
	SampleParams P;
	init_params(P, m->dev, dmin, cell, invert);
	if (mode == 0)
		layout_range(P, res, a0, a1);
	else
		layout_shard(P, res, (int)a0, (int)a1);
	P.mask = mask;
	P.out = out;
	P.brick_blocking = g_brick_blocking;
	Stats st;
	int err = 0;
	// heavy-brick scratch exactly as dg_capi.cpp attaches it
	uint32_t ovf_count = 0;
	std::unique_ptr<uint32_t[]> ovf_brick;
	std::unique_ptr<double[]> saved_d2, cand_d2; // uninitialised on purpose: only parked slots are ever touched
	std::unique_ptr<int32_t[]> saved_tri, cand_tri;
	std::memset(&P.ovf, 0, sizeof(P.ovf));
	const uint32_t slots = g_heavy_slots == kAutoSlots ? overflow_slots_for(P.total_bricks) : g_heavy_slots;
	if (slots > 0 && P.mesh.n_sub >= 2)
	{
		ovf_brick.reset(new uint32_t[slots]);
		saved_d2.reset(new double[(size_t)slots * 64]);
		saved_tri.reset(new int32_t[(size_t)slots * 64]);
		cand_d2.reset(new double[(size_t)slots * kSubtrees * 64]);
		cand_tri.reset(new int32_t[(size_t)slots * kSubtrees * 64]);
		P.ovf.count = &ovf_count;
		P.ovf.brick = ovf_brick.get();
		P.ovf.saved_d2 = saved_d2.get();
		P.ovf.saved_tri = saved_tri.get();
		P.ovf.cand_d2 = cand_d2.get();
		P.ovf.cand_tri = cand_tri.get();
		P.ovf.slots = slots;
		P.ovf.heavy_work = g_heavy_work > 0 ? g_heavy_work : heavy_work_for(P.mesh.n_positions);
	}
	P.filtered = g_fast ? 1 : 0;
	auto write_nodes = [&](const LaneNode* ln, const bool* sample, const Wave& w) {
		for (int l = 0; l < 64; ++l)
		{
			if (!ln[l].valid)
				continue;
			double v = 1.7976931348623157e308;
			if (sample[l] && w.q[l].best_tri >= 0)
			{
				const LaneResult r = finish_query(P.mesh.tris, P.mesh.pn, w.q[l], HostSqrt());
				v = P.invert ? -1.0 * r.signed_dist : r.signed_dist;
			}
			out[ln[l].out_idx] = v;
			if (written)
			{
#pragma omp atomic
				written[ln[l].out_idx]++;
			}
		}
	};
	auto init_wave = [&](uint64_t brick, LaneNode* ln, bool* sample, Wave& w) {
		bool any = false;
		for (int l = 0; l < 64; ++l)
		{
			ln[l] = map_lane(P, brick, l);
			sample[l] = ln[l].valid && (!mask || mask[ln[l].out_idx] != 0);
			double x[3];
			node_position(ln[l].cls, ln[l].a, ln[l].b, ln[l].s, P.dmin, P.cell, x);
			init_query(P.mesh.origin, P.mesh.mesh_l1, sample[l], x[0], x[1], x[2], w.q[l]);
			any = any || sample[l];
		}
		return any;
	};
	// same block/XCD enumeration as the kernel (coverage of the remap is part of the test)
	const uint32_t grid = P.blocks_per_xcd * 8u;
#pragma omp parallel for schedule(dynamic, 16)
	for (long long bid = 0; bid < (long long)grid; ++bid)
	{
		uint32_t blk;
		if (!logical_block(P, (uint32_t)bid, &blk))
			continue;
		Stats ls;
		for (int wave = 0; wave < kWavesPerBlock; ++wave)
		{
			const uint64_t brick = (uint64_t)blk * (uint64_t)kWavesPerBlock + (uint64_t)wave;
			if (brick >= P.total_bricks)
				continue;
			Wave w;
			LaneNode ln[64];
			bool sample[64];
			const bool any = init_wave(brick, ln, sample, w);
			ls.bricks++;
			int slot = -1;
			if (any && P.filtered)
			{
				// k_sample_fast
				FastLane fl[64];
				FastLists lists;
				FastStats fs;
				fs.bricks = 1;
				bool exact[64];
				bool any_fast = false, any_exact = false;
				for (int l = 0; l < 64; ++l)
				{
					fl[l].a = make_approx_lane(w.q[l].px - P.mesh.origin[0], w.q[l].py - P.mesh.origin[1],
											   w.q[l].pz - P.mesh.origin[2], P.mesh.mesh_l1);
					exact[l] = sample[l] && !(fl[l].a.E < __builtin_inff());
					init_fast_lane(fl[l], sample[l] && !exact[l], FastLists::base(l));
					any_fast = any_fast || (sample[l] && !exact[l]);
				}
				if (any_fast && g_seed_study)
				{
					// design study: what would an ORACLE-TIGHT initial upper bound save?  A first traversal (not counted) gives every
					// lane's final upper bound; the counted traversal starts from it
					FastStats discard;
					FastLists scratch_lists;
					FastLane pre[64];
					for (int l = 0; l < 64; ++l)
						pre[l] = fl[l];
					(void)walk_fast(P.mesh, pre, scratch_lists, discard, nullptr);
					for (int l = 0; l < 64; ++l)
						if (sample[l] && !exact[l] && pre[l].U < __builtin_inff())
						{
							float theta, kappa;
							fl[l].U = pre[l].U * (1.0f + 1.0e-5f);
							if (g_seed_study == 2) // what a coarse pre-pass could hand down: the distance at the brick's centre + the brick's radius
							{
								const float radius = 1.5f * sqrtf((float)(P.cell[0] * P.cell[0] + P.cell[1] * P.cell[1] + P.cell[2] * P.cell[2]));
								const float d = sqrtf(pre[l].U) + 2.0f * radius; // (centre value <= own + radius; own <= centre + radius)
								fl[l].U = d * d;
							}
							approx_err_terms(fl[l].a.E, fl[l].U, &theta, &kappa);
							fl[l].Uprune = __builtin_fmaf(fl[l].U, 1.0f + theta, kappa);
						}
				}
				if (any_fast)
				{
					slot = walk_fast(P.mesh, fl, lists, fs, &P.ovf);
					if (slot >= 0) // parked as a heavy brick with the upper bounds as seeds
					{
						fs.parked = 1;
						P.ovf.brick[slot] = (uint32_t)brick;
						for (int l = 0; l < 64; ++l)
						{
							P.ovf.saved_d2[slot * 64 + l] = exact[l] ? 1.7976931348623157e308 : (double)fl[l].U;
							P.ovf.saved_tri[slot * 64 + l] = kSeedOnly;
						}
					}
					else
						for (int l = 0; l < 64; ++l)
							exact[l] = exact[l] || (sample[l] && (slot == -2 || FastLists::count(fl[l], l) >= kFastListCap));
				}
				if (slot < 0)
				{
					for (int l = 0; l < 64; ++l)
						any_exact = any_exact || exact[l];
					Wave wx; // the exact traversal of the lanes the filter could not serve
					int slot2 = -1;
					if (any_exact)
					{
						fs.redo_bricks = 1;
						for (int l = 0; l < 64; ++l)
						{
							init_query(P.mesh.origin, P.mesh.mesh_l1, exact[l], w.q[l].px, w.q[l].py, w.q[l].pz, wx.q[l]);
							if (exact[l] && fl[l].U > 0.0f)
								wx.q[l].bestf = best_as_float((double)fl[l].U);
						}
						slot2 = walk_exact(P.mesh, wx, ls, P.mesh.root_info, P.ovf.count ? &P.ovf : nullptr, true);
					}
					if (slot2 >= 0)
					{
						fs.parked = 1;
						P.ovf.brick[slot2] = (uint32_t)brick;
						for (int l = 0; l < 64; ++l)
						{
							const bool have = exact[l] && wx.q[l].best_tri >= 0;
							P.ovf.saved_d2[slot2 * 64 + l] = have ? wx.q[l].best_d2 : (exact[l] ? 1.7976931348623157e308 : (double)fl[l].U);
							P.ovf.saved_tri[slot2 * 64 + l] = have ? wx.q[l].best_tri : kSeedOnly;
						}
					}
					else
					{
						int mx = 0, mx2 = 0;
						// The epilogue of k_sample_fast as the device runs it (dg_kernels_k1.hip): every lane tests the first candidate of its
						// list itself; what is left of the lists is POOLED -- (owner, triangle) pairs laid out contiguously in list order
						// behind the pairs of the lanes below (`before`, by bit planes of the tail lengths as the device's ballots count
						// them), tested 64 per round with the OWNER's point, the values handed back to the owners in list order -- unless
						// the pool does not hold them (stack_levels x 32 items, (kFastListCap + 1) x 32 results, or g_pool_cap): then lane
						// by lane.  Same values in the same order either way: what this model pins is the index arithmetic.
						int n_cand[64], tail_n[64];
						for (int l = 0; l < 64; ++l)
						{
							n_cand[l] = (sample[l] && !exact[l]) ? FastLists::count(fl[l], l) : 0;
							tail_n[l] = n_cand[l] > 1 ? n_cand[l] - 1 : 0;
							if (!sample[l])
								continue;
							if (exact[l])
							{
								w.q[l].best_d2 = wx.q[l].best_d2;
								w.q[l].best_tri = wx.q[l].best_tri;
								continue;
							}
							fs.lanes++;
							fs.sum_list += n_cand[l];
							fs.hist[n_cand[l]]++;
							mx = n_cand[l] > mx ? n_cand[l] : mx;
							int need = 0;
							for (int k = 0; k < n_cand[l]; ++k)
								need += tri_closest<false>(P.mesh.tris[lists.at(l, k)], w.q[l].px, w.q[l].py, w.q[l].pz).d2 <= (double)fl[l].U * 1.00001;
							g_need_sum += need;
							mx2 = need > mx2 ? need : mx2;
							if (n_cand[l] > 0) // the head round
							{
								const int t = lists.at(l, 0);
								offer(w.q[l], tri_closest<false>(P.mesh.tris[t], w.q[l].px, w.q[l].py, w.q[l].pz).d2, t);
							}
						}
						uint32_t before[64], T = 0;
						for (int l = 0; l < 64; ++l)
							before[l] = 0;
						for (int b = 3; b >= 0; --b)
						{
							unsigned long long m = 0;
							for (int l = 0; l < 64; ++l)
								m |= (unsigned long long)((tail_n[l] >> b) & 1) << l;
							for (int l = 0; l < 64; ++l)
								before[l] += (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull)) << b; // mbcnt: set lanes BELOW l
							T += (uint32_t)__builtin_popcountll(m) << b;
						}
						if (T > 0)
						{
							const uint32_t item_cap = (uint32_t)P.mesh.stack_levels * 32u, res_cap = (uint32_t)(kFastListCap + 1) * 32u;
							if (T <= item_cap && T <= res_cap && T <= g_pool_cap)
							{
								fs.pooled_waves++;
								std::vector<uint32_t> items(T, 0xffffffffu);
								std::vector<double> res(T, -1.0);
								for (int l = 0; l < 64; ++l)
									for (int j = 0; j < tail_n[l]; ++j)
										items[before[l] + (uint32_t)j] = ((uint32_t)l << 26) | (uint32_t)lists.at(l, 1 + j);
								for (uint32_t base = 0; base < T; base += 64u)
									for (int l = 0; l < 64; ++l)
									{
										const uint32_t i = base + (uint32_t)l;
										if (i >= T)
											continue;
										const int owner = (int)(items[i] >> 26);
										res[i] = tri_closest<false>(P.mesh.tris[items[i] & 0x3ffffffu], w.q[owner].px, w.q[owner].py, w.q[owner].pz).d2;
									}
								for (int l = 0; l < 64; ++l)
									for (int j = 0; j < tail_n[l]; ++j)
										offer(w.q[l], res[before[l] + (uint32_t)j], (int)(items[before[l] + (uint32_t)j] & 0x3ffffffu));
							}
							else
							{
								fs.unpooled_waves++;
								for (int l = 0; l < 64; ++l)
									for (int k = 1; k < n_cand[l]; ++k)
									{
										const int t = lists.at(l, k);
										offer(w.q[l], tri_closest<false>(P.mesh.tris[t], w.q[l].px, w.q[l].py, w.q[l].pz).d2, t);
									}
							}
						}
						g_need_max += mx2;
						fs.sum_max_list += mx;
						write_nodes(ln, sample, w);
					}
				}
				if (getenv("EMU_LEAF_DEBUG") && slot < 0)
				{
					int needed = 0, winners = 0;
					for (auto const& lf : g_leaf_log)
					{
						bool cand = false, win = false;
						for (int l = 0; l < 64 && !win; ++l)
						{
							if (!sample[l]) continue;
							for (int k = 0; k < FastLists::count(fl[l], l); ++k)
								if (lists.at(l, k) >= lf.first && lists.at(l, k) < lf.first + lf.second)
									cand = true;
							if (w.q[l].best_tri >= lf.first && w.q[l].best_tri < lf.first + lf.second)
								win = true;
						}
						needed += cand || win;
						winners += win;
					}
					const double rx = w.q[21].px - P.mesh.origin[0], ry = w.q[21].py - P.mesh.origin[1], rz = w.q[21].pz - P.mesh.origin[2];
#pragma omp critical
					fprintf(stderr, "leaf %.3f %d %d %d\n", sqrt(rx*rx+ry*ry+rz*rz), (int)g_leaf_log.size(), needed, winners);
				}
				if (getenv("EMU_COST_DEBUG"))
				{
					const double rx = w.q[21].px - P.mesh.origin[0], ry = w.q[21].py - P.mesh.origin[1], rz = w.q[21].pz - P.mesh.origin[2];
#pragma omp critical
					fprintf(stderr, "cost %.4f %llu %llu %llu %llu\n", sqrt(rx*rx+ry*ry+rz*rz), (unsigned long long)fs.pair_steps, (unsigned long long)fs.tri_pairs, (unsigned long long)fs.sum_max_list, (unsigned long long)fs.parked);
				}
#pragma omp critical
				g_fs.add(fs);
				continue;
			}
			if (any)
				slot = walk_exact(P.mesh, w, ls, P.mesh.root_info, P.ovf.count ? &P.ovf : nullptr);
			if (slot >= 0) // k_sample_nodes parks the wave
			{
				P.ovf.brick[slot] = (uint32_t)brick;
				for (int l = 0; l < 64; ++l)
				{
					P.ovf.saved_d2[slot * 64 + l] = w.q[l].best_d2;
					P.ovf.saved_tri[slot * 64 + l] = w.q[l].best_tri;
				}
				continue;
			}
			write_nodes(ln, sample, w);
		}
#pragma omp critical
		{
			st.bricks += ls.bricks;
			st.node_visits += ls.node_visits;
			st.leaf_visits += ls.leaf_visits;
			st.tri_tests += ls.tri_tests;
			st.descent_nodes += ls.descent_nodes;
			st.slab_tests += ls.slab_tests;
			st.lane_interest += ls.lane_interest;
			st.useful_tests += ls.useful_tests;
			st.leaf_groups += ls.leaf_groups;
			st.pops += ls.pops;
			st.stale_pops += ls.stale_pops;
		}
	}
	// k_heavy_subtrees, k_heavy_finish
	const uint32_t parked = P.ovf.count ? std::min(ovf_count, P.ovf.slots) : 0u;
	st.heavy_bricks = parked;
	for (uint32_t slot = 0; slot < parked; ++slot)
	{
		LaneNode ln[64];
		bool sample[64];
		for (int s = 0; s < P.mesh.n_sub; ++s)
		{
			Wave w;
			init_wave(P.ovf.brick[slot], ln, sample, w);
			for (int l = 0; l < 64; ++l)
			{
				if (sample[l] && P.ovf.saved_tri[slot * 64 + l] >= 0)
					offer(w.q[l], P.ovf.saved_d2[slot * 64 + l], P.ovf.saved_tri[slot * 64 + l]);
				else if (sample[l] && P.ovf.saved_tri[slot * 64 + l] == kSeedOnly)
					w.q[l].bestf = fmin2(w.q[l].bestf, best_as_float(P.ovf.saved_d2[slot * 64 + l]));
			}
			if (g_heavy_study)
			{
				// the job as the filtered walk would run it (seeded by the parked upper bounds); counted, results discarded
				FastLane fl[64];
				FastLists lists;
				FastStats fs;
				bool any = false, all_seeded = true;
				for (int l = 0; l < 64; ++l)
				{
					fl[l].a = make_approx_lane(w.q[l].px - P.mesh.origin[0], w.q[l].py - P.mesh.origin[1], w.q[l].pz - P.mesh.origin[2], P.mesh.mesh_l1);
					const bool serve = sample[l] && fl[l].a.E < __builtin_inff();
					all_seeded = all_seeded && (!sample[l] || (serve && P.ovf.saved_tri[slot * 64 + l] == kSeedOnly && P.ovf.saved_d2[slot * 64 + l] < 1.0e300));
					init_fast_lane(fl[l], serve, FastLists::base(l));
					if (serve && P.ovf.saved_d2[slot * 64 + l] < 1.0e300)
					{
						float theta, kappa;
						fl[l].U = (float)P.ovf.saved_d2[slot * 64 + l];
						approx_err_terms(fl[l].a.E, fl[l].U, &theta, &kappa);
						fl[l].Uprune = __builtin_fmaf(fl[l].U, 1.0f + theta, kappa);
					}
					any = any || serve;
				}
				Stats before = st;
				walk_exact(P.mesh, w, st, P.mesh.sub_roots[s], nullptr);
				const uint64_t e_steps = (st.node_visits - before.node_visits) / 2, e_groups = st.leaf_groups - before.leaf_groups, e_tests = st.tri_tests - before.tri_tests;
				g_hs[0]++;
				g_hs[1] += e_steps;
				g_hs[2] += e_groups;
				g_hs[3] += e_tests;
				g_hs[14] += e_steps <= 1 && e_tests == 0;
				g_hs_hist[e_tests == 0 ? 0 : e_tests < 10 ? 1 : e_tests < 50 ? 2 : e_tests < 100 ? 3 : e_tests < 200 ? 4 : e_tests < 300 ? 5 : e_tests < 380 ? 6 : 7]++;
				bool fallback = !any || !all_seeded;
				if (!fallback)
				{
					const int r = walk_fast(P.mesh, fl, lists, fs, nullptr, P.mesh.sub_roots[s]);
					uint64_t cand = 0;
					for (int l = 0; l < 64; ++l)
					{
						const int c = sample[l] ? FastLists::count(fl[l], l) : 0;
						cand += (uint64_t)c;
						fallback = fallback || c >= kFastListCap;
					}
					fallback = fallback || r == -2;
					g_hs[4] += fs.pair_steps;
					g_hs[5] += fs.leaf_visits;
					g_hs[6] += fs.hist[16];
					g_hs[7] += fs.tri_pairs;
					g_hs[8] += cand;
					g_hs[9] += (cand + 63) / 64;
				}
				if (fallback)
				{
					g_hs[10]++;
					g_hs[11] += e_steps;
					g_hs[12] += e_groups;
					g_hs[13] += e_tests;
				}
			}
			else
				walk_exact(P.mesh, w, st, P.mesh.sub_roots[s], nullptr);
			for (int l = 0; l < 64; ++l)
			{
				const size_t at = ((size_t)slot * kSubtrees + (size_t)s) * 64 + (size_t)l;
				P.ovf.cand_d2[at] = w.q[l].best_d2;
				P.ovf.cand_tri[at] = w.q[l].best_tri;
			}
		}
		Wave w;
		init_wave(P.ovf.brick[slot], ln, sample, w);
		for (int l = 0; l < 64; ++l)
			if (sample[l])
				for (int s = 0; s < P.mesh.n_sub; ++s)
				{
					const size_t at = ((size_t)slot * kSubtrees + (size_t)s) * 64 + (size_t)l;
					if (P.ovf.cand_tri[at] >= 0)
						offer(w.q[l], P.ovf.cand_d2[at], P.ovf.cand_tri[at]);
				}
		write_nodes(ln, sample, w);
	}
	if (stats)
	{
		stats[0] = st.bricks;
		stats[1] = st.node_visits;
		stats[2] = st.leaf_visits;
		stats[3] = st.tri_tests;
		stats[4] = st.descent_nodes;
		stats[5] = st.slab_tests;
		stats[6] = st.lane_interest;
		stats[7] = st.useful_tests;
		stats[8] = st.leaf_groups;
		stats[9] = st.pops;
		stats[10] = st.stale_pops;
		stats[11] = st.heavy_bricks;
	}
	return err;
}

// the product's single-point host evaluator (dg_host_query.h), called from `threads` OpenMP threads at once
void emu_host_signed_distance(void* h, const double* xyz, uint64_t n, int threads, double* dist, int32_t* tri, int32_t* entity,
							  double* nearest)
{
	auto m = static_cast<HostMesh*>(h);
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
	for (long long i = 0; i < (long long)n; ++i)
	{
		LaneResult r;
		if (!dg::host::signed_distance_point(m->dev, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], r))
		{
			dist[i] = 1.7976931348623157e308;
			if (tri) tri[i] = -1;
			if (entity) entity[i] = -1;
			continue;
		}
		dist[i] = r.signed_dist;
		if (tri) tri[i] = r.tri_id;
		if (entity) entity[i] = r.entity;
		if (nearest)
			for (int d = 0; d < 3; ++d)
				nearest[3 * i + d] = r.nearest[d];
	}
}

void emu_signed_distance(void* h, const double* xyz, uint64_t n, double* dist, int32_t* tri, int32_t* entity,
						 double* nearest)
{
	auto m = static_cast<HostMesh*>(h);
	const long long n_waves = (long long)((n + 63) / 64);
#pragma omp parallel for schedule(dynamic, 4)
	for (long long wv = 0; wv < n_waves; ++wv)
	{
		Wave w;
		Stats st;
		bool act[64];
		for (int l = 0; l < 64; ++l)
		{
			const uint64_t gid = (uint64_t)wv * 64 + l;
			const bool valid = gid < n;
			const uint64_t g = valid ? gid : n - 1;
			init_query(m->dev.origin, m->dev.mesh_l1, valid, xyz[3 * g], xyz[3 * g + 1], xyz[3 * g + 2], w.q[l]);
			act[l] = valid;
		}
		if (g_fast)
		{
			FastStats fs;
			fast_wave_unparked(m->dev, w, act, fs, st);
#pragma omp critical
			g_fs.add(fs);
		}
		else
			walk_exact(m->dev, w, st, m->dev.root_info, nullptr);
		for (int l = 0; l < 64; ++l)
		{
			const uint64_t gid = (uint64_t)wv * 64 + l;
			if (gid >= n)
				continue;
			if (w.q[l].best_tri < 0)
			{
				dist[gid] = 1.7976931348623157e308;
				if (tri) tri[gid] = -1;
				if (entity) entity[gid] = -1;
				continue;
			}
			const LaneResult r = finish_query(m->dev.tris, m->dev.pn, w.q[l], HostSqrt());
			dist[gid] = r.signed_dist;
			if (tri) tri[gid] = r.tri_id;
			if (entity) entity[gid] = r.entity;
			if (nearest)
				for (int d = 0; d < 3; ++d)
					nearest[3 * gid + d] = r.nearest[d];
		}
	}
}

// work counters of the packet traversal for caller-supplied points, 64 consecutive points per wave
// (design studies: how does the grouping of lattice nodes into waves change the work?)
void emu_points_work(void* h, const double* xyz, uint64_t n, uint64_t* stats /*4: waves, node_visits, tri_tests, leaf_groups*/)
{
	auto m = static_cast<HostMesh*>(h);
	const long long n_waves = (long long)((n + 63) / 64);
	uint64_t nv = 0, tt = 0, lg = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : nv, tt, lg)
	for (long long wv = 0; wv < n_waves; ++wv)
	{
		Wave w;
		Stats st;
		for (int l = 0; l < 64; ++l)
		{
			const uint64_t gid = (uint64_t)wv * 64 + l;
			const uint64_t g = gid < n ? gid : n - 1;
			init_query(m->dev.origin, m->dev.mesh_l1, gid < n, xyz[3 * g], xyz[3 * g + 1], xyz[3 * g + 2], w.q[l]);
		}
		walk_exact(m->dev, w, st, m->dev.root_info, nullptr);
		nv += st.node_visits;
		tt += st.tri_tests;
		lg += st.leaf_groups;
	}
	stats[0] = (uint64_t)n_waves;
	stats[1] = nv;
	stats[2] = tt;
	stats[3] = lg;
}

// Float filter vs the double test: for every (triangle, point) the interval [q - err, q + err] the filtered
// kernel would form (error terms around the TRUE distance as d0 and around a 100 x larger one) must contain
// tri_closest's double value.  origin = the mesh origin the records are relative to; mesh_l1 as dg_build
// computes it.  Returns the number of violations; worst = max of (|q - d2| / err) seen.
uint64_t emu_filter_check(const double* tri /* n_tri x 9 */, size_t n_tri, const double* pts /* n_pts x 3 */, size_t n_pts,
						  const double origin[3], double* worst, uint64_t* checked)
{
	float l1 = 0.0f;
	for (size_t t = 0; t < n_tri; ++t)
		for (int k = 0; k < 3; ++k)
		{
			const double* v = tri + 9 * t + 3 * k;
			const double a = std::fabs(v[0] - origin[0]) + std::fabs(v[1] - origin[1]) + std::fabs(v[2] - origin[2]);
			l1 = std::max(l1, std::nextafterf((float)a, INFINITY));
		}
	uint64_t bad = 0, n = 0;
	double w = 0.0;
	for (size_t t = 0; t + 1 < n_tri; t += 2)
	{
		TriApproxPair rec;
		std::memset(&rec, 0, sizeof(rec));
		TriPacket pk[2];
		for (int side = 0; side < 2; ++side)
		{
			const double* v = tri + 9 * (t + side);
			make_tri_approx(v, v + 3, v + 6, origin, rec, side);
			make_packet(v, v + 3, v + 6, (int32_t)(t + side), pk[side]);
		}
		for (size_t i = 0; i < n_pts; ++i)
		{
			const double* p = pts + 3 * i;
			const ApproxLane a = make_approx_lane(p[0] - origin[0], p[1] - origin[1], p[2] - origin[2], l1);
			if (!(a.E < __builtin_inff()))
				continue;
			TriFrame fr;
			const f2 qlb = tri_approx_frame(&rec.f[0][0], a, &fr);
			const f2 q = tri_approx_rest(&rec.f[0][0], a, fr);
			for (int side = 0; side < 2; ++side)
			{
				if (rec.valid[side] != 1)
					continue;
				const double d2 = tri_closest<false>(pk[side], p[0], p[1], p[2]).d2;
				const float qs = side ? q.y : q.x;
				for (double scale : {1.0, 100.0, 0.01})
				{
					float theta, kappa;
					approx_err_terms(a.E, (float)(d2 * scale * scale), &theta, &kappa);
					const float err = __builtin_fmaf(qs, theta, kappa);
					const float lo = qs - err, up = qs + err;
					++n;
					const float qb = side ? qlb.y : qlb.x; // the rectangle bound must stay below the double value too
					const float lo_b = qb - __builtin_fmaf(qb, theta, kappa);
					if (!((double)lo <= d2 && d2 <= (double)up && (double)lo_b <= d2))
					{
						++bad;
						if (getenv("EMU_FILTER_DEBUG") && scale == 1.0)
						{
							const double* v = tri + 9 * (t + side);
							double e0[3], e1[3], e2[3];
							for (int d = 0; d < 3; ++d) { e0[d] = v[3+d]-v[d]; e1[d] = v[6+d]-v[d]; e2[d] = v[6+d]-v[3+d]; }
							auto len = [](const double* e) { return std::sqrt(e[0]*e[0]+e[1]*e[1]+e[2]*e[2]); };
							const double n0 = e0[1]*e1[2]-e0[2]*e1[1], n1 = e0[2]*e1[0]-e0[0]*e1[2], n2 = e0[0]*e1[1]-e0[1]*e1[0];
							const double area2 = std::sqrt(n0*n0+n1*n1+n2*n2);
							fprintf(stderr, "viol q %.9g d2 %.17g err %.3g E %.3g sides %.3g %.3g %.3g area2/lmax^2 %.3g |p-o| %.3g\n", (double)qs, d2, (double)err, (double)a.E,
									len(e0), len(e1), len(e2), area2 / std::pow(std::max(len(e0), std::max(len(e1), len(e2))), 2), std::fabs(p[0]-origin[0])+std::fabs(p[1]-origin[1])+std::fabs(p[2]-origin[2]));
						}
					}
					if (scale == 1.0)
						w = std::max(w, std::fabs((double)qs - d2) / (double)err);
				}
			}
		}
	}
	*worst = w;
	*checked = n;
	return bad;
}

uint64_t emu_shard_count(const uint32_t res[3], int rank, int nranks) { return shard_count(res, rank, nranks); }

void emu_unpack(const uint32_t res[3], int nranks, const double* gathered, uint64_t stride, double* field)
{
	UnpackParams U;
	layout_unpack(U, res, nranks);
	U.stride = stride;
	const uint64_t total = U.class_off[4];
	for (uint64_t l = 0; l < total; ++l)
		field[l] = gathered[unpack_source(U, l)];
}

// mirrors k_unpack_ranks: scatter of the slots [r0, r1) (field entries of other ranks untouched)
void emu_unpack_ranks(const uint32_t res[3], int nranks, const double* gathered, uint64_t stride, int r0, int r1,
					  double* field)
{
	UnpackParams U;
	layout_unpack(U, res, nranks);
	U.stride = stride;
	for (int r = r0; r < r1; ++r)
		for (uint64_t off = 0; off < U.count[r]; ++off)
			field[unpack_dest(U, (uint32_t)r, off)] = gathered[(uint64_t)r * stride + off];
}

int g_emu_tiles = 0; // 1: emu_interpolate / emu_density_map read an (unreduced) field through a tile-major copy, 2: through the x-major copy
void emu_set_tile_major(int on) { g_emu_tiles = on; }
// the tile-major copy exactly as k_expand_tiles builds it
static std::vector<double> build_tiles(FieldDev& F)
{
	for (int d = 0; d < 3; ++d)
		F.ntile[d] = (F.res[d] + kTmCells - 1) / kTmCells;
	const uint64_t n_tiles = (uint64_t)F.ntile[0] * F.ntile[1] * F.ntile[2];
	std::vector<double> t(n_tiles * kTmNodes);
	for (uint64_t e = 0; e < t.size(); ++e)
	{
		const uint64_t tile = e / kTmNodes;
		const uint32_t slot = (uint32_t)(e - tile * kTmNodes);
		const uint32_t node = tile_slot_node(slot, (uint32_t)(tile % F.ntile[0]), (uint32_t)((tile / F.ntile[0]) % F.ntile[1]),
											 (uint32_t)(tile / ((uint64_t)F.ntile[0] * F.ntile[1])), F.res);
		t[e] = node == 0xffffffffu ? 0.0 : F.coeffs[node];
	}
	for (uint64_t tile = 0; tile < n_tiles; ++tile) // k_tile_flags
	{
		uint64_t flags = 0;
		for (uint32_t c = 0; c < 64; ++c)
		{
			uint32_t slots[32];
			tile_node_slots(c & 3u, (c >> 2) & 3u, c >> 4, slots);
			bool nov = false;
			for (int q = 0; q < 32; ++q)
				nov = nov || (t[tile * kTmNodes + slots[q]] == 1.7976931348623157e308);
			flags |= (uint64_t)nov << c;
		}
		std::memcpy(&t[tile * kTmNodes + kTmFlags], &flags, 8);
	}
	return t;
}
// the x-major copy of the Y and Z classes and the per-cell "no value" bits exactly as k_xmajor_copy / k_xmajor_flags build them
struct XMajorCopy
{
	std::vector<double> pairs;
	std::vector<uint64_t> flags;
};
static void build_xmajor(FieldDev& F, XMajorCopy& X)
{
	const uint64_t n_pairs = xmajor_doubles(F.res) / 2;
	X.pairs.resize(2 * n_pairs + 2);
	for (uint64_t e = 0; e < n_pairs; ++e)
	{
		const uint32_t node = xmajor_pair_node(e, F.res);
		X.pairs[2 * e] = F.coeffs[node];
		X.pairs[2 * e + 1] = F.coeffs[node + 1];
	}
	F.xmajor = X.pairs.data();
	const uint32_t words = xmajor_flag_words(F.res);
	X.flags.assign((size_t)F.res[2] * F.res[1] * words, 0ull);
	for (uint32_t k = 0; k < F.res[2]; ++k)
		for (uint32_t j = 0; j < F.res[1]; ++j)
			for (uint32_t i = 0; i < F.res[0]; ++i)
			{
				double cf[32];
				fetch_cell<kFieldXMajor>(F, i, j, k, 0u, cf);
				bool nov = false;
				for (int q = 0; q < 32; ++q)
					nov = nov || (cf[q] == 1.7976931348623157e308);
				if (nov)
					X.flags[((size_t)k * F.res[1] + j) * words + (i >> 6)] |= 1ull << (i & 63u);
			}
	F.xmajor_flags = X.flags.data();
}
extern int g_density_skip;
// K3 with one lane per lattice point (dg_density_cells.h) on the host: every wave and lane of a k_density_cells launch over
// [begin, end) -- layout_density_cells(), logical_block(), row_wave_map(), k3c_lane() with the host's wave context (the
// table entries computed where they are read).  out[l - begin]; hits[l - begin] += 1 for every node a lane writes (the
// caller checks "exactly once").  mask (nullable) is indexed like out.
void emu_density_cells(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3],
					   const double* coeffs, double h, double rho0, int band, uint64_t begin, uint64_t end, const uint8_t* mask,
					   const uint32_t block[3], double* out, uint32_t* hits)
{
	FieldDev F;
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = domain[d];
		F.dmax[d] = domain[3 + d];
		F.cell[d] = cell[d];
		F.inv_cell[d] = inv_cell[d];
		F.res[d] = res[d];
	}
	F.coeffs = coeffs;
	F.cells = nullptr;
	F.cell_map = nullptr;
	F.cell_major = nullptr;
	F.tile_major = nullptr;
	F.ntile[0] = F.ntile[1] = F.ntile[2] = 0;
	XMajorCopy xm;
	build_xmajor(F, xm);
	DensityParams P;
	std::vector<double> w;
	init_density_params(P, h, rho0, cell, band, w);
	P.wtab = w.data();
	uint32_t flags = 0;
	{
		dg::ClassGeom cg[4];
		const uint64_t n_coeffs = dg::class_geometry(res, cg);
		for (uint64_t i = 0; i < n_coeffs; ++i)
		{
			if (density_value_unsafe(coeffs[i]))
				flags |= 1u;
			if (coeffs[i] == 1.7976931348623157e308)
				flags |= 2u;
		}
	}
	P.skip_mode = (h >= 1.0e-12 && g_density_skip) ? 2 : 0;
	P.unsafe = &flags;
	SampleParams L;
	std::memset(&L, 0, sizeof(L));
	for (int d = 0; d < 3; ++d)
	{
		L.dmin[d] = domain[d];
		L.cell[d] = cell[d];
	}
	layout_density_cells(P, L, res, block);
	P.row_node_begin = begin;
	P.row_node_end = end;
	L.mask = mask;
	L.out = out;
	const K3CellsGeom G = k3c_geometry(F);
	const uint32_t grid = L.blocks_per_xcd * 8u;
	for (uint64_t l = 0; l < end - begin; ++l)
		out[l] = -7.0; // (no node value is negative: 0, a density, or DBL_MAX)
#pragma omp parallel for schedule(dynamic, 4)
	for (long long b = 0; b < (long long)grid; ++b)
	{
		uint32_t blk;
		if (!logical_block(L, (uint32_t)b, &blk) || blk >= P.row_prefix[4])
			continue;
		const RowWave m = row_wave_map(P, blk);
		for (int lane = 0; lane < 64; ++lane)
		{
			K3HostWave hw;
			k3c_lane(hw, L, F, P, G, m, lane);
		}
	}
	// a second pass that only records which nodes the launch's lanes own (valid lanes' nodes inside the range)
	for (uint32_t b = 0; b < grid; ++b)
	{
		uint32_t blk;
		if (!logical_block(L, b, &blk) || blk >= P.row_prefix[4])
			continue;
		const RowWave m = row_wave_map(P, blk);
		for (int lane = 0; lane < 64; ++lane)
		{
			uint32_t i = m.w[0] * (uint32_t)kK3cLx + ((uint32_t)lane & 15u), j = m.w[1] * (uint32_t)kK3cLy + (((uint32_t)lane >> 4) & 1u),
					 k = m.w[2] * (uint32_t)kK3cLz + ((uint32_t)lane >> 5);
			const bool valid = i <= res[0] && j <= res[1] && k <= res[2];
			i = std::min(i, res[0]);
			j = std::min(j, res[1]);
			k = std::min(k, res[2]);
			const K3PointNodes pn = k3c_point_nodes(res, i, j, k, valid);
			for (int e = 0; e < 4; ++e)
				for (uint64_t s = 0; s < (e == 0 ? 1u : 2u); ++s)
					if (pn.valid[e] && pn.node[e] + s >= begin && pn.node[e] + s < end)
						hits[pn.node[e] + s - begin] += 1;
		}
	}
}
// k3c_div_h() against the division: n random numerators per call -- uniform bit patterns in the allowed range, values
// around h, and numerators built so that the quotient lands next to a rounding boundary (q = m + ulp / 2 for a random m,
// d = RN(q h) and its neighbours).  Returns the number of quotients that differ from d / h.
uint64_t emu_div_h_check(double h, uint64_t n, uint64_t seed)
{
	const double y = 1.0 / h;
	uint64_t bad = 0;
	uint64_t s = seed * 0x9e3779b97f4a7c15ull + 1;
	auto next = [&s]() {
		s ^= s << 13;
		s ^= s >> 7;
		s ^= s << 17;
		return s;
	};
	for (uint64_t t = 0; t < n; ++t)
	{
		double d;
		const uint64_t r = next();
		const int kind = (int)(r & 3u);
		if (kind == 0)
		{
			// any exponent in [-900, 900], random significand and sign
			const int e = (int)(next() % 1801) - 900;
			const double m = 1.0 + (double)(next() >> 12) * 0x1p-52;
			d = std::ldexp((next() & 1u) ? -m : m, e);
		}
		else if (kind == 1)
		{
			d = h * (((double)(next() >> 11) * 0x1p-53) * 4.0 - 2.0); // [-2h, 2h)
		}
		else
		{
			// quotient next to a midpoint between two doubles
			const int e = (int)(next() % 41) - 20;
			const double m = std::ldexp(1.0 + (double)(next() >> 12) * 0x1p-52, e);
			const long double mid = (long double)m + (long double)std::ldexp(0x1p-53, e);
			d = (double)(mid * (long double)h);
			const int step = (int)(next() % 5) - 2;
			for (int q = 0; q < (step < 0 ? -step : step); ++q)
				d = std::nextafter(d, step < 0 ? -INFINITY : INFINITY);
			if (next() & 1u)
				d = -d;
		}
		if (k3c_div_h_unsafe(d) || !(std::fabs(d) <= 1.0e300))
			continue;
		if (k3c_div_h(d, h, y) != d / h)
			++bad;
	}
	return bad;
}
// The band-limited cell-major copy on the host: flags -> bit / rank words exactly as k_band_flags / the scan / k_band_expand
// build them, then K2 through band_row_of() -- mapped queries from the copy's rows, the others from the field -- with the
// product's locate_query / evaluate_cell.  rows_out: number of rows in the copy.
void emu_interpolate_band(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3], const double* coeffs,
						  const uint32_t* cells, const uint32_t* cell_map, uint64_t n_rows, double lo, double hi, const double* xyz, uint64_t n,
						  double* phi, double* grad, uint64_t* rows_out, uint64_t* mapped_out)
{
	FieldDev F;
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = domain[d];
		F.dmax[d] = domain[3 + d];
		F.cell[d] = cell[d];
		F.inv_cell[d] = inv_cell[d];
		F.res[d] = res[d];
	}
	F.coeffs = coeffs;
	F.cells = cells;
	F.cell_map = cell_map;
	F.cell_major = nullptr;
	F.tile_major = nullptr;
	F.ntile[0] = F.ntile[1] = F.ntile[2] = 0;
	auto indices = [&](uint64_t row, uint32_t idx[32]) {
		if (cells)
			for (int j = 0; j < 32; ++j)
				idx[j] = cells[32 * row + j];
		else
		{
			const uint32_t n01 = res[0] * res[1];
			const uint32_t k = (uint32_t)(row / n01), r = (uint32_t)(row % n01);
			cell_node_indices(r % res[0], r / res[0], k, res, idx);
		}
	};
	std::vector<uint64_t> bits((n_rows + 63) / 64, 0);
	std::vector<uint32_t> rank((n_rows + 63) / 64, 0);
	std::vector<double> rows;
	uint32_t count = 0;
	for (uint64_t row = 0; row < n_rows; ++row)
	{
		if ((row & 63u) == 0u)
			rank[row >> 6] = count;
		uint32_t idx[32];
		indices(row, idx);
		double mn = coeffs[idx[0]], mx = mn;
		for (int j = 1; j < 32; ++j)
		{
			const double v = coeffs[idx[j]];
			mn = v < mn ? v : mn;
			mx = v > mx ? v : mx;
		}
		if (mn <= hi && mx >= lo)
		{
			bits[row >> 6] |= 1ull << (row & 63u);
			for (int j = 0; j < 32; ++j)
				rows.push_back(coeffs[idx[j]]);
			++count;
		}
	}
	if (rows.empty())
		rows.resize(32, 0.0);
	F.band_rows = rows.data();
	F.band_bits = bits.data();
	F.band_rank = rank.data();
	*rows_out = count;
	uint64_t mapped = 0;
	for (uint64_t i = 0; i < n; ++i)
	{
		const CellQuery q = locate_query(F, xyz + 3 * i);
		double g[3] = {0.0, 0.0, 0.0};
		double v = 1.7976931348623157e308;
		if (q.valid)
		{
			double cf[32];
			const uint32_t r = band_row_of(F, q.row);
			if (r != 0xffffffffu)
			{
				++mapped;
				for (int j = 0; j < 32; ++j)
					cf[j] = F.band_rows[32 * (size_t)r + j];
			}
			else if (cells)
				fetch_cell<kFieldTable>(F, q.mi[0], q.mi[1], q.mi[2], q.row, cf);
			else
				fetch_cell<kFieldClosed>(F, q.mi[0], q.mi[1], q.mi[2], q.row, cf);
			v = grad ? evaluate_cell<true>(cf, q.xi, q.c0, g) : evaluate_cell<false>(cf, q.xi, q.c0, g);
		}
		phi[i] = v;
		if (grad)
			for (int d = 0; d < 3; ++d)
				grad[3 * i + d] = g[d];
	}
	*mapped_out = mapped;
}
void emu_interpolate(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3],
					 const double* coeffs, const uint32_t* cells, const uint32_t* cell_map, const double* xyz,
					 uint64_t n, double* phi, double* grad)
{
	FieldDev F;
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = domain[d];
		F.dmax[d] = domain[3 + d];
		F.cell[d] = cell[d];
		F.inv_cell[d] = inv_cell[d];
		F.res[d] = res[d];
	}
	F.coeffs = coeffs;
	F.cells = cells;
	F.cell_map = cell_map;
	F.cell_major = nullptr;
	F.tile_major = nullptr;
	F.ntile[0] = F.ntile[1] = F.ntile[2] = 0;
	std::vector<double> tiles;
	XMajorCopy xm;
	if (g_emu_tiles == 1 && !cells && !cell_map)
	{
		tiles = build_tiles(F);
		F.tile_major = tiles.data();
	}
	if (g_emu_tiles == 2 && !cells && !cell_map)
		build_xmajor(F, xm);
#pragma omp parallel for schedule(static)
	for (long long q = 0; q < (long long)n; ++q)
	{
		double g[3];
		phi[q] = grad ? interpolate_point<true>(F, xyz + 3 * q, g) : interpolate_point<false>(F, xyz + 3 * q, g);
		if (grad)
			for (int d = 0; d < 3; ++d)
				grad[3 * q + d] = g[d];
	}
}

// k_interpolate_rows on the host: the cell-major copy as k_expand_cells builds it (n_rows rows of 32 doubles), waves
// of 64 queries whose rows are staged with the kernel's lane -> (row, piece) map into padded rows, then locate_query /
// evaluate_cell per lane -- the two halves of interpolate_point_mode the kernel uses.
void emu_interpolate_rows(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3],
						  const double* coeffs, const uint32_t* cells, uint64_t n_rows, const uint32_t* cell_map, const double* xyz,
						  uint64_t n, double* phi, double* grad)
{
	FieldDev F;
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = domain[d];
		F.dmax[d] = domain[3 + d];
		F.cell[d] = cell[d];
		F.inv_cell[d] = inv_cell[d];
		F.res[d] = res[d];
	}
	F.coeffs = coeffs;
	F.cells = cells;
	F.cell_map = cell_map;
	F.tile_major = nullptr;
	F.ntile[0] = F.ntile[1] = F.ntile[2] = 0;
	std::vector<double> cm((size_t)n_rows * 32);
	for (uint64_t row = 0; row < n_rows; ++row) // k_expand_cells
	{
		uint32_t idx[32];
		if (cells)
			for (int j = 0; j < 32; ++j)
				idx[j] = cells[32 * row + j];
		else
		{
			const uint32_t n01 = res[0] * res[1];
			const uint32_t k = (uint32_t)(row / n01), r = (uint32_t)(row % n01);
			cell_node_indices(r % res[0], r / res[0], k, res, idx);
		}
		for (int j = 0; j < 32; ++j)
			cm[32 * row + j] = coeffs[idx[j]];
	}
	F.cell_major = cm.data();
	const int stride = 33;
#pragma omp parallel for schedule(static)
	for (long long base = 0; base < (long long)n; base += 64)
	{
		std::vector<double> rows(64 * stride, 0.0);
		CellQuery q[64];
		uint32_t my_row[64];
		for (int lane = 0; lane < 64; ++lane)
		{
			const uint64_t gid = (uint64_t)base + (uint64_t)lane;
			const bool have = gid < n;
			double x[3] = {0.0, 0.0, 0.0};
			if (have)
				for (int d = 0; d < 3; ++d)
					x[d] = xyz[3 * gid + d];
			q[lane] = locate_query(F, x);
			q[lane].valid = q[lane].valid && have;
			my_row[lane] = q[lane].valid ? q[lane].row : 0u;
		}
		for (int k = 0; k < 16; ++k)
			for (int lane = 0; lane < 64; ++lane)
			{
				const int sub = lane & 15, grp = lane >> 4, owner = 4 * k + grp;
				const double* v = F.cell_major + 32 * (size_t)my_row[owner] + 2 * sub;
				rows[owner * stride + 2 * sub] = v[0];
				rows[owner * stride + 2 * sub + 1] = v[1];
			}
		for (int lane = 0; lane < 64; ++lane)
		{
			const uint64_t gid = (uint64_t)base + (uint64_t)lane;
			if (gid >= n)
				continue;
			double cf[32], g[3] = {0.0, 0.0, 0.0};
			for (int j = 0; j < 32; ++j)
				cf[j] = rows[lane * stride + j];
			double v = 1.7976931348623157e308;
			if (q[lane].valid)
				v = grad ? evaluate_cell<true>(cf, q[lane].xi, q[lane].c0, g) : evaluate_cell<false>(cf, q[lane].xi, q[lane].c0, g);
			phi[gid] = v;
			if (grad)
				for (int d = 0; d < 3; ++d)
					grad[3 * gid + d] = g[d];
		}
	}
}

int g_density_skip = 1;
void emu_set_density_skip(int on) { g_density_skip = on; }
// K3 on the host: the product's density_prefilter / density_integral and launch constants
void emu_density_map(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3],
					 const double* coeffs, const uint32_t* cells, const uint32_t* cell_map, double h, double rho0, int band,
					 uint64_t begin, uint64_t end, double* out)
{
	FieldDev F;
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = domain[d];
		F.dmax[d] = domain[3 + d];
		F.cell[d] = cell[d];
		F.inv_cell[d] = inv_cell[d];
		F.res[d] = res[d];
	}
	F.coeffs = coeffs;
	F.cells = cells;
	F.cell_map = cell_map;
	F.cell_major = nullptr;
	F.tile_major = nullptr;
	F.ntile[0] = F.ntile[1] = F.ntile[2] = 0;
	std::vector<double> tiles;
	XMajorCopy xm;
	if (g_emu_tiles == 1 && !cells && !cell_map)
	{
		tiles = build_tiles(F);
		F.tile_major = tiles.data();
	}
	if (g_emu_tiles == 2 && !cells && !cell_map)
		build_xmajor(F, xm);
	DensityParams P;
	std::vector<double> w;
	init_density_params(P, h, rho0, cell, band, w);
	P.wtab = w.data();
	// the product's rule for skipping the zero-weight quadrature points (dg_capi_field.cpp + k_field_check)
	{
		dg::ClassGeom cg[4];
		const uint64_t n_coeffs = dg::class_geometry(res, cg);
		bool unsafe = false;
		if (cells == nullptr && cell_map == nullptr) // (a reduced field has fewer coefficients: leave it at "evaluate all")
			for (uint64_t i = 0; i < n_coeffs; ++i)
				unsafe = unsafe || density_value_unsafe(coeffs[i]);
		else
			unsafe = true;
		P.skip_mode = (!unsafe && h >= 1.0e-12 && g_density_skip) ? 1 : 0;
	}
#pragma omp parallel for schedule(dynamic, 16)
	for (long long l = (long long)begin; l < (long long)end; ++l)
	{
		double x[3], v;
		node_position_flat((uint64_t)l, F.res, F.dmin, F.cell, x);
		out[l - begin] = density_prefilter(F, P, x, &v) ? density_integral(F, P, x) : v;
	}
}

} // extern "C"
