// tests/emu/wave_emu.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host-side lock-step emulation of the K1 / K1p / K2 / unpack kernels of
// discregrid_amd/csrc/dg_kernels.hip.  It runs the PRODUCT's own data structures (dg_build:
// flattened BVH, triangle packets, pseudonormals), the product's own lattice decomposition
// (dg_layout.h, map_lane) and the product's own per-lane arithmetic (dg_geom.h) -- only the
// wave-level control flow (ballots over 64 lanes) is re-expressed as loops over lane arrays.
// It exists because the build container has no GPU: the algorithmic logic (does the packet
// traversal with conservative float boxes find the reference's distances bit for bit? does
// the brick decomposition cover every node exactly once?) is checked here on CPU before GPU
// minutes are spent.  The product never links this file and has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../discregrid_amd/csrc/dg_build.h"
#include "../../discregrid_amd/csrc/dg_kernels.h"
#include "../../discregrid_amd/csrc/dg_layout.h"

#ifndef DG_TRI_BOX
#define DG_TRI_BOX 1
#endif

using namespace dg;

namespace
{

struct Stats
{
	uint64_t bricks = 0, node_visits = 0, leaf_visits = 0, tri_tests = 0, descent_nodes = 0, slab_tests = 0, lane_interest = 0, useful_tests = 0, leaf_groups = 0;
};

struct HostSqrt
{
	double operator()(double x) const { return std::sqrt(x); }
};

struct Wave
{
	LaneQuery q[64];
};

void test_leaf(const MeshDev& M, int first, int cnt, const float* leaf_lb2 /*per lane or null*/, Wave& w, Stats& st)
{
	for (int g0 = 0; g0 < cnt; g0 += 4)
	{
		const int gfirst = first + g0;
		const int gcnt = (cnt - g0) < 4 ? (cnt - g0) : 4;
		unsigned want = 0;
		st.leaf_groups++;
		for (int t = 0; t < 4; ++t)
		{
			const TriSlab& sl = M.slabs[gfirst + t];
			bool any = false;
			int n_int = 0;
			for (int l = 0; l < 64; ++l)
			{
				float lb = slab_lb2(sl.u[0], sl.u[1], sl.u[2], sl.lo, sl.hi, w.q[l].fp);
#if DG_TRI_BOX
				lb = fmax2(lb, box_lb2(sl.blo, sl.bhi, w.q[l].fp));
#endif
				const bool hit = (fmax2(lb, leaf_lb2 ? leaf_lb2[l] : 0.0f) < w.q[l].bestf);
				any = any || hit;
				n_int += hit;
			}
			if (t < gcnt && any)
			{
				want |= 1u << t;
				st.lane_interest += n_int;
			}
		}
		for (int t = 0; t < gcnt; ++t)
		{
			st.slab_tests++;
			if (!((want >> t) & 1u))
				continue;
			st.tri_tests++;
			const TriPacket& T = M.tris[gfirst + t];
			bool useful = false;
			for (int l = 0; l < 64; ++l)
			{
				const Hit h = tri_closest<false>(T, w.q[l].px, w.q[l].py, w.q[l].pz);
				useful = useful || (h.d2 < w.q[l].best_d2);
				offer(w.q[l], h.d2, gfirst + t);
			}
			st.useful_tests += useful;
		}
	}
}

// mirrors traverse() of dg_kernels.hip: near-first packet traversal with a wave-shared stack
void traverse(const MeshDev& M, Wave& w, Stats& st)
{
	const BvhNode* nodes = M.nodes;
	int stack[128];
	int sp = 0;
	int node = 0;
	bool have = true; // `node` is valid and already known to be hit by some lane
	float lbcur[64];
	for (int k = 0; k < 64; ++k) lbcur[k] = 0.0f;
	while (true)
	{
		if (!have)
		{
			if (sp == 0)
				break;
			node = stack[--sp];
			const BvhNode& nd = nodes[node];
			st.node_visits++;
			bool any = false;
			for (int k = 0; k < 64; ++k)
			{
				lbcur[k] = node_lb2(nd.lo, nd.hi, nd.su, nd.slo, nd.shi, w.q[k].fp);
				any = any || (lbcur[k] < w.q[k].bestf);
			}
			if (!any)
				continue;
		}
		const BvhNode& nd = nodes[node];
		if (nd.info < 0)
		{
			st.leaf_visits++;
			const unsigned code = ~(unsigned)nd.info;
			test_leaf(M, (int)(code >> kLeafBits), (int)(code & (unsigned)(kMaxLeaf - 1)) + 1, lbcur, w, st);
			have = false;
			continue;
		}
		const int li = node + 1, ri = nd.info;
		const BvhNode& l = nodes[li];
		const BvhNode& r = nodes[ri];
		st.node_visits += 2;
		float lbl[64], lbr[64];
		bool anyl = false, anyr = false;
		int pref = 0, act = 0;
		for (int k = 0; k < 64; ++k)
		{
			lbl[k] = node_lb2(l.lo, l.hi, l.su, l.slo, l.shi, w.q[k].fp);
			lbr[k] = node_lb2(r.lo, r.hi, r.su, r.slo, r.shi, w.q[k].fp);
			const bool hl = lbl[k] < w.q[k].bestf, hr = lbr[k] < w.q[k].bestf;
			anyl = anyl || hl;
			anyr = anyr || hr;
			if (hl || hr)
			{
				act++;
				pref += (lbl[k] <= lbr[k]);
			}
		}
		if (anyl && anyr)
		{
			const bool left_first = 2 * pref >= act;
			stack[sp++] = left_first ? ri : li;
			node = left_first ? li : ri;
			for (int k = 0; k < 64; ++k) lbcur[k] = left_first ? lbl[k] : lbr[k];
			have = true;
		}
		else if (anyl || anyr)
		{
			node = anyl ? li : ri;
			for (int k = 0; k < 64; ++k) lbcur[k] = anyl ? lbl[k] : lbr[k];
			have = true;
		}
		else
			have = false;
	}
}

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
	m->dev.nodes = m->B.nodes.data();
	m->dev.tris = m->B.tris.data();
	m->dev.pn = m->B.pn.data();
	m->dev.slabs = m->B.slabs.data();
	m->dev.mesh_l1 = m->B.mesh_l1;
	m->dev.n_nodes = (int32_t)m->B.nodes.size();
	m->dev.n_tris = (int32_t)m->B.tris.size();
	for (int d = 0; d < 3; ++d)
		m->dev.origin[d] = m->B.origin[d];
	return m;
}
void emu_mesh_free(void* h) { delete static_cast<HostMesh*>(h); }
void emu_mesh_info(void* h, uint64_t* n_nodes, uint32_t* depth, uint32_t* flags)
{
	auto m = static_cast<HostMesh*>(h);
	*n_nodes = m->B.nodes.size();
	*depth = m->B.depth;
	*flags = m->B.not_watertight;
}
// pseudonormals in the CALLER's triangle order: pn[t][slot][3] (slot 7 unused)
void emu_mesh_pseudonormals(void* h, double* pn)
{
	auto m = static_cast<HostMesh*>(h);
	for (size_t k = 0; k < m->B.tris.size(); ++k)
		std::memcpy(pn + (size_t)m->B.tris[k].tri_id * kPnSlots * 3, m->B.pn.data() + k * kPnSlots * 3,
					kPnSlots * 3 * sizeof(double));
}
// structural self-check of the flattened BVH: every triangle in exactly one leaf, boxes
// contain their triangles (with the float rounding), skip pointers consistent
int emu_mesh_check(void* h, const double* verts, const uint32_t* tris)
{
	auto m = static_cast<HostMesh*>(h);
	const auto& N = m->B.nodes;
	std::vector<int> seen(m->B.tris.size(), 0);
	for (size_t i = 0; i < N.size(); ++i)
	{
		if (N[i].skip <= (int)i || N[i].skip > (int)N.size())
			return 1;
		if (N[i].info >= 0)
		{
			if (N[i].info <= (int)i + 1 || N[i].info >= N[i].skip)
				return 2;
			if (N[i + 1].skip != N[i].info || N[N[i].info].skip != N[i].skip)
				return 3;
		}
		else
		{
			if (N[i].skip != (int)i + 1)
				return 4;
			const unsigned code = ~(unsigned)N[i].info;
			for (unsigned t = code >> kLeafBits; t <= (code >> kLeafBits) + (code & (unsigned)(kMaxLeaf - 1)); ++t)
			{
				if (t >= seen.size())
					return 5;
				seen[t]++;
				const uint32_t id = (uint32_t)m->B.tris[t].tri_id;
				for (int k = 0; k < 3; ++k)
					for (int d = 0; d < 3; ++d)
					{
						const double v = verts[3 * tris[3 * id + k] + d] - m->B.origin[d];
						// every ancestor must contain it too: checked through the leaf's own box
						// being inside its ancestors (below)
						if (!((double)N[i].lo[d] <= v && v <= (double)N[i].hi[d]))
							return 6;
					}
			}
		}
	}
	for (int s : seen)
		if (s != 1)
			return 7;
	// child boxes inside parent boxes
	for (size_t i = 0; i < N.size(); ++i)
		if (N[i].info >= 0)
			for (int ch : {(int)i + 1, N[i].info})
				for (int d = 0; d < 3; ++d)
					if (N[ch].lo[d] < N[i].lo[d] || N[ch].hi[d] > N[i].hi[d])
						return 8;
	return 0;
}

// mode 0: flat range [a0, a1) -> out[l - a0];  mode 1: shard (rank = a0, nranks = a1) -> packed
int emu_sample_nodes(void* h, const double dmin[3], const double cell[3], const uint32_t res[3], int invert, int mode,
					 uint64_t a0, uint64_t a1, const uint8_t* mask, double* out, uint8_t* written /*nullable*/,
					 uint64_t* stats /*6, nullable*/)
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
	Stats st;
	int err = 0;
	// same block/XCD enumeration as the kernel (coverage of the remap is part of the test)
	const uint32_t grid = P.blocks_per_xcd * 8u;
#pragma omp parallel for schedule(dynamic, 16)
	for (long long bid = 0; bid < (long long)grid; ++bid)
	{
		const uint32_t xcd = (uint32_t)bid & 7u, within = (uint32_t)bid >> 3;
		const uint32_t blk = xcd * P.blocks_per_xcd + within;
		if (within >= P.blocks_per_xcd || blk >= P.n_blocks)
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
			ls.bricks++;
			if (any)
				traverse(P.mesh, w, ls);
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
		}
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
	}
	return err;
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
		for (int l = 0; l < 64; ++l)
		{
			const uint64_t gid = (uint64_t)wv * 64 + l;
			const bool valid = gid < n;
			const uint64_t g = valid ? gid : n - 1;
			init_query(m->dev.origin, m->dev.mesh_l1, valid, xyz[3 * g], xyz[3 * g + 1], xyz[3 * g + 2], w.q[l]);
		}
		traverse(m->dev, w, st);
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
	DensityParams P;
	std::vector<double> w;
	init_density_params(P, h, rho0, cell, band, w);
	P.wtab = w.data();
#pragma omp parallel for schedule(dynamic, 16)
	for (long long l = (long long)begin; l < (long long)end; ++l)
	{
		double x[3], v;
		node_position_flat((uint64_t)l, F.res, F.dmin, F.cell, x);
		out[l - begin] = density_prefilter(F, P, x, &v) ? density_integral(F, P, x) : v;
	}
}

} // extern "C"
