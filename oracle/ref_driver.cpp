// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// C entry points over the UNMODIFIED reference sources under /root/reference,
// compiled in place by oracle/Makefile (target `ref`) into
// oracle/_ref/libdiscregrid_ref.so against the stand-in Eigen header in
// discregrid_amd/cpp/third_party/eigen_min.  Nothing from the reference is copied into this repo: this
// file only *includes* its public headers and calls its public API the way
// cmd/generate_sdf/main.cpp:70-120 does.
//
// Used by tests/ to pin the CPU restatement (oracle/discregrid_oracle.cpp), to
// generate the golden vectors under tests/golden/, and by bench.py as the
// `cpu_baseline` leg (kind = "reference").
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <memory>
#include <string>
#include <vector>
// every standard header the reference's public headers pull in, included BEFORE the
// access-specifier override below so that libstdc++ itself is parsed untouched
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <fstream>
#include <functional>
#include <iostream>
#include <iterator>
#include <limits>
#include <set>
#include <sstream>
#include <streambuf>
#include <unordered_map>
#include <Eigen/Dense>

// The grid keeps its coefficient vectors private; the driver needs to read them
// without a 3 GB round trip through save().  Access specifiers do not change
// the layout produced by g++, and this applies to this translation unit only.
#define private public
#define protected public
#include <Discregrid/All>
#undef private
#undef protected

// the density-map tool's helpers (cmd/generate_density_map): header-only kernel + quadrature
#include "sph_kernel.hpp"
#include "gauss_quadrature.hpp"

using namespace Discregrid;

namespace
{
struct RefGrid
{
	std::unique_ptr<TriangleMesh> mesh;
	std::unique_ptr<TriangleMeshDistance> md;
	std::unique_ptr<CubicLagrangeDiscreteGrid> grid;
};
} // namespace

// GenerateDensityMap's lambdas (cmd/generate_density_map/main.cpp:83-133) around the grid's field 0, re-typed
// around the reference's CubicKernel, GaussQuadrature and DiscreteGrid::interpolate; `use(density_func, predicate)`
// runs with them in scope.
template <class Use>
static void with_density_lambdas(CubicLagrangeDiscreteGrid* sdf, double h, double rho0, int no_reduction, Use use)
{
	auto sph_kernel = CubicKernel{};
	sph_kernel.setRadius(h);
	auto gamma = [&](Eigen::Vector3d const& x) {
		auto ar = sph_kernel.getRadius();
		auto dist = sdf->interpolate(0u, x);
		if (dist > ar)
			return 0.0;
		return 1.0 - dist / ar;
	};
	auto int_domain = Eigen::AlignedBox3d(Eigen::Vector3d::Constant(-h), Eigen::Vector3d::Constant(h));
	auto density_func = [&](Eigen::Vector3d const& x) {
		auto dist = sdf->interpolate(0u, x);
		if (dist > 2.0 * sph_kernel.getRadius())
			return 0.0;
		auto integrand = [&sph_kernel, &gamma, &x](Eigen::Vector3d const& xi) {
			auto res = gamma(x + xi) * sph_kernel.W(xi);
			return res;
		};
		auto res = GaussQuadrature::integrate(integrand, int_domain, 30);
		return rho0 * res;
	};
	auto cell_diag = sdf->cellSize().norm();
	auto predicate = [&](Eigen::Vector3d const& x_) {
		if (no_reduction)
			return true;
		auto x = x_.cwiseMax(sdf->domain().min()).cwiseMin(sdf->domain().max());
		auto dist = sdf->interpolate(0u, x);
		if (dist == std::numeric_limits<double>::max())
			return false;
		return -6.0 * h < dist + cell_diag && dist - cell_diag < 2.0 * h;
	};
	use(density_func, predicate);
}


extern "C"
{

// Default domain rule of cmd/generate_sdf/main.cpp:83-91 (bbox grown by 1e-3*|diag|,
// max first, then min with the already grown diagonal).
void ref_default_domain(const double* verts, size_t nv, double out[6])
{
	Eigen::AlignedBox3d domain;
	domain.setEmpty();
	for (size_t i = 0; i < nv; ++i)
		domain.extend(Eigen::Vector3d(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]));
	domain.max() += 1.0e-3 * domain.diagonal().norm() * Eigen::Vector3d::Ones();
	domain.min() -= 1.0e-3 * domain.diagonal().norm() * Eigen::Vector3d::Ones();
	for (int d = 0; d < 3; ++d)
	{
		out[d] = domain.min()[d];
		out[3 + d] = domain.max()[d];
	}
}

// Mesh (raw arrays) -> TriangleMesh -> TriangleMeshDistance -> grid(domain,res),
// i.e. cmd/generate_sdf/main.cpp:70-93 without the OBJ parser.
void* ref_grid_create(const double* verts, size_t nv, const unsigned* faces, size_t nf,
					  const double domain[6], const unsigned res[3])
{
	auto g = new RefGrid;
	if (verts && nv && faces && nf)
	{
		g->mesh.reset(new TriangleMesh(verts, faces, nv, nf));
		g->md.reset(new TriangleMeshDistance(*g->mesh));
	}
	Eigen::AlignedBox3d dom(Eigen::Vector3d(domain[0], domain[1], domain[2]),
							Eigen::Vector3d(domain[3], domain[4], domain[5]));
	g->grid.reset(new CubicLagrangeDiscreteGrid(dom, {{res[0], res[1], res[2]}}));
	return g;
}

void* ref_grid_load(const char* path)
{
	auto g = new RefGrid;
	g->grid.reset(new CubicLagrangeDiscreteGrid(std::string(path)));
	return g;
}

void ref_grid_free(void* h) { delete static_cast<RefGrid*>(h); }

// addFunction with the SDF lambda of cmd/generate_sdf/main.cpp:95-105.  Returns the
// wall time of the addFunction call in seconds (negative on error).
double ref_grid_add_sdf(void* h, int invert)
{
	auto g = static_cast<RefGrid*>(h);
	if (!g->md)
		return -1.0;
	auto& md = *g->md;
	auto func = DiscreteGrid::ContinuousFunction{};
	if (invert)
		func = [&md](Eigen::Vector3d const& xi) { return -1.0 * md.signed_distance(xi).distance; };
	else
		func = [&md](Eigen::Vector3d const& xi) { return md.signed_distance(xi).distance; };
	auto t0 = std::chrono::high_resolution_clock::now();
	g->grid->addFunction(func, false);
	auto t1 = std::chrono::high_resolution_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

// Only the node-sampling loop (cubic_lagrange_discrete_grid.cpp:806-831) over a node
// range, without the cell table: used for the bounded-sample CPU baseline.  Runs the
// reference's own indexToNodePosition + signed_distance under the same OpenMP
// schedule(static).
double ref_sample_nodes(void* h, int invert, unsigned begin, unsigned end, double* out)
{
	auto g = static_cast<RefGrid*>(h);
	if (!g->md)
		return -1.0;
	auto& md = *g->md;
	auto& grid = *g->grid;
	const double sgn = invert ? -1.0 : 1.0;
	auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(static)
	for (long long l = begin; l < (long long)end; ++l)
	{
		auto x = grid.indexToNodePosition((unsigned)l);
		double d = md.signed_distance(x).distance;
		out[l - begin] = invert ? sgn * d : d; // -1.0 * distance, main.cpp:97
	}
	auto t1 = std::chrono::high_resolution_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

size_t ref_grid_n_fields(void* h) { return static_cast<RefGrid*>(h)->grid->m_nodes.size(); }
size_t ref_grid_n_nodes(void* h, unsigned f) { return static_cast<RefGrid*>(h)->grid->m_nodes[f].size(); }
size_t ref_grid_n_cells(void* h, unsigned f) { return static_cast<RefGrid*>(h)->grid->m_cells[f].size(); }
void ref_grid_get_nodes(void* h, unsigned f, double* out)
{
	auto& v = static_cast<RefGrid*>(h)->grid->m_nodes[f];
	std::memcpy(out, v.data(), v.size() * sizeof(double));
}
void ref_grid_get_cells(void* h, unsigned f, unsigned* out)
{
	auto& v = static_cast<RefGrid*>(h)->grid->m_cells[f];
	std::memcpy(out, v.data(), v.size() * 32 * sizeof(unsigned));
}
void ref_grid_get_cell_map(void* h, unsigned f, unsigned* out)
{
	auto& v = static_cast<RefGrid*>(h)->grid->m_cell_map[f];
	std::memcpy(out, v.data(), v.size() * sizeof(unsigned));
}
void ref_grid_get_header(void* h, double domain[6], unsigned res[3], double cell[3], double inv_cell[3])
{
	auto& g = *static_cast<RefGrid*>(h)->grid;
	for (int d = 0; d < 3; ++d)
	{
		domain[d] = g.domain().min()[d];
		domain[3 + d] = g.domain().max()[d];
		res[d] = g.resolution()[d];
		cell[d] = g.cellSize()[d];
		inv_cell[d] = g.invCellSize()[d];
	}
}
// Adds a field whose coefficients are given (node order of indexToNodePosition):
// appends the vector and lets the reference build its own cell table by running
// addFunction on a dummy function first.
unsigned ref_grid_add_coeffs(void* h, const double* coeffs, size_t n)
{
	auto& g = *static_cast<RefGrid*>(h)->grid;
	unsigned f = g.addFunction([](Eigen::Vector3d const&) { return 0.0; }, false);
	if (g.m_nodes[f].size() != n)
		return ~0u;
	std::memcpy(g.m_nodes[f].data(), coeffs, n * sizeof(double));
	return f;
}

void ref_grid_save(void* h, const char* path) { static_cast<RefGrid*>(h)->grid->save(path); }

void ref_node_positions(void* h, unsigned begin, unsigned end, double* xyz)
{
	auto& g = *static_cast<RefGrid*>(h)->grid;
	for (unsigned l = begin; l < end; ++l)
	{
		auto x = g.indexToNodePosition(l);
		xyz[3 * (size_t)(l - begin) + 0] = x[0];
		xyz[3 * (size_t)(l - begin) + 1] = x[1];
		xyz[3 * (size_t)(l - begin) + 2] = x[2];
	}
}

// interpolate(field, x, grad*) for a batch (cubic_lagrange_discrete_grid.cpp:977-1063),
// under `omp parallel for` like cmd/discrete_field_to_bitmap/main.cpp:118.
double ref_grid_interpolate(void* h, unsigned field, const double* xyz, size_t n, double* phi,
							double* grad /*nullable, 3n*/)
{
	auto& g = *static_cast<RefGrid*>(h)->grid;
	auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(static)
	for (long long q = 0; q < (long long)n; ++q)
	{
		Eigen::Vector3d x(xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]);
		if (grad)
		{
			Eigen::Vector3d gr;
			phi[q] = g.interpolate(field, x, &gr);
			grad[3 * q] = gr[0];
			grad[3 * q + 1] = gr[1];
			grad[3 * q + 2] = gr[2];
		}
		else
			phi[q] = g.interpolate(field, x);
	}
	auto t1 = std::chrono::high_resolution_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

// reduceField(field, |v| < bound)  -- predicate shape used by
// cmd/generate_density_map/main.cpp:138-145 (value-only predicates).
void ref_grid_reduce_abs_lt(void* h, unsigned field, double bound)
{
	static_cast<RefGrid*>(h)->grid->reduceField(
		field, [bound](Eigen::Vector3d const&, double v) { return std::abs(v) < bound; });
}

// GenerateDensityMap's addFunction call on the grid's field 0.  Returns the wall time of addFunction.
double ref_grid_add_density_map(void* hh, double h, double rho0, int no_reduction)
{
	auto* sdf = static_cast<RefGrid*>(hh)->grid.get();
	double secs = 0.0;
	with_density_lambdas(sdf, h, rho0, no_reduction, [&](auto& density_func, auto& predicate) {
		auto t0 = std::chrono::high_resolution_clock::now();
		sdf->addFunction(density_func, false, predicate);
		auto t1 = std::chrono::high_resolution_clock::now();
		secs = std::chrono::duration<double>(t1 - t0).count();
	});
	return secs;
}
// The same lambdas over a node RANGE, the way addFunction's node loop applies them
// (cubic_lagrange_discrete_grid.cpp:806-831: `pred && !pred(x) ? DBL_MAX : func(x)`), without the cell table:
// full-lattice digests in resumable chunks (tests/golden/make_digests.py) and the bounded-sample CPU baseline of K3.
// Nodes are independent, so the dynamic schedule changes no bit.
double ref_density_nodes(void* hh, double h, double rho0, int no_reduction, unsigned begin, unsigned end, double* out)
{
	auto* sdf = static_cast<RefGrid*>(hh)->grid.get();
	double secs = 0.0;
	with_density_lambdas(sdf, h, rho0, no_reduction, [&](auto& density_func, auto& predicate) {
		auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(dynamic, 64)
		for (long long l = begin; l < (long long)end; ++l)
		{
			auto x = sdf->indexToNodePosition((unsigned)l);
			out[l - begin] = !predicate(x) ? std::numeric_limits<double>::max() : density_func(x);
		}
		auto t1 = std::chrono::high_resolution_clock::now();
		secs = std::chrono::duration<double>(t1 - t0).count();
	});
	return secs;
}
// the two reduceField calls of main.cpp:135-147
void ref_grid_reduce_density(void* hh, double h, double rho0)
{
	auto* sdf = static_cast<RefGrid*>(hh)->grid.get();
	auto cell_diag = sdf->cellSize().norm();
	sdf->reduceField(0u, [&](const Eigen::Vector3d&, double v) { return -6.0 * h < v + cell_diag && v - cell_diag < 2.0 * h; });
	sdf->reduceField(1u, [&](const Eigen::Vector3d&, double v) { return 0.0 <= v && v <= 3.0 * rho0; });
}

// TriangleMeshDistance::signed_distance for a batch of points
// (TriangleMeshDistance.h:269-308): distance, triangle id, entity, nearest point.
int ref_signed_distance(void* h, const double* xyz, size_t n, double* dist, int* tri, int* entity,
						double* nearest /*nullable 3n*/)
{
	auto g = static_cast<RefGrid*>(h);
	if (!g->md)
		return -1;
	auto& md = *g->md;
#pragma omp parallel for schedule(dynamic, 256)
	for (long long q = 0; q < (long long)n; ++q)
	{
		std::array<double, 3> p = {{xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]}};
		Result r = md.signed_distance(p);
		dist[q] = r.distance;
		if (tri)
			tri[q] = r.triangle_id;
		if (entity)
			entity[q] = (int)r.nearest_entity;
		if (nearest)
		{
			nearest[3 * q] = r.nearest_point[0];
			nearest[3 * q + 1] = r.nearest_point[1];
			nearest[3 * q + 2] = r.nearest_point[2];
		}
	}
	return 0;
}

// Pseudonormals + BVH produced by the reference's _construct (for pinning the
// restatement's host-side construction).
void ref_md_sizes(void* h, size_t* n_nodes, size_t* n_tris, size_t* n_verts)
{
	auto& md = *static_cast<RefGrid*>(h)->md;
	*n_nodes = md.nodes.size();
	*n_tris = md.triangles.size();
	*n_verts = md.vertices.size();
}
void ref_md_get(void* h, double* pn_tri, double* pn_edge, double* pn_vert, double* node_spheres /*8 per node*/,
				int* node_children /*2 per node*/)
{
	auto& md = *static_cast<RefGrid*>(h)->md;
	for (size_t i = 0; i < md.triangles.size(); ++i)
		for (int d = 0; d < 3; ++d)
		{
			pn_tri[3 * i + d] = md.pseudonormals_triangles[i][d];
			for (int e = 0; e < 3; ++e)
				pn_edge[9 * i + 3 * e + d] = md.pseudonormals_edges[i][e][d];
		}
	for (size_t i = 0; i < md.vertices.size(); ++i)
		for (int d = 0; d < 3; ++d)
			pn_vert[3 * i + d] = md.pseudonormals_vertices[i][d];
	for (size_t i = 0; i < md.nodes.size(); ++i)
	{
		auto& nd = md.nodes[i];
		for (int d = 0; d < 3; ++d)
		{
			node_spheres[8 * i + d] = nd.bv_left.center[d];
			node_spheres[8 * i + 4 + d] = nd.bv_right.center[d];
		}
		node_spheres[8 * i + 3] = nd.bv_left.radius;
		node_spheres[8 * i + 7] = nd.bv_right.radius;
		node_children[2 * i] = nd.left;
		node_children[2 * i + 1] = nd.right;
	}
}

} // extern "C"
