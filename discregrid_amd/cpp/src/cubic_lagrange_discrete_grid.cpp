// CubicLagrangeDiscreteGrid -- host side of the drop-in (reference:
// discregrid/src/cubic_lagrange_discrete_grid.cpp).  The node-sampling loop of addFunction
// (:806-831) for mesh SDFs and the batched interpolate run on the GPU through
// include/discregrid_hip.h; file format (:678-778), reduceField (:1065-1174), the scalar
// evaluator (:901-1063) and the generic-callback sampling loop are host code.
#include <Discregrid/cubic_lagrange_discrete_grid.hpp>
#include <Discregrid/geometry/TriangleMeshDistance.h>
#include <Discregrid/utility/serialize.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>

#include "discregrid_hip.h"
#include "dg_lattice.h"
#include "dg_force.h"

namespace Discregrid
{

namespace
{
const double kNoValue = std::numeric_limits<double>::max();
const unsigned int kNoCell = std::numeric_limits<unsigned int>::max();

struct ClassLayout
{
	uint32_t D[4][3];
	uint64_t off[5];
};
ClassLayout class_layout(std::array<unsigned int, 3> const& res)
{
	ClassLayout L;
	uint64_t o = 0;
	const uint32_t r[3] = {res[0], res[1], res[2]};
	for (int c = 0; c < 4; ++c)
	{
		dg::class_dims(c, r, L.D[c]);
		L.off[c] = o;
		o += (uint64_t)L.D[c][0] * L.D[c][1] * L.D[c][2];
	}
	L.off[4] = o;
	return L;
}

dg_grid_desc make_desc(Eigen::AlignedBox3d const& dom, std::array<unsigned int, 3> const& res,
					   Eigen::Vector3d const& cell, Eigen::Vector3d const& inv)
{
	dg_grid_desc g;
	std::memset(&g, 0, sizeof(g));
	for (int d = 0; d < 3; ++d)
	{
		g.domain_min[d] = dom.min()[d];
		g.domain_max[d] = dom.max()[d];
		g.resolution[d] = res[d];
		g.cell_size[d] = cell[d];
		g.inv_cell_size[d] = inv[d];
	}
	return g;
}

// Morton key of the reference's zValue()/morton_lut(): dg::reference_z_value (dg_lattice.h), shared with
// the device version of reduceField.
inline uint64_t z_value(const double x[3], double inv_cell) { return dg::reference_z_value(x, inv_cell); }
int env_flag(const char* name, int fallback)
{
	const char* e = std::getenv(name);
	return e ? std::atoi(e) : fallback;
}
} // namespace

// ONE device handle per field: the array K1 / K3 wrote for a field the GPU produced (dg_sdf_sample_field,
// dg_density_map_field), or an upload made once, on first use, for fields that came from a file or a host callback.
// K2 batches, K3 (which ignores a cell-major copy K2 may have built on the handle) and the device reduceField all
// read it.  `pending[f]`: the asynchronous copy into m_nodes[f] has not been collected yet.
struct CubicLagrangeDiscreteGrid::DeviceCache
{
	std::vector<dg_field*> fields;
	std::vector<char> pending;
	std::atomic<unsigned int> n_pending{0u};
	std::mutex mutex; // guards `pending` and the creation of handles (const methods may run concurrently)
	~DeviceCache()
	{
		for (auto f : fields)
			dg_field_destroy(f); // waits for a copy that is still running
	}
};

CubicLagrangeDiscreteGrid::CubicLagrangeDiscreteGrid(std::string const& filename) : m_dev(new DeviceCache)
{
	load(filename);
}

CubicLagrangeDiscreteGrid::CubicLagrangeDiscreteGrid(Eigen::AlignedBox3d const& domain,
													 std::array<unsigned int, 3> const& resolution)
	: DiscreteGrid(domain, resolution), m_dev(new DeviceCache)
{
}

CubicLagrangeDiscreteGrid::~CubicLagrangeDiscreteGrid() = default; // m_dev goes first (declared last): copies end before m_nodes is freed

CubicLagrangeDiscreteGrid::CubicLagrangeDiscreteGrid(CubicLagrangeDiscreteGrid const& other)
	: DiscreteGrid((other.waitForHostData(), static_cast<DiscreteGrid const&>(other))), m_nodes(other.m_nodes), m_cells(other.m_cells),
	  m_cell_map(other.m_cell_map), m_dev(new DeviceCache), m_last_total_s(other.m_last_total_s),
	  m_last_sampling_s(other.m_last_sampling_s), m_last_used_gpu(other.m_last_used_gpu),
	  m_last_reduce_used_gpu(other.m_last_reduce_used_gpu)
{
	m_dev->fields.resize(m_nodes.size(), nullptr);
	m_dev->pending.resize(m_nodes.size(), 0);
}

CubicLagrangeDiscreteGrid& CubicLagrangeDiscreteGrid::operator=(CubicLagrangeDiscreteGrid const& other)
{
	if (this != &other)
	{
		CubicLagrangeDiscreteGrid tmp(other);
		*this = std::move(tmp);
	}
	return *this;
}

CubicLagrangeDiscreteGrid::CubicLagrangeDiscreteGrid(CubicLagrangeDiscreteGrid&& other) noexcept
	: DiscreteGrid(static_cast<DiscreteGrid const&>(other)), m_nodes(std::move(other.m_nodes)), m_cells(std::move(other.m_cells)),
	  m_cell_map(std::move(other.m_cell_map)), m_dev(std::move(other.m_dev)), m_last_total_s(other.m_last_total_s),
	  m_last_sampling_s(other.m_last_sampling_s), m_last_used_gpu(other.m_last_used_gpu),
	  m_last_reduce_used_gpu(other.m_last_reduce_used_gpu)
{
	// the heap buffers of the field vectors moved with them, so copies in flight keep writing to the right place
	other.m_dev.reset(new DeviceCache); // (never null: every member function dereferences it)
	other.m_nodes.clear();
	other.m_cells.clear();
	other.m_cell_map.clear();
	other.m_n_fields = 0;
}

CubicLagrangeDiscreteGrid& CubicLagrangeDiscreteGrid::operator=(CubicLagrangeDiscreteGrid&& other) noexcept
{
	if (this != &other)
	{
		m_dev.reset(); // ends this grid's copies before its vectors are replaced
		DiscreteGrid::operator=(static_cast<DiscreteGrid const&>(other));
		m_nodes = std::move(other.m_nodes);
		m_cells = std::move(other.m_cells);
		m_cell_map = std::move(other.m_cell_map);
		m_dev = std::move(other.m_dev);
		m_last_total_s = other.m_last_total_s;
		m_last_sampling_s = other.m_last_sampling_s;
		m_last_used_gpu = other.m_last_used_gpu;
		m_last_reduce_used_gpu = other.m_last_reduce_used_gpu;
		other.m_dev.reset(new DeviceCache); // (never null: every member function dereferences it)
		other.m_nodes.clear();
		other.m_cells.clear();
		other.m_cell_map.clear();
		other.m_n_fields = 0;
	}
	return *this;
}

void CubicLagrangeDiscreteGrid::hostReady(unsigned int f) const
{
	if (m_dev->n_pending.load(std::memory_order_acquire) == 0u)
		return;
	std::lock_guard<std::mutex> lock(m_dev->mutex);
	if (f >= m_dev->pending.size() || !m_dev->pending[f])
		return;
	const dg_status st = dg_field_host_wait(m_dev->fields[f]);
	m_dev->pending[f] = 0;
	m_dev->n_pending.fetch_sub(1u, std::memory_order_release);
	if (st != DG_OK)
		throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid: copy of a field to the host failed: ") + dg_last_error());
}

void CubicLagrangeDiscreteGrid::waitForHostData() const
{
	for (unsigned int f = 0; f < m_nodes.size(); ++f)
		hostReady(f);
}

void CubicLagrangeDiscreteGrid::adoptDeviceField(unsigned int f, void* handle, bool host_pending)
{
	std::lock_guard<std::mutex> lock(m_dev->mutex);
	if (m_dev->fields.size() <= f)
		m_dev->fields.resize(f + 1, nullptr);
	if (m_dev->pending.size() <= f)
		m_dev->pending.resize(f + 1, 0);
	m_dev->fields[f] = static_cast<dg_field*>(handle);
	if (host_pending)
	{
		m_dev->pending[f] = 1;
		m_dev->n_pending.fetch_add(1u, std::memory_order_release);
	}
}

void* CubicLagrangeDiscreteGrid::deviceField(unsigned int field_id) const
{
	if (field_id >= m_nodes.size())
		throw std::out_of_range("CubicLagrangeDiscreteGrid: no such field");
	std::lock_guard<std::mutex> lock(m_dev->mutex);
	if (m_dev->fields.size() < m_nodes.size())
		m_dev->fields.resize(m_nodes.size(), nullptr);
	if (m_dev->pending.size() < m_nodes.size())
		m_dev->pending.resize(m_nodes.size(), 0);
	dg_field*& f = m_dev->fields[field_id];
	if (f == nullptr)
	{
		// a field that came from a file, a host callback or addNodeData: uploaded once
		const dg_grid_desc g = make_desc(m_domain, m_resolution, m_cell_size, m_inv_cell_size);
		auto const& cells = m_cells[field_id];
		auto const& map = m_cell_map[field_id];
		static_assert(sizeof(unsigned int) == sizeof(uint32_t), "unsigned int must be 32 bits");
		if (dg_field_create(&g, m_nodes[field_id].data(), m_nodes[field_id].size(),
							cells.empty() ? nullptr : reinterpret_cast<const uint32_t*>(cells[0].data()), cells.size(),
							map.empty() ? nullptr : reinterpret_cast<const uint32_t*>(map.data()), &f) != DG_OK)
			throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid (GPU): ") + dg_last_error());
	}
	return f;
}

unsigned int CubicLagrangeDiscreteGrid::nNodesFull() const
{
	return static_cast<unsigned int>(class_layout(m_resolution).off[4]);
}

// cubic_lagrange_discrete_grid.cpp:604-665 through the class decomposition of dg_geom.h
Eigen::Vector3d CubicLagrangeDiscreteGrid::indexToNodePosition(unsigned int l) const
{
	const ClassLayout L = class_layout(m_resolution);
	int c = 0;
	while (c < 3 && l >= L.off[c + 1])
		++c;
	const uint64_t lc = l - L.off[c];
	const uint32_t a = (uint32_t)(lc % L.D[c][0]);
	const uint32_t b = (uint32_t)((lc / L.D[c][0]) % L.D[c][1]);
	const uint32_t s = (uint32_t)(lc / ((uint64_t)L.D[c][0] * L.D[c][1]));
	const double dmin[3] = {m_domain.min()[0], m_domain.min()[1], m_domain.min()[2]};
	const double cell[3] = {m_cell_size[0], m_cell_size[1], m_cell_size[2]};
	double x[3];
	dg::node_position(c, a, b, s, dmin, cell, x);
	return Eigen::Vector3d(x[0], x[1], x[2]);
}

void CubicLagrangeDiscreteGrid::cellRow(unsigned int field_id, unsigned int row, unsigned int out[32]) const
{
	if (!m_cells[field_id].empty())
	{
		std::memcpy(out, m_cells[field_id][row].data(), 32 * sizeof(unsigned int));
		return;
	}
	const unsigned int n01 = m_resolution[0] * m_resolution[1];
	const unsigned int k = row / n01, r = row % n01;
	const uint32_t res[3] = {m_resolution[0], m_resolution[1], m_resolution[2]};
	dg::cell_node_indices(r % m_resolution[0], r / m_resolution[0], k, res, out);
}

void CubicLagrangeDiscreteGrid::materializeCells(unsigned int f)
{
	if (!m_cells[f].empty() || m_n_cells == 0)
		return;
	m_cells[f].resize(m_n_cells);
	m_cell_map[f].resize(m_n_cells);
#pragma omp parallel for schedule(static)
	for (long long l = 0; l < (long long)m_n_cells; ++l)
	{
		const unsigned int n01 = m_resolution[0] * m_resolution[1];
		const unsigned int k = (unsigned int)l / n01, r = (unsigned int)l % n01;
		const uint32_t res[3] = {m_resolution[0], m_resolution[1], m_resolution[2]};
		dg::cell_node_indices(r % m_resolution[0], r / m_resolution[0], k, res, m_cells[f][l].data());
		m_cell_map[f][l] = (unsigned int)l;
	}
}

void CubicLagrangeDiscreteGrid::invalidateDevice(unsigned int f) const
{
	hostReady(f); // a copy in flight reads the handle's array
	std::lock_guard<std::mutex> lock(m_dev->mutex);
	if (f < m_dev->fields.size() && m_dev->fields[f])
	{
		dg_field_destroy(m_dev->fields[f]);
		m_dev->fields[f] = nullptr;
	}
}

// ---------------------------------------------------------------------------------------------
// addFunction
// ---------------------------------------------------------------------------------------------
unsigned int CubicLagrangeDiscreteGrid::addFunction(ContinuousFunction const& func, bool verbose,
													SamplePredicate const& pred)
{
	using clock = std::chrono::high_resolution_clock;
	const auto t_begin = clock::now();
	const unsigned int n_nodes = nNodesFull();

	m_nodes.push_back({});
	auto& coeffs = m_nodes.back();
	coeffs.resize(n_nodes);
	m_cells.push_back({});    // implicit (closed form) until a reduction needs the table
	m_cell_map.push_back({}); // implicit identity

	const auto t_sample = clock::now();
	MeshSDF const* sdf = func.target<MeshSDF>();
	m_last_used_gpu = false;
	// A failure below must leave the grid as it was: the three vectors lose the entry pushed above, the device
	// bookkeeping of the field that never came to be is dropped, and m_n_fields was not touched yet.
	struct Unwind
	{
		CubicLagrangeDiscreteGrid* g;
		bool armed = true;
		~Unwind()
		{
			if (!armed)
				return;
			const std::size_t id = g->m_nodes.size() - 1;
			{
				std::lock_guard<std::mutex> lock(g->m_dev->mutex);
				if (id < g->m_dev->fields.size() && g->m_dev->fields[id])
				{
					dg_field_destroy(g->m_dev->fields[id]); // (joins a copy job that still writes into the host vector)
					g->m_dev->fields[id] = nullptr;
				}
				if (id < g->m_dev->pending.size())
					g->m_dev->pending[id] = 0;
			}
			g->m_nodes.pop_back();
			g->m_cells.pop_back();
			g->m_cell_map.pop_back();
		}
	} unwind{this};
	// The typed functor on a host-only mesh handle (no HIP device, or DG_FORCE_CPU=1): the reference's OpenMP node loop
	// (:806-831) over the per-point query -- the product's own BVH and arithmetic on the calling threads, the bits of
	// the kernel (tests/test_host_api.py).  DG_REQUIRE_GPU=1 refuses instead (a deployment that must not degrade silently).
	const bool host_only_mesh = sdf != nullptr && sdf->distance != nullptr && sdf->distance->deviceMesh() != nullptr &&
								dg_mesh_device(static_cast<const dg_mesh*>(sdf->distance->deviceMesh())) < 0;
	if (host_only_mesh && env_flag("DG_REQUIRE_GPU", 0) != 0)
		throw std::runtime_error("CubicLagrangeDiscreteGrid::addFunction: no HIP device for the MeshSDF functor and DG_REQUIRE_GPU=1");
	if (sdf != nullptr && sdf->distance != nullptr && !host_only_mesh)
	{
		// GPU path.  The predicate is opaque host code: evaluate it into a byte mask first.
		std::vector<uint8_t> mask;
		if (pred)
		{
			mask.resize(n_nodes);
#pragma omp parallel for schedule(static)
			for (long long l = 0; l < (long long)n_nodes; ++l)
				mask[l] = pred(indexToNodePosition((unsigned int)l)) ? 1 : 0;
		}
		const dg_grid_desc g = make_desc(m_domain, m_resolution, m_cell_size, m_inv_cell_size);
		const dg_mesh* mesh = static_cast<const dg_mesh*>(sdf->distance->deviceMesh());
		if (mesh == nullptr)
		{
			std::cout << "DistanceTriangleMesh error: not constructed." << std::endl;
			throw std::runtime_error("DistanceTriangleMesh error: not constructed.");
		}
		// one GPU, or -- when the mesh has replicas on further devices (DG_DEVICES) -- all of them,
		// each with its own copy pipeline writing into `coeffs`
		auto const& all = sdf->distance->deviceMeshes();
		// verbose: the reference's progress line, at most once per second and at the end (:819-829)
		struct ProgressScope
		{
			explicit ProgressScope(bool on)
			{
				if (on)
					dg_set_progress_callback(
						[](uint64_t done, uint64_t total, void*) {
							std::cout << "\r"
									  << "Construction " << std::setw(20)
									  << 100.0 * static_cast<double>(done) / static_cast<double>(total) << "%" << std::flush;
						},
						nullptr);
			}
			~ProgressScope() { dg_set_progress_callback(nullptr, nullptr); }
		} progress_scope(verbose);
		dg_status st;
		const unsigned int id = static_cast<unsigned int>(m_nodes.size() - 1);
		if (all.size() > 1)
			st = dg_sdf_sample_nodes_multi(reinterpret_cast<const dg_mesh* const*>(all.data()), (int)all.size(), &g,
										   sdf->invert ? 1 : 0, 0, n_nodes, pred ? mask.data() : nullptr, coeffs.data());
		else
		{
			// One GPU: the field stays in the device array K1 writes (the handle K2 / K3 / reduceField will read) and the
			// host vector is filled by an asynchronous copy -- this call does not wait for it (see the header).
			dg_field* produced = nullptr;
			// lazy: nobody waits for the host vector inside this call.  dg_sdf_sample_field takes the argument as a HINT: one chunk
			// policy serves both kinds of consumer (seven chunks whose copies run under the following chunks: field complete on
			// the device after 18.5 ms, in the host vector after 21.7 ms at 256^3; DG_FIELD_ONE_LAUNCH=1 is one launch with the
			// copy behind it: 16.3 / 34.6 ms) -- this call cannot know who consumes the field next
			const bool lazy = !verbose && env_flag("DG_LAZY_HOST", 1) != 0;
			st = dg_sdf_sample_field(mesh, &g, sdf->invert ? 1 : 0, pred ? mask.data() : nullptr, coeffs.data(), lazy ? 0 : 1, &produced);
			if (st == DG_OK)
			{
				adoptDeviceField(id, produced, true);
				if (verbose || env_flag("DG_LAZY_HOST", 1) == 0)
				{
					hostReady(id);
					if (verbose)
						std::cout << "\r"
								  << "Construction " << std::setw(20) << 100.0 << "%" << std::flush;
				}
			}
		}
		if (st != DG_OK)
			throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid::addFunction (GPU): ") + dg_last_error());
		m_last_used_gpu = true;
	}
	else
	{
		// Arbitrary callable (or the typed functor without a device: its operator() is the per-point query): can only run
		// on the host (same loop as the reference, :806-831, with a dynamic schedule because the cost per node is very uneven).
		std::atomic_uint counter(0u);
		auto t0 = clock::now();
#pragma omp parallel for schedule(dynamic, 256)
		for (long long l = 0; l < (long long)n_nodes; ++l)
		{
			const Eigen::Vector3d x = indexToNodePosition((unsigned int)l);
			coeffs[l] = (!pred || pred(x)) ? func(x) : kNoValue;
			if (verbose)
			{
				const unsigned int done = ++counter;
				if ((done & 0xFFFu) == 0u || done == n_nodes)
				{
#pragma omp critical(discregrid_progress)
					{
						if (done == n_nodes ||
							std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - t0).count() > 1000)
						{
							t0 = clock::now();
							std::cout << "\r"
									  << "Construction " << std::setw(20)
									  << 100.0 * static_cast<double>(done) / static_cast<double>(n_nodes) << "%"
									  << std::flush;
						}
					}
				}
			}
		}
	}
	const auto t_end = clock::now();
	m_last_sampling_s = std::chrono::duration<double>(t_end - t_sample).count();
	m_last_total_s = std::chrono::duration<double>(t_end - t_begin).count();
	if (verbose)
		std::cout << "\rConstruction took " << std::setw(15)
				  << static_cast<double>(std::chrono::duration_cast<std::chrono::milliseconds>(t_end - t_begin).count()) /
						 1000.0
				  << "s" << std::endl;
	{
		std::lock_guard<std::mutex> lock(m_dev->mutex);
		if (m_dev->fields.size() < m_nodes.size())
			m_dev->fields.resize(m_nodes.size(), nullptr);
		if (m_dev->pending.size() < m_nodes.size())
			m_dev->pending.resize(m_nodes.size(), 0);
	}
	unwind.armed = false;
	return static_cast<unsigned int>(m_n_fields++);
}

unsigned int CubicLagrangeDiscreteGrid::addNodeData(FieldVector coeffs)
{
	if (coeffs.size() != nNodesFull())
		throw std::invalid_argument("CubicLagrangeDiscreteGrid::addNodeData: expected " + std::to_string(nNodesFull()) +
									" coefficients, got " + std::to_string(coeffs.size()));
	m_nodes.push_back(std::move(coeffs));
	m_cells.push_back({});
	m_cell_map.push_back({});
	{
		std::lock_guard<std::mutex> lock(m_dev->mutex);
		m_dev->fields.resize(m_nodes.size(), nullptr);
		m_dev->pending.resize(m_nodes.size(), 0);
	}
	return static_cast<unsigned int>(m_n_fields++);
}

unsigned int CubicLagrangeDiscreteGrid::addDensityMap(unsigned int sdf_field, double support_radius, double rho0,
													  bool band_predicate, bool verbose)
{
	using clock = std::chrono::high_resolution_clock;
	const auto t0 = clock::now();
	if (sdf_field >= m_nodes.size())
		throw std::out_of_range("CubicLagrangeDiscreteGrid::addDensityMap: no such field");
	const unsigned int n_nodes = nNodesFull();
	// the SDF's ONE device handle: the array K1 wrote if addFunction(MeshSDF) produced the field in this process, an
	// upload made once otherwise.  K3 writes the new field's device array; its host vector is filled asynchronously.
	dg_field* sdf = static_cast<dg_field*>(deviceField(sdf_field));
	m_nodes.push_back(FieldVector(n_nodes));
	m_cells.push_back({});
	m_cell_map.push_back({});
	const unsigned int id = static_cast<unsigned int>(m_nodes.size() - 1);
	dg_field* produced = nullptr;
	if (dg_density_map_field(sdf, support_radius, rho0, band_predicate ? 1 : 0, nullptr, m_nodes[id].data(), &produced) != DG_OK)
	{
		m_nodes.pop_back();
		m_cells.pop_back();
		m_cell_map.pop_back();
		throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid::addDensityMap (GPU): ") + dg_last_error());
	}
	adoptDeviceField(id, produced, true);
	if (verbose || env_flag("DG_LAZY_HOST", 1) == 0)
		hostReady(id);
	m_last_used_gpu = true;
	const auto t1 = clock::now();
	m_last_total_s = m_last_sampling_s = std::chrono::duration<double>(t1 - t0).count();
	if (verbose)
		std::cout << "\rConstruction took " << std::setw(15)
				  << static_cast<double>(std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count()) / 1000.0
				  << "s" << std::endl;
	return static_cast<unsigned int>(m_n_fields++);
}

// ---------------------------------------------------------------------------------------------
// evaluation
// ---------------------------------------------------------------------------------------------
namespace
{
dg::FieldDev host_field(Eigen::AlignedBox3d const& dom, std::array<unsigned int, 3> const& res,
						Eigen::Vector3d const& cell, Eigen::Vector3d const& inv, FieldVector const& coeffs,
						std::vector<std::array<unsigned int, 32>> const& cells, std::vector<unsigned int> const& map)
{
	dg::FieldDev F;
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = dom.min()[d];
		F.dmax[d] = dom.max()[d];
		F.cell[d] = cell[d];
		F.inv_cell[d] = inv[d];
		F.res[d] = res[d];
	}
	F.coeffs = coeffs.data();
	F.cells = cells.empty() ? nullptr : cells[0].data();
	F.cell_map = map.empty() ? nullptr : map.data();
	F.cell_major = nullptr;
	F.tile_major = nullptr;
	F.ntile[0] = F.ntile[1] = F.ntile[2] = 0;
	return F;
}
} // namespace

// Single point, on the host (:977-1063).  Same arithmetic, in the same order, as the GPU kernel:
// both instantiate dg::interpolate_point.
double CubicLagrangeDiscreteGrid::interpolate(unsigned int field_id, Eigen::Vector3d const& x,
											  Eigen::Vector3d* gradient) const
{
	hostReady(field_id); // one relaxed load unless a copy is in flight
	const dg::FieldDev F = host_field(m_domain, m_resolution, m_cell_size, m_inv_cell_size, m_nodes[field_id],
									  m_cells[field_id], m_cell_map[field_id]);
	const double p[3] = {x[0], x[1], x[2]};
	double g[3];
	if (!gradient)
		return dg::interpolate_point<false>(F, p, g);
	const double phi = dg::interpolate_point<true>(F, p, g);
	// the reference leaves *gradient untouched when x is outside the domain or in a removed cell
	// (:981-994) and zeroes it when a coefficient is missing (:1050-1054); writing zero in all
	// three cases is a superset of that behaviour
	(*gradient)[0] = g[0];
	(*gradient)[1] = g[1];
	(*gradient)[2] = g[2];
	return phi;
}

void CubicLagrangeDiscreteGrid::interpolate(unsigned int field_id, double const* xyz, std::size_t n, double* phi,
											double* grad) const
{
	if (field_id >= m_nodes.size())
		throw std::out_of_range("CubicLagrangeDiscreteGrid::interpolate: no such field");
	dg_field* f = static_cast<dg_field*>(deviceField(field_id));
	if (dg_interpolate_batch(f, xyz, n, phi, grad) != DG_OK)
		throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid::interpolate (GPU): ") + dg_last_error());
}

bool CubicLagrangeDiscreteGrid::determineShapeFunctions(unsigned int field_id, Eigen::Vector3d const& x,
														std::array<unsigned int, 32>& cell, Eigen::Vector3d& c0,
														Eigen::Matrix<double, 32, 1>& N,
														Eigen::Matrix<double, 32, 3>* dN) const
{
	for (int d = 0; d < 3; ++d)
		if (!(m_domain.min()[d] <= x[d] && x[d] <= m_domain.max()[d]))
			return false;
	unsigned int mi[3];
	for (int d = 0; d < 3; ++d)
	{
		mi[d] = static_cast<unsigned int>((x[d] - m_domain.min()[d]) * m_inv_cell_size[d]);
		if (mi[d] >= m_resolution[d])
			mi[d] = m_resolution[d] - 1;
	}
	const unsigned int ci = multiToSingleIndex({{mi[0], mi[1], mi[2]}});
	const unsigned int cm = m_cell_map[field_id].empty() ? ci : m_cell_map[field_id][ci];
	if (cm == kNoCell)
		return false;
	double xi[3];
	for (int d = 0; d < 3; ++d)
	{
		const double lo = m_domain.min()[d] + static_cast<double>(mi[d]) * m_cell_size[d];
		const double hi = lo + m_cell_size[d];
		const double den = hi - lo;
		c0[d] = 2.0 / den;
		const double c1 = (hi + lo) / den;
		xi[d] = c0[d] * x[d] - c1;
	}
	cellRow(field_id, cm, cell.data());
	double n[32], dx[32], dy[32], dz[32];
	if (dN)
	{
		dg::shape_functions<true>(xi[0], xi[1], xi[2], n, dx, dy, dz);
		for (int j = 0; j < 32; ++j)
		{
			(*dN)(j, 0) = dx[j];
			(*dN)(j, 1) = dy[j];
			(*dN)(j, 2) = dz[j];
		}
	}
	else
		dg::shape_functions<false>(xi[0], xi[1], xi[2], n, dx, dy, dz);
	for (int j = 0; j < 32; ++j)
		N[j] = n[j];
	return true;
}

double CubicLagrangeDiscreteGrid::interpolate(unsigned int field_id, Eigen::Vector3d const& /*xi*/,
											  const std::array<unsigned int, 32>& cell, const Eigen::Vector3d& c0,
											  const Eigen::Matrix<double, 32, 1>& N, Eigen::Vector3d* gradient,
											  Eigen::Matrix<double, 32, 3>* dN) const
{
	hostReady(field_id);
	auto const& coeffs = m_nodes[field_id];
	double phi = 0.0;
	if (!gradient)
	{
		for (unsigned int j = 0; j < 32u; ++j)
		{
			const double c = coeffs[cell[j]];
			if (c == kNoValue)
				return kNoValue;
			phi += c * N[j];
		}
		return phi;
	}
	double g[3] = {0.0, 0.0, 0.0};
	for (unsigned int j = 0; j < 32u; ++j)
	{
		const double c = coeffs[cell[j]];
		if (c == kNoValue)
		{
			gradient->setZero();
			return kNoValue;
		}
		phi += c * N[j];
		g[0] += c * (*dN)(j, 0);
		g[1] += c * (*dN)(j, 1);
		g[2] += c * (*dN)(j, 2);
	}
	for (int d = 0; d < 3; ++d)
		(*gradient)[d] = g[d] * c0[d];
	return phi;
}

// ---------------------------------------------------------------------------------------------
// reduceField (:1065-1174)
// ---------------------------------------------------------------------------------------------
// The device version (dg_reduce_field) for an unreduced field and a value predicate.  Returns false -- and
// leaves the field untouched -- when two surviving nodes share a Morton key: the reference's numbering is
// then whatever libstdc++'s unstable sort makes of the tie, which only the host algorithm below reproduces.
bool CubicLagrangeDiscreteGrid::reduceFieldOnDevice(unsigned int field_id, ValuePredicate const& pred)
{
	// on the field's device handle: no upload if the GPU produced the field or a batch / density map used it before
	dg_reduction* red = nullptr;
	if (dg_reduce_field_device(static_cast<dg_field*>(deviceField(field_id)), pred.closed ? 1 : 0, pred.lo, pred.hi, pred.offset,
							   &red) != DG_OK)
		throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid::reduceField (GPU): ") + dg_last_error());
	struct Guard
	{
		dg_reduction* r;
		~Guard() { dg_reduction_destroy(r); }
	} guard{red};
	uint64_t m = 0, rows = 0;
	int tied = 0;
	dg_reduction_info(red, &m, &rows, &tied);
	if (tied)
		return false;
	FieldVector out(m);
	std::vector<std::array<unsigned int, 32>> cells(rows);
	std::vector<unsigned int> map(m_n_cells);
	static_assert(sizeof(std::array<unsigned int, 32>) == 32 * sizeof(uint32_t), "cell rows must be packed");
	if (dg_reduction_fetch(red, out.data(), rows ? reinterpret_cast<uint32_t*>(cells[0].data()) : nullptr,
						   reinterpret_cast<uint32_t*>(map.data())) != DG_OK)
		throw std::runtime_error(std::string("CubicLagrangeDiscreteGrid::reduceField (GPU): ") + dg_last_error());
	invalidateDevice(field_id); // (waits for a host copy of the unreduced field that is still in flight)
	m_nodes[field_id].swap(out);
	m_cells[field_id].swap(cells);
	m_cell_map[field_id].swap(map);
	// the reduced field's device arrays become its handle: a batched interpolate that follows uploads nothing
	dg_field* reduced = nullptr;
	if (dg_reduction_to_field(red, &reduced) == DG_OK)
		adoptDeviceField(field_id, reduced, false);
	return true;
}

void CubicLagrangeDiscreteGrid::reduceField(unsigned int field_id, Predicate pred)
{
	m_last_reduce_used_gpu = false;
	if (ValuePredicate const* vp = pred.target<ValuePredicate>())
		if (m_cells[field_id].empty() && m_cell_map[field_id].empty() && m_nodes[field_id].size() == nNodesFull() &&
			!dg::force_set("reduce_on_host") && reduceFieldOnDevice(field_id, *vp))
		{
			m_last_reduce_used_gpu = true;
			return;
		}
	using clock = std::chrono::steady_clock;
	const bool timing = dg::force_set("reduce_timing");
	auto tick = [&](const char* what) {
		static thread_local clock::time_point last = clock::now();
		const auto now = clock::now();
		if (timing)
			std::cerr << "reduceField: " << what << " " << std::chrono::duration<double>(now - last).count() << " s" << std::endl;
		last = now;
	};
	tick("start");
	hostReady(field_id);
	materializeCells(field_id);
	invalidateDevice(field_id);
	tick("materialize cells");
	auto& coeffs = m_nodes[field_id];
	auto& cells = m_cells[field_id];
	auto& cell_map = m_cell_map[field_id];
	const std::size_t n = coeffs.size();

	// nodes that satisfy the predicate (user code: called serially, as the reference does)
	std::vector<char> keep(n);
	for (std::size_t l = 0; l < n; ++l)
		keep[l] = pred(indexToNodePosition((unsigned int)l), coeffs[l]) && coeffs[l] != kNoValue;
	tick("predicate + keys");
	// keep a cell if any of its 32 nodes is kept
	const std::vector<std::array<unsigned int, 32>> old_cells = cells;
	cells.clear();
	cell_map.assign(m_n_cells, 0u);
	std::iota(cell_map.begin(), cell_map.end(), 0u);
	for (std::size_t i = 0; i < old_cells.size(); ++i)
	{
		bool any = false;
		for (unsigned int v : old_cells[i])
			any = any || keep[v];
		if (any)
		{
			cells.push_back(old_cells[i]);
			cell_map[i] = static_cast<unsigned int>(cells.size() - 1);
		}
		else
			cell_map[i] = kNoCell;
	}

	tick("cells");
	// nodes referenced by a surviving cell stay; the others are removed by moving the current
	// last node into the hole, scanning from the back (:1131-1150) -- simulated on a permutation
	std::vector<char> used(n, 0);
	for (auto const& cell : cells)
		for (unsigned int v : cell)
			used[v] = 1;
	std::vector<unsigned int> at(n); // at[pos] = original node stored at pos
	std::iota(at.begin(), at.end(), 0u);
	long long last = (long long)n - 1;
	for (long long i = (long long)n - 1; i >= 0; --i)
		if (!used[i])
		{
			std::swap(at[i], at[last]);
			--last;
		}
	const std::size_t m = (std::size_t)(last + 1);

	tick("compaction");
	// Morton sort of the survivors: the reference's std::sort on the reference's keys, over the same
	// sequence (so that even tied keys -- possible only on strongly anisotropic lattices -- end up
	// in the order libstdc++ gives them there).  Keys are computed for the survivors only.
	const double zscale = 4.0 * std::min(std::min(m_inv_cell_size[0], m_inv_cell_size[1]), m_inv_cell_size[2]);
	std::vector<uint64_t> z(m);
#pragma omp parallel for schedule(static)
	for (long long a = 0; a < (long long)m; ++a)
	{
		const Eigen::Vector3d x = indexToNodePosition(at[(std::size_t)a]);
		const double p[3] = {x[0], x[1], x[2]};
		z[(std::size_t)a] = z_value(p, zscale);
	}
	std::vector<unsigned int> order(m);
	std::iota(order.begin(), order.end(), 0u);
	std::sort(order.begin(), order.end(), [&](unsigned int a, unsigned int b) { return z[a] < z[b]; });

	tick("sort");
	std::vector<unsigned int> new_id(n, kNoCell);
	FieldVector out(m);
	for (std::size_t i = 0; i < m; ++i)
	{
		out[i] = coeffs[at[order[i]]];
		new_id[at[order[i]]] = (unsigned int)i;
	}
	for (auto& cell : cells)
		for (auto& v : cell)
			v = new_id[v];
	coeffs.swap(out);
	tick("renumber");
}

void CubicLagrangeDiscreteGrid::forEachCell(
	unsigned int /*field_id*/, std::function<void(unsigned int, Eigen::AlignedBox3d const&, unsigned int)> const& cb) const
{
	const unsigned int n = m_resolution[0] * m_resolution[1] * m_resolution[2];
	for (unsigned int i = 0; i < n; ++i)
		cb(i, subdomain(i), 0);
}

// ---------------------------------------------------------------------------------------------
// save / load (:678-778): packed little-endian, no magic:
//   domain 6 f64 | resolution 3 u32 | cell_size 3 f64 | inv_cell_size 3 f64 | n_cells u64 | n_fields u64 |
//   u64 F { u64 n ; n f64 } | u64 F { u64 n ; n x 32 u32 } | u64 F { u64 n ; n u32 }
// Written with bulk I/O; implicit cell tables are generated on the fly.
// ---------------------------------------------------------------------------------------------
void CubicLagrangeDiscreteGrid::save(std::string const& filename) const
{
	waitForHostData();
	std::ofstream out(filename, std::ios::binary);
	std::streambuf& b = *out.rdbuf();
	for (int d = 0; d < 3; ++d)
		serialize::write(b, m_domain.min()[d]);
	for (int d = 0; d < 3; ++d)
		serialize::write(b, m_domain.max()[d]);
	serialize::write(b, m_resolution);
	for (int d = 0; d < 3; ++d)
		serialize::write(b, m_cell_size[d]);
	for (int d = 0; d < 3; ++d)
		serialize::write(b, m_inv_cell_size[d]);
	serialize::write(b, m_n_cells);
	serialize::write(b, m_n_fields);

	serialize::write(b, m_nodes.size());
	for (auto const& nodes : m_nodes)
	{
		serialize::write(b, nodes.size());
		b.sputn(reinterpret_cast<const char*>(nodes.data()), (std::streamsize)(nodes.size() * sizeof(double)));
	}
	serialize::write(b, m_cells.size());
	for (std::size_t f = 0; f < m_cells.size(); ++f)
	{
		if (!m_cells[f].empty() || m_n_cells == 0 || !m_cell_map[f].empty())
		{
			serialize::write(b, m_cells[f].size());
			if (!m_cells[f].empty())
				b.sputn(reinterpret_cast<const char*>(m_cells[f][0].data()),
						(std::streamsize)(m_cells[f].size() * 32 * sizeof(unsigned int)));
			continue;
		}
		serialize::write(b, m_n_cells);
		std::vector<unsigned int> chunk;
		const std::size_t step = 1u << 16;
		for (std::size_t l0 = 0; l0 < m_n_cells; l0 += step)
		{
			const std::size_t l1 = std::min(m_n_cells, l0 + step);
			chunk.resize((l1 - l0) * 32);
#pragma omp parallel for schedule(static)
			for (long long l = (long long)l0; l < (long long)l1; ++l)
				cellRow((unsigned int)f, (unsigned int)l, &chunk[(l - l0) * 32]);
			b.sputn(reinterpret_cast<const char*>(chunk.data()), (std::streamsize)(chunk.size() * sizeof(unsigned int)));
		}
	}
	serialize::write(b, m_cell_map.size());
	for (std::size_t f = 0; f < m_cell_map.size(); ++f)
	{
		if (!m_cell_map[f].empty() || m_n_cells == 0)
		{
			serialize::write(b, m_cell_map[f].size());
			b.sputn(reinterpret_cast<const char*>(m_cell_map[f].data()),
					(std::streamsize)(m_cell_map[f].size() * sizeof(unsigned int)));
			continue;
		}
		serialize::write(b, m_n_cells);
		std::vector<unsigned int> iota(std::min<std::size_t>(m_n_cells, 1u << 20));
		for (std::size_t l0 = 0; l0 < m_n_cells; l0 += iota.size())
		{
			const std::size_t cnt = std::min(iota.size(), m_n_cells - l0);
			for (std::size_t i = 0; i < cnt; ++i)
				iota[i] = (unsigned int)(l0 + i);
			b.sputn(reinterpret_cast<const char*>(iota.data()), (std::streamsize)(cnt * sizeof(unsigned int)));
		}
	}
	out.close();
}

void CubicLagrangeDiscreteGrid::load(std::string const& filename)
{
	std::ifstream in(filename, std::ios::binary);
	if (!in.good())
	{
		std::cerr << "ERROR: Discrete grid can not be loaded. Input file does not exist!" << std::endl;
		return;
	}
	m_dev.reset(new DeviceCache); // ends copies into the vectors that are about to be replaced
	std::streambuf& b = *in.rdbuf();
	Eigen::Vector3d lo, hi;
	for (int d = 0; d < 3; ++d)
		serialize::read(b, lo[d]);
	for (int d = 0; d < 3; ++d)
		serialize::read(b, hi[d]);
	m_domain = Eigen::AlignedBox3d(lo, hi);
	serialize::read(b, m_resolution);
	for (int d = 0; d < 3; ++d)
		serialize::read(b, m_cell_size[d]);
	for (int d = 0; d < 3; ++d)
		serialize::read(b, m_inv_cell_size[d]);
	serialize::read(b, m_n_cells);
	serialize::read(b, m_n_fields);

	std::size_t nf = 0;
	serialize::read(b, nf);
	m_nodes.assign(nf, {});
	for (auto& nodes : m_nodes)
	{
		std::size_t n = 0;
		serialize::read(b, n);
		nodes.resize(n);
		b.sgetn(reinterpret_cast<char*>(nodes.data()), (std::streamsize)(n * sizeof(double)));
	}
	serialize::read(b, nf);
	m_cells.assign(nf, {});
	for (auto& cells : m_cells)
	{
		std::size_t n = 0;
		serialize::read(b, n);
		cells.resize(n);
		if (n)
			b.sgetn(reinterpret_cast<char*>(cells[0].data()), (std::streamsize)(n * 32 * sizeof(unsigned int)));
	}
	serialize::read(b, nf);
	m_cell_map.assign(nf, {});
	for (auto& map : m_cell_map)
	{
		std::size_t n = 0;
		serialize::read(b, n);
		map.resize(n);
		b.sgetn(reinterpret_cast<char*>(map.data()), (std::streamsize)(n * sizeof(unsigned int)));
	}
	in.close();

	// A field whose table is the identity/closed-form one goes back to the implicit form (saves
	// 132 bytes per cell of host memory and lets the GPU evaluator compute the indices).
	const unsigned int full = nNodesFull();
	for (std::size_t f = 0; f < m_nodes.size() && f < m_cells.size() && f < m_cell_map.size(); ++f)
	{
		if (m_nodes[f].size() != full || m_cells[f].size() != m_n_cells || m_cell_map[f].size() != m_n_cells)
			continue;
		bool implicit = true;
		unsigned int row[32];
		const uint32_t res[3] = {m_resolution[0], m_resolution[1], m_resolution[2]};
		for (std::size_t l = 0; l < m_n_cells && implicit; ++l)
		{
			const unsigned int n01 = m_resolution[0] * m_resolution[1];
			const unsigned int k = (unsigned int)l / n01, r = (unsigned int)l % n01;
			dg::cell_node_indices(r % m_resolution[0], r / m_resolution[0], k, res, row);
			implicit = m_cell_map[f][l] == (unsigned int)l &&
					   std::memcmp(row, m_cells[f][l].data(), sizeof(row)) == 0;
		}
		if (implicit)
		{
			std::vector<std::array<unsigned int, 32>>().swap(m_cells[f]);
			std::vector<unsigned int>().swap(m_cell_map[f]);
		}
	}
	m_dev->fields.resize(m_nodes.size(), nullptr);
	m_dev->pending.resize(m_nodes.size(), 0);
}

} // namespace Discregrid
