// Discregrid::CubicLagrangeDiscreteGrid -- API-compatible with the reference class
// (discregrid/include/Discregrid/cubic_lagrange_discrete_grid.hpp:9-73) with the node-sampling
// and batched-evaluation hot paths running on an MI355X through the C ABI of
// include/discregrid_hip.h.
//
// What is the same: every public signature, the node / cell numbering, the on-disk format
// (save/load are byte-compatible with the reference's .cdf/.cdm files), the DBL_MAX "no value"
// sentinel, reduceField semantics.
//
// What is new (additive):
//   * addFunction() recognises a Discregrid::MeshSDF functor inside the std::function
//     (func.target<MeshSDF>()) and then samples all nodes on the GPU; any other callable is
//     opaque host code and is evaluated by an OpenMP loop exactly like the reference does.
//   * interpolate(field, points, n, phi, grad): batched evaluation on the GPU.
//   * The 32-index cell table (2.1 GB at 256^3 in the reference) is implicit until something
//     needs it materialised (reduceField, or a file that was saved after a reduction).
//   * ONE device-resident copy per field.  The reference has one m_nodes vector per field (:69); here a field the
//     GPU produced (addFunction(MeshSDF), addDensityMap) stays in the device array its kernel wrote, and batched
//     interpolate, addDensityMap and reduceField(ValuePredicate) read that array -- nothing is uploaded again.
//     The host vector is filled by an asynchronous copy that addFunction starts and does NOT wait for: the call
//     returns once the work is enqueued, and the first HOST reader of the field (interpolate(x), save,
//     reduceField with an opaque predicate, nodeData, a copy of the grid) waits for the copy.  DG_LAZY_HOST=0
//     makes addFunction / addDensityMap wait before they return.
#pragma once

#include "discrete_grid.hpp"
#include "utility/field_allocator.hpp"

#include <memory>

namespace Discregrid
{

// A reduceField predicate on the node value alone, of the two forms the reference's GenerateDensityMap
// uses (cmd/generate_density_map/main.cpp:138-145):
//     band(lo, hi, offset):  lo < v + offset && v - offset < hi          range(lo, hi):  lo <= v && v <= hi
// It is an ordinary callable -- it converts to DiscreteGrid::Predicate and behaves like the lambda it
// replaces -- but CubicLagrangeDiscreteGrid::reduceField recognises it (pred.target<ValuePredicate>()) and
// then runs the selection, compaction and Morton renumbering on the GPU.
struct ValuePredicate
{
	double lo = 0.0, hi = 0.0, offset = 0.0;
	bool closed = false;
	static ValuePredicate band(double lo, double hi, double offset) { return ValuePredicate{lo, hi, offset, false}; }
	static ValuePredicate range(double lo, double hi) { return ValuePredicate{lo, hi, 0.0, true}; }
	bool operator()(Eigen::Vector3d const&, double v) const
	{
		return closed ? (lo <= v && v <= hi) : (lo < v + offset && v - offset < hi);
	}
};

class CubicLagrangeDiscreteGrid : public DiscreteGrid
{
public:
	CubicLagrangeDiscreteGrid(std::string const& filename);
	CubicLagrangeDiscreteGrid(Eigen::AlignedBox3d const& domain, std::array<unsigned int, 3> const& resolution);
	~CubicLagrangeDiscreteGrid() override;
	// copyable and movable like the reference class (implicit members there, :9-73): a copy gets the host state
	// (waiting for copies still in flight) and fresh, empty device handles of its own
	CubicLagrangeDiscreteGrid(CubicLagrangeDiscreteGrid const& other);
	CubicLagrangeDiscreteGrid& operator=(CubicLagrangeDiscreteGrid const& other);
	CubicLagrangeDiscreteGrid(CubicLagrangeDiscreteGrid&& other) noexcept;
	CubicLagrangeDiscreteGrid& operator=(CubicLagrangeDiscreteGrid&& other) noexcept;

	void save(std::string const& filename) const override;
	void load(std::string const& filename) override;

	unsigned int addFunction(ContinuousFunction const& func, bool verbose = false,
							 SamplePredicate const& pred = nullptr) override;

	std::size_t nCells() const { return m_n_cells; }

	double interpolate(unsigned int field_id, Eigen::Vector3d const& xi,
					   Eigen::Vector3d* gradient = nullptr) const override;
	using DiscreteGrid::interpolate;

	bool determineShapeFunctions(unsigned int field_id, Eigen::Vector3d const& x, std::array<unsigned int, 32>& cell,
								 Eigen::Vector3d& c0, Eigen::Matrix<double, 32, 1>& N,
								 Eigen::Matrix<double, 32, 3>* dN = nullptr) const override;

	double interpolate(unsigned int field_id, Eigen::Vector3d const& xi, const std::array<unsigned int, 32>& cell,
					   const Eigen::Vector3d& c0, const Eigen::Matrix<double, 32, 1>& N,
					   Eigen::Vector3d* gradient = nullptr, Eigen::Matrix<double, 32, 3>* dN = nullptr) const override;

	void reduceField(unsigned int field_id, Predicate pred) override;

	void forEachCell(unsigned int field_id,
					 std::function<void(unsigned int, Eigen::AlignedBox3d const&, unsigned int)> const& cb) const;

	// ---- additions ---------------------------------------------------------------------------
	// Batched evaluation on the GPU (no CPU fallback: throws std::runtime_error if the HIP
	// library reports an error).  xyz: 3n doubles; phi: n; grad: 3n or nullptr.  Same
	// semantics per point as interpolate(field_id, x, gradient); grad is zero where phi is
	// DBL_MAX.
	void interpolate(unsigned int field_id, double const* xyz, std::size_t n, double* phi,
					 double* grad = nullptr) const;

	// The SPH boundary density map of the reference's GenerateDensityMap tool
	// (cmd/generate_density_map/main.cpp:83-133) as ONE call, evaluated on the GPU: appends the
	// field  rho0 * integral_{[-h,h]^3} gamma(x + xi) W(xi) dxi  (gamma from field `sdf_field`, W the
	// cubic spline kernel of support radius h, 16^3 Gauss points) sampled on this grid's lattice.
	// band_predicate applies the tool's node predicate (nodes outside the band around the surface
	// get DBL_MAX); pass false for the tool's --no-reduction behaviour.  Returns the new field id.
	unsigned int addDensityMap(unsigned int sdf_field, double support_radius, double rho0, bool band_predicate = true,
							   bool verbose = false);

	// Adopts a coefficient vector that was computed elsewhere for THIS grid's lattice (full, unreduced
	// node order [V | X | Y | Z], nNodes() values) -- e.g. by dg_sdf_sample_allgather_device on several
	// GPUs -- as a new field; returns its id.  Throws std::invalid_argument on a size mismatch.
	unsigned int addNodeData(FieldVector coeffs);
	unsigned int nNodes() const { return nNodesFull(); }

	std::size_t nFields() const { return m_n_fields; }
	FieldVector const& nodeData(unsigned int field_id) const
	{
		hostReady(field_id);
		return m_nodes[field_id];
	}
	// Blocks until the host vector of the field (of every field) is complete; returns at once if it already is.
	void waitForHostData(unsigned int field_id) const { hostReady(field_id); }
	void waitForHostData() const;
	// The field's device handle (dg_field* of include/discregrid_hip.h, created on first use), for callers that
	// chain further device work; nullptr never (throws if the GPU call fails).
	void* deviceField(unsigned int field_id) const;
	// Seconds spent in the last addFunction call (whole call) and in its node-sampling stage.
	// wall time of the last addFunction CALL and of its sampling part.  With the typed MeshSDF functor in the default lazy mode
	// the call returns when the work is enqueued: these exclude the kernels and the copy into the host vector
	// (waitForHostData() / the first host reader waits for that; DG_LAZY_HOST=0 makes the call itself wait, as the reference does)
	double lastAddFunctionSeconds() const { return m_last_total_s; }
	double lastSamplingSeconds() const { return m_last_sampling_s; }
	bool lastAddFunctionUsedGpu() const { return m_last_used_gpu; }
	bool lastReduceFieldUsedGpu() const { return m_last_reduce_used_gpu; }

private:
	Eigen::Vector3d indexToNodePosition(unsigned int l) const;
	unsigned int nNodesFull() const;
	void cellRow(unsigned int field_id, unsigned int cell_row, unsigned int out[32]) const;
	void materializeCells(unsigned int field_id);
	void invalidateDevice(unsigned int field_id) const;
	void hostReady(unsigned int field_id) const; // waits for the field's host copy if one is in flight
	void adoptDeviceField(unsigned int field_id, void* handle, bool host_pending);

private:
	// one FieldVector per field: a std::vector<double> whose allocator skips the zero fill and
	// hands out huge-page-advised memory (utility/field_allocator.hpp)
	std::vector<FieldVector> m_nodes;
	// m_cells[f] / m_cell_map[f] are EMPTY while field f is unreduced (identity map, closed-form
	// rows); they hold the reference's tables once the field has been reduced or loaded reduced.
	std::vector<std::vector<std::array<unsigned int, 32>>> m_cells;
	std::vector<std::vector<unsigned int>> m_cell_map;

	struct DeviceCache;
	mutable std::unique_ptr<DeviceCache> m_dev;
	double m_last_total_s = 0.0, m_last_sampling_s = 0.0;
	bool m_last_used_gpu = false;
	bool m_last_reduce_used_gpu = false;
	bool reduceFieldOnDevice(unsigned int field_id, ValuePredicate const& pred);
};

} // namespace Discregrid
