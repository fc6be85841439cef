// Discregrid::DiscreteGrid -- abstract interface of a regular grid that stores one or more
// scalar fields.  API-compatible with the reference class of the same name
// (discregrid/include/Discregrid/discrete_grid.hpp:11-98: same member functions, same typedefs,
// same protected data so that subclasses written against the reference still compile); this is
// an independent implementation.
#pragma once

#include <array>
#include <cstddef>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include <Eigen/Dense>

namespace Discregrid
{

class DiscreteGrid
{
public:
	// ---- callback and index types --------------------------------------------------------
	using MultiIndex = std::array<unsigned int, 3>;
	using CoefficientVector = Eigen::Matrix<double, 32, 1>;
	using ContinuousFunction = std::function<double(Eigen::Vector3d const&)>;
	using SamplePredicate = std::function<bool(Eigen::Vector3d const&)>;          // "sample this node?"
	using Predicate = std::function<bool(Eigen::Vector3d const&, double)>;        // "keep this node?" (reduceField)

	// ---- construction ----------------------------------------------------------------------
	DiscreteGrid() = default; // members stay unset until load()
	DiscreteGrid(Eigen::AlignedBox3d const& domain, std::array<unsigned int, 3> const& resolution);
	virtual ~DiscreteGrid() = default;

	// ---- geometry of the grid --------------------------------------------------------------
	Eigen::AlignedBox3d const& domain() const { return m_domain; }
	std::array<unsigned int, 3> const& resolution() const { return m_resolution; }
	Eigen::Vector3d const& cellSize() const { return m_cell_size; }
	Eigen::Vector3d const& invCellSize() const { return m_inv_cell_size; }

	unsigned int multiToSingleIndex(MultiIndex const& ijk) const; // i fastest
	MultiIndex singleToMultiIndex(unsigned int cell) const;
	Eigen::AlignedBox3d subdomain(MultiIndex const& ijk) const;   // box of one cell
	Eigen::AlignedBox3d subdomain(unsigned int cell) const;

	// ---- fields ------------------------------------------------------------------------------
	// Samples `func` at every node (nodes rejected by `pred` get DBL_MAX) and returns the id of
	// the new field.
	virtual unsigned int addFunction(ContinuousFunction const& func, bool verbose = false,
									 SamplePredicate const& pred = nullptr) = 0;

	// Drops the cells none of whose nodes satisfies `pred` (no-op in the base class).
	virtual void reduceField(unsigned int /*field_id*/, Predicate /*pred*/) {}
	void reduce_field(unsigned int field_id, Predicate pred) { reduceField(field_id, std::move(pred)); } // README spelling

	// ---- evaluation --------------------------------------------------------------------------
	// Value (and optionally gradient) of field `field_id` at x; DBL_MAX if there is no value.
	virtual double interpolate(unsigned int field_id, Eigen::Vector3d const& x,
							   Eigen::Vector3d* gradient = nullptr) const = 0;
	double interpolate(Eigen::Vector3d const& x, Eigen::Vector3d* gradient = nullptr) const
	{
		return interpolate(0u, x, gradient); // field 0
	}

	// Two-step form for callers that evaluate several fields at one point: first the cell, the
	// scaling vector c0 and the 32 shape functions N (and their derivatives dN) at x ...
	virtual bool determineShapeFunctions(unsigned int field_id, Eigen::Vector3d const& x,
										 std::array<unsigned int, 32>& cell, Eigen::Vector3d& c0,
										 Eigen::Matrix<double, 32, 1>& N,
										 Eigen::Matrix<double, 32, 3>* dN = nullptr) const = 0;
	// ... then the value of one field from them.
	virtual double interpolate(unsigned int field_id, Eigen::Vector3d const& x,
							   const std::array<unsigned int, 32>& cell, const Eigen::Vector3d& c0,
							   const Eigen::Matrix<double, 32, 1>& N, Eigen::Vector3d* gradient = nullptr,
							   Eigen::Matrix<double, 32, 3>* dN = nullptr) const = 0;

	// ---- persistence -------------------------------------------------------------------------
	virtual void save(std::string const& filename) const = 0;
	virtual void load(std::string const& filename) = 0;

protected:
	Eigen::AlignedBox3d m_domain;
	std::array<unsigned int, 3> m_resolution;
	Eigen::Vector3d m_cell_size;     // domain diagonal ./ resolution
	Eigen::Vector3d m_inv_cell_size; // 1 ./ cell size
	std::size_t m_n_cells;
	std::size_t m_n_fields;
};

} // namespace Discregrid
