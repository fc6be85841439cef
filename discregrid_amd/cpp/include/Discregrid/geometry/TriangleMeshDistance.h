// Discregrid::TriangleMeshDistance -- API-compatible with the reference's header-only class
// (discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:36-208).  Construction builds the
// angle-weighted pseudonormals exactly like the reference and a flattened BVH of this
// library's own design, and uploads both to the GPU (dg_mesh_create).  Batches of points and whole
// lattices (hand a MeshSDF to CubicLagrangeDiscreteGrid::addFunction) are evaluated on the GPU.  The
// reference's single-point signed_distance / unsigned_distance keep their signatures and contract
// (const, thread safe) and are evaluated on the calling thread against the same BVH with the same
// arithmetic -- an unchanged caller such as the lambda of cmd/generate_sdf/main.cpp:97-101 works,
// at host speed.
//
// Deviations from the reference, on purpose:
//   * errors (query before construct(), empty triangle list) throw std::runtime_error after
//     printing the reference's message instead of calling exit(-1);
//   * the raw-pointer construct() does not reproduce the reference's 3x over-allocation
//     (TriangleMeshDistance.h:235,242 -- phantom degenerate triangles, a bug).
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <limits>
#include <memory>
#include <vector>

#include "../mesh/triangle_mesh.hpp"

namespace Discregrid
{

template <typename FLOAT>
class Vec3r
{
public:
	std::array<FLOAT, 3> v;
	Vec3r() {}
	template <typename FLOAT_I>
	Vec3r(const FLOAT_I& x, const FLOAT_I& y, const FLOAT_I& z)
	{
		v[0] = static_cast<FLOAT>(x);
		v[1] = static_cast<FLOAT>(y);
		v[2] = static_cast<FLOAT>(z);
	}
	template <typename SIZE_T>
	const FLOAT& operator[](const SIZE_T& i) const { return v[i]; }
	template <typename SIZE_T>
	FLOAT& operator[](const SIZE_T& i) { return v[i]; }
	FLOAT dot(const Vec3r& u) const { return v[0] * u[0] + v[1] * u[1] + v[2] * u[2]; }
	Vec3r operator+(const Vec3r& u) const { return Vec3r(v[0] + u[0], v[1] + u[1], v[2] + u[2]); }
	Vec3r operator-(const Vec3r& u) const { return Vec3r(v[0] - u[0], v[1] - u[1], v[2] - u[2]); }
	FLOAT squaredNorm() const { return dot(*this); }
	FLOAT norm() const { return std::sqrt(squaredNorm()); }
};
using Vec3d = Vec3r<double>;

enum class NearestEntity { V0, V1, V2, E01, E12, E02, F };

struct Result
{
	double distance = std::numeric_limits<double>::max();
	Vec3d nearest_point;
	NearestEntity nearest_entity;
	int triangle_id = -1;
};

class TriangleMeshDistance
{
public:
	TriangleMeshDistance() = default;

	template <typename FLOAT, typename INT, typename SIZE_T>
	TriangleMeshDistance(const FLOAT* vertices, const SIZE_T n_vertices, const INT* triangles, const SIZE_T n_triangles)
	{
		construct(vertices, n_vertices, triangles, n_triangles);
	}
	template <typename IndexableVector3double, typename IndexableVector3int>
	TriangleMeshDistance(const std::vector<IndexableVector3double>& vertices,
						 const std::vector<IndexableVector3int>& triangles)
	{
		construct(vertices, triangles);
	}
	TriangleMeshDistance(const TriangleMesh& mesh) { construct(mesh.vertex_data(), mesh.face_data()); }

	template <typename FLOAT, typename INT, typename SIZE_T>
	void construct(const FLOAT* vertices, const SIZE_T n_vertices, const INT* triangles, const SIZE_T n_triangles)
	{
		std::vector<double> v(3 * (std::size_t)n_vertices);
		std::vector<unsigned int> t(3 * (std::size_t)n_triangles);
		for (std::size_t i = 0; i < v.size(); ++i)
			v[i] = (double)vertices[i];
		for (std::size_t i = 0; i < t.size(); ++i)
			t[i] = (unsigned int)triangles[i];
		constructFlat(v, t);
	}
	template <typename IndexableVector3double, typename IndexableVector3int>
	void construct(const std::vector<IndexableVector3double>& vertices, const std::vector<IndexableVector3int>& triangles)
	{
		std::vector<double> v(3 * vertices.size());
		std::vector<unsigned int> t(3 * triangles.size());
		for (std::size_t i = 0; i < vertices.size(); ++i)
			for (int d = 0; d < 3; ++d)
				v[3 * i + d] = (double)vertices[i][d];
		for (std::size_t i = 0; i < triangles.size(); ++i)
			for (int d = 0; d < 3; ++d)
				t[3 * i + d] = (unsigned int)triangles[i][d];
		constructFlat(v, t);
	}

	// single-point queries (reference signatures)
	template <typename IndexableVector3double>
	Result unsigned_distance(const IndexableVector3double& point) const
	{
		return unsigned_distance(std::array<double, 3>{{(double)point[0], (double)point[1], (double)point[2]}});
	}
	Result unsigned_distance(const std::array<double, 3>& point) const;
	template <typename IndexableVector3double>
	Result signed_distance(const IndexableVector3double& point) const
	{
		return signed_distance(std::array<double, 3>{{(double)point[0], (double)point[1], (double)point[2]}});
	}
	Result signed_distance(const std::array<double, 3>& point) const;

	// ---- additions: batched queries (xyz = 3n doubles) ------------------------------------------
	void signed_distance(const double* xyz, std::size_t n, double* distance, int* triangle_id = nullptr,
						 int* nearest_entity = nullptr, double* nearest_point = nullptr) const;
	std::vector<Result> signed_distance(const std::vector<std::array<double, 3>>& points) const;

	bool isConstructed() const { return (bool)m_impl; }
	bool isWatertight() const;
	std::size_t nTriangles() const;
	// opaque dg_mesh* of include/discregrid_hip.h (used by CubicLagrangeDiscreteGrid)
	const void* deviceMesh() const;
	// the primary mesh followed by its replicas on the other devices of DG_DEVICES (if any)
	const std::vector<const void*>& deviceMeshes() const;

private:
	void constructFlat(const std::vector<double>& v, const std::vector<unsigned int>& t);
	struct Impl;
	std::shared_ptr<Impl> m_impl;
};

// The typed SDF functor CubicLagrangeDiscreteGrid::addFunction recognises.  It is an ordinary
// callable, so it can be stored in a DiscreteGrid::ContinuousFunction like the lambda of
// cmd/generate_sdf/main.cpp:97-101 that it replaces:
//     sdf.addFunction(Discregrid::MeshSDF{&md, invert}, true);
struct MeshSDF
{
	const TriangleMeshDistance* distance = nullptr;
	bool invert = false;
	double operator()(Eigen::Vector3d const& x) const
	{
		const double d = distance->signed_distance(x).distance;
		return invert ? -1.0 * d : d;
	}
};

} // namespace Discregrid
