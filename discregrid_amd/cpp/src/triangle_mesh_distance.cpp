// TriangleMeshDistance: a thin owner of a dg_mesh handle (include/discregrid_hip.h).  All
// distance queries run on the GPU; there is no host implementation of the query here.
#include <Discregrid/geometry/TriangleMeshDistance.h>

#include <iostream>
#include <stdexcept>
#include <string>

#include "discregrid_hip.h"

namespace Discregrid
{

struct TriangleMeshDistance::Impl
{
	dg_mesh* mesh = nullptr;
	dg_mesh_info info;
	~Impl() { dg_mesh_destroy(mesh); }
};

namespace
{
[[noreturn]] void fail(const char* what)
{
	throw std::runtime_error(std::string(what) + ": " + dg_last_error());
}
} // namespace

void TriangleMeshDistance::constructFlat(const std::vector<double>& v, const std::vector<unsigned int>& t)
{
	if (t.empty())
	{
		std::cout << "DistanceTriangleMesh error: Empty triangle list." << std::endl;
		throw std::runtime_error("DistanceTriangleMesh error: Empty triangle list.");
	}
	auto impl = std::make_shared<Impl>();
	static_assert(sizeof(unsigned int) == sizeof(uint32_t), "unsigned int must be 32 bits");
	if (dg_mesh_create(v.data(), v.size() / 3, reinterpret_cast<const uint32_t*>(t.data()), t.size() / 3,
					   &impl->mesh) != DG_OK)
		fail("TriangleMeshDistance::construct");
	dg_mesh_get_info(impl->mesh, &impl->info);
	if (impl->info.not_watertight & 1u)
		std::cout << "DistanceTriangleMesh warning: mesh is not watertight. At least one edge found belonging to "
					 "just one triangle."
				  << std::endl;
	if (impl->info.not_watertight & 2u)
		std::cout << "DistanceTriangleMesh warning: mesh is not watertight. At least one edge found belonging to "
					 "more than two triangle."
				  << std::endl;
	m_impl = impl;
}

void TriangleMeshDistance::signed_distance(const double* xyz, std::size_t n, double* distance, int* triangle_id,
											int* nearest_entity, double* nearest_point) const
{
	if (!m_impl)
	{
		std::cout << "DistanceTriangleMesh error: not constructed." << std::endl;
		throw std::runtime_error("DistanceTriangleMesh error: not constructed.");
	}
	static_assert(sizeof(int) == sizeof(int32_t), "int must be 32 bits");
	if (dg_signed_distance(m_impl->mesh, xyz, n, distance, reinterpret_cast<int32_t*>(triangle_id),
						   reinterpret_cast<int32_t*>(nearest_entity), nearest_point) != DG_OK)
		fail("TriangleMeshDistance::signed_distance");
}

Result TriangleMeshDistance::signed_distance(const std::array<double, 3>& point) const
{
	Result r;
	int tri = -1, ent = 0;
	double np[3] = {0, 0, 0};
	signed_distance(point.data(), 1, &r.distance, &tri, &ent, np);
	r.triangle_id = tri;
	r.nearest_entity = static_cast<NearestEntity>(ent);
	r.nearest_point = Vec3d(np[0], np[1], np[2]);
	return r;
}

Result TriangleMeshDistance::unsigned_distance(const std::array<double, 3>& point) const
{
	Result r = signed_distance(point);
	r.distance = std::abs(r.distance);
	return r;
}

std::vector<Result> TriangleMeshDistance::signed_distance(const std::vector<std::array<double, 3>>& points) const
{
	const std::size_t n = points.size();
	std::vector<double> d(n), np(3 * n);
	std::vector<int> tri(n), ent(n);
	signed_distance(n ? points[0].data() : nullptr, n, d.data(), tri.data(), ent.data(), np.data());
	std::vector<Result> out(n);
	for (std::size_t i = 0; i < n; ++i)
	{
		out[i].distance = d[i];
		out[i].triangle_id = tri[i];
		out[i].nearest_entity = static_cast<NearestEntity>(ent[i]);
		out[i].nearest_point = Vec3d(np[3 * i], np[3 * i + 1], np[3 * i + 2]);
	}
	return out;
}

bool TriangleMeshDistance::isWatertight() const { return m_impl && m_impl->info.not_watertight == 0; }
std::size_t TriangleMeshDistance::nTriangles() const { return m_impl ? (std::size_t)m_impl->info.n_triangles : 0; }
const void* TriangleMeshDistance::deviceMesh() const { return m_impl ? m_impl->mesh : nullptr; }

} // namespace Discregrid
