// TriangleMeshDistance: owner of a dg_mesh handle (include/discregrid_hip.h).  Batches of points and
// whole lattices (MeshSDF) run on the GPU; the reference's single-point signed_distance /
// unsigned_distance -- const, thread safe, called per node or per particle by user code -- are
// evaluated on the calling thread by dg_signed_distance_point (same BVH, same arithmetic, same bits).
#include <Discregrid/geometry/TriangleMeshDistance.h>

#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

#include "discregrid_hip.h"

namespace Discregrid
{

struct TriangleMeshDistance::Impl
{
	dg_mesh* mesh = nullptr;          // on the device that was current at construction: all point queries
	std::vector<dg_mesh*> replicas;   // further copies, one per extra device of DG_DEVICES (addFunction only)
	std::vector<const void*> all;     // mesh + replicas
	dg_mesh_info info;
	~Impl()
	{
		dg_mesh_destroy(mesh);
		for (dg_mesh* m : replicas)
			dg_mesh_destroy(m);
	}
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
	impl->all.push_back(impl->mesh);
	// DG_DEVICES = "all" or a comma-separated list of device ordinals: replicate the mesh there so that
	// CubicLagrangeDiscreteGrid::addFunction can spread the node lattice over those GPUs
	const char* env_devices = dg_mesh_device(impl->mesh) < 0 ? nullptr : std::getenv("DG_DEVICES"); // (a host-only handle has no replicas)
	if (const char* env = env_devices)
	{
		int count = 0, current = 0;
		dg_device_count(&count);
		std::vector<int> devices;
		const std::string spec = env;
		std::size_t at = 0;
		while (at <= spec.size())
		{
			const std::size_t comma = spec.find(',', at);
			const std::string tok = spec.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
			if (tok == "all")
				for (int d = 0; d < count; ++d)
					devices.push_back(d);
			else if (!tok.empty())
			{
				char* end = nullptr;
				const long d = std::strtol(tok.c_str(), &end, 10);
				if (end == tok.c_str() || *end != '\0')
					throw std::runtime_error("DG_DEVICES: cannot parse '" + tok + "'");
				devices.push_back((int)d);
			}
			if (comma == std::string::npos)
				break;
			at = comma + 1;
		}
		if (dg_current_device(&current) != DG_OK)
			fail("TriangleMeshDistance::construct");
		bool first = true; // the first mention of the current device is the primary mesh itself
		for (int d : devices)
		{
			if (d < 0 || d >= count)
				throw std::runtime_error("DG_DEVICES names device " + std::to_string(d) + ", but only " +
										 std::to_string(count) + " device(s) are visible");
			if (d == current && first)
			{
				first = false;
				continue;
			}
			dg_mesh* m = nullptr;
			const bool ok = dg_set_device(d) == DG_OK &&
							dg_mesh_create(v.data(), v.size() / 3, reinterpret_cast<const uint32_t*>(t.data()), t.size() / 3, &m) == DG_OK;
			dg_set_device(current);
			if (!ok)
				fail("TriangleMeshDistance::construct (replica)");
			impl->replicas.push_back(m);
			impl->all.push_back(m);
		}
	}
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
	if (dg_mesh_device(m_impl->mesh) < 0)
	{
		// host-only handle (no HIP device / DG_FORCE_CPU=1): the per-point query on every core, same bits as the kernel
		bool ok = true;
#pragma omp parallel for schedule(dynamic, 64) reduction(&& : ok)
		for (long long i = 0; i < (long long)n; ++i)
		{
			int32_t tri = -1, ent = -1;
			double np[3] = {0.0, 0.0, 0.0};
			ok = ok && dg_signed_distance_point(m_impl->mesh, xyz + 3 * i, distance + i, &tri, &ent, np) == DG_OK;
			if (triangle_id)
				triangle_id[i] = tri;
			if (nearest_entity)
				nearest_entity[i] = ent;
			if (nearest_point)
			{
				nearest_point[3 * i] = np[0];
				nearest_point[3 * i + 1] = np[1];
				nearest_point[3 * i + 2] = np[2];
			}
		}
		if (!ok)
			fail("TriangleMeshDistance::signed_distance");
		return;
	}
	if (dg_signed_distance(m_impl->mesh, xyz, n, distance, reinterpret_cast<int32_t*>(triangle_id),
						   reinterpret_cast<int32_t*>(nearest_entity), nearest_point) != DG_OK)
		fail("TriangleMeshDistance::signed_distance");
}

Result TriangleMeshDistance::signed_distance(const std::array<double, 3>& point) const
{
	if (!m_impl)
	{
		std::cout << "DistanceTriangleMesh error: not constructed." << std::endl;
		throw std::runtime_error("DistanceTriangleMesh error: not constructed.");
	}
	Result r;
	int32_t tri = -1, ent = 0;
	double np[3] = {0, 0, 0};
	if (dg_signed_distance_point(m_impl->mesh, point.data(), &r.distance, &tri, &ent, np) != DG_OK)
		fail("TriangleMeshDistance::signed_distance");
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
const std::vector<const void*>& TriangleMeshDistance::deviceMeshes() const
{
	static const std::vector<const void*> none;
	return m_impl ? m_impl->all : none;
}

} // namespace Discregrid
