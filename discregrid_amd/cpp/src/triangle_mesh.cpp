// TriangleMesh: constructors, OBJ subset reader/writer (reference:
// discregrid/src/mesh/triangle_mesh.cpp:70-148, 207-215) and the closedness diagnostic.
#include <Discregrid/mesh/triangle_mesh.hpp>

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_map>

namespace Discregrid
{

TriangleMesh::TriangleMesh(std::vector<Eigen::Vector3d> const& vertices,
						   std::vector<std::array<unsigned int, 3>> const& faces)
	: m_vertices(vertices), m_faces(faces)
{
	construct();
}

TriangleMesh::TriangleMesh(double const* vertices, unsigned int const* faces, std::size_t nv, std::size_t nf)
	: m_vertices(nv), m_faces(nf)
{
	for (std::size_t i = 0; i < nv; ++i)
		m_vertices[i] = Eigen::Vector3d(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
	for (std::size_t i = 0; i < nf; ++i)
		m_faces[i] = {{faces[3 * i], faces[3 * i + 1], faces[3 * i + 2]}};
	construct();
}

TriangleMesh::TriangleMesh(std::string const& path)
{
	std::ifstream in(path, std::ios::in);
	if (!in)
	{
		std::cerr << "Cannot open " << path << std::endl;
		return;
	}
	std::string line;
	while (std::getline(in, line))
	{
		if (line.compare(0, 2, "v ") == 0)
		{
			std::istringstream s(line.substr(2));
			Eigen::Vector3d v;
			s >> v[0];
			s >> v[1];
			s >> v[2];
			m_vertices.push_back(v);
		}
		else if (line.compare(0, 2, "f ") == 0)
		{
			std::istringstream s(line.substr(2));
			std::array<unsigned int, 3> f;
			for (unsigned int j = 0; j < 3; ++j)
			{
				std::string tok;
				s >> tok;
				tok = tok.substr(0, tok.find_first_of('/'));
				f[j] = static_cast<unsigned int>(std::stoi(tok) - 1);
			}
			m_faces.push_back(f);
		}
	}
	construct();
}

void TriangleMesh::exportOBJ(std::string const& filename) const
{
	std::ofstream out(filename.c_str());
	out << "# Created by discregrid_amd\n";
	out << "g default\n";
	for (auto const& v : m_vertices)
		out << "v " << v[0] << " " << v[1] << " " << v[2] << "\n";
	for (auto const& f : m_faces)
		out << "f " << (f[0] + 1) << " " << (f[1] + 1) << " " << (f[2] + 1) << "\n";
}

Eigen::Vector3d TriangleMesh::computeFaceNormal(unsigned int f) const
{
	Eigen::Vector3d const& x0 = vertex(faceVertex(f, 0));
	Eigen::Vector3d const& x1 = vertex(faceVertex(f, 1));
	Eigen::Vector3d const& x2 = vertex(faceVertex(f, 2));
	return (x1 - x0).cross(x2 - x0).normalized();
}

// Counts directed edges without an oppositely directed partner (what the reference's half-edge
// construction calls border edges) and prints its warning when the mesh is not closed.
void TriangleMesh::construct()
{
	std::unordered_map<std::uint64_t, int> directed;
	directed.reserve(m_faces.size() * 3);
	const std::uint64_t nv = m_vertices.size() + 1;
	for (auto const& f : m_faces)
		for (int e = 0; e < 3; ++e)
			directed[(std::uint64_t)f[e] * nv + f[(e + 1) % 3]] += 1;
	std::size_t border = 0;
	for (auto const& kv : directed)
	{
		const std::uint64_t a = kv.first / nv, b = kv.first % nv;
		if (directed.find(b * nv + a) == directed.end())
			border += (std::size_t)kv.second;
	}
	m_n_border_edges = border;
	if (border != 0)
		std::cout << std::endl << "WARNING: Mesh not closed!" << std::endl;
}

} // namespace Discregrid
