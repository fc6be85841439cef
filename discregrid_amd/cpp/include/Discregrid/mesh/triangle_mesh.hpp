// Discregrid::TriangleMesh -- the input-side surface of the reference class
// (discregrid/include/Discregrid/mesh/triangle_mesh.hpp:16-106) that the SDF path needs:
// constructors (vectors / raw arrays / OBJ file), vertex and face data access, OBJ export,
// face normals, border-edge count.  The half-edge containers/iterators of the reference
// (faces(), incident_faces(), Halfedge ...) are not part of the hot path and are not provided.
#pragma once

#include <array>
#include <cstddef>
#include <string>
#include <vector>

#include <Eigen/Dense>

namespace Discregrid
{

class TriangleMesh
{
public:
	TriangleMesh(std::vector<Eigen::Vector3d> const& vertices, std::vector<std::array<unsigned int, 3>> const& faces);
	TriangleMesh(double const* vertices, unsigned int const* faces, std::size_t nv, std::size_t nf);
	// OBJ subset of the reference (triangle_mesh.cpp:91-124): only "v " and "f " lines, the index
	// in front of the first '/', the first three indices of a face, 1-based.
	TriangleMesh(std::string const& filename);

	void exportOBJ(std::string const& filename) const;

	std::size_t nFaces() const { return m_faces.size(); }
	std::size_t nVertices() const { return m_vertices.size(); }
	std::size_t nBorderEdges() const { return m_n_border_edges; }

	unsigned int const& faceVertex(unsigned int f, unsigned int i) const { return m_faces[f][i]; }
	unsigned int& faceVertex(unsigned int f, unsigned int i) { return m_faces[f][i]; }
	Eigen::Vector3d const& vertex(unsigned int i) const { return m_vertices[i]; }
	Eigen::Vector3d& vertex(unsigned int i) { return m_vertices[i]; }
	std::array<unsigned int, 3> const& face(unsigned int i) const { return m_faces[i]; }
	std::array<unsigned int, 3>& face(unsigned int i) { return m_faces[i]; }

	// iteration over vertex positions: `for (auto const& x : mesh.vertices())`
	std::vector<Eigen::Vector3d> const& vertices() const { return m_vertices; }

	std::vector<Eigen::Vector3d> const& vertex_data() const { return m_vertices; }
	std::vector<Eigen::Vector3d>& vertex_data() { return m_vertices; }
	std::vector<std::array<unsigned int, 3>> const& face_data() const { return m_faces; }
	std::vector<std::array<unsigned int, 3>>& face_data() { return m_faces; }

	Eigen::Vector3d computeFaceNormal(unsigned int f) const;

private:
	void construct();

	std::vector<Eigen::Vector3d> m_vertices;
	std::vector<std::array<unsigned int, 3>> m_faces;
	std::size_t m_n_border_edges = 0;
};

} // namespace Discregrid
