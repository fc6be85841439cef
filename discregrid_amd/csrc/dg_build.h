// dg_build.h -- host-side construction of the device data set for one triangle mesh:
//   * angle-weighted pseudonormals, computed exactly as the reference does
//     (discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:359-420) -- the sign of the
//     distance depends on them bit for bit;
//   * triangle packets (dg_geom.h) in BVH leaf order;
//   * a flattened BVH of this library's own design, stored as sibling-pair records (the reference's bounding-sphere tree
//     is not reproduced: traversal order and bounding volumes only prune).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "dg_geom.h"

namespace dg
{

struct MeshBuild
{
	std::vector<PairRec> pairs;     // node pairs; an inner node's record holds its two children
	std::vector<PairRec> tri_pairs; // triangle bound pairs: position t -> tri_pairs[t / 2], side t & 1
	std::vector<TriPacket> tris;    // one per position (leaf order; padding slots have tri_id = -1)
	std::vector<TriApproxPair> tri_approx; // float filter data of the triangles: position t -> tri_approx[t / 2], side t & 1
	std::vector<double> pn;         // kPnSlots * 3 doubles per position
	int32_t root_info = 0;          // info word of the root (dg_geom.h)
	std::vector<int32_t> sub_roots; // info words of <= kSubtrees disjoint subtrees covering the tree (level-order cut)
	double origin[3];               // bounds are relative to this point
	float mesh_l1 = 0;              // max over vertices of |v - origin|_1, rounded up
	double mean_edge = 0;           // edge of the equilateral triangle with the mesh's mean triangle area (kernel choice, dg_capi.cpp)
	uint32_t depth = 0;
	uint32_t not_watertight = 0;    // bit0 single edge, bit1 edge shared by > 2 faces
	uint64_t n_vertices = 0, n_triangles = 0;
};

// max_leaf: triangles per leaf (1..16).  Returns false on invalid input (no triangles, index
// out of range).
bool build_mesh(const double* verts, size_t n_vertices, const uint32_t* tris, size_t n_triangles, int max_leaf,
				MeshBuild& out);

} // namespace dg
