// tests/cpp/host_api_driver.cpp -- exercises the Discregrid-compatible C++ host API
// (discregrid_amd/cpp) the way a downstream user (PBD / SPlisHSPlasH / the reference's CLIs)
// would; the pytest files compare the artefacts it writes with the oracle / golden files.
//
//   host_api_driver roundtrip  in.cdf out.cdf            load -> save
//   host_api_driver poly       out.cdf                    generic (host callback) addFunction
//   host_api_driver reduce     in.cdf bound out.cdf       load -> reduceField(|v| < bound) -> save
//   host_api_driver reduceband in.cdf lo hi out.cdf gpu|host  the same through the typed ValuePredicate (GPU path)
//   host_api_driver eval       in.cdf pts.bin out.bin     scalar interpolate (value + gradient)
//   host_api_driver evalsplit  in.cdf pts.bin out.bin     determineShapeFunctions + split interpolate
//   host_api_driver gpu        mesh.obj pts.bin out.bin   GPU: MeshSDF addFunction (with a predicate
//                                                         on field 1), batched vs scalar interpolate,
//                                                         batched vs single signed_distance
#include <Discregrid/All>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

using namespace Discregrid;

static std::vector<double> read_doubles(const std::string& p)
{
	std::ifstream in(p, std::ios::binary | std::ios::ate);
	std::vector<double> v((size_t)in.tellg() / sizeof(double));
	in.seekg(0);
	in.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(double)));
	return v;
}
static void write_doubles(const std::string& p, const std::vector<double>& v)
{
	std::ofstream out(p, std::ios::binary);
	out.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(double)));
}

int main(int argc, char** argv)
{
	if (argc < 2)
		return 2;
	const std::string cmd = argv[1];
	if (cmd == "roundtrip" && argc == 4)
	{
		CubicLagrangeDiscreteGrid g{std::string(argv[2])};
		g.save(argv[3]);
		return 0;
	}
	if (cmd == "poly" && argc == 3)
	{
		Eigen::AlignedBox3d dom(Eigen::Vector3d(-1.0, -0.5, 0.0), Eigen::Vector3d(1.5, 0.75, 2.0));
		CubicLagrangeDiscreteGrid g(dom, {{3, 4, 2}});
		auto f = [](Eigen::Vector3d const& p) {
			const double x = p[0], y = p[1], z = p[2];
			return 1 + x - 2 * y + 0.5 * z + x * y - y * z + x * x * z - 0.3 * y * y * y + 0.7 * x * y * z;
		};
		unsigned id0 = g.addFunction(f);
		unsigned id1 = g.addFunction(f, false, [](Eigen::Vector3d const& p) { return p[0] < 0.25; });
		if (id0 != 0 || id1 != 1 || g.lastAddFunctionUsedGpu())
			return 3;
		g.save(argv[2]);
		return 0;
	}
	if (cmd == "reduce" && argc == 5)
	{
		CubicLagrangeDiscreteGrid g{std::string(argv[2])};
		const double bound = std::stod(argv[3]);
		g.reduceField(0u, [bound](Eigen::Vector3d const&, double v) { return std::abs(v) < bound; });
		g.save(argv[4]);
		return 0;
	}
	if (cmd == "reduceband" && argc == 7)
	{
		// typed predicate |v| < bound as band(-bound, bound, 0): GPU path where the node order is unique;
		// prints which path ran.  argv[6] = "host" forces the host algorithm through the same typed predicate.
		CubicLagrangeDiscreteGrid g{std::string(argv[2])};
		const double lo = std::stod(argv[3]), hi = std::stod(argv[4]);
		if (std::string(argv[6]) == "host")
			setenv("DG_REDUCE_ON_HOST", "1", 1);
		g.reduceField(0u, ValuePredicate::band(lo, hi, 0.0));
		std::printf("%s\n", g.lastReduceFieldUsedGpu() ? "gpu" : "host");
		g.save(argv[5]);
		return 0;
	}
	if ((cmd == "eval" || cmd == "evalsplit") && argc == 5)
	{
		CubicLagrangeDiscreteGrid g{std::string(argv[2])};
		const std::vector<double> pts = read_doubles(argv[3]);
		const size_t n = pts.size() / 3;
		std::vector<double> out(5 * n, 0.0); // phi (no grad), phi (with grad), grad xyz
		for (size_t q = 0; q < n; ++q)
		{
			Eigen::Vector3d x(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]), gr(0, 0, 0);
			if (cmd == "eval")
			{
				out[5 * q] = g.interpolate(0u, x);
				out[5 * q + 1] = g.interpolate(0u, x, &gr);
			}
			else
			{
				std::array<unsigned int, 32> cell;
				Eigen::Vector3d c0;
				Eigen::Matrix<double, 32, 1> N;
				Eigen::Matrix<double, 32, 3> dN;
				const double nv = std::numeric_limits<double>::max();
				out[5 * q] = g.determineShapeFunctions(0u, x, cell, c0, N) ? g.interpolate(0u, x, cell, c0, N) : nv;
				out[5 * q + 1] =
					g.determineShapeFunctions(0u, x, cell, c0, N, &dN) ? g.interpolate(0u, x, cell, c0, N, &gr, &dN) : nv;
			}
			out[5 * q + 2] = gr[0];
			out[5 * q + 3] = gr[1];
			out[5 * q + 4] = gr[2];
		}
		write_doubles(argv[4], out);
		return 0;
	}
	if (cmd == "gpu" && argc == 5)
	{
		TriangleMesh mesh{std::string(argv[2])};
		TriangleMeshDistance md(mesh);
		Eigen::AlignedBox3d dom;
		dom.setEmpty();
		for (auto const& x : mesh.vertices())
			dom.extend(x);
		dom.max() += 1.0e-3 * dom.diagonal().norm() * Eigen::Vector3d::Ones();
		dom.min() -= 1.0e-3 * dom.diagonal().norm() * Eigen::Vector3d::Ones();
		CubicLagrangeDiscreteGrid g(dom, {{9, 7, 8}});
		DiscreteGrid::ContinuousFunction fn = MeshSDF{&md, false};
		g.addFunction(fn, true);
		if (!g.lastAddFunctionUsedGpu())
			return 3;
		g.addFunction(MeshSDF{&md, true}, false, [](Eigen::Vector3d const& p) { return p[2] > 0.0; });
		const std::vector<double> pts = read_doubles(argv[3]);
		const size_t n = pts.size() / 3;
		// batched (GPU) vs scalar (host) evaluation
		std::vector<double> phi(n), grad(3 * n), phi1(n);
		g.interpolate(0u, pts.data(), n, phi.data(), grad.data());
		g.interpolate(1u, pts.data(), n, phi1.data());
		size_t bad = 0;
		for (size_t q = 0; q < n; ++q)
		{
			Eigen::Vector3d x(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]), gr;
			const double s = g.interpolate(0u, x, &gr);
			bad += !(s == phi[q] && gr[0] == grad[3 * q] && gr[1] == grad[3 * q + 1] && gr[2] == grad[3 * q + 2]);
			bad += !(g.interpolate(1u, x) == phi1[q]);
		}
		// batched vs single-point distance queries
		std::vector<double> d(n);
		md.signed_distance(pts.data(), n, d.data());
		for (size_t q = 0; q < n && q < 64; ++q)
		{
			Result r = md.signed_distance(Eigen::Vector3d(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]));
			bad += !(r.distance == d[q]);
			Result u = md.unsigned_distance(std::array<double, 3>{{pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]}});
			bad += !(u.distance == std::abs(d[q]));
		}
		std::vector<double> out;
		out.push_back((double)bad);
		out.insert(out.end(), g.nodeData(0).begin(), g.nodeData(0).end());
		out.insert(out.end(), g.nodeData(1).begin(), g.nodeData(1).end());
		out.insert(out.end(), d.begin(), d.end());
		out.insert(out.end(), phi.begin(), phi.end());
		write_doubles(argv[4], out);
		std::printf("gpu driver: %zu mismatches, addFunction %.4f s (sampling %.4f s)\n", bad, g.lastAddFunctionSeconds(),
					g.lastSamplingSeconds());
		return bad ? 4 : 0;
	}
	std::fprintf(stderr, "bad arguments\n");
	return 2;
}
