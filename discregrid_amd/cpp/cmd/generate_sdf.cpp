// GenerateSDF -- command-line tool with the reference's options (cmd/generate_sdf/main.cpp:33-40:
// -r/--resolution "x y z", -d/--domain "minX minY minZ maxX maxY maxZ", -i/--invert, -o/--output,
// positional OBJ file) producing a byte-compatible .cdf file, with the node sampling on the GPU:
// the only functional change w.r.t. the reference's main() is that the lambda handed to
// addFunction is replaced by the typed Discregrid::MeshSDF functor.
#include <Discregrid/All>

#include <array>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace
{
void usage(const char* argv0)
{
	std::cout << "Generates a signed distance field from a closed two-manifold triangle mesh.\n"
				 "Usage:\n  "
			  << argv0
			  << " [OPTION...] [input OBJ file]\n\n"
				 "  -h, --help            Prints this help text\n"
				 "  -r, --resolution arg  Grid resolution (default: 10 10 10)\n"
				 "  -d, --domain arg      Domain extents (bounding box), format: \"minX minY minZ maxX maxY maxZ\"\n"
				 "  -i, --invert          Invert SDF\n"
				 "  -o, --output arg      Ouput file in cdf format (default: \"\")\n\n\n"
				 "Example: GenerateSDF -r \"50 50 50\" dragon.obj"
			  << std::endl;
}
} // namespace

int main(int argc, char* argv[])
{
	std::array<unsigned int, 3> resolution = {{10, 10, 10}};
	Eigen::AlignedBox3d domain;
	domain.setEmpty();
	bool invert = false;
	std::string output, input;
	for (int i = 1; i < argc; ++i)
	{
		const std::string a = argv[i];
		auto value = [&](std::string& dst) {
			const auto eq = a.find('=');
			if (a.rfind("--", 0) == 0 && eq != std::string::npos)
				dst = a.substr(eq + 1);
			else if (i + 1 < argc)
				dst = argv[++i];
			else
			{
				std::cout << "error parsing options: Option " << a << " is missing an argument" << std::endl;
				exit(1);
			}
		};
		std::string v;
		if (a == "-h" || a == "--help")
		{
			usage(argv[0]);
			return 0;
		}
		else if (a == "-i" || a == "--invert")
			invert = true;
		else if (a == "-r" || a.rfind("--resolution", 0) == 0)
		{
			value(v);
			std::istringstream s(v);
			s >> resolution[0] >> resolution[1] >> resolution[2];
			if (!s)
			{
				std::cout << "error parsing options: Argument '" << v << "' failed to parse" << std::endl;
				return 1;
			}
		}
		else if (a == "-d" || a.rfind("--domain", 0) == 0)
		{
			value(v);
			std::istringstream s(v);
			s >> domain.min()[0] >> domain.min()[1] >> domain.min()[2] >> domain.max()[0] >> domain.max()[1] >>
				domain.max()[2];
			if (!s)
			{
				std::cout << "error parsing options: Argument '" << v << "' failed to parse" << std::endl;
				return 1;
			}
		}
		else if (a == "-o" || a.rfind("--output", 0) == 0)
			value(output);
		else if (!a.empty() && a[0] == '-')
		{
			std::cout << "error parsing options: Option '" << a << "' does not exist" << std::endl;
			return 1;
		}
		else if (input.empty())
			input = a;
	}
	if (input.empty())
	{
		std::cout << "ERROR: No input mesh given." << std::endl;
		usage(argv[0]);
		return 1;
	}
	if (!std::ifstream(input).good())
	{
		std::cerr << "ERROR: Input file does not exist!" << std::endl;
		return 1;
	}

	std::cout << "Load mesh...";
	Discregrid::TriangleMesh mesh(input);
	std::cout << "DONE" << std::endl;

	std::cout << "Set up data structures...";
	Discregrid::TriangleMeshDistance md(mesh);
	std::cout << "DONE" << std::endl;

	if (domain.isEmpty())
	{
		// default domain: bounding box grown by 1e-3 * |diagonal|, max first, then min with the
		// already grown diagonal (cmd/generate_sdf/main.cpp:83-91)
		for (auto const& x : mesh.vertices())
			domain.extend(x);
		domain.max() += 1.0e-3 * domain.diagonal().norm() * Eigen::Vector3d::Ones();
		domain.min() -= 1.0e-3 * domain.diagonal().norm() * Eigen::Vector3d::Ones();
	}

	Discregrid::CubicLagrangeDiscreteGrid sdf(domain, resolution);
	std::cout << "Generate discretization..." << std::endl;
	sdf.addFunction(Discregrid::MeshSDF{&md, invert}, true);
	std::cout << "DONE" << std::endl;

	std::cout << "Serialize discretization...";
	if (output.empty())
	{
		output = input;
		const auto dot = output.find_last_of('.');
		if (dot != std::string::npos)
			output = output.substr(0, dot);
		output += ".cdf";
	}
	sdf.save(output);
	std::cout << "DONE" << std::endl;
	return 0;
}
