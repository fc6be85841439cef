// GenerateDensityMap -- the reference tool's options (cmd/generate_density_map/main.cpp:33-42:
// -r/--rest_density, -s/--smoothing_length, -o/--output, --no-reduction, -i/--invert (accepted and
// ignored, as in the reference, which never reads it), positional .cdf file with the SDF in
// field 0) producing a byte-compatible .cdm file.  The tool's addFunction(density_func, true,
// predicate) call (:119-133) -- 4097 interpolations per lattice node -- is one GPU call here:
// CubicLagrangeDiscreteGrid::addDensityMap.
#include <Discregrid/All>

#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <string>

int main(int argc, char* argv[])
{
	double rho0 = 1000.0, h = 0.1;
	bool no_reduction = false;
	std::string output, input;
	for (int i = 1; i < argc; ++i)
	{
		const std::string a = argv[i];
		auto value = [&]() -> std::string {
			const auto eq = a.find('=');
			if (a.rfind("--", 0) == 0 && eq != std::string::npos)
				return a.substr(eq + 1);
			if (i + 1 < argc)
				return argv[++i];
			std::cout << "error parsing options: Option " << a << " is missing an argument" << std::endl;
			exit(1);
		};
		if (a == "-h" || a == "--help")
		{
			std::cout << "Generates an SPH boundary density map from a discrete signed distance field.\n"
						 "Usage:\n  "
					  << argv[0]
					  << " [OPTION...] [input .cdf file]\n\n"
						 "  -h, --help                  Prints this help text\n"
						 "  -r, --rest_density arg      Rest density rho0 of the fluid (default: 1000.0)\n"
						 "  -i, --invert                Invert field\n"
						 "  -s, --smoothing_length arg  Kernel smoothing length (default: 0.1)\n"
						 "  -o, --output arg            Ouput file in cdf format (default: \"\")\n"
						 "      --no-reduction          Disables discarding of cells for sparse layout.\n"
					  << std::endl;
			return 0;
		}
		else if (a == "-r" || a.rfind("--rest_density", 0) == 0)
			rho0 = std::stod(value());
		else if (a == "-s" || a.rfind("--smoothing_length", 0) == 0)
			h = std::stod(value());
		else if (a == "-o" || a.rfind("--output", 0) == 0)
			output = value();
		else if (a == "--no-reduction")
			no_reduction = true;
		else if (a == "-i" || a == "--invert")
		{
		}
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
		std::cout << "ERROR: No input SDF given." << std::endl;
		return 1;
	}
	if (!std::ifstream(input).good())
	{
		std::cerr << "ERROR: Input file does not exist!" << std::endl;
		return 1;
	}
	const auto dot = input.find_last_of('.');
	const std::string extension = dot == std::string::npos ? "" : input.substr(dot + 1);

	std::cout << "Load SDF...";
	std::unique_ptr<Discregrid::CubicLagrangeDiscreteGrid> sdf;
	if (extension == "cdf")
		sdf.reset(new Discregrid::CubicLagrangeDiscreteGrid(input));
	std::cout << "DONE" << std::endl;
	if (!sdf)
	{
		std::cerr << "ERROR: unsupported input file type (expected .cdf)" << std::endl;
		return 1;
	}

	const double cell_diag = sdf->cellSize().norm();
	std::cout << "Generate density map..." << std::endl;
	sdf->addDensityMap(0u, h, rho0, !no_reduction, true);

	if (!no_reduction)
	{
		std::cout << "Reduce discrete fields...";
		// the reference's two lambdas (main.cpp:138-145) as typed predicates: same arithmetic, evaluated on the GPU
		sdf->reduceField(0u, Discregrid::ValuePredicate::band(-6.0 * h, 2.0 * h, cell_diag));
		sdf->reduceField(1u, Discregrid::ValuePredicate::range(0.0, 3.0 * rho0));
		std::cout << "DONE" << std::endl;
	}

	std::cout << "Serialize discretization...";
	if (output.empty())
	{
		output = input;
		if (dot != std::string::npos)
			output = output.substr(0, dot);
		output += ".cdm";
	}
	sdf->save(output);
	std::cout << "DONE" << std::endl;
	return 0;
}
