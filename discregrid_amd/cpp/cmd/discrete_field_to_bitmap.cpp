// DiscreteFieldToBitmap -- writes a planar slice of a field of a .cdf/.cdm file as a 24-bit BMP,
// with the reference tool's options (cmd/discrete_field_to_bitmap/main.cpp:35-44: -f/--field_id,
// -s/--samples, -p/--plane, -d/--depth, -o/--output, -c/--colormap, positional input file), sample
// positions (:108-137), normalisation (:157) and colour maps (:15-28).  The xsamples*ysamples
// evaluations are ONE batched CubicLagrangeDiscreteGrid::interpolate call on the GPU (kernel K2)
// instead of the reference's OpenMP loop over the per-point overload.
//
// File layout as written by the reference (cmd/discrete_field_to_bitmap/bmp_file.cpp:71-121):
// 14-byte file header whose size field holds 40 (sizeof the info header, not the file size), a
// 40-byte BITMAPINFOHEADER with 4000 pels/m, rows in sample order (j = 0 first) padded to a
// multiple of 4 bytes, pixels stored B,G,R.  The reference writes biSizeImage before assigning it
// (uninitialised stack bytes, file offsets 34..37); 0 is written here, which BI_RGB permits.
#include <Discregrid/All>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace
{
void usage(const char* argv0)
{
	std::cout << "Transforms a slice of a discrete SDF to a bitmap image.\n"
				 "Usage:\n  "
			  << argv0
			  << " [OPTION...] [input SDF file]\n\n"
				 "  -h, --help          Prints this help text\n"
				 "  -f, --field_id arg  ID in which the SDF to export is stored. (default: 0)\n"
				 "  -s, --samples arg   Number of samples in width direction (default: 1024)\n"
				 "  -p, --plane arg     Plane in which the image slice is extracted (default: xy)\n"
				 "  -d, --depth arg     Relative depth value between -1 and 1 in direction of the axis\n"
				 "                      orthogonal to the plane (default: 0)\n"
				 "  -o, --output arg    Output (in bmp format) (default: \"\")\n"
				 "  -c, --colormap arg  Color map options: redsequential (rs), green blue inverse\n"
				 "                      diverging (gb) (suitable for visualiztion of signed distance\n"
				 "                      fields) (default: gb)\n\n\n"
				 "Example: SDFToBitmap -p xz file.sdf"
			  << std::endl;
}

unsigned char clamp_byte(double x) { return static_cast<unsigned char>(std::min(std::max(x, 0.0), 255.0)); }

void put_u16(std::vector<unsigned char>& b, std::size_t at, std::uint16_t v) { std::memcpy(&b[at], &v, 2); }
void put_u32(std::vector<unsigned char>& b, std::size_t at, std::uint32_t v) { std::memcpy(&b[at], &v, 4); }

// rgb: 3 bytes per pixel, row j at rgb + 3*width*j
bool write_bmp(std::string const& path, unsigned int width, unsigned int height, unsigned char const* rgb)
{
	const std::size_t row = ((static_cast<std::size_t>(width) * 3u + 3u) >> 2) << 2;
	std::vector<unsigned char> file(54u + row * height, 0);
	file[0] = 'B';
	file[1] = 'M';
	put_u32(file, 2, 40u); // the reference stores the info-header size here
	put_u32(file, 10, 54u);
	put_u32(file, 14, 40u);
	put_u32(file, 18, width);
	put_u32(file, 22, height);
	put_u16(file, 26, 1u);
	put_u16(file, 28, 24u);
	put_u32(file, 38, 4000u);
	put_u32(file, 42, 4000u);
	for (unsigned int j = 0; j < height; ++j)
	{
		unsigned char* dst = &file[54u + row * j];
		unsigned char const* src = rgb + static_cast<std::size_t>(j) * width * 3u;
		for (unsigned int i = 0; i < width; ++i, dst += 3, src += 3)
		{
			dst[0] = src[2];
			dst[1] = src[1];
			dst[2] = src[0];
		}
	}
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f)
		return false;
	const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
	return (std::fclose(f) == 0) && ok;
}
} // namespace

int main(int argc, char* argv[])
{
	unsigned int field_id = 0, xsamples = 1024;
	std::string plane = "xy", out_file, cm = "gb", filename;
	double depth = 0.0;
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
		auto number = [&](auto& dst) {
			const std::string v = value();
			std::istringstream s(v);
			s >> dst;
			if (!s)
			{
				std::cout << "error parsing options: Argument '" << v << "' failed to parse" << std::endl;
				exit(1);
			}
		};
		if (a == "-h" || a == "--help")
		{
			usage(argv[0]);
			return 0;
		}
		else if (a == "-f" || a.rfind("--field_id", 0) == 0)
			number(field_id);
		else if (a == "-s" || a.rfind("--samples", 0) == 0)
			number(xsamples);
		else if (a == "-d" || a.rfind("--depth", 0) == 0)
			number(depth);
		else if (a == "-p" || a.rfind("--plane", 0) == 0)
			plane = value();
		else if (a == "-o" || a.rfind("--output", 0) == 0)
			out_file = value();
		else if (a == "-c" || a.rfind("--colormap", 0) == 0)
			cm = value();
		else if (!a.empty() && a[0] == '-')
		{
			std::cout << "error parsing options: Option '" << a << "' does not exist" << std::endl;
			return 1;
		}
		else if (filename.empty())
			filename = a;
	}
	if (filename.empty())
	{
		std::cout << "ERROR: No input file given." << std::endl;
		usage(argv[0]);
		return 1;
	}

	const auto dot = filename.find_last_of('.');
	const std::string extension = dot == std::string::npos ? std::string() : filename.substr(dot + 1);
	if (extension != "cdf" && extension != "cdm")
	{
		// the reference dereferences a null grid here (main.cpp:72-80)
		std::cerr << "ERROR: Input file must be a .cdf or .cdm file." << std::endl;
		return 1;
	}
	std::cout << "Load SDF...";
	std::unique_ptr<Discregrid::CubicLagrangeDiscreteGrid> sdf(new Discregrid::CubicLagrangeDiscreteGrid(filename));
	std::cout << "DONE" << std::endl;
	if (field_id >= sdf->nFields())
	{
		std::cerr << "ERROR: The file holds " << sdf->nFields() << " field(s); field " << field_id << " requested."
				  << std::endl;
		return 1;
	}

	auto const& domain = sdf->domain();
	const Eigen::Vector3d diag = domain.diagonal();

	if (plane.empty() || (plane.length() != 2 && plane[0] != plane[1])) // main.cpp:85 ("xx"-like strings pass)
	{
		std::cerr << "ERROR: Invalid option for plane provided. Should be one of the following options: xy, xz, yz, yx"
				  << std::endl;
		return 1;
	}
	// axis of the image's width, of its height and the one orthogonal to the slice (main.cpp:91-103;
	// letters other than y and z select x, as there)
	int dir[3] = {0, 0, 0};
	if (plane[0] == 'y')
		dir[0] = 1;
	else if (plane[0] == 'z')
		dir[0] = 2;
	if (plane[1] == 'y')
		dir[1] = 1;
	else if (plane[1] == 'z')
		dir[1] = 2;
	if (dir[0] != 1 && dir[1] != 1)
		dir[2] = 1;
	if (dir[0] != 2 && dir[1] != 2)
		dir[2] = 2;

	const unsigned int ysamples =
		static_cast<unsigned int>(std::round(diag[dir[1]] / diag[dir[0]] * static_cast<double>(xsamples)));
	const double xwidth = diag[dir[0]] / xsamples;
	const double ywidth = diag[dir[1]] / ysamples;
	const std::size_t n = static_cast<std::size_t>(xsamples) * ysamples;
	if (n == 0)
	{
		std::cerr << "ERROR: Empty image (" << xsamples << " x " << ysamples << ")." << std::endl;
		return 1;
	}

	std::cout << "Sample field...";
	std::vector<double> xyz(3 * n), data(n);
	const double w = domain.min()[dir[2]] + 0.5 * (1.0 + depth) * diag[dir[2]];
#pragma omp parallel for
	for (long long k = 0; k < static_cast<long long>(n); ++k)
	{
		const unsigned int i = static_cast<unsigned int>(k % xsamples);
		const unsigned int j = static_cast<unsigned int>(k / xsamples);
		const double xr = static_cast<double>(i) / static_cast<double>(xsamples);
		const double yr = static_cast<double>(j) / static_cast<double>(ysamples);
		// written in the reference's order: a degenerate plane such as "xx" overwrites a component
		double* p = &xyz[3 * k];
		p[0] = p[1] = p[2] = 0.0;
		p[dir[0]] = domain.min()[dir[0]] + xr * diag[dir[0]] + 0.5 * xwidth;
		p[dir[1]] = domain.min()[dir[1]] + yr * diag[dir[1]] + 0.5 * ywidth;
		p[dir[2]] = w;
	}
	sdf->interpolate(field_id, xyz.data(), n, data.data(), nullptr);
	for (auto& v : data)
		if (v == std::numeric_limits<double>::max())
			v = 0.0;
	std::cout << "DONE" << std::endl;

	const double min_v = *std::min_element(data.begin(), data.end());
	const double max_v = *std::max_element(data.begin(), data.end());

	if (out_file.empty())
	{
		out_file = filename;
		const auto last = out_file.find_last_of('.');
		if (last != std::string::npos)
			out_file = out_file.substr(0, last);
		out_file += ".bmp";
	}
	std::cout << "Ouput file: " << out_file << std::endl;

	std::cout << "Export BMP...";
	if (cm != "gb" && cm != "rs")
		std::cerr << "WARNING: Unknown color map option. Fallback to mode 'gb'." << std::endl;
	// As in the reference (main.cpp:168-171) an unknown option leaves the image black.
	std::vector<unsigned char> rgb(3 * n, 0);
	const double pos_scale = std::abs(max_v), neg_scale = std::abs(min_v);
	for (std::size_t k = 0; k < n; ++k)
	{
		const double v = data[k] >= 0.0 ? data[k] / pos_scale : data[k] / neg_scale;
		if (cm == "gb")
		{
			if (v >= 0.0)
				rgb[3 * k + 1] = clamp_byte(255.0 * (1.0 - v));
			else
				rgb[3 * k + 2] = clamp_byte(255.0 * (1.0 + v));
		}
		else if (cm == "rs")
			rgb[3 * k] = clamp_byte(255.0 * v);
	}
	if (!write_bmp(out_file, xsamples, ysamples, rgb.data()))
	{
		std::cerr << "ERROR: Could not write " << out_file << std::endl;
		return 1;
	}
	std::cout << "DONE" << std::endl;

	std::cout << std::endl << "Statistics:" << std::endl;
	std::cout << "\tdomain         = " << domain.min()[0] << " " << domain.min()[1] << " " << domain.min()[2] << ", "
			  << domain.max()[0] << " " << domain.max()[1] << " " << domain.max()[2] << std::endl;
	std::cout << "\tmin value      = " << min_v << std::endl;
	std::cout << "\tmax value      = " << max_v << std::endl;
	std::cout << "\tbmp resolution = " << xsamples << " x " << ysamples << std::endl;
	return 0;
}
