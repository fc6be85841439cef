// GenerateSDFMultiGPU -- GenerateSDF over several GPUs of one node, one process per GPU (no reference
// counterpart: the reference is a single OpenMP process; options as GenerateSDF, cmd/generate_sdf/main.cpp:33-40,
// plus -g/--gpus N, --pieces C, --steps K).
//
// The parent forks N ranks before touching HIP.  Rank r makes device r current, uploads the mesh
// (replicated), joins the RCCL communicator (rank 0 publishes the unique id through a file) and calls
// dg_sdf_sample_allgather_device: its shards of the node lattice are sampled, all-gathered over xGMI and
// unpacked, in pieces that overlap, so that EVERY rank holds the whole coefficient vector on its GPU
// (ready for the batched interpolate / density-map kernels).  Rank 0 copies it to the host and writes the
// same .cdf file GenerateSDF writes.  With --steps K the sampling step is repeated K times after one
// warm-up and the parent prints one JSON line: whole-job Mnodes/s = nodes * K / max over ranks of the
// time of the K steps (the figure bench.py reports for N GPUs).
#include <algorithm>
#include <Discregrid/All>

#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include <sys/wait.h>
#include <unistd.h>

#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "discregrid_hip.h"

namespace
{
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Options
{
	std::array<unsigned int, 3> resolution = {{10, 10, 10}};
	Eigen::AlignedBox3d domain;
	bool invert = false;
	int gpus = 1, pieces = 4, steps = 1;
	int exchange = 0; // flags of dg_sdf_sample_exchange_device (--inplace, --p2p, --copy)
	bool host_vector = false; // --host: dg_sdf_sample_to_host_field (a shared-memory host vector; no RCCL, no device IPC)
	bool shm_control = false; // --copy-shm: the copy form on dg_comm_create_shm (control plane in shared memory: no RCCL)
	std::string output, input;
};

[[noreturn]] void die(int rank, const std::string& what)
{
	std::cerr << "GenerateSDFMultiGPU rank " << rank << ": " << what << std::endl;
	_exit(1);
}
void check(int rank, dg_status s, const char* what)
{
	if (s != DG_OK)
		die(rank, std::string(what) + ": " + dg_last_error());
}
void check_hip(int rank, hipError_t e, const char* what)
{
	if (e != hipSuccess)
		die(rank, std::string(what) + ": " + hipGetErrorString(e));
}

int run_rank(const Options& opt, int rank, const std::string& id_file, const std::string& time_file)
{
	int n_devices = 0;
	check(rank, dg_device_count(&n_devices), "dg_device_count");
	if (n_devices < 1)
		die(rank, "no HIP device");
	if (opt.gpus > n_devices)
		die(rank, "asked for " + std::to_string(opt.gpus) + " GPUs, " + std::to_string(n_devices) + " visible");
	check(rank, dg_set_device(rank), "dg_set_device");

	Discregrid::TriangleMesh mesh(opt.input);
	Eigen::AlignedBox3d domain = opt.domain;
	if (domain.isEmpty())
	{
		for (auto const& x : mesh.vertices())
			domain.extend(x);
		domain.max() += 1.0e-3 * domain.diagonal().norm() * Eigen::Vector3d::Ones(); // main.cpp:83-91
		domain.min() -= 1.0e-3 * domain.diagonal().norm() * Eigen::Vector3d::Ones();
	}
	Discregrid::TriangleMeshDistance md(mesh); // uploads BVH + packets to this rank's GPU
	Discregrid::CubicLagrangeDiscreteGrid sdf(domain, opt.resolution);
	dg_grid_desc grid;
	std::memset(&grid, 0, sizeof(grid));
	for (int d = 0; d < 3; ++d)
	{
		grid.domain_min[d] = domain.min()[d];
		grid.domain_max[d] = domain.max()[d];
		grid.resolution[d] = opt.resolution[d];
		grid.cell_size[d] = sdf.cellSize()[d];
		grid.inv_cell_size[d] = sdf.invCellSize()[d];
	}
	const uint64_t n_nodes = dg_grid_n_nodes(&grid);

	hipStream_t stream = nullptr;
	check_hip(rank, hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
	const dg_mesh* dmesh = static_cast<const dg_mesh*>(md.deviceMesh());
	if (opt.host_vector)
	{
		// the form that needs neither RCCL nor device IPC: every rank copies its chunks into ONE shared-memory vector (the
		// segment's name is derived from the id file's, which is unique to this run); rank 0 saves from it
		std::string name = "dg_sdfmulti_" + id_file.substr(id_file.find_last_of('/') + 1);
		dg_host_field* hf = nullptr;
		check(rank, dg_host_field_open(name.c_str(), n_nodes, rank, opt.gpus, &hf), "dg_host_field_open");
		double* d_mine = nullptr;
		check_hip(rank, hipMalloc(reinterpret_cast<void**>(&d_mine), n_nodes * sizeof(double)), "hipMalloc");
		check(rank, dg_sdf_sample_to_host_field(dmesh, &grid, opt.invert ? 1 : 0, hf, opt.pieces, nullptr, d_mine, stream), "sample to the host vector");
		double seconds = 0.0;
		if (opt.steps > 1 || !time_file.empty())
		{
			const double t0 = now();
			for (int k = 0; k < opt.steps; ++k)
				check(rank, dg_sdf_sample_to_host_field(dmesh, &grid, opt.invert ? 1 : 0, hf, opt.pieces, nullptr, d_mine, stream), "sample to the host vector");
			seconds = now() - t0;
		}
		if (!time_file.empty())
			std::ofstream(time_file) << seconds << "\n";
		if (rank == 0 && !opt.output.empty())
		{
			const double* v = dg_host_field_data(hf);
			sdf.addNodeData(Discregrid::FieldVector(v, v + n_nodes));
			sdf.save(opt.output);
		}
		check(rank, dg_host_field_barrier(hf), "barrier"); // (rank 0 has read the vector)
		dg_host_field_close(hf);
		(void)hipFree(d_mine);
		(void)hipStreamDestroy(stream);
		return 0;
	}

	dg_comm* comm = nullptr;
	if (opt.shm_control)
	{
		// no RCCL at all: the copy form's two small collectives run over a shared-memory segment named after this run's id file
		const std::string name = "dg_sdfmulti_ctl_" + id_file.substr(id_file.find_last_of('/') + 1);
		check(rank, dg_comm_create_shm(name.c_str(), rank, opt.gpus, &comm), "dg_comm_create_shm");
	}
	else
	{
		// communicator: rank 0 publishes the RCCL unique id through a file (written whole, then renamed)
		uint8_t id[DG_UNIQUE_ID_BYTES];
		if (rank == 0)
		{
			check(rank, dg_comm_unique_id(id), "dg_comm_unique_id");
			const std::string tmp = id_file + ".tmp";
			const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0600); // (never through a file somebody else made)
			if (fd < 0 || write(fd, id, sizeof(id)) != (ssize_t)sizeof(id) || close(fd) != 0 || std::rename(tmp.c_str(), id_file.c_str()) != 0)
				die(rank, "cannot publish the communicator id");
		}
		else
		{
			const double t0 = now();
			while (true)
			{
				std::ifstream in(id_file, std::ios::binary);
				if (in.good() && in.read(reinterpret_cast<char*>(id), sizeof(id)))
					break;
				if (now() - t0 > 120.0)
					die(rank, "timed out waiting for the communicator id");
				std::this_thread::sleep_for(std::chrono::milliseconds(20));
			}
		}
		check(rank, dg_comm_create(id, rank, opt.gpus, &comm), "dg_comm_create");
	}

	// the copy form's field comes from the communicator (chunks the peers can map whatever the field's size); the RCCL forms
	// take any device array
	const bool from_comm = (opt.exchange & DG_EXCHANGE_COPY) != 0;
	double* d_field = nullptr;
	if (from_comm)
		check(rank, dg_comm_field_alloc(comm, n_nodes, &d_field), "dg_comm_field_alloc");
	else
		check_hip(rank, hipMalloc(reinterpret_cast<void**>(&d_field), n_nodes * sizeof(double)), "hipMalloc");

	// one untimed step (first-use allocations, RCCL channel set-up), then the timed ones
	check(rank, dg_sdf_sample_exchange_device(dmesh, &grid, opt.invert ? 1 : 0, comm, opt.pieces, opt.exchange, 0, nullptr, d_field, stream), "sample + exchange");
	check_hip(rank, hipStreamSynchronize(stream), "synchronize");
	double seconds = 0.0;
	if (opt.steps > 1 || !time_file.empty())
	{
		const double t0 = now();
		for (int k = 0; k < opt.steps; ++k)
			check(rank, dg_sdf_sample_exchange_device(dmesh, &grid, opt.invert ? 1 : 0, comm, opt.pieces, opt.exchange, 0, nullptr, d_field, stream), "sample + exchange");
		check_hip(rank, hipStreamSynchronize(stream), "synchronize");
		seconds = now() - t0;
	}
	if (!time_file.empty())
		std::ofstream(time_file) << seconds << "\n";

	if (rank == 0 && !opt.output.empty())
	{
		Discregrid::FieldVector coeffs(n_nodes);
		check_hip(rank, hipMemcpy(coeffs.data(), d_field, n_nodes * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
		sdf.addNodeData(std::move(coeffs));
		sdf.save(opt.output);
	}
	dg_comm_destroy(comm); // (releases a field of dg_comm_field_alloc as well)
	if (!from_comm)
		(void)hipFree(d_field);
	(void)hipStreamDestroy(stream);
	return 0;
}
} // namespace

int main(int argc, char* argv[])
{
	Options opt;
	opt.domain.setEmpty();
	bool timing = false;
	for (int i = 1; i < argc; ++i)
	{
		const std::string a = argv[i];
		auto value = [&]() -> std::string {
			if (i + 1 >= argc)
			{
				std::cout << "error parsing options: Option " << a << " is missing an argument" << std::endl;
				exit(1);
			}
			return argv[++i];
		};
		if (a == "-h" || a == "--help")
		{
			std::cout << "Usage: " << argv[0]
					  << " [-r \"x y z\"] [-d \"minX minY minZ maxX maxY maxZ\"] [-i] [-g gpus] [--pieces c] [--inplace | --p2p | --copy | --copy-shm | --host] [--steps k] [-o out.cdf] mesh.obj"
					  << std::endl;
			return 0;
		}
		else if (a == "-i" || a == "--invert")
			opt.invert = true;
		else if (a == "-r" || a == "--resolution")
		{
			std::istringstream s(value());
			s >> opt.resolution[0] >> opt.resolution[1] >> opt.resolution[2];
		}
		else if (a == "-d" || a == "--domain")
		{
			std::istringstream s(value());
			s >> opt.domain.min()[0] >> opt.domain.min()[1] >> opt.domain.min()[2] >> opt.domain.max()[0] >> opt.domain.max()[1] >>
				opt.domain.max()[2];
		}
		else if (a == "-g" || a == "--gpus")
			opt.gpus = std::atoi(value().c_str());
		else if (a == "--pieces")
			opt.pieces = std::atoi(value().c_str());
		else if (a == "--inplace") // contiguous chunks sampled into place, grouped broadcasts, no unpack
			opt.exchange |= DG_EXCHANGE_INPLACE;
		else if (a == "--p2p") // ... exchanged with send / recv pairs instead
			opt.exchange |= DG_EXCHANGE_INPLACE | DG_EXCHANGE_P2P;
		else if (a == "--copy-shm") // the copy form with its control plane in shared memory: the whole field on every GPU, no RCCL
		{
			opt.exchange = DG_EXCHANGE_INPLACE | DG_EXCHANGE_COPY;
			opt.shm_control = true;
		}
		else if (a == "--host") // every rank copies its chunks into a shared-memory host vector: no RCCL, no device IPC
			opt.host_vector = true;
		else if (a == "--copy") // ... pushed into the peers' fields by the copy engines (fields of dg_comm_field_alloc), no collective kernel
			opt.exchange = DG_EXCHANGE_INPLACE | DG_EXCHANGE_COPY;
		else if (a == "--steps")
		{
			opt.steps = std::atoi(value().c_str());
			timing = true;
		}
		else if (a == "-o" || a == "--output")
			opt.output = value();
		else if (!a.empty() && a[0] == '-')
		{
			std::cout << "error parsing options: Option '" << a << "' does not exist" << std::endl;
			return 1;
		}
		else if (opt.input.empty())
			opt.input = a;
	}
	if (opt.input.empty() || !std::ifstream(opt.input).good())
	{
		std::cerr << "ERROR: Input file does not exist!" << std::endl;
		return 1;
	}
	if (opt.gpus < 1 || opt.gpus > 64 || opt.steps < 1)
	{
		std::cerr << "ERROR: --gpus must be 1..64, --steps >= 1" << std::endl;
		return 1;
	}
	// rendezvous files live in a directory of our own (mkdtemp: mode 0700, unpredictable name): nobody else can plant
	// or read the communicator id
	char dir_template[] = "/tmp/dg_multi_gpu_XXXXXX";
	if (mkdtemp(dir_template) == nullptr)
	{
		std::perror("mkdtemp");
		return 1;
	}
	const std::string base = std::string(dir_template) + "/r";
	const std::string id_file = base + ".id";
	std::vector<pid_t> kids;
	for (int r = 0; r < opt.gpus; ++r)
	{
		const pid_t pid = fork(); // before any HIP call: every rank initialises its own runtime
		if (pid < 0)
		{
			std::perror("fork");
			return 1;
		}
		if (pid == 0)
			_exit(run_rank(opt, r, id_file, timing ? base + ".t" + std::to_string(r) : std::string()));
		kids.push_back(pid);
	}
	// Wait for the ranks in the order they finish.  A rank that fails leaves the others inside a collective (or the
	// communicator set-up) that can never complete: they are terminated instead of being waited for.
	int failed = 0;
	for (size_t left = kids.size(); left > 0; --left)
	{
		int st = 0;
		const pid_t pid = wait(&st);
		if (pid < 0)
			break;
		// (a reaped pid may be handed to an unrelated process at any moment: only ranks still running are ever signalled)
		kids.erase(std::remove(kids.begin(), kids.end(), pid), kids.end());
		if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0))
		{
			if (failed++ == 0)
				for (pid_t k : kids)
					kill(k, SIGTERM);
		}
	}
	std::remove(id_file.c_str());
	double slowest = 0.0;
	for (int r = 0; r < opt.gpus && timing; ++r)
	{
		const std::string f = base + ".t" + std::to_string(r);
		double s = 0.0;
		std::ifstream(f) >> s;
		slowest = std::max(slowest, s);
		std::remove(f.c_str());
	}
	rmdir(dir_template);
	if (failed)
	{
		std::cerr << "GenerateSDFMultiGPU: " << failed << " rank(s) failed or were stopped" << std::endl;
		return 1;
	}
	if (timing)
	{
		const uint64_t nx = opt.resolution[0], ny = opt.resolution[1], nz = opt.resolution[2];
		const uint64_t n_nodes = (nx + 1) * (ny + 1) * (nz + 1) + 2 * (nx * (ny + 1) * (nz + 1) + (nx + 1) * ny * (nz + 1) + (nx + 1) * (ny + 1) * nz);
		std::printf("{\"metric\": \"Mnodes/s SDF sampling, sample + all-gather + unpack\", \"value\": %.3f, \"unit\": \"Mnodes/s\", "
					"\"n_gpus\": %d, \"steps\": %d, \"pieces\": %d, \"ms_per_step\": %.4f, \"nodes\": %llu}\n",
					(double)n_nodes * opt.steps / slowest / 1e6, opt.gpus, opt.steps, opt.pieces, slowest / opt.steps * 1e3,
					(unsigned long long)n_nodes);
	}
	return 0;
}
