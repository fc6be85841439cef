// tests/cpp/sanitize_main.cpp -- runs the host-side product code (BVH / pseudonormal builder,
// lattice decomposition, shard bookkeeping, per-lane arithmetic through the wave emulator, the
// staged density integral) under AddressSanitizer + UndefinedBehaviorSanitizer.
// Built and run by tests/test_sanitizers.py:
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all sanitize_main.cpp ../emu/wave_emu.cpp
//       ../../discregrid_amd/csrc/dg_build.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C"
{
void* emu_mesh_create(const double* verts, size_t nv, const uint32_t* tris, size_t nt, int max_leaf);
void emu_mesh_free(void* h);
int emu_mesh_check(void* h, const double* verts, const uint32_t* tris);
int emu_sample_nodes(void* h, const double dmin[3], const double cell[3], const uint32_t res[3], int invert, int mode,
					 uint64_t a0, uint64_t a1, const uint8_t* mask, double* out, uint8_t* written, uint64_t* stats);
void emu_signed_distance(void* h, const double* xyz, uint64_t n, double* dist, int32_t* tri, int32_t* entity, double* nearest);
uint64_t emu_shard_count(const uint32_t res[3], int rank, int nranks);
void emu_unpack(const uint32_t res[3], int nranks, const double* gathered, uint64_t stride, double* field);
void emu_interpolate(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3],
					 const double* coeffs, const uint32_t* cells, const uint32_t* cell_map, const double* xyz, uint64_t n,
					 double* phi, double* grad);
void emu_density_map(const double domain[6], const double cell[3], const double inv_cell[3], const uint32_t res[3],
					 const double* coeffs, const uint32_t* cells, const uint32_t* cell_map, double h, double rho0, int band,
					 uint64_t begin, uint64_t end, double* out);
}

int main()
{
	// torus mesh
	const int nu = 20, nv = 10;
	std::vector<double> V;
	std::vector<uint32_t> F;
	for (int i = 0; i < nu; ++i)
		for (int j = 0; j < nv; ++j)
		{
			const double a = 2 * M_PI * i / nu, b = 2 * M_PI * j / nv;
			V.push_back((1.0 + 0.35 * std::cos(b)) * std::cos(a));
			V.push_back((1.0 + 0.35 * std::cos(b)) * std::sin(a));
			V.push_back(0.35 * std::sin(b));
		}
	for (int i = 0; i < nu; ++i)
		for (int j = 0; j < nv; ++j)
		{
			const uint32_t p00 = i * nv + j, p10 = ((i + 1) % nu) * nv + j, p01 = i * nv + (j + 1) % nv,
						   p11 = ((i + 1) % nu) * nv + (j + 1) % nv;
			F.insert(F.end(), {p00, p10, p11, p00, p11, p01});
		}
	int bad = 0;
	for (int leaf : {1, 3, 8, 16})
	{
		void* m = emu_mesh_create(V.data(), V.size() / 3, F.data(), F.size() / 3, leaf);
		bad += emu_mesh_check(m, V.data(), F.data());
		const double dmin[3] = {-1.4, -1.4, -0.4}, cell[3] = {2.8 / 7, 2.8 / 5, 0.8 / 6}, inv[3] = {7 / 2.8, 5 / 2.8, 6 / 0.8};
		const double dom[6] = {-1.4, -1.4, -0.4, 1.4, 1.4, 0.4};
		const uint32_t res[3] = {7, 5, 6};
		const uint64_t n = 8 * 6 * 7 + 2 * (7 * 6 * 7 + 8 * 5 * 7 + 8 * 6 * 6);
		std::vector<double> out(n), field(n);
		std::vector<uint8_t> written(n, 0);
		emu_sample_nodes(m, dmin, cell, res, 0, 0, 0, n, nullptr, out.data(), written.data(), nullptr);
		for (auto w : written)
			bad += (w != 1);
		// ragged range + mask
		std::vector<uint8_t> mask(n, 1);
		for (uint64_t i = 0; i < n; i += 3)
			mask[i] = 0;
		std::vector<double> part(n - 150);
		emu_sample_nodes(m, dmin, cell, res, 1, 0, 100, n - 50, mask.data() + 0, part.data(), nullptr, nullptr);
		// shards
		for (int nr : {2, 5})
		{
			uint64_t stride = 0;
			for (int r = 0; r < nr; ++r)
				stride = std::max<uint64_t>(stride, emu_shard_count(res, r, nr));
			stride = (stride + 63) / 64 * 64;
			std::vector<double> G(stride * nr, 0.0);
			for (int r = 0; r < nr; ++r)
				emu_sample_nodes(m, dmin, cell, res, 0, 1, r, nr, nullptr, G.data() + r * stride, nullptr, nullptr);
			emu_unpack(res, nr, G.data(), stride, field.data());
			for (uint64_t i = 0; i < n; ++i)
				bad += (field[i] != out[i]);
		}
		// points, interpolation, density map
		std::vector<double> P;
		for (int i = 0; i < 333; ++i)
		{
			P.push_back(-1.6 + 3.2 * (i % 17) / 16.0);
			P.push_back(-1.6 + 3.2 * (i % 13) / 12.0);
			P.push_back(-0.5 + 1.0 * (i % 7) / 6.0);
		}
		std::vector<double> d(333), phi(333), grad(999), near(999);
		std::vector<int32_t> tri(333), ent(333);
		emu_signed_distance(m, P.data(), 333, d.data(), tri.data(), ent.data(), near.data());
		emu_interpolate(dom, cell, inv, res, out.data(), nullptr, nullptr, P.data(), 333, phi.data(), grad.data());
		std::vector<double> rho(64);
		emu_density_map(dom, cell, inv, res, out.data(), nullptr, nullptr, 0.15, 1000.0, 1, 300, 364, rho.data());
		emu_mesh_free(m);
	}
	std::printf("sanitize_main: %d problems\n", bad);
	return bad != 0;
}
