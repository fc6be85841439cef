// dg_capi_shm.h -- a POSIX shared-memory segment that the ranks of one node (one process per GPU) map, with a barrier that lives
// in it: the base of the host-vector exchange form (dg_capi_hostfield.cpp) and of the RCCL-free control plane of the copy form
// (dg_comm_create_shm, dg_capi_comm.cpp).  Rank 0 creates the segment (claiming its pages at once: a tmpfs that is too small
// answers ENOSPC, not SIGBUS at the first touch), the others wait for it; nobody waits longer than the deadline.  Internal.
#pragma once
#include "dg_capi_internal.h"

#include <atomic>
#include <csignal>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace dgshm
{
constexpr uint64_t kMagic = 0x64675f73686d3031ull; // "dg_shm01"
constexpr size_t kHeaderBytes = 4096;
struct Header // the first page of every segment; every member is address-free
{
	std::atomic<uint64_t> magic; // set last by the creating rank
	uint64_t payload_bytes;
	uint32_t nranks;
	uint32_t kind;                    // what the payload is (the opener's tag: a segment is not mistaken for another kind)
	std::atomic<uint32_t> arrived;    // barrier: ranks that have arrived in the current generation
	std::atomic<uint32_t> generation; // barrier: bumped by the last arrival
	std::atomic<uint32_t> attached;   // ranks that mapped the segment
	std::atomic<uint32_t> failed;     // a barrier of this segment timed out: its counters mean nothing any more, every later barrier fails at once
	int64_t creator_pid;              // the creating rank's process (a segment whose creator is gone is a leftover of a crashed job)
	uint64_t user[495]; // the owner's words (the cuts' hashes of the host-vector form, the slots of the control plane)
};
static_assert(sizeof(Header) <= kHeaderBytes, "header page");
static_assert(std::atomic<uint32_t>::is_always_lock_free && std::atomic<uint64_t>::is_always_lock_free, "shared-memory atomics");

struct Segment
{
	std::string name;
	int fd = -1;
	void* map = nullptr;
	size_t map_bytes = 0;
	Header* hdr = nullptr;
	char* payload = nullptr;
	int rank = 0, nranks = 1;
	double timeout_s = 180.0;
	bool (*stale_check)(Segment&) = nullptr; // set during open() on ranks > 0: has the name been given to another segment meanwhile?
};
// is the file behind s.fd still what s.name refers to?  (false: the name is gone -- rank 0 removes it after the first barrier -- or the same)
inline bool name_moved_on(Segment& s)
{
	struct stat mine, now;
	if (fstat(s.fd, &mine) != 0)
		return false;
	const std::string path = "/dev/shm" + s.name;
	if (stat(path.c_str(), &now) != 0)
		return false;
	return now.st_ino != mine.st_ino || now.st_dev != mine.st_dev;
}

// sense-reversing barrier of all ranks; fails after the deadline instead of hanging
inline dg_status barrier(Segment& s)
{
	if (s.nranks <= 1)
		return DG_OK;
	Header* h = s.hdr;
	if (h->failed.load(std::memory_order_acquire) != 0u)
		return fail(DG_ERR_HIP, "shared-memory barrier: an earlier barrier of this segment timed out (rank %d of %d): the segment is unusable", s.rank, s.nranks);
	const uint32_t gen = h->generation.load(std::memory_order_acquire);
	if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)s.nranks)
	{
		h->arrived.store(0, std::memory_order_relaxed);
		h->generation.store(gen + 1, std::memory_order_release);
		return DG_OK;
	}
	const auto t0 = std::chrono::steady_clock::now();
	for (uint64_t spins = 0; h->generation.load(std::memory_order_acquire) == gen; ++spins)
	{
		if (spins < 2000)
			continue;
		(void)sched_yield();
		if ((spins & 1023) == 0 && h->failed.load(std::memory_order_acquire) != 0u)
			return fail(DG_ERR_HIP, "shared-memory barrier: another rank gave up on this segment (rank %d of %d)", s.rank, s.nranks);
		if ((spins & 1023) == 0 && s.stale_check && s.stale_check(s))
			return DG_ERR_INVALID; // (open(): the name now belongs to another segment -- the caller detaches and looks again)
		if ((spins & 1023) == 0 && s.timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s.timeout_s)
		{
			h->failed.store(1u, std::memory_order_release); // (`arrived` stays incremented: nobody may trust the counters again)
			return fail(DG_ERR_HIP, "shared-memory barrier: the other ranks did not arrive within %.0f s (rank %d of %d): is every rank running?", s.timeout_s,
						s.rank, s.nranks);
		}
	}
	return DG_OK;
}

inline void close(Segment& s)
{
	if (s.map)
		(void)munmap(s.map, s.map_bytes);
	if (s.fd >= 0)
		(void)::close(s.fd);
	s.map = nullptr;
	s.fd = -1;
}

// Collective over the ranks: rank 0 creates "/name" with `payload_bytes` behind the header page, the others wait for it to
// appear at its full size and to be initialised; everybody maps it, meets at the barrier, and rank 0 removes the name (the memory
// lives until the last rank unmaps it).  `name` must be unique to the job.
inline dg_status open(Segment& s, const char* name, size_t payload_bytes, uint32_t kind, int rank, int nranks,
					  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now())
{
	s.name = name[0] == '/' ? std::string(name) : "/" + std::string(name);
	s.rank = rank;
	s.nranks = nranks;
	s.timeout_s = (double)env_int("DG_COMM_TIMEOUT_S", 180, 0, 86400);
	s.map_bytes = kHeaderBytes + payload_bytes;
	// (t0: when the caller's attempt began -- a rank that detaches from a leftover and looks again keeps ITS deadline)
	auto expired = [&]() { return s.timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s.timeout_s; };
	if (rank == 0)
	{
		(void)shm_unlink(s.name.c_str()); // (a segment a crashed job left behind)
		s.fd = shm_open(s.name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
		int falloc = 0;
		if (s.fd < 0 || ftruncate(s.fd, (off_t)s.map_bytes) != 0 || (falloc = posix_fallocate(s.fd, 0, (off_t)s.map_bytes)) != 0)
		{
			const int err = falloc != 0 ? falloc : errno;
			if (s.fd >= 0)
			{
				(void)::close(s.fd);
				(void)shm_unlink(s.name.c_str());
			}
			s.fd = -1;
			return fail(DG_ERR_ALLOC, "shared-memory segment %s of %.2f GB: %s", name, (double)payload_bytes * 1e-9, std::strerror(err));
		}
	}
	else
	{
		while (true) // the creating rank may be later than this one
		{
			s.fd = shm_open(s.name.c_str(), O_RDWR, 0600);
			struct stat st;
			if (s.fd >= 0 && fstat(s.fd, &st) == 0 && (size_t)st.st_size == s.map_bytes)
				break;
			if (s.fd >= 0)
				(void)::close(s.fd);
			s.fd = -1;
			if (expired())
				return fail(DG_ERR_HIP, "shared-memory segment %s did not appear within the deadline (rank %d of %d): is rank 0 running?", name, rank, nranks);
			(void)usleep(2000);
		}
	}
	s.map = mmap(nullptr, s.map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, s.fd, 0);
	if (s.map == MAP_FAILED)
	{
		const int err = errno;
		s.map = nullptr;
		close(s);
		if (rank == 0)
			(void)shm_unlink(s.name.c_str());
		return fail(DG_ERR_ALLOC, "mapping the shared-memory segment %s: %s", name, std::strerror(err));
	}
	s.hdr = static_cast<Header*>(s.map);
	s.payload = static_cast<char*>(s.map) + kHeaderBytes;
	if (rank == 0)
	{
		s.hdr->payload_bytes = payload_bytes; // (fresh pages are zero: counters and user words start at 0)
		s.hdr->nranks = (uint32_t)nranks;
		s.hdr->kind = kind;
		s.hdr->creator_pid = (int64_t)getpid();
		s.hdr->magic.store(kMagic, std::memory_order_release);
	}
	else
	{
		// A segment of this name that a crashed job left behind (rank 0 removes the name after the first barrier, so only a crash
		// during set-up leaves one) must not be taken for ours: rank 0 unlinks and re-creates the name when it arrives, and an early
		// rank that attached to the leftover would sit in a different segment until the deadline.  Three guards: the creator's
		// process must be alive, no more ranks than the job has may have attached, and while this rank waits at the first barrier
		// it keeps checking that the name still refers to the file it mapped; on any of them it detaches and looks again.
		bool stale = false;
		while (!stale && s.hdr->magic.load(std::memory_order_acquire) != kMagic)
		{
			if (expired())
			{
				close(s);
				return fail(DG_ERR_HIP, "shared-memory segment %s was never initialised by rank 0", name);
			}
			stale = name_moved_on(s);
			(void)usleep(1000);
		}
		stale = stale || s.hdr->failed.load(std::memory_order_acquire) != 0u || (s.hdr->creator_pid > 0 && kill((pid_t)s.hdr->creator_pid, 0) != 0 && errno == ESRCH) ||
				s.hdr->attached.load(std::memory_order_acquire) >= (uint32_t)nranks;
		if (!stale && (s.hdr->payload_bytes != payload_bytes || s.hdr->nranks != (uint32_t)nranks || s.hdr->kind != kind))
		{
			if (!name_moved_on(s))
			{
				close(s);
				return fail(DG_ERR_INVALID, "shared-memory segment %s was created for another size, rank count or purpose", name);
			}
			stale = true;
		}
		if (stale)
		{
			close(s);
			if (expired())
				return fail(DG_ERR_HIP, "shared-memory segment %s: only a leftover of an earlier job was found (rank %d of %d): is rank 0 running?", name, rank, nranks);
			(void)usleep(5000);
			return open(s, name, payload_bytes, kind, rank, nranks, t0);
		}
		s.stale_check = name_moved_on;
	}
	s.hdr->attached.fetch_add(1, std::memory_order_acq_rel);
	const dg_status bs = barrier(s);
	s.stale_check = nullptr;
	if (bs == DG_ERR_INVALID && rank != 0) // the name went to another segment while this rank waited in a leftover
	{
		close(s);
		return open(s, name, payload_bytes, kind, rank, nranks, t0);
	}
	if (rank == 0)
		(void)shm_unlink(s.name.c_str());
	if (bs != DG_OK)
		close(s);
	return bs;
}
} // namespace dgshm
