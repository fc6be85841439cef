// dg_capi_vmm.h -- device arrays that OTHER PROCESSES can map whatever their size (DG_EXCHANGE_COPY beyond 2 GiB).
//
// hipIpcOpenMemHandle does not return for allocations above 2 GiB on the platform this was developed on (ROCm 7.2,
// dmabuf IPC).  An array allocated here is a contiguous virtual range backed by hipMemCreate chunks of at most
// kVmmChunkBytes; every chunk is exported as a POSIX file descriptor (hipMemExportToShareableHandle), the descriptors
// travel to the peers over a unix-domain socket (SCM_RIGHTS -- the only portable way to hand a descriptor to another
// process), and a peer imports and maps them side by side into a contiguous range of its own.  No reference counterpart
// (the reference is one process).  Internal to dg_capi_comm.cpp.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <ctime>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <poll.h>
#include <sys/random.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

namespace dgvmm
{
constexpr size_t kVmmChunkBytes = (size_t)512 << 20; // well below the 2 GiB beyond which whole-allocation IPC hung

struct Array // a mapped range: created here (handles owned) or imported from a peer
{
	char* base = nullptr;
	size_t bytes = 0; // mapped size = sum of the chunk sizes
	size_t chunk = 0; // size of every chunk but the last
	std::vector<hipMemGenericAllocationHandle_t> handles;
	std::vector<size_t> chunk_bytes;
	int device = -1;
};

inline hipMemAllocationProp chunk_prop(int device)
{
	hipMemAllocationProp prop;
	std::memset(&prop, 0, sizeof(prop));
	prop.type = hipMemAllocationTypePinned;
	prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = device;
	return prop;
}

inline void destroy(Array& a)
{
	if (a.base)
	{
		size_t off = 0;
		for (size_t i = 0; i < a.handles.size(); ++i)
		{
			(void)hipMemUnmap(a.base + off, a.chunk_bytes[i]);
			off += a.chunk_bytes[i];
		}
		(void)hipMemAddressFree(a.base, a.bytes);
	}
	for (hipMemGenericAllocationHandle_t h : a.handles)
		(void)hipMemRelease(h);
	a = Array();
}

inline hipError_t set_access(const Array& a)
{
	hipMemAccessDesc desc;
	std::memset(&desc, 0, sizeof(desc));
	desc.location.type = hipMemLocationTypeDevice;
	desc.location.id = a.device;
	desc.flags = hipMemAccessFlagsProtReadWrite;
	return hipMemSetAccess(a.base, a.bytes, &desc, 1);
}

// `bytes` of device memory on `device` as chunks of at most `chunk` bytes behind one contiguous range
inline hipError_t create(Array& a, size_t bytes, int device, size_t chunk = kVmmChunkBytes)
{
	const hipMemAllocationProp prop = chunk_prop(device);
	size_t gran = 0;
	hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
	if (e != hipSuccess)
		return e;
	if (gran == 0)
		gran = (size_t)2 << 20;
	chunk = std::max(gran, chunk / gran * gran);
	const size_t total = (std::max<size_t>(bytes, 1) + gran - 1) / gran * gran;
	a.device = device;
	a.chunk = chunk;
	void* va = nullptr;
	e = hipMemAddressReserve(&va, total, gran, nullptr, 0);
	if (e != hipSuccess)
		return e;
	a.base = static_cast<char*>(va);
	a.bytes = total;
	size_t off = 0;
	while (off < total && e == hipSuccess)
	{
		const size_t len = std::min(chunk, total - off);
		hipMemGenericAllocationHandle_t h = nullptr;
		e = hipMemCreate(&h, len, &prop, 0);
		if (e != hipSuccess)
			break;
		a.handles.push_back(h);
		a.chunk_bytes.push_back(0); // (not mapped yet: destroy() must not unmap it)
		e = hipMemMap(a.base + off, len, 0, h, 0);
		if (e != hipSuccess)
			break;
		a.chunk_bytes.back() = len;
		off += len;
	}
	if (e == hipSuccess)
		e = set_access(a);
	if (e != hipSuccess)
	{
		// (unmap what was mapped: chunk_bytes of an unmapped chunk is 0, hipMemUnmap of 0 bytes is skipped)
		Array dead = a;
		a = Array();
		size_t o = 0;
		for (size_t i = 0; i < dead.handles.size(); ++i)
		{
			if (dead.chunk_bytes[i])
				(void)hipMemUnmap(dead.base + o, dead.chunk_bytes[i]);
			o += dead.chunk_bytes[i];
			(void)hipMemRelease(dead.handles[i]);
		}
		(void)hipMemAddressFree(dead.base, dead.bytes);
	}
	return e;
}

// one POSIX descriptor per chunk (the caller closes them)
inline hipError_t export_fds(const Array& a, std::vector<int>& fds)
{
	for (hipMemGenericAllocationHandle_t h : a.handles)
	{
		int fd = -1;
		const hipError_t e = hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0);
		if (e != hipSuccess || fd < 0)
		{
			for (int f : fds)
				(void)close(f);
			fds.clear();
			return e != hipSuccess ? e : hipErrorInvalidValue;
		}
		fds.push_back(fd);
	}
	return hipSuccess;
}

// a peer's chunks (descriptors in order, sizes as the peer announced them) mapped side by side on `device`
inline hipError_t import_fds(Array& a, const std::vector<int>& fds, const std::vector<size_t>& sizes, int device)
{
	a.device = device;
	size_t total = 0;
	for (size_t s : sizes)
		total += s;
	void* va = nullptr;
	hipError_t e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
	if (e != hipSuccess)
		return e;
	a.base = static_cast<char*>(va);
	a.bytes = total;
	a.chunk = sizes.empty() ? 0 : sizes[0];
	size_t off = 0;
	for (size_t i = 0; i < fds.size() && e == hipSuccess; ++i)
	{
		hipMemGenericAllocationHandle_t h = nullptr;
		// The HIP runtime has read this argument both ways over its releases: as a POINTER to the descriptor and (like
		// CUDA) as the descriptor's VALUE.  Pointer first -- a runtime that wants the value sees a number that is no open
		// descriptor and fails cleanly, whereas a runtime that wants the pointer would dereference the small integer.
		int fd = fds[i];
		e = hipMemImportFromShareableHandle(&h, &fd, hipMemHandleTypePosixFileDescriptor);
		if (e != hipSuccess)
		{
			(void)hipGetLastError();
			e = hipMemImportFromShareableHandle(&h, reinterpret_cast<void*>(static_cast<uintptr_t>(fds[i])), hipMemHandleTypePosixFileDescriptor);
		}
		if (e != hipSuccess)
			break;
		a.handles.push_back(h);
		a.chunk_bytes.push_back(0);
		e = hipMemMap(a.base + off, sizes[i], 0, h, 0);
		if (e != hipSuccess)
			break;
		a.chunk_bytes.back() = sizes[i];
		off += sizes[i];
	}
	if (e == hipSuccess)
		e = set_access(a);
	if (e != hipSuccess)
	{
		Array dead = a;
		a = Array();
		size_t o = 0;
		for (size_t i = 0; i < dead.handles.size(); ++i)
		{
			if (dead.chunk_bytes[i])
				(void)hipMemUnmap(dead.base + o, dead.chunk_bytes[i]);
			o += dead.chunk_bytes[i];
			(void)hipMemRelease(dead.handles[i]);
		}
		(void)hipMemAddressFree(dead.base, dead.bytes);
	}
	return e;
}

// ---- descriptors between processes: an abstract unix socket per exporting rank ---------------------------------------
inline sockaddr_un abstract_address(const std::string& name, socklen_t* len)
{
	sockaddr_un addr;
	std::memset(&addr, 0, sizeof(addr));
	addr.sun_family = AF_UNIX;
	const size_t n = std::min(name.size(), sizeof(addr.sun_path) - 2);
	std::memcpy(addr.sun_path + 1, name.data(), n); // sun_path[0] == 0: abstract namespace, nothing to unlink
	*len = (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + n);
	return addr;
}

constexpr int kFdsPerMessage = 32;

inline bool send_fds(int sock, const int* fds, int n)
{
	char payload = (char)n;
	iovec iov = {&payload, 1};
	alignas(cmsghdr) char control[CMSG_SPACE(sizeof(int) * kFdsPerMessage)];
	std::memset(control, 0, sizeof(control));
	msghdr msg;
	std::memset(&msg, 0, sizeof(msg));
	msg.msg_iov = &iov;
	msg.msg_iovlen = 1;
	msg.msg_control = control;
	msg.msg_controllen = CMSG_SPACE(sizeof(int) * (size_t)n);
	cmsghdr* cm = CMSG_FIRSTHDR(&msg);
	cm->cmsg_level = SOL_SOCKET;
	cm->cmsg_type = SCM_RIGHTS;
	cm->cmsg_len = CMSG_LEN(sizeof(int) * (size_t)n);
	std::memcpy(CMSG_DATA(cm), fds, sizeof(int) * (size_t)n);
	ssize_t r;
	do
		r = sendmsg(sock, &msg, MSG_NOSIGNAL);
	while (r < 0 && errno == EINTR);
	return r == 1;
}

inline bool recv_fds(int sock, std::vector<int>& out, int timeout_ms)
{
	pollfd p = {sock, POLLIN, 0};
	int pr;
	do
		pr = poll(&p, 1, timeout_ms);
	while (pr < 0 && errno == EINTR);
	if (pr <= 0)
		return false;
	char payload = 0;
	iovec iov = {&payload, 1};
	alignas(cmsghdr) char control[CMSG_SPACE(sizeof(int) * kFdsPerMessage)];
	msghdr msg;
	std::memset(&msg, 0, sizeof(msg));
	msg.msg_iov = &iov;
	msg.msg_iovlen = 1;
	msg.msg_control = control;
	msg.msg_controllen = sizeof(control);
	ssize_t r;
	do
		r = recvmsg(sock, &msg, MSG_CMSG_CLOEXEC);
	while (r < 0 && errno == EINTR);
	if (r != 1)
		return false;
	for (cmsghdr* cm = CMSG_FIRSTHDR(&msg); cm != nullptr; cm = CMSG_NXTHDR(&msg, cm))
		if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS)
		{
			const size_t n = (cm->cmsg_len - CMSG_LEN(0)) / sizeof(int);
			const size_t at = out.size();
			out.resize(at + n);
			std::memcpy(out.data() + at, CMSG_DATA(cm), n * sizeof(int));
		}
	return true;
}

// 64 random bits for the socket's name (getrandom; /dev/urandom; as a last resort clock and address noise)
inline uint64_t random_token()
{
	uint64_t t = 0;
	if (getrandom(&t, sizeof(t), 0) == (ssize_t)sizeof(t) && t != 0)
		return t;
	if (FILE* f = std::fopen("/dev/urandom", "rb"))
	{
		const size_t got = std::fread(&t, sizeof(t), 1, f);
		std::fclose(f);
		if (got == 1 && t != 0)
			return t;
	}
	timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ((uint64_t)ts.tv_nsec * 6364136223846793005ull) ^ ((uint64_t)ts.tv_sec << 32) ^ (uint64_t)(uintptr_t)&t ^ ((uint64_t)getpid() << 17);
}

// Serves this rank's descriptors to `expect` peers, one connection each, on a thread of its own.  The descriptors give read / write
// access to the field's device memory, and an abstract unix socket carries no permissions: the name ends in 64 random bits that
// travel through the communicator's control plane only, the server accepts nothing before allow() has named the peers' process
// ids, and a connection whose SO_PEERCRED is another user's or another process' is closed WITHOUT counting as one of the
// `expect` (a stranger can neither obtain descriptors nor starve a real peer of its turn).
struct FdServer
{
	int listen_fd = -1;
	std::thread thread;
	std::atomic<bool> stop{false}, armed{false};
	std::vector<int> fds;
	std::vector<int32_t> allowed; // written before `armed` is set

	bool start(const std::string& name, const std::vector<int>& descriptors, int expect)
	{
		listen_fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
		if (listen_fd < 0)
			return false;
		socklen_t len = 0;
		const sockaddr_un addr = abstract_address(name, &len);
		if (bind(listen_fd, reinterpret_cast<const sockaddr*>(&addr), len) != 0 || listen(listen_fd, 64) != 0)
		{
			(void)close(listen_fd);
			listen_fd = -1;
			return false; // (the caller still owns the descriptors: exactly one owner closes them)
		}
		fds = descriptors;
		thread = std::thread([this, expect]() {
			int served = 0;
			while (served < expect && !stop.load())
			{
				if (!armed.load(std::memory_order_acquire)) // (the peers learn the name in the exchange that also tells us who they are)
				{
					(void)usleep(500);
					continue;
				}
				pollfd p = {listen_fd, POLLIN, 0};
				if (poll(&p, 1, 50) <= 0)
					continue;
				const int conn = accept4(listen_fd, nullptr, nullptr, SOCK_CLOEXEC);
				if (conn < 0)
					continue;
				ucred cred;
				socklen_t cl = sizeof(cred);
				const bool known = getsockopt(conn, SOL_SOCKET, SO_PEERCRED, &cred, &cl) == 0 && cred.uid == geteuid() &&
								   std::find(allowed.begin(), allowed.end(), (int32_t)cred.pid) != allowed.end();
				if (!known)
				{
					(void)close(conn);
					continue;
				}
				for (size_t at = 0; at < fds.size(); at += kFdsPerMessage)
					if (!send_fds(conn, fds.data() + at, (int)std::min<size_t>(kFdsPerMessage, fds.size() - at)))
						break;
				(void)close(conn);
				++served;
			}
		});
		return true;
	}
	void allow(const std::vector<int32_t>& pids)
	{
		allowed = pids;
		armed.store(true, std::memory_order_release);
	}
	void finish() // (after every peer has reported: nobody connects any more)
	{
		stop.store(true);
		if (thread.joinable())
			thread.join();
		if (listen_fd >= 0)
			(void)close(listen_fd);
		listen_fd = -1;
		for (int f : fds)
			(void)close(f);
		fds.clear();
	}
	~FdServer() { finish(); }
};

// the `n` descriptors the peer behind `name` serves (the caller closes them)
inline bool fetch_fds(const std::string& name, int n, std::vector<int>& out, int timeout_ms)
{
	const int sock = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
	if (sock < 0)
		return false;
	socklen_t len = 0;
	const sockaddr_un addr = abstract_address(name, &len);
	bool ok = false;
	for (int waited = 0; waited <= timeout_ms && !ok; waited += 20) // (the listener exists before its name travels; retries cover a full backlog)
	{
		if (connect(sock, reinterpret_cast<const sockaddr*>(&addr), len) == 0)
			ok = true;
		else if (errno == ECONNREFUSED || errno == EAGAIN || errno == EINTR)
			(void)usleep(20000);
		else
			break;
	}
	while (ok && (int)out.size() < n)
		if (!recv_fds(sock, out, timeout_ms))
			ok = false;
	(void)close(sock);
	if (!ok || (int)out.size() != n)
	{
		for (int f : out)
			(void)close(f);
		out.clear();
		return false;
	}
	return true;
}
} // namespace dgvmm
