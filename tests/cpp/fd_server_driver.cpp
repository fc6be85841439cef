// tests/cpp/fd_server_driver.cpp -- TEST TOOLING: the descriptor server of the copy exchange (discregrid_amd/csrc/dg_capi_vmm.h: FdServer /
// fetch_fds) without a GPU.  The server holds ONE descriptor (a pipe's write end) for ONE expected peer.  A stranger -- a process that
// knows the socket's name but is not among the allowed process ids -- connects first: it must receive nothing and must not use up the
// peer's turn.  Then the peer fetches the descriptor and writes through it; the parent reads what it wrote.  Prints one JSON line.
#include "../../discregrid_amd/csrc/dg_capi_vmm.h"

#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>

int main()
{
	int data[2], go[2];
	if (pipe(data) != 0 || pipe(go) != 0)
		return 2;
	const uint64_t token = dgvmm::random_token(), token2 = dgvmm::random_token();
	const std::string name = "dg_vmm_test_" + std::to_string((int)getpid()) + "_" + std::to_string(token);
	dgvmm::FdServer server;
	std::vector<int> fds = {dup(data[1])};
	const bool started = server.start(name, fds, 1);
	// a second server cannot take the name (what a squatter would see)
	dgvmm::FdServer second;
	std::vector<int> none;
	const bool name_taken = !second.start(name, none, 1);
	const pid_t stranger = fork();
	if (stranger == 0)
	{
		std::vector<int> got;
		const bool ok = dgvmm::fetch_fds(name, 1, got, 1500); // (connects: the listener exists; is never served)
		_exit(ok ? 1 : 0);
	}
	const pid_t peer = fork();
	if (peer == 0)
	{
		char c;
		if (read(go[0], &c, 1) != 1)
			_exit(3);
		std::vector<int> got;
		if (!dgvmm::fetch_fds(name, 1, got, 5000) || got.size() != 1)
			_exit(4);
		const char msg[] = "through the served descriptor";
		_exit(write(got[0], msg, sizeof(msg)) == (ssize_t)sizeof(msg) ? 0 : 5);
	}
	(void)usleep(300000); // the stranger is connected (or queued) by now; the server is not armed yet
	server.allow({(int32_t)peer});
	int st_stranger = -1, st_peer = -1;
	(void)waitpid(stranger, &st_stranger, 0); // refused: its connection was closed without a descriptor
	if (write(go[1], "g", 1) != 1)
		return 2;
	(void)waitpid(peer, &st_peer, 0);
	char buf[64] = {0};
	ssize_t n = 0;
	if (WIFEXITED(st_peer) && WEXITSTATUS(st_peer) == 0)
		n = read(data[0], buf, sizeof(buf) - 1);
	server.finish();
	std::printf("{\"started\": %s, \"name_taken\": %s, \"tokens_differ\": %s, \"stranger_got_nothing\": %s, \"peer_exit\": %d, \"peer_wrote\": \"%s\", \"bytes\": %d}\n",
				started ? "true" : "false", name_taken ? "true" : "false", token != token2 && token != 0 ? "true" : "false",
				(WIFEXITED(st_stranger) && WEXITSTATUS(st_stranger) == 0) ? "true" : "false", WIFEXITED(st_peer) ? WEXITSTATUS(st_peer) : -1, buf, (int)n);
	return 0;
}
