// CLI: render_amd [--gpus N] [scene]  (reference: src/main.cpp = Scene(path).render())
// --gpus N: one process per GPU.  The parent forks N-1 workers BEFORE anything touches the HIP runtime, rank 0 creates
// the RCCL id (rtx_comm_unique_id) and hands it to the workers through pipes; every rank loads the scene, renders its
// bands of rows on its own GPU and Scene::render() collects the image on rank 0 (rtx_gather), which writes the BMP.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#include "../../../include/rtx.h"
#include "scene.h"

namespace {

int renderRank(const std::string& path, int nRanks, int rank, const unsigned char* id)
{
	rtx_comm* comm = nullptr;
	if (nRanks > 1 && rtx_comm_create(id, nRanks, rank, rank, &comm) != RTX_OK) {
		std::fprintf(stderr, "rank %d: rtx_comm_create failed: %s\n", rank, rtx_last_error());
		return 1;
	}
	{
		if (rank != 0) { options::enableOutput = false; options::outputProgress = false; }
		options::acBuildDevice = rank;          // the acceleration structure is built on this rank's own GPU
		Scene scene(path);
		if (rank != 0) options::enableOutput = false;      // (scene files may switch it back on)
		scene.device = rank;
		scene.attachComm(comm, nRanks, rank);
		scene.render();
	}
	rtx_comm_destroy(comm);
	return 0;
}

} // namespace

int main(int argc, char** argv)
{
	int nGpus = 1;
	std::string path = "scenes/cfg1_simple_shapes.scene";
	for (int i = 1; i < argc; ++i) {
		if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) nGpus = std::atoi(argv[++i]);
		else path = argv[i];
	}
	if (nGpus <= 1) {
		Scene(path).render();
		return 0;
	}
	// fork first, HIP later: a forked child must not inherit an initialised runtime
	std::vector<int> toChild(nGpus, -1);
	std::vector<pid_t> kids;
	int rank = 0, fromParent = -1;
	for (int r = 1; r < nGpus; ++r) {
		int fd[2];
		if (pipe(fd) != 0) { std::perror("pipe"); return 1; }
		const pid_t pid = fork();
		if (pid < 0) { std::perror("fork"); return 1; }
		if (pid == 0) {
			rank = r; fromParent = fd[0]; close(fd[1]);
			for (int k = 1; k < r; ++k) close(toChild[k]);
			kids.clear();
			break;
		}
		close(fd[0]); toChild[r] = fd[1]; kids.push_back(pid);
	}
	unsigned char id[RTX_COMM_ID_BYTES];
	if (rank == 0) {
		int devices = 0;
		if (rtx_device_count(&devices) != RTX_OK || devices < nGpus) {
			std::fprintf(stderr, "--gpus %d: only %d GPU(s) visible (one rank per GPU)\n", nGpus, devices);
			std::memset(id, 0, sizeof(id));
			for (int r = 1; r < nGpus; ++r) { unsigned char bad = 0; (void)!write(toChild[r], &bad, 1); close(toChild[r]); }
			for (pid_t k : kids) waitpid(k, nullptr, 0);
			return 2;
		}
		if (rtx_comm_unique_id(id) != RTX_OK) { std::fprintf(stderr, "rtx_comm_unique_id: %s\n", rtx_last_error()); return 1; }
		for (int r = 1; r < nGpus; ++r) {
			unsigned char ok = 1;
			if (write(toChild[r], &ok, 1) != 1 || write(toChild[r], id, sizeof(id)) != (ssize_t)sizeof(id)) { std::perror("write"); return 1; }
			close(toChild[r]);
		}
	}
	else {
		unsigned char ok = 0;
		if (read(fromParent, &ok, 1) != 1 || !ok) return 2;
		size_t got = 0;
		while (got < sizeof(id)) { const ssize_t n = read(fromParent, id + got, sizeof(id) - got); if (n <= 0) return 1; got += (size_t)n; }
		close(fromParent);
	}
	int rc = renderRank(path, nGpus, rank, id);
	if (rank == 0)
		for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 3; }
	return rc;
}
