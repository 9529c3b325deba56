// Multi-GPU part of the C ABI (include/rtx.h): one process per GPU, the framebuffer shards are collected with grouped
// point-to-point RCCL transfers over xGMI (reference seam: Scene::render, scene.cpp:595-606 -- the
// reference has one address space, so "collecting the frame" has no counterpart there).
// RCCL is loaded on first use (dlopen of librccl.so.1: the copy a host process such as PyTorch has already loaded is
// reused), so the single-GPU path has no dependency on it.  Included by rtx_api.hip (uses fail / HIPCHK).
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
	void* lib = nullptr;
	ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*groupStart)() = nullptr;
	ncclResult_t (*groupEnd)() = nullptr;
	ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*allReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char* (*errorString)(ncclResult_t) = nullptr;
};
RcclApi gRccl;

int loadRccl()
{
	if (gRccl.lib) return RTX_OK;
	void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!lib) return fail(RTX_ERR_UNSUPPORTED, std::string("RCCL is not available: ") + dlerror());
	RcclApi a;
	a.lib = lib;
#define RTX_SYM(field, name) if (!(*(void**)&a.field = dlsym(lib, name))) return fail(RTX_ERR_UNSUPPORTED, "RCCL lacks " name)
	RTX_SYM(getUniqueId, "ncclGetUniqueId"); RTX_SYM(commInitRank, "ncclCommInitRank"); RTX_SYM(commDestroy, "ncclCommDestroy");
	RTX_SYM(groupStart, "ncclGroupStart"); RTX_SYM(groupEnd, "ncclGroupEnd"); RTX_SYM(send, "ncclSend"); RTX_SYM(recv, "ncclRecv");
	RTX_SYM(errorString, "ncclGetErrorString"); RTX_SYM(allReduce, "ncclAllReduce");
#undef RTX_SYM
	gRccl = a;
	return RTX_OK;
}

#define NCCLCHK(expr)                                                                                                      \
	do {                                                                                                                   \
		ncclResult_t r_ = (expr);                                                                                          \
		if (r_ != ncclSuccess) return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + gRccl.errorString(r_));            \
	} while (0)

} // namespace

struct rtx_comm {
	ncclComm_t comm = nullptr;
	int nRanks = 1, rank = 0, device = 0;
	int* flag = nullptr;      // one device word for rtx_comm_agree
};

extern "C" {

int rtx_gather_plan(uint32_t height, uint32_t band_height, uint32_t n_parts, size_t row_bytes, int bottom_up, uint32_t max_bands,
                    uint32_t* owner, size_t* offset, size_t* bytes, uint32_t* n_bands)
{
	if (!n_bands || band_height == 0 || n_parts == 0) return fail(RTX_ERR_ARG, "rtx_gather_plan: bad argument");
	uint32_t n = 0;
	for (uint32_t y0 = 0, b = 0; y0 < height; y0 += band_height, b++) {
		const uint32_t y1 = y0 + band_height < height ? y0 + band_height : height;
		if (n < max_bands && owner && offset && bytes) {
			owner[n] = b % n_parts;
			// image row y is stored at row H-1-y of a bottom-up image (util.cpp:50): the band is still one contiguous slab
			offset[n] = (size_t)(bottom_up ? height - y1 : y0) * row_bytes;
			bytes[n] = (size_t)(y1 - y0) * row_bytes;
		}
		n++;
	}
	*n_bands = n;
	return RTX_OK;
}

int rtx_comm_unique_id(void* id128)
{
	if (!id128) return fail(RTX_ERR_ARG, "id is NULL");
	int rc = loadRccl();
	if (rc) return rc;
	static_assert(sizeof(ncclUniqueId) == RTX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	ncclUniqueId id;
	NCCLCHK(gRccl.getUniqueId(&id));
	memcpy(id128, &id, sizeof(id));
	return RTX_OK;
}

int rtx_comm_create(const void* id128, int n_ranks, int rank, int device, rtx_comm** out)
{
	if (!id128 || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RTX_ERR_ARG, "rtx_comm_create: bad argument");
	*out = nullptr;
	int rc = loadRccl();
	if (rc) return rc;
	HIPCHK(hipSetDevice(device));
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	rtx_comm* c = new rtx_comm;
	c->nRanks = n_ranks; c->rank = rank; c->device = device;
	ncclResult_t r = gRccl.commInitRank(&c->comm, n_ranks, id, rank);
	if (r != ncclSuccess) { delete c; return fail(RTX_ERR_DEVICE, std::string("ncclCommInitRank: ") + gRccl.errorString(r)); }
	if (hipMalloc((void**)&c->flag, sizeof(int)) != hipSuccess) { (void)gRccl.commDestroy(c->comm); delete c; return fail(RTX_ERR_DEVICE, "hipMalloc"); }
	*out = c;
	return RTX_OK;
}

void rtx_comm_destroy(rtx_comm* c)
{
	if (!c) return;
	if (c->comm && gRccl.commDestroy) (void)gRccl.commDestroy(c->comm);
	if (c->flag) (void)hipFree(c->flag);
	delete c;
}

int rtx_comm_info(const rtx_comm* c, int* n_ranks, int* rank)
{
	if (!c) return fail(RTX_ERR_ARG, "comm is NULL");
	if (n_ranks) *n_ranks = c->nRanks;
	if (rank) *rank = c->rank;
	return RTX_OK;
}

// Do all ranks agree that things went well so far?  A rank that failed must not leave the others waiting in rtx_gather for
// bands that will never be sent: every rank calls this with its own verdict BEFORE the gather (one 4-byte ncclAllReduce, min)
// and all of them skip the gather and report the error when *all_ok comes back 0.  Synchronises `stream`.
int rtx_comm_agree(rtx_comm* c, int ok, int* all_ok, void* stream)
{
	if (!c || !all_ok) return fail(RTX_ERR_ARG, "rtx_comm_agree: NULL argument");
	*all_ok = ok ? 1 : 0;
	if (c->nRanks == 1) return RTX_OK;
	HIPCHK(hipSetDevice(c->device));
	hipStream_t st = (hipStream_t)stream;
	const int mine = ok ? 1 : 0;
	HIPCHK(hipMemcpyAsync(c->flag, &mine, sizeof(int), hipMemcpyHostToDevice, st));
	NCCLCHK(gRccl.allReduce(c->flag, c->flag, 1, ncclInt, ncclMin, c->comm, st));
	int all = 0;
	HIPCHK(hipMemcpyAsync(&all, c->flag, sizeof(int), hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	*all_ok = all;
	return RTX_OK;
}

int rtx_gather(rtx_scene* s, rtx_comm* c, void* img_dev, size_t row_bytes, int bottom_up, int root, void* stream)
{
	if (!s || !c || !img_dev || row_bytes == 0) return fail(RTX_ERR_ARG, "rtx_gather: NULL argument");
	if (root < 0 || root >= c->nRanks) return fail(RTX_ERR_ARG, "rtx_gather: bad root");
	if (c->nRanks == 1) return RTX_OK;
	const Params& p = s->params;
	if (p.bandH == 0 || (int)p.nParts != c->nRanks || (int)p.part != c->rank)
		return fail(RTX_ERR_ARG, "rtx_gather: rtx_set_row_ownership(band, n_ranks, rank) must describe this communicator");
	HIPCHK(hipSetDevice(s->device));
	const uint32_t H = p.view.height;
	const uint32_t nBands = (H + p.bandH - 1) / p.bandH;
	std::vector<uint32_t> owner(nBands);
	std::vector<size_t> off(nBands), len(nBands);
	uint32_t n = 0;
	int rc = rtx_gather_plan(H, p.bandH, p.nParts, row_bytes, bottom_up, nBands, owner.data(), off.data(), len.data(), &n);
	if (rc) return rc;
	hipStream_t st = (hipStream_t)stream;
	// every band travels as ONE message from its owner straight into its final place in root's image: xGMI links are
	// point-to-point, so the transfers of the different owners run on different links in parallel (no ring, no staging)
	NCCLCHK(gRccl.groupStart());
	for (uint32_t b = 0; b < n; b++) {
		if ((int)owner[b] == root) continue;
		char* ptr = (char*)img_dev + off[b];
		ncclResult_t r = ncclSuccess;
		if (c->rank == root) r = gRccl.recv(ptr, len[b], ncclChar, (int)owner[b], c->comm, st);
		else if ((int)owner[b] == c->rank) r = gRccl.send(ptr, len[b], ncclChar, root, c->comm, st);
		if (r != ncclSuccess) { (void)gRccl.groupEnd(); return fail(RTX_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + gRccl.errorString(r)); }
	}
	NCCLCHK(gRccl.groupEnd());
	return RTX_OK;
}

} // extern "C"
