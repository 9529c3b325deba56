// Acceleration-structure build on the device: reproduces the reference's spatial-split
// builder (objects.cpp:470-526 recursion, 633-689 cost / bisection, 737-760 partition) bit for bit, but
// level-synchronously instead of recursively:
//
//   level d holds every node of depth d with its triangle ids in one node-major buffer (order inside a node =
//   the reference's vector order).  Per level: pick the axis, run the reference's bisection for all nodes at once
//   (each step = one counting pass over the level's ids: #{lo <= s} / #{hi >= s} at the two probe positions, reduced
//   per wave segment and added to the node with one atomic per (wave, node)), count the two sides at the final
//   position, decide leaf / inner (objects.cpp:477, 498), then scatter the ids of inner nodes stably into the
//   next level (global exclusive scans of the two side flags give the ranks).  When no inner node is left, subtree
//   sizes are summed bottom-up, pre-order indices and first-reference offsets are handed out top-down, and the
//   nodes / references are written in the reference's DFS-left-first order (objects.cpp:601-629) -- the layout
//   rtx_mesh and the kernels use.
//
// All arithmetic that decides the topology is the reference's own: fp32, no contraction, counts converted to
// float in the cost (objects.cpp:672), the 1.5x duplication stop in double (objects.cpp:498).
// Included by rtx_api.hip (single translation unit).

namespace bvhb {

struct BNode {
	float lo[3], hi[3];          // bounds[0], bounds[1]
	uint32_t begin, count;       // segment of this level's id buffer
	float left, right;           // bisection interval (objects.cpp:676-689)
	float s, s1, s2;             // split position; probe positions mid -+ 0.05
	int32_t axis;
	uint32_t state;              // kLeaf / kActive / kFinalCount / kFinished
	uint32_t cnt[4];             // nLeft(s1), nRight(s1), nLeft(s2), nRight(s2); after the final count: nLeft(s), nRight(s)
	uint32_t hasCounts;
	uint32_t child;              // index of the left child in the next level (right = child + 1)
	uint32_t childOff;           // first slot of the left child's ids in the next level's buffer
	uint32_t subNodes, subRefs;  // size of the subtree in nodes / leaf references
	uint32_t pre, refStart;      // pre-order index; first leaf reference of the subtree
};

enum : uint32_t { kLeaf = 0, kActive = 1, kFinalCount = 2, kFinished = 3 };

// min / max of the three vertices per axis: "some vertex <= s" == lo <= s, "some vertex >= s" == hi >= s
// (objects.cpp:640-668, 741-757)
__global__ void triExtentKernel(const float* __restrict__ pos, uint32_t n, float* __restrict__ ext)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float* p = pos + (size_t)i * 9;
	for (int ax = 0; ax < 3; ax++) {
		const float a = p[ax], b = p[3 + ax], c = p[6 + ax];
		ext[(size_t)(2 * ax) * n + i] = fminf(a, fminf(b, c));
		ext[(size_t)(2 * ax + 1) * n + i] = fmaxf(a, fmaxf(b, c));
	}
}

__global__ void iotaKernel(uint32_t* ids, uint32_t* owner, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { ids[i] = i; owner[i] = 0; }
}

// objects.cpp:477 (leaf by size) and 486-490 (axis = strictly longest dimension, else y over z)
__global__ void levelInitKernel(BNode* nodes, uint32_t n, uint32_t depth, int32_t penalty)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	BNode& nd = nodes[i];
	nd.hasCounts = 0; nd.child = 0; nd.childOff = 0;
	nd.cnt[0] = nd.cnt[1] = nd.cnt[2] = nd.cnt[3] = 0;
	const unsigned long long limit = (unsigned long long)depth * (unsigned long long)(long long)penalty;   // depth * (size_t)penalty
	if ((unsigned long long)nd.count <= limit) { nd.state = kLeaf; nd.axis = -1; return; }
	const float dx = nd.hi[0] - nd.lo[0], dy = nd.hi[1] - nd.lo[1], dz = nd.hi[2] - nd.lo[2];
	int ax = 2;
	if (dx > dy && dx > dz) ax = 0;
	else if (dy > dz) ax = 1;
	nd.axis = ax;
	nd.left = nd.lo[ax]; nd.right = nd.hi[ax];
	nd.state = kActive;
}

// One step of the bisection for every node (objects.cpp:676-689): consume the counts of the previous probe pair,
// then either finish (interval narrower than 0.1) or publish the next probe pair.
__global__ void bisectStepKernel(BNode* nodes, uint32_t n, uint32_t* unfinished)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	BNode& nd = nodes[i];
	if (nd.state == kFinalCount) { nd.state = kFinished; return; }
	if (nd.state != kActive) return;
	const float mn = nd.lo[nd.axis], mx = nd.hi[nd.axis];
	if (nd.hasCounts) {
		// nLeft * (s - min) + nRight * (max - s), counts promoted to float (objects.cpp:672)
		const float c1 = (float)(int)nd.cnt[0] * (nd.s1 - mn) + (float)(int)nd.cnt[1] * (mx - nd.s1);
		const float c2 = (float)(int)nd.cnt[2] * (nd.s2 - mn) + (float)(int)nd.cnt[3] * (mx - nd.s2);
		const float mid = nd.right - (nd.right - nd.left) / 2;
		if (c1 < c2) nd.right = mid; else nd.left = mid;
	}
	const float mid = nd.right - (nd.right - nd.left) / 2;
	nd.cnt[0] = nd.cnt[1] = nd.cnt[2] = nd.cnt[3] = 0;
	if (nd.right - nd.left < 0.1f) { nd.s = mid; nd.s1 = mid; nd.s2 = mid; nd.state = kFinalCount; }
	else { nd.s1 = mid - 0.05f; nd.s2 = mid + 0.05f; nd.hasCounts = 1; }
	atomicAdd(unfinished, 1u);
}

// Counting pass over the ids of the level.  A block covers 256 * perThread consecutive ids (coalesced; perThread grows
// with the level's size, so that big levels have few blocks per node).
// Ids of the node the block STARTS in are counted in registers and reduced once per block (wave shuffles, LDS, one
// atomic per counter) -- at the top of the tree, where a few nodes own all ids, this replaces thousands of atomics
// on the same four words; ids of other nodes (deep levels: many small nodes per block) are reduced per wave segment
// with ballots and added with one atomic per (wave, node, counter).
__global__ void __launch_bounds__(256) countKernel(BNode* nodes, const uint32_t* __restrict__ owner, const uint32_t* __restrict__ ids, uint32_t total,
                                                   const float* __restrict__ ext, uint32_t nTris, uint32_t perThread)
{
	__shared__ uint32_t part[4][4];
	const uint32_t base = blockIdx.x * 256 * perThread;
	const uint32_t home = owner[base];                    // base < total by construction of the grid
	const int lane = (int)__lane_id();
	uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
	for (uint32_t k = 0; k < perThread; k++) {
		const uint32_t e = base + k * 256 + threadIdx.x;
		bool active = false;
		uint32_t node = 0;
		bool p0 = false, p1 = false, p2 = false, p3 = false;
		if (e < total) {
			node = owner[e];
			const BNode& nd = nodes[node];
			const uint32_t st = nd.state;
			if (st == kActive || st == kFinalCount) {
				active = true;
				const uint32_t id = ids[e];
				const float l = ext[(size_t)(2 * nd.axis) * nTris + id], h = ext[(size_t)(2 * nd.axis + 1) * nTris + id];
				p0 = l <= nd.s1; p1 = h >= nd.s1; p2 = l <= nd.s2; p3 = h >= nd.s2;
			}
		}
		if (active && node == home) { h0 += p0; h1 += p1; h2 += p2; h3 += p3; active = false; }
		unsigned long long todo = __ballot(active);
		while (todo) {
			const int leader = __builtin_ctzll(todo);
			const uint32_t ln = (uint32_t)__builtin_amdgcn_readlane((int)node, leader);
			const bool mine = active && node == ln;
			const unsigned long long same = __ballot(mine);
			const uint32_t c0 = (uint32_t)__popcll(__ballot(mine && p0)), c1 = (uint32_t)__popcll(__ballot(mine && p1));
			const uint32_t c2 = (uint32_t)__popcll(__ballot(mine && p2)), c3 = (uint32_t)__popcll(__ballot(mine && p3));
			if (lane == leader) {
				BNode& nd = nodes[ln];
				if (c0) atomicAdd(&nd.cnt[0], c0);
				if (c1) atomicAdd(&nd.cnt[1], c1);
				if (nd.state == kActive) { if (c2) atomicAdd(&nd.cnt[2], c2); if (c3) atomicAdd(&nd.cnt[3], c3); }
			}
			todo &= ~same;
		}
	}
	// the home node's counts: wave reduction, then across the four waves
	for (int d = 32; d >= 1; d >>= 1) { h0 += __shfl_down(h0, d, 64); h1 += __shfl_down(h1, d, 64); h2 += __shfl_down(h2, d, 64); h3 += __shfl_down(h3, d, 64); }
	if (lane == 0) { uint32_t* q = part[threadIdx.x >> 6]; q[0] = h0; q[1] = h1; q[2] = h2; q[3] = h3; }
	__syncthreads();
	if (threadIdx.x < 4) {
		const uint32_t c = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
		BNode& nd = nodes[home];
		if (c && (threadIdx.x < 2 || nd.state == kActive)) atomicAdd(&nd.cnt[threadIdx.x], c);
	}
}

// objects.cpp:498: leaf if one side is empty or the split duplicates too much (>= 1.5x, compared in double)
__global__ void decideKernel(BNode* nodes, uint32_t n, uint32_t* childCount, uint32_t* childIds)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	BNode& nd = nodes[i];
	uint32_t cc = 0, ci = 0;
	if (nd.state == kFinished) {
		const uint32_t nl = nd.cnt[0], nr = nd.cnt[1];
		const bool leaf = nl == 0 || nr == 0 || (double)((unsigned long long)nl + nr) >= (double)nd.count * 1.5;
		if (leaf) nd.state = kLeaf;
		else { cc = 2; ci = nl + nr; }
	}
	childCount[i] = cc; childIds[i] = ci;
}

// ---- exclusive scan of uint32 (blocks of 1024 elements, recursive over the block sums) ----
constexpr uint32_t kScanBlock = 1024;

__global__ void __launch_bounds__(256) scanBlockKernel(uint32_t* data, uint32_t n, uint32_t* blockSums)
{
	__shared__ uint32_t waveSum[4];
	const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4;
	uint32_t v[4];
	for (int k = 0; k < 4; k++) v[k] = base + k < n ? data[base + k] : 0u;
	const uint32_t mine = v[0] + v[1] + v[2] + v[3];
	// inclusive scan of `mine` across the wave
	uint32_t x = mine;
	const int lane = (int)__lane_id();
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
		if (lane >= d) x += y;
	}
	const int wave = threadIdx.x >> 6;
	if (lane == 63) waveSum[wave] = x;
	__syncthreads();
	uint32_t off = 0;
	for (int w = 0; w < wave; w++) off += waveSum[w];
	uint32_t run = off + x - mine;
	for (int k = 0; k < 4; k++) { if (base + k < n) data[base + k] = run; run += v[k]; }
	if (threadIdx.x == 255 && blockSums) blockSums[blockIdx.x] = off + x;
}

__global__ void scanAddKernel(uint32_t* data, uint32_t n, const uint32_t* blockOffsets)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) data[i] += blockOffsets[i / kScanBlock];
}

// side flags of every id whose node is split (objects.cpp:741-757): flagL = some vertex <= s, flagR = some vertex >= s
__global__ void flagKernel(const BNode* __restrict__ nodes, const uint32_t* __restrict__ owner, const uint32_t* __restrict__ ids, uint32_t total,
                           const float* __restrict__ ext, uint32_t nTris, uint32_t* flagL, uint32_t* flagR)
{
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= total) return;
	const BNode& nd = nodes[owner[e]];
	uint32_t fl = 0, fr = 0;
	if (nd.state == kFinished) {
		const uint32_t id = ids[e];
		fl = ext[(size_t)(2 * nd.axis) * nTris + id] <= nd.s;
		fr = ext[(size_t)(2 * nd.axis + 1) * nTris + id] >= nd.s;
	}
	flagL[e] = fl; flagR[e] = fr;
}

// stable scatter into the next level: [left ids][right ids] per split node, order preserved (objects.cpp:737-760)
__global__ void scatterKernel(const BNode* __restrict__ nodes, const uint32_t* __restrict__ owner, const uint32_t* __restrict__ ids, uint32_t total,
                              const float* __restrict__ ext, uint32_t nTris, const uint32_t* __restrict__ scanL, const uint32_t* __restrict__ scanR,
                              uint32_t* nextIds, uint32_t* nextOwner)
{
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= total) return;
	const BNode& nd = nodes[owner[e]];
	if (nd.state != kFinished) return;
	const uint32_t id = ids[e];
	if (ext[(size_t)(2 * nd.axis) * nTris + id] <= nd.s) {
		const uint32_t dst = nd.childOff + (scanL[e] - scanL[nd.begin]);
		nextIds[dst] = id; nextOwner[dst] = nd.child;
	}
	if (ext[(size_t)(2 * nd.axis + 1) * nTris + id] >= nd.s) {
		const uint32_t dst = nd.childOff + nd.cnt[0] + (scanR[e] - scanR[nd.begin]);
		nextIds[dst] = id; nextOwner[dst] = nd.child + 1;
	}
}

// children of every split node: the parent's box cut at s on the split axis (objects.cpp:510-521)
__global__ void childrenKernel(BNode* nodes, uint32_t n, const uint32_t* __restrict__ childIndex, const uint32_t* __restrict__ childOff, BNode* next)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	BNode& nd = nodes[i];
	if (nd.state != kFinished) return;
	nd.child = childIndex[i]; nd.childOff = childOff[i];
	BNode l, r;
	memset(&l, 0, sizeof(l)); memset(&r, 0, sizeof(r));
	for (int c = 0; c < 3; c++) { l.lo[c] = r.lo[c] = nd.lo[c]; l.hi[c] = r.hi[c] = nd.hi[c]; }
	l.hi[nd.axis] = nd.s; r.lo[nd.axis] = nd.s;
	l.begin = nd.childOff; l.count = nd.cnt[0];
	r.begin = nd.childOff + nd.cnt[0]; r.count = nd.cnt[1];
	next[nd.child] = l; next[nd.child + 1] = r;
}

__global__ void subtreeKernel(BNode* nodes, uint32_t n, const BNode* __restrict__ next)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	BNode& nd = nodes[i];
	if (nd.state == kLeaf) { nd.subNodes = 1; nd.subRefs = nd.count; }
	else { nd.subNodes = 1 + next[nd.child].subNodes + next[nd.child + 1].subNodes; nd.subRefs = next[nd.child].subRefs + next[nd.child + 1].subRefs; }
}

__global__ void preorderKernel(const BNode* __restrict__ nodes, uint32_t n, BNode* next)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const BNode& nd = nodes[i];
	if (nd.state == kLeaf) return;
	BNode& l = next[nd.child];
	BNode& r = next[nd.child + 1];
	l.pre = nd.pre + 1; l.refStart = nd.refStart;
	r.pre = nd.pre + 1 + l.subNodes; r.refStart = nd.refStart + l.subRefs;
}

__global__ void emitNodesKernel(const BNode* __restrict__ nodes, uint32_t n, float* bounds, int32_t* skip, int32_t* leafBegin, int32_t* leafCount)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const BNode& nd = nodes[i];
	const uint32_t p = nd.pre;
	for (int c = 0; c < 3; c++) { bounds[(size_t)p * 6 + c] = nd.lo[c]; bounds[(size_t)p * 6 + 3 + c] = nd.hi[c]; }
	skip[p] = (int32_t)(p + nd.subNodes);
	const bool leaf = nd.state == kLeaf;
	leafBegin[p] = leaf ? (int32_t)nd.refStart : -1;
	leafCount[p] = leaf ? (int32_t)nd.count : -1;
}

__global__ void emitRefsKernel(const BNode* __restrict__ nodes, const uint32_t* __restrict__ owner, const uint32_t* __restrict__ ids, uint32_t total, uint32_t* refs)
{
	const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= total) return;
	const BNode& nd = nodes[owner[e]];
	if (nd.state == kLeaf) refs[nd.refStart + (e - nd.begin)] = ids[e];
}

} // namespace bvhb

struct rtx_bvh {
	int device = 0;
	uint32_t nNodes = 0, nRefs = 0, maxDepth = 0, launches = 0;
	float buildMs = 0;
	float* bounds = nullptr; int32_t* skip = nullptr; int32_t* leafBegin = nullptr; int32_t* leafCount = nullptr; uint32_t* refs = nullptr;
};

namespace {

struct BvhLevel { bvhb::BNode* nodes = nullptr; uint32_t n = 0; uint32_t* ids = nullptr; uint32_t* owner = nullptr; uint32_t total = 0; };

inline unsigned gridFor(uint32_t n, unsigned block = 256) { return n ? (n + block - 1) / block : 1; }

// exclusive scan in place; `tmp` holds the block sums of every recursion level (capacity >= n / 1024 * 1.01 + 8)
int scanExclusive(uint32_t* data, uint32_t n, uint32_t* tmp, hipStream_t st, uint32_t& launches)
{
	if (n == 0) return RTX_OK;
	const uint32_t blocks = (n + bvhb::kScanBlock - 1) / bvhb::kScanBlock;
	hipLaunchKernelGGL(bvhb::scanBlockKernel, dim3(blocks), dim3(256), 0, st, data, n, blocks > 1 ? tmp : nullptr);
	launches++;
	if (blocks > 1) {
		int rc = scanExclusive(tmp, blocks, tmp + blocks, st, launches);
		if (rc) return rc;
		hipLaunchKernelGGL(bvhb::scanAddKernel, dim3(gridFor(n)), dim3(256), 0, st, data, n, tmp);
		launches++;
	}
	return RTX_OK;
}

} // namespace

extern "C" {

int rtx_bvh_build(const float* tri_pos, uint32_t n_tris, const float* root_lo, const float* root_hi, int32_t ac_penalty, int device, rtx_bvh** out)
{
	using namespace bvhb;
	if (!out) return fail(RTX_ERR_ARG, "out is NULL");
	*out = nullptr;
	if ((n_tris && !tri_pos) || !root_lo || !root_hi) return fail(RTX_ERR_ARG, "tri_pos / root bounds missing");
	if (n_tris > 0x7fffffffu / 2) return fail(RTX_ERR_ARG, "too many triangles");
	int nDev = 0;
	if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) return fail(RTX_ERR_NO_DEVICE, "no HIP device");
	if (device < 0 || device >= nDev) return fail(RTX_ERR_ARG, "bad device index");
	HIPCHK(hipSetDevice(device));

	// bump arena over a few large slabs (one hipMalloc per ~27 levels x 8 buffers would dominate the wall time);
	// everything in it is freed on every exit path
	std::vector<void*> scratch;
	struct Cleanup { std::vector<void*>& v; ~Cleanup() { for (void* p : v) (void)hipFree(p); } } cleanup{ scratch };
	char* slab = nullptr; size_t slabLeft = 0;
	const size_t slabBytes = std::max<size_t>((size_t)32 << 20, (size_t)n_tris * 160);
	auto dalloc = [&](void** p, size_t bytes) -> hipError_t {
		bytes = (std::max<size_t>(bytes, 4) + 255) & ~(size_t)255;
		if (bytes > slabLeft) {
			const size_t sz = std::max(slabBytes, bytes);
			void* q = nullptr;
			hipError_t e = hipMalloc(&q, sz);
			if (e != hipSuccess) return e;
			scratch.push_back(q);
			slab = (char*)q; slabLeft = sz;
		}
		*p = slab; slab += bytes; slabLeft -= bytes;
		return hipSuccess;
	};
	hipStream_t st = nullptr;
	hipEvent_t ev0, ev1;
	HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
	struct EvCleanup { hipEvent_t a, b; ~EvCleanup() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evCleanup{ ev0, ev1 };

	float* dPos = nullptr; float* dExt = nullptr;
	HIPCHK(dalloc((void**)&dPos, (size_t)n_tris * 9 * sizeof(float)));
	HIPCHK(dalloc((void**)&dExt, (size_t)n_tris * 6 * sizeof(float)));
	if (n_tris) HIPCHK(hipMemcpy(dPos, tri_pos, (size_t)n_tris * 9 * sizeof(float), hipMemcpyHostToDevice));
	uint32_t* dFlags = nullptr;      // [0] unfinished nodes of the running bisection
	HIPCHK(dalloc((void**)&dFlags, 256 * 4));
	uint32_t launches = 0;
	HIPCHK(hipEventRecord(ev0, st));
	if (n_tris) { hipLaunchKernelGGL(triExtentKernel, dim3(gridFor(n_tris)), dim3(256), 0, st, dPos, n_tris, dExt); launches++; }

	std::vector<BvhLevel> levels;
	{
		BvhLevel l0;
		l0.n = 1; l0.total = n_tris;
		HIPCHK(dalloc((void**)&l0.nodes, sizeof(BNode)));
		HIPCHK(dalloc((void**)&l0.ids, (size_t)n_tris * 4));
		HIPCHK(dalloc((void**)&l0.owner, (size_t)n_tris * 4));
		BNode root;
		memset(&root, 0, sizeof(root));
		for (int c = 0; c < 3; c++) { root.lo[c] = root_lo[c]; root.hi[c] = root_hi[c]; }
		root.begin = 0; root.count = n_tris;
		HIPCHK(hipMemcpyAsync(l0.nodes, &root, sizeof(root), hipMemcpyHostToDevice, st));
		HIPCHK(hipStreamSynchronize(st));
		if (n_tris) { hipLaunchKernelGGL(iotaKernel, dim3(gridFor(n_tris)), dim3(256), 0, st, l0.ids, l0.owner, n_tris); launches++; }
		levels.push_back(l0);
	}
	// the bisection halves [min, max] of the split axis until it is narrower than 0.1: the root's largest extent
	// bounds the number of steps of every node below it
	float ext = 0;
	for (int c = 0; c < 3; c++) ext = std::max(ext, root_hi[c] - root_lo[c]);
	constexpr uint32_t kMaxSteps = 256;
	int steps = 3;
	if (std::isfinite(ext)) for (float w = ext; w >= 0.1f && steps < 200; w *= 0.5f) steps++;
	else return fail(RTX_ERR_ARG, "root bounds are not finite");

	for (uint32_t depth = 1;; depth++) {
		if (depth > 4096) return fail(RTX_ERR_ARG, "acceleration structure deeper than 4096 levels");
		BvhLevel& L = levels.back();
		const unsigned gN = gridFor(L.n), gE = gridFor(L.total);
		const uint32_t perThread = std::min<uint32_t>(16u, std::max<uint32_t>(1u, L.total >> 16));
		hipLaunchKernelGGL(levelInitKernel, dim3(gN), dim3(256), 0, st, L.nodes, L.n, depth, ac_penalty);
		launches++;
		// `steps` bisection steps, each followed by its counting pass; step k reports in dFlags[k] how many nodes were
		// still bisecting when it started.  The batch is long enough when its last step found none; the first such
		// step sizes the batch of the next level (boxes only shrink on the way down).
		for (int round = 0;; round++) {
			if (round > 16) return fail(RTX_ERR_DEVICE, "bisection did not terminate (non-finite triangle coordinates?)");
			HIPCHK(hipMemsetAsync(dFlags, 0, kMaxSteps * 4, st));
			for (int k = 0; k < steps; k++) {
				hipLaunchKernelGGL(bisectStepKernel, dim3(gN), dim3(256), 0, st, L.nodes, L.n, dFlags + k);
				if (L.total) hipLaunchKernelGGL(countKernel, dim3(gridFor(L.total, 256 * perThread)), dim3(256), 0, st, L.nodes, L.owner, L.ids, L.total, dExt, n_tris, perThread);
				launches += 2;
			}
			uint32_t busy[kMaxSteps];
			HIPCHK(hipMemcpyAsync(busy, dFlags, (size_t)steps * 4, hipMemcpyDeviceToHost, st));
			HIPCHK(hipStreamSynchronize(st));
			if (busy[steps - 1] != 0) { steps = std::min(steps + 4, (int)kMaxSteps); continue; }
			int needed = 1;
			while (needed < steps && busy[needed - 1] != 0) needed++;
			steps = std::max(needed, 2);
			break;
		}
		// leaf / inner decision, child indices and id offsets (scans over the level's nodes)
		uint32_t* dChildIdx = nullptr; uint32_t* dChildOff = nullptr; uint32_t* dTmp = nullptr;
		HIPCHK(dalloc((void**)&dChildIdx, ((size_t)L.n + 1) * 4));
		HIPCHK(dalloc((void**)&dChildOff, ((size_t)L.n + 1) * 4));
		const size_t tmpWords = std::max<size_t>(L.n, L.total) / kScanBlock * 2 + 64;
		HIPCHK(dalloc((void**)&dTmp, tmpWords * 4));
		HIPCHK(hipMemsetAsync(dChildIdx + L.n, 0, 4, st));
		HIPCHK(hipMemsetAsync(dChildOff + L.n, 0, 4, st));
		hipLaunchKernelGGL(decideKernel, dim3(gN), dim3(256), 0, st, L.nodes, L.n, dChildIdx, dChildOff);
		launches++;
		int rc;
		if ((rc = scanExclusive(dChildIdx, L.n + 1, dTmp, st, launches))) return rc;
		if ((rc = scanExclusive(dChildOff, L.n + 1, dTmp, st, launches))) return rc;
		uint32_t nextN = 0, nextTotal = 0;
		HIPCHK(hipMemcpyAsync(&nextN, dChildIdx + L.n, 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipMemcpyAsync(&nextTotal, dChildOff + L.n, 4, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		if (nextN == 0) break;
		BvhLevel nx;
		nx.n = nextN; nx.total = nextTotal;
		HIPCHK(dalloc((void**)&nx.nodes, (size_t)nextN * sizeof(BNode)));
		HIPCHK(dalloc((void**)&nx.ids, (size_t)nextTotal * 4));
		HIPCHK(dalloc((void**)&nx.owner, (size_t)nextTotal * 4));
		uint32_t* dScanL = nullptr; uint32_t* dScanR = nullptr;
		HIPCHK(dalloc((void**)&dScanL, (size_t)L.total * 4));
		HIPCHK(dalloc((void**)&dScanR, (size_t)L.total * 4));
		hipLaunchKernelGGL(childrenKernel, dim3(gN), dim3(256), 0, st, L.nodes, L.n, dChildIdx, dChildOff, nx.nodes);
		hipLaunchKernelGGL(flagKernel, dim3(gE), dim3(256), 0, st, L.nodes, L.owner, L.ids, L.total, dExt, n_tris, dScanL, dScanR);
		launches += 2;
		if ((rc = scanExclusive(dScanL, L.total, dTmp, st, launches))) return rc;
		if ((rc = scanExclusive(dScanR, L.total, dTmp, st, launches))) return rc;
		hipLaunchKernelGGL(scatterKernel, dim3(gE), dim3(256), 0, st, L.nodes, L.owner, L.ids, L.total, dExt, n_tris, dScanL, dScanR, nx.ids, nx.owner);
		launches++;
		levels.push_back(nx);       // (invalidates L)
	}

	// bottom-up subtree sizes, top-down pre-order indices, emission in DFS-left-first order
	for (size_t l = levels.size(); l-- > 0;) {
		hipLaunchKernelGGL(subtreeKernel, dim3(gridFor(levels[l].n)), dim3(256), 0, st, levels[l].nodes, levels[l].n,
		                   l + 1 < levels.size() ? levels[l + 1].nodes : nullptr);
		launches++;
	}
	BNode root;
	HIPCHK(hipMemcpyAsync(&root, levels[0].nodes, sizeof(root), hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	rtx_bvh* b = new rtx_bvh();
	b->device = device; b->nNodes = root.subNodes; b->nRefs = root.subRefs; b->maxDepth = (uint32_t)levels.size();
	auto bailOut = [&](int code) { rtx_bvh_destroy(b); return code; };
	if (hipMalloc((void**)&b->bounds, (size_t)b->nNodes * 6 * sizeof(float)) != hipSuccess || hipMalloc((void**)&b->skip, (size_t)b->nNodes * 4) != hipSuccess ||
	    hipMalloc((void**)&b->leafBegin, (size_t)b->nNodes * 4) != hipSuccess || hipMalloc((void**)&b->leafCount, (size_t)b->nNodes * 4) != hipSuccess ||
	    hipMalloc((void**)&b->refs, (size_t)std::max<uint32_t>(b->nRefs, 1) * 4) != hipSuccess)
		return bailOut(fail(RTX_ERR_DEVICE, "out of device memory for the acceleration structure"));
	for (size_t l = 0; l < levels.size(); l++) {
		const BvhLevel& L = levels[l];
		if (l + 1 < levels.size()) { hipLaunchKernelGGL(preorderKernel, dim3(gridFor(L.n)), dim3(256), 0, st, L.nodes, L.n, levels[l + 1].nodes); launches++; }
		hipLaunchKernelGGL(emitNodesKernel, dim3(gridFor(L.n)), dim3(256), 0, st, L.nodes, L.n, b->bounds, b->skip, b->leafBegin, b->leafCount);
		if (L.total) hipLaunchKernelGGL(emitRefsKernel, dim3(gridFor(L.total)), dim3(256), 0, st, L.nodes, L.owner, L.ids, L.total, b->refs);
		launches += 2;
	}
	if (hipEventRecord(ev1, st) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess || hipGetLastError() != hipSuccess)
		return bailOut(fail(RTX_ERR_DEVICE, "acceleration-structure build failed on the device"));
	(void)hipEventElapsedTime(&b->buildMs, ev0, ev1);
	b->launches = launches;
	*out = b;
	return RTX_OK;
}

int rtx_bvh_info(const rtx_bvh* b, uint32_t* n_nodes, uint32_t* n_refs, uint32_t* max_depth, float* build_ms)
{
	if (!b) return fail(RTX_ERR_ARG, "bvh is NULL");
	if (n_nodes) *n_nodes = b->nNodes;
	if (n_refs) *n_refs = b->nRefs;
	if (max_depth) *max_depth = b->maxDepth;
	if (build_ms) *build_ms = b->buildMs;
	return RTX_OK;
}

int rtx_bvh_read(const rtx_bvh* b, float* node_bounds, int32_t* node_skip, int32_t* leaf_begin, int32_t* leaf_count, uint32_t* refs)
{
	if (!b) return fail(RTX_ERR_ARG, "bvh is NULL");
	HIPCHK(hipSetDevice(b->device));
	if (node_bounds) HIPCHK(hipMemcpy(node_bounds, b->bounds, (size_t)b->nNodes * 6 * sizeof(float), hipMemcpyDeviceToHost));
	if (node_skip) HIPCHK(hipMemcpy(node_skip, b->skip, (size_t)b->nNodes * 4, hipMemcpyDeviceToHost));
	if (leaf_begin) HIPCHK(hipMemcpy(leaf_begin, b->leafBegin, (size_t)b->nNodes * 4, hipMemcpyDeviceToHost));
	if (leaf_count) HIPCHK(hipMemcpy(leaf_count, b->leafCount, (size_t)b->nNodes * 4, hipMemcpyDeviceToHost));
	if (refs && b->nRefs) HIPCHK(hipMemcpy(refs, b->refs, (size_t)b->nRefs * 4, hipMemcpyDeviceToHost));
	return RTX_OK;
}

void rtx_bvh_destroy(rtx_bvh* b)
{
	if (!b) return;
	(void)hipSetDevice(b->device);
	if (b->bounds) (void)hipFree(b->bounds);
	if (b->skip) (void)hipFree(b->skip);
	if (b->leafBegin) (void)hipFree(b->leafBegin);
	if (b->leafCount) (void)hipFree(b->leafCount);
	if (b->refs) (void)hipFree(b->refs);
	delete b;
}

} // extern "C"
