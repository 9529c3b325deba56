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

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same builder as ONE persistent launch (the level-synchronous form above needs ~27 levels x ~27 launches and three
// host round trips per level -- 740 launches, 6.8 ms for 250 000 triangles, all of it launch latency).
//
// Work item = a node, processed by one block of 256 threads from the reference's leaf test to its two children (objects.cpp:470-526): the bisection
// (676-689) as a loop inside the block -- each step one counting pass over the node's ids, reduced in the block --, the 1.5x duplication stop (498),
// a STABLE partition of the ids into the two children (737-760: order inside a node = the reference's vector order; where a node or its ids sit in
// the pools is irrelevant to the result) and the children's boxes (510-521).  Nodes are created by bumping one 64-bit counter (node slots + id
// slots); block b handles the nodes b, b + G, b + 2G ... and waits for each to be published by the block that created it (its own flag: no address
// is polled by more than one block).  Subtree sizes travel bottom-up as the leaves finish (the second child to report continues at the parent);
// when the root's are known every slot a block could still be waiting for is marked "no node".  Pre-order indices, first references and the emission
// in the DFS-left-first layout are then one plain launch (thread per node: the path to the root gives the index; wave per leaf for the references).
//
// The eight L2s are not coherent with each other: what one block writes for another (node records, id lists) is stored write-through (agent-scope
// atomic stores) into 128-byte-aligned allocations (no line is shared between two writers or read before it is complete), acknowledged
// (s_waitcnt vmcnt(0)) before the flag that publishes it; flags and sizes are read with agent-scope atomic loads.  A 2-s watchdog turns a bug into
// an error (the level-synchronous build then runs instead), not a hung GPU.
namespace bvhq {

struct alignas(128) QNode {
	float lo[3], hi[3];
	uint32_t begin, count;       // ids of this node: idPool[begin, begin + count)
	uint32_t depth, parent;      // root: depth 1, parent ~0
	uint32_t state;              // bvhb::kLeaf, or kFinished = split
	uint32_t child;              // left child (right = child + 1)
	uint32_t subNodes, subRefs;  // written when the subtree is complete
	uint32_t arrived;            // children that have reported
	uint32_t ready;              // 1: the record is complete, 3: no such node (the build is over)
	uint32_t refStart;           // finishKernel
	uint32_t processed;          // the node has been looked at (leaf or split): the second launch skips it
	uint32_t pad[14];
};
static_assert(sizeof(QNode) == 128, "one node record per 128-byte line");

struct Ctl {                     // every word that several blocks touch on a line of its own
	unsigned long long alloc; uint32_t padA[14];     // [63:40] node slots handed out, [39:0] id slots handed out
	uint32_t error; uint32_t padB[15];               // 1 pool exhausted, 2 watchdog
	uint32_t finished; uint32_t nodes; uint32_t refs; uint32_t maxDepth; uint32_t padC[12];
	uint32_t big; uint32_t padD[15];                 // nodes of more than kSmall ids that are published and not yet processed
};
static_assert(sizeof(Ctl) == 256, "control block");

#define BVHQ_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#ifdef RTX_BVHQ_DBG
__device__ unsigned long long gBvhqDbg[64][4];      // per node < 64: start, after the bisection, after the partition, count
#endif
__device__ __forceinline__ void qstore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, BVHQ_AGENT); }
__device__ __forceinline__ void qstoref(float* p, float v) { __hip_atomic_store(p, v, BVHQ_AGENT); }
__device__ __forceinline__ uint32_t qload(const uint32_t* p) { return __hip_atomic_load(p, BVHQ_AGENT); }
__device__ __forceinline__ float qloadf(const float* p) { return __hip_atomic_load(p, BVHQ_AGENT); }
__device__ __forceinline__ void acknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A block is 1 024 threads: the nodes at the top of the tree hold hundreds of thousands of ids and ONE block walks them (six bisection steps + the partition);
// with 256 threads the root alone took a millisecond and a half (the passes are chains of dependent loads: id, then its extent).
constexpr uint32_t kQB = 1024, kQW = kQB / 64;
// sum of four per-thread counters over the block (every thread gets the result)
__device__ __forceinline__ void blockSum4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t (*part)[4])
{
	for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); }
	__syncthreads();
	if ((threadIdx.x & 63) == 0) { uint32_t* q = part[threadIdx.x >> 6]; q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
	__syncthreads();
	a = b = c = d = 0;
	for (uint32_t w = 0; w < kQW; w++) { a += part[w][0]; b += part[w][1]; c += part[w][2]; d += part[w][3]; }
}

// loPool / hiPool: beside every id of a node its extent along THE NODE'S OWN split axis (written by the parent's partition, which knows the child's box and
// therefore its axis): the six counting passes of a bisection stream two float arrays instead of chasing id -> extent (two dependent loads per id and pass).
__device__ __forceinline__ int splitAxis(const float* lo, const float* hi)      // objects.cpp:486-490: the strictly longest dimension, else y over z
{
	const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	if (dx > dy && dx > dz) return 0;
	return dy > dz ? 1 : 2;
}
// WAVE = false: a node is processed by the whole block (1 024 threads) -- the launch for the BIG nodes (more than kSmall ids: the top of the tree, where one
// node's passes are long) -- and small nodes are left alone; WAVE = true: a node is processed by ONE wave (no barrier anywhere), four nodes at a time per block
// of 256 -- the second launch, which finds the small nodes the first one left (published, not processed) and everything below them.  Worker k of W handles the
// nodes k, k + W, k + 2 W ...  Markers in QNode::ready: 1 the record is complete, 2 "the launch for the big nodes is over" (to the second launch: not made yet),
// 3 "the build is over".
constexpr uint32_t kSmall = 1024;
template <bool WAVE>
__global__ void __launch_bounds__(WAVE ? 256 : kQB) buildKernel(QNode* nodes, uint32_t nodeCap, uint32_t* idPool, float* loPool, float* hiPool, unsigned long long idCap, Ctl* ctl,
                                                                const float* __restrict__ ext, uint32_t nTris, int32_t penalty, uint32_t slack)
{
	__shared__ uint32_t part[kQW][4];
	__shared__ uint32_t shNode[12];
	__shared__ uint32_t shAlloc[2];
	__shared__ uint32_t waveCnt[kQW][2];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t tid = WAVE ? lane : threadIdx.x;                    // index in the group that shares a node
	const uint32_t GS = WAVE ? 64u : kQB, nSeg = WAVE ? 1u : kQW;      // its size; wave segments of the final count / partition
	const uint32_t W = WAVE ? gridDim.x * 4u : gridDim.x, me = WAVE ? blockIdx.x * 4u + wave : blockIdx.x;
	const unsigned long long started = wall_clock64();
	auto gsync = [&]() { if (!WAVE) __syncthreads(); };
	for (uint32_t idx = me;; idx += W) {
		if (idx >= nodeCap) { if (tid == 0) atomicMax(&ctl->error, 1u); return; }
		// ---- wait for the node to be published (or for the end of this launch)
		uint32_t verdict = 0;
		if (tid == 0) {
			uint32_t spins = 0;
			for (;;) {
				const uint32_t r = qload(&nodes[idx].ready);
				if (r == 1 || r == 3 || (!WAVE && r == 2)) { verdict = r; break; }
				if ((++spins & 63u) == 0) {
					if (qload(&ctl->error) != 0) { verdict = 3; break; }
					if (wall_clock64() - started > 200000000ull) { atomicMax(&ctl->error, 2u); verdict = 3; break; }
				}
				__builtin_amdgcn_s_sleep(20);
			}
		}
		float lo[3], hi[3];
		uint32_t begin, count, depth, parent;
		if (WAVE) {
			verdict = (uint32_t)__builtin_amdgcn_readfirstlane((int)verdict);
			if (verdict != 1) return;
			const QNode* n = nodes + idx;
			if (qload(&n->processed) != 0) continue;                   // (done by the launch for the big nodes)
			for (int c = 0; c < 3; c++) { lo[c] = qloadf(&n->lo[c]); hi[c] = qloadf(&n->hi[c]); }
			begin = qload(&n->begin); count = qload(&n->count); depth = qload(&n->depth); parent = qload(&n->parent);
		}
		else {
			if (tid == 0) {
				shNode[11] = verdict;
				if (verdict == 1) {
					const QNode* n = nodes + idx;
					for (int c = 0; c < 3; c++) { shNode[c] = __float_as_uint(qloadf(&n->lo[c])); shNode[3 + c] = __float_as_uint(qloadf(&n->hi[c])); }
					shNode[6] = qload(&n->begin); shNode[7] = qload(&n->count); shNode[8] = qload(&n->depth); shNode[9] = qload(&n->parent);
				}
			}
			__syncthreads();
			if (shNode[11] != 1) return;
			for (int c = 0; c < 3; c++) { lo[c] = __uint_as_float(shNode[c]); hi[c] = __uint_as_float(shNode[3 + c]); }
			begin = shNode[6]; count = shNode[7]; depth = shNode[8]; parent = shNode[9];
			__syncthreads();
			if (count <= kSmall) continue;                             // (the second launch's)
		}
#ifdef RTX_BVHQ_DBG
		if (tid == 0 && idx < 64) gBvhqDbg[idx][0] = wall_clock64();
#endif
		const uint32_t* ids = idPool + begin;
		const float* L = loPool + begin;
		const float* Hh = hiPool + begin;
		const uint32_t seg = WAVE ? count : ((count + kQW - 1) / kQW + 63u) & ~63u;      // entries per wave in the final count and the partition
		// ---- objects.cpp:477: leaf by size; 486-490: the axis
		bool leaf = (unsigned long long)count <= (unsigned long long)depth * (unsigned long long)(long long)penalty;
		uint32_t nl = 0, nr = 0;
		float s = 0;
		int ax = 2;
		if (!leaf) {
			ax = splitAxis(lo, hi);
			const float mn = lo[ax], mx = hi[ax];
			float left = mn, right = mx;
			// objects.cpp:676-689: halve [left, right] until it is narrower than 0.1, comparing the cost 0.05 either side of the middle
			for (int step = 0; step < 4096; step++) {
				const float mid = right - (right - left) / 2;
				if (right - left < 0.1f) { s = mid; break; }
				const float s1 = mid - 0.05f, s2 = mid + 0.05f;
				uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
				for (uint32_t e = tid; e < count; e += 4 * GS) {      // (four entries in flight per thread)
					float l[4], h[4];
					for (int k = 0; k < 4; k++) { const bool in = e + k * GS < count; l[k] = in ? L[e + k * GS] : __builtin_inff(); h[k] = in ? Hh[e + k * GS] : -__builtin_inff(); }
					for (int k = 0; k < 4; k++) { c0 += l[k] <= s1; c1 += h[k] >= s1; c2 += l[k] <= s2; c3 += h[k] >= s2; }
				}
				if (WAVE) { for (int o = 32; o >= 1; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); c3 += __shfl_xor(c3, o, 64); } }
				else blockSum4(c0, c1, c2, c3, part);
				// nLeft * (s - min) + nRight * (max - s), counts promoted to float (objects.cpp:672)
				const float k1 = (float)(int)c0 * (s1 - mn) + (float)(int)c1 * (mx - s1);
				const float k2 = (float)(int)c2 * (s2 - mn) + (float)(int)c3 * (mx - s2);
				if (k1 < k2) right = mid; else left = mid;
				if (step == 4095) { s = right - (right - left) / 2; if (tid == 0) atomicMax(&ctl->error, 3u); }      // (non-finite coordinates: the interval never narrows)
			}
			// The final count (objects.cpp:741-757 decide the sides with the same compares) by WAVE SEGMENTS: wave w owns the entries [w seg, (w + 1) seg) -- its totals
			// are where its part of either child's list begins, so the partition below needs no barrier per chunk.
			{
				const uint32_t e0 = (WAVE ? 0u : wave) * seg, e1 = min(e0 + seg, count);
				uint32_t a = 0, b = 0;
				for (uint32_t e = e0 + lane; e < e1; e += 256) {
					float l[4], h[4];
					for (int k = 0; k < 4; k++) { const bool in = e + 64 * k < e1; l[k] = in ? L[e + 64 * k] : __builtin_inff(); h[k] = in ? Hh[e + 64 * k] : -__builtin_inff(); }
					for (int k = 0; k < 4; k++) { a += l[k] <= s; b += h[k] >= s; }
				}
				for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
				if (WAVE) { nl = a; nr = b; }
				else {
					__syncthreads();
					if (lane == 0) { waveCnt[wave][0] = a; waveCnt[wave][1] = b; }
					__syncthreads();
					for (uint32_t w = 0; w < kQW; w++) { nl += waveCnt[w][0]; nr += waveCnt[w][1]; }
				}
			}
			// objects.cpp:498: a leaf if one side is empty or the split duplicates too much (>= 1.5x, compared in double)
			leaf = nl == 0 || nr == 0 || (double)((unsigned long long)nl + nr) >= (double)count * 1.5;
		}
#ifdef RTX_BVHQ_DBG
		if (tid == 0 && idx < 64) { gBvhqDbg[idx][1] = wall_clock64(); gBvhqDbg[idx][3] = count; }
#endif
		const int bigDelta = -1;                                       // (WAVE = false: big nodes in flight: this one is done ...)
		bool rootDone = false;
		if (!leaf) {
			// ---- two node slots and nl + nr id slots (rounded up to whole 128-byte lines) with one atomic
			const unsigned long long idsWanted = (((unsigned long long)nl + 31u) & ~31ull) + (((unsigned long long)nr + 31u) & ~31ull);
			uint32_t child = 0, idBase32 = 0;
			if (tid == 0) {
				const unsigned long long old = atomicAdd(&ctl->alloc, (2ull << 40) | idsWanted);
				const unsigned long long idBase = old & ((1ull << 40) - 1), nodeBase = old >> 40;
				const bool ok = nodeBase + 2 + slack <= nodeCap && idBase + idsWanted <= idCap && idBase + idsWanted < (1ull << 32);
				if (!ok) atomicMax(&ctl->error, 1u);
				child = ok ? (uint32_t)nodeBase : 0xffffffffu; idBase32 = (uint32_t)idBase;
				if (!WAVE) { shAlloc[0] = child; shAlloc[1] = idBase32; }
			}
			if (WAVE) { child = (uint32_t)__builtin_amdgcn_readfirstlane((int)child); idBase32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)idBase32); }
			else { __syncthreads(); child = shAlloc[0]; idBase32 = shAlloc[1]; }
			if (child == 0xffffffffu) return;
			const uint32_t lBegin = idBase32, rBegin = lBegin + ((nl + 31u) & ~31u);
			// ---- stable partition (objects.cpp:737-760): [ids with a vertex <= s] and [ids with a vertex >= s], each in the node's order, with their extents
			// along the CHILD's split axis (its box is the parent's cut at s: objects.cpp:510-521).  Every wave writes its own segment's part of either list
			// (offsets from the final count): no barrier inside.
			float cl[3], ch[3];
			for (int a = 0; a < 3; a++) { cl[a] = lo[a]; ch[a] = hi[a]; }
			ch[ax] = s;
			const int axL = splitAxis(cl, ch);
			ch[ax] = hi[ax]; cl[ax] = s;
			const int axR = splitAxis(cl, ch);
			const float* elL = ext + (size_t)(2 * axL) * nTris; const float* ehL = ext + (size_t)(2 * axL + 1) * nTris;
			const float* elR = ext + (size_t)(2 * axR) * nTris; const float* ehR = ext + (size_t)(2 * axR + 1) * nTris;
			const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
			uint32_t offL = lBegin, offR = rBegin;
			if (!WAVE) for (uint32_t w = 0; w < wave; w++) { offL += waveCnt[w][0]; offR += waveCnt[w][1]; }
			const uint32_t e0 = (WAVE ? 0u : wave) * seg, e1 = min(e0 + seg, count);
			for (uint32_t base = e0; base < e1; base += 256) {      // (four chunks of 64 in flight per wave)
				uint32_t id[4]; float l[4], h[4];
				for (int k = 0; k < 4; k++) {
					const uint32_t e = base + 64 * k + lane;
					const bool in = e < e1;
					id[k] = in ? ids[e] : 0xffffffffu; l[k] = in ? L[e] : __builtin_inff(); h[k] = in ? Hh[e] : -__builtin_inff();
				}
				for (int k = 0; k < 4; k++) {
					const bool fl = l[k] <= s, fr = h[k] >= s;
					const unsigned long long bl = __ballot(fl), br = __ballot(fr);
					if (fl) {
						const uint32_t at = offL + (uint32_t)__popcll(bl & below);
						qstore(idPool + at, id[k]); qstoref(loPool + at, axL == ax ? l[k] : elL[id[k]]); qstoref(hiPool + at, axL == ax ? h[k] : ehL[id[k]]);
					}
					if (fr) {
						const uint32_t at = offR + (uint32_t)__popcll(br & below);
						qstore(idPool + at, id[k]); qstoref(loPool + at, axR == ax ? l[k] : elR[id[k]]); qstoref(hiPool + at, axR == ax ? h[k] : ehR[id[k]]);
					}
					offL += (uint32_t)__popcll(bl); offR += (uint32_t)__popcll(br);
				}
			}
			acknowledged();
			gsync();
#ifdef RTX_BVHQ_DBG
			if (tid == 0 && idx < 64) gBvhqDbg[idx][2] = wall_clock64();
#endif
			if (tid == 0) {
				qstore(&nodes[idx].child, child);
				qstore(&nodes[idx].state, bvhb::kFinished);
				qstore(&nodes[idx].processed, 1u);
				// the children: the parent's box cut at s on the split axis (objects.cpp:510-521)
				for (int k = 0; k < 2; k++) {
					QNode* c = nodes + child + k;
					for (int a = 0; a < 3; a++) { qstoref(&c->lo[a], (k == 1 && a == ax) ? s : lo[a]); qstoref(&c->hi[a], (k == 0 && a == ax) ? s : hi[a]); }
					qstore(&c->begin, k ? rBegin : lBegin); qstore(&c->count, k ? nr : nl);
					qstore(&c->depth, depth + 1); qstore(&c->parent, idx);
				}
				// (... its big children are counted BEFORE they are published: a child that finishes first must not see the counter at 0 for a moment and end
				// the launch while its sibling is still on its way -- ADVICE r5)
				if (!WAVE) { const uint32_t kids = (uint32_t)(nl > kSmall) + (uint32_t)(nr > kSmall); if (kids) atomicAdd(&ctl->big, kids); }
				acknowledged();
				qstore(&nodes[child].ready, 1u); qstore(&nodes[child + 1].ready, 1u);
			}
		}
		else if (tid == 0) {
			// ---- a leaf: its subtree is complete; the second child to report completes the parent, and so on up to the root
			qstore(&nodes[idx].state, bvhb::kLeaf);
			qstore(&nodes[idx].processed, 1u);
			qstore(&nodes[idx].subNodes, 1u); qstore(&nodes[idx].subRefs, count);
			uint32_t n = idx, par = parent;
			for (;;) {
				acknowledged();
				if (par == 0xffffffffu) {
					// the root: the build is over.  Every slot some worker may still be waiting for is marked "no node" (by this whole wave, below).
					const unsigned long long a = __hip_atomic_load(&ctl->alloc, BVHQ_AGENT);
					qstore(&ctl->nodes, (uint32_t)(a >> 40)); qstore(&ctl->refs, qload(&nodes[n].subRefs));
					qstore(&ctl->finished, 1u);
					rootDone = true;
					break;
				}
				if (atomicAdd(&nodes[par].arrived, 1u) == 0) break;      // the first child: the other one will carry on
				const uint32_t c = qload(&nodes[par].child);
				qstore(&nodes[par].subNodes, 1u + qload(&nodes[c].subNodes) + qload(&nodes[c + 1].subNodes));
				qstore(&nodes[par].subRefs, qload(&nodes[c].subRefs) + qload(&nodes[c + 1].subRefs));
				n = par; par = qload(&nodes[n].parent);
			}
		}
		// ---- the end of this launch: the build is over (3), or -- the launch for the big nodes -- no big node is left (2).  The markers go to every slot a worker
		// may still be waiting for: the `slack` slots behind the last node made.
		uint32_t mark = 0;
		if (tid == 0) {
			acknowledged();
			if (rootDone) mark = 3;
			else if (!WAVE && atomicAdd(&ctl->big, (uint32_t)bigDelta) + (uint32_t)bigDelta == 0u) mark = 2;
		}
		if (WAVE) mark = (uint32_t)__builtin_amdgcn_readfirstlane((int)mark);
		else { __syncthreads(); if (tid == 0) shAlloc[0] = mark; __syncthreads(); mark = shAlloc[0]; __syncthreads(); }
		if (mark != 0) {
			const uint32_t nNodes = (uint32_t)(__hip_atomic_load(&ctl->alloc, BVHQ_AGENT) >> 40);
			for (uint32_t k = tid; k < slack && nNodes + k < nodeCap; k += GS) qstore(&nodes[nNodes + k].ready, mark);
		}
	}
}

// Pre-order index and first reference of every node from its path to the root (objects.cpp:601-629: a node, its left subtree, its right subtree), the
// node records in that order, and -- a wave per 64 nodes -- the references of the leaves among them.
__global__ void __launch_bounds__(256) finishKernel(const QNode* __restrict__ nodes, uint32_t n, const uint32_t* __restrict__ idPool, Ctl* ctl,
                                                    float* bounds, int32_t* skip, int32_t* leafBegin, int32_t* leafCount, uint32_t* refs)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t pre = 0, refStart = 0, count = 0, begin = 0, depth = 0;
	bool leaf = false;
	if (i < n) {
		const QNode& nd = nodes[i];
		uint32_t c = i, p = nd.parent;
		while (p != 0xffffffffu) {
			const QNode& pn = nodes[p];
			pre += 1;
			if (c == pn.child + 1) { pre += nodes[pn.child].subNodes; refStart += nodes[pn.child].subRefs; }
			c = p; p = pn.parent;
		}
		leaf = nd.state == bvhb::kLeaf; count = nd.count; begin = nd.begin; depth = nd.depth;
		for (int a = 0; a < 3; a++) { bounds[(size_t)pre * 6 + a] = nd.lo[a]; bounds[(size_t)pre * 6 + 3 + a] = nd.hi[a]; }
		skip[pre] = (int32_t)(pre + nd.subNodes);
		leafBegin[pre] = leaf ? (int32_t)refStart : -1;
		leafCount[pre] = leaf ? (int32_t)count : -1;
	}
	uint32_t d = depth;
	for (int o = 32; o >= 1; o >>= 1) d = max(d, (uint32_t)__shfl_xor((int)d, o, 64));
	if ((threadIdx.x & 63) == 0 && d) atomicMax(&ctl->maxDepth, d);
	unsigned long long todo = __ballot(leaf && count != 0);
	const uint32_t lane = threadIdx.x & 63;
	while (todo) {
		const int k = __builtin_ctzll(todo);
		todo &= todo - 1;
		const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)begin, k), cnt = (uint32_t)__builtin_amdgcn_readlane((int)count, k), r0 = (uint32_t)__builtin_amdgcn_readlane((int)refStart, k);
		for (uint32_t e = lane; e < cnt; e += 64) refs[r0 + e] = idPool[b + e];
	}
}

__global__ void initKernel(QNode* nodes, uint32_t* idPool, float* loPool, float* hiPool, Ctl* ctl, const float* __restrict__ ext, uint32_t nTris,
                           float lx, float ly, float lz, float hx, float hy, float hz)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const float rl[3] = { lx, ly, lz }, rh[3] = { hx, hy, hz };
	const int ax = splitAxis(rl, rh);
	if (i < nTris) { idPool[i] = i; loPool[i] = ext[(size_t)(2 * ax) * nTris + i]; hiPool[i] = ext[(size_t)(2 * ax + 1) * nTris + i]; }
	if (i == 0) {
		QNode& r = nodes[0];
		r.lo[0] = lx; r.lo[1] = ly; r.lo[2] = lz; r.hi[0] = hx; r.hi[1] = hy; r.hi[2] = hz;
		r.begin = 0; r.count = nTris; r.depth = 1; r.parent = 0xffffffffu; r.ready = 1;
		ctl->alloc = (1ull << 40) | (((unsigned long long)nTris + 31u) & ~31ull);
		ctl->big = nTris > kSmall ? 1u : 0u;
	}
}

} // namespace bvhq

static int gBvhBuildMode = 0;      // rtx_bvh_build_mode (include/rtx_debug.h): 0 = the persistent launches, level by level only as their fallback; 1 = level by level

struct rtx_bvh {
	int device = 0;
	uint32_t nNodes = 0, nRefs = 0, maxDepth = 0, launches = 0;
	float buildMs = 0;
	bool queued = false;         // built by the single persistent launch (bvhq) -- else level by level (bvhb)
	void* slab = nullptr;        // bvhq: the five arrays below are one allocation
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

// The build as one persistent launch (bvhq).  RTX_OK with *out = nullptr: not done here (a pool ran out, the watchdog fired, RTX_BVH_BUILD=levels) -- the
// caller builds level by level.
int buildQueued(const float* tri_pos, uint32_t n_tris, const float* root_lo, const float* root_hi, int32_t ac_penalty, int device, rtx_bvh** out)
{
	using namespace bvhq;
	*out = nullptr;
	if (gBvhBuildMode == 1) return RTX_OK;
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	int occ = 0, occW = 0;
	HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, buildKernel<false>, (int)kQB, 0));
	HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occW, buildKernel<true>, 256, 0));
	if (occ < 1 || occW < 1) return RTX_OK;
	// every worker resident: a worker waits for nodes other workers make
	const uint32_t G = (uint32_t)std::min(occ, 2) * (uint32_t)prop.multiProcessorCount, GW = (uint32_t)std::min(occW, 8) * (uint32_t)prop.multiProcessorCount;
	const uint32_t slack = std::max(G, GW * 4u);
	// (mode 2, tests: pools far too small -- the launches must notice, say so and leave the build to the level-by-level path)
	const uint32_t nodeCap = (gBvhBuildMode == 2 ? 64u : std::max<uint32_t>(2 * n_tris, 1024)) + slack + 64;
	const unsigned long long idCap = std::min<unsigned long long>(32ull * n_tris + (1ull << 20), (1ull << 32) - 64);      // (ids of ALL levels: 27 levels x 2.6 n at the headline; three arrays)
	std::vector<void*> scratch;
	struct Cleanup { std::vector<void*>& v; ~Cleanup() { for (void* p : v) (void)hipFree(p); } } cleanup{ scratch };
	auto dalloc = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 256)); if (e == hipSuccess) scratch.push_back(*p); return e; };
	float* dPos = nullptr; float* dExt = nullptr; QNode* dNodes = nullptr; uint32_t* dIds = nullptr; Ctl* dCtl = nullptr;
	// (one allocation for everything transient: positions, extents, node pool, id pool, control block)
	const size_t szPos = ((size_t)n_tris * 9 * 4 + 255) & ~(size_t)255, szExt = ((size_t)n_tris * 6 * 4 + 255) & ~(size_t)255;
	const size_t szNodes = (size_t)nodeCap * sizeof(QNode), szIds = (size_t)idCap * 4, szCtl = 256;      // (szIds: each of the three pools)
	char* base = nullptr;
	if (dalloc((void**)&base, szPos + szExt + szNodes + 3 * szIds + szCtl) != hipSuccess) { (void)hipGetLastError(); return RTX_OK; }      // (not enough memory for the pools: level by level)
	dNodes = (QNode*)base; dIds = (uint32_t*)(base + szNodes); float* dLo = (float*)(base + szNodes + szIds); float* dHi = (float*)(base + szNodes + 2 * szIds);
	dCtl = (Ctl*)(base + szNodes + 3 * szIds); dPos = (float*)(base + szNodes + 3 * szIds + szCtl); dExt = (float*)((char*)dPos + szPos);
	hipStream_t st = nullptr;
	hipEvent_t ev0, ev1;
	HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
	struct EvCleanup { hipEvent_t a, b; ~EvCleanup() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evCleanup{ ev0, ev1 };
	if (n_tris) HIPCHK(hipMemcpy(dPos, tri_pos, (size_t)n_tris * 9 * sizeof(float), hipMemcpyHostToDevice));
	uint32_t launches = 0;
	HIPCHK(hipEventRecord(ev0, st));
	HIPCHK(hipMemsetAsync(dNodes, 0, szNodes, st));
	HIPCHK(hipMemsetAsync(dCtl, 0, szCtl, st));
	launches += 2;
	if (n_tris) { hipLaunchKernelGGL(bvhb::triExtentKernel, dim3(gridFor(n_tris)), dim3(256), 0, st, dPos, n_tris, dExt); launches++; }
	hipLaunchKernelGGL(initKernel, dim3(gridFor(std::max(n_tris, 1u))), dim3(256), 0, st, dNodes, dIds, dLo, dHi, dCtl, (const float*)dExt, n_tris, root_lo[0], root_lo[1], root_lo[2], root_hi[0], root_hi[1], root_hi[2]);
	if (n_tris > kSmall) { hipLaunchKernelGGL(buildKernel<false>, dim3(G), dim3(kQB), 0, st, dNodes, nodeCap, dIds, dLo, dHi, idCap, dCtl, (const float*)dExt, n_tris, ac_penalty, slack); launches++; }
	hipLaunchKernelGGL(buildKernel<true>, dim3(GW), dim3(256), 0, st, dNodes, nodeCap, dIds, dLo, dHi, idCap, dCtl, (const float*)dExt, n_tris, ac_penalty, slack);
	launches += 2;
	Ctl ctl;
	HIPCHK(hipMemcpyAsync(&ctl, dCtl, sizeof(ctl), hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	if (ctl.error != 0 || !ctl.finished) return RTX_OK;
	rtx_bvh* b = new rtx_bvh();
	b->device = device; b->nNodes = ctl.nodes; b->nRefs = ctl.refs; b->queued = true;
	const size_t nN = b->nNodes, nR = std::max<uint32_t>(b->nRefs, 1);
	const size_t oSkip = (nN * 6 * 4 + 255) & ~(size_t)255, oLb = oSkip + ((nN * 4 + 255) & ~(size_t)255), oLc = oLb + ((nN * 4 + 255) & ~(size_t)255), oRefs = oLc + ((nN * 4 + 255) & ~(size_t)255);
	if (hipMalloc(&b->slab, oRefs + nR * 4) != hipSuccess) { (void)hipGetLastError(); delete b; return fail(RTX_ERR_DEVICE, "out of device memory for the acceleration structure"); }
	b->bounds = (float*)b->slab; b->skip = (int32_t*)((char*)b->slab + oSkip); b->leafBegin = (int32_t*)((char*)b->slab + oLb); b->leafCount = (int32_t*)((char*)b->slab + oLc);
	b->refs = (uint32_t*)((char*)b->slab + oRefs);
	hipLaunchKernelGGL(finishKernel, dim3(gridFor(b->nNodes)), dim3(256), 0, st, (const QNode*)dNodes, b->nNodes, (const uint32_t*)dIds, dCtl, b->bounds, b->skip, b->leafBegin, b->leafCount, b->refs);
	launches++;
	uint32_t maxDepth = 0;
	if (hipMemcpyAsync(&maxDepth, &dCtl->maxDepth, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(ev1, st) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess ||
	    hipGetLastError() != hipSuccess) { rtx_bvh_destroy(b); return fail(RTX_ERR_DEVICE, "acceleration-structure build failed on the device"); }
	(void)hipEventElapsedTime(&b->buildMs, ev0, ev1);
	b->maxDepth = maxDepth; b->launches = launches;
#ifdef RTX_BVHQ_DBG
	{
		unsigned long long d[64][4];
		(void)hipMemcpyFromSymbol(d, HIP_SYMBOL(gBvhqDbg), sizeof(d));
		for (int i = 0; i < 16 && i < (int)b->nNodes; i++) fprintf(stderr, "[bvhq] node %d count %llu: starts at %.1f us, bisection %.1f us, partition %.1f us\n", i, d[i][3], (d[i][0] - d[0][0]) * 0.01, (d[i][1] - d[i][0]) * 0.01, d[i][2] > d[i][1] ? (d[i][2] - d[i][1]) * 0.01 : 0.0);
	}
#endif
	*out = b;
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
	if (!std::isfinite(root_lo[0] + root_lo[1] + root_lo[2] + root_hi[0] + root_hi[1] + root_hi[2])) return fail(RTX_ERR_ARG, "root bounds are not finite");
	{
		// one persistent launch (bvhq); level by level (below) only when its pools ran out or its watchdog fired
		rtx_bvh* q = nullptr;
		const int rcq = buildQueued(tri_pos, n_tris, root_lo, root_hi, ac_penalty, device, &q);
		if (rcq) return rcq;
		if (q) { *out = q; return RTX_OK; }
	}

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

int rtx_bvh_build_mode(int mode) { if (mode < 0 || mode > 2) return fail(RTX_ERR_ARG, "rtx_bvh_build_mode: 0, 1 or 2"); gBvhBuildMode = mode; return RTX_OK; }
int rtx_bvh_launches(const rtx_bvh* b, uint32_t* launches, int* queued)
{
	if (!b) return fail(RTX_ERR_ARG, "bvh is NULL");
	if (launches) *launches = b->launches;
	if (queued) *queued = b->queued ? 1 : 0;
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
	if (b->slab) { (void)hipFree(b->slab); delete b; return; }
	if (b->bounds) (void)hipFree(b->bounds);
	if (b->skip) (void)hipFree(b->skip);
	if (b->leafBegin) (void)hipFree(b->leafBegin);
	if (b->leafCount) (void)hipFree(b->leafCount);
	if (b->refs) (void)hipFree(b->refs);
	delete b;
}

} // extern "C"
