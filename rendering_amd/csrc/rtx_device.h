// Device-side data layout of the flattened scene (gfx950).  Shared by the kernels and the upload code.
//
// The 64 lanes of a wave hold 64 different rays (an 8x8 pixel tile or 16 SSAA pixels x 4 samples).  BVH nodes are
// wave-uniform and read through the scalar unit (one node = one 32-byte s_load_dwordx8, operands in SGPRs); leaf
// references are read one per lane with coalesced vector loads.
#pragma once
#include <stdint.h>

namespace rtxd {

// Pre-order BVH node (reference: AccelerationStructure, objects.h:125-164).
//   link  > 0 : inner node; link = pre-order index of the first node after this subtree ("skip")
//   link  < 0 : leaf with ~link triangles; `first` = index of its first leaf reference (refA / refB / refC)
struct Node {
	float b[6];      // lo.x hi.x | lo.y hi.y | lo.z hi.z : (lo_i, hi_i) pairs are the operands of the packed slab test
	int32_t link;
	int32_t first;
};
static_assert(sizeof(Node) == 32, "node must be one s_load_dwordx8");

// Four slots = the grandchildren of a binary node (a child that is a leaf takes one slot itself), left to right, in the
// layout of four Node records.  link > 0: inner node, link - 1 = index of its own WideNode; link < 0: leaf with ~link
// references from `first`; link == 0: empty slot.  One WideNode = two s_load_dwordx16: one dependent fetch per TWO levels
// of the reference's tree.  Skipping the boxes of the levels in between is exact: every box lies inside its parent's, and
// for nested boxes the slab products are monotone in the bounds, so a ray that passes a box passes every box around it --
// a ray reaches a leaf (objects.cpp:587-631) iff it passes the leaf's OWN box (no NaN: see meshWalk).
// RTX_WIDE_LEVELS binary levels per wide node: 2 (four slots, rounds 2-4) or 3 (eight slots: the descendants three levels down -- the walk is a chain
// of dependent fetches, one per wide level, and a third fewer levels is a third fewer visits; the slots are fetched four at a time).
#ifndef RTX_WIDE_LEVELS
#define RTX_WIDE_LEVELS 3
#endif
constexpr int kWideLevels = RTX_WIDE_LEVELS;
constexpr int kWideSlots = 1 << kWideLevels;
// entries of the walk's per-wave stack in LDS (meshWalk): at most kWideSlots - 1 per wide level + 1 -- rtx_scene_create checks a mesh's depth against it
// (a deeper tree is walked in the binary form).  Five blocks per CU hold 31 744 B of LDS each (the allocation granule): 72 entries x 16 B x 4 waves fit
// beside 25 parked fields and the six axis records of pruneEval8 (rtx_kernels.hip, pruneUni).
constexpr int kWideStackEntries = kWideSlots == 4 ? 56 : (kWideSlots == 8 ? 72 : 124);      // (72: ten wide levels = thirty binary ones need 71; 256 bytes of LDS went to pruneUni's axis records)
struct WideNode { Node slot[kWideSlots]; };
static_assert(sizeof(WideNode) == 32 * kWideSlots && (kWideSlots == 4 || kWideSlots == 8 || kWideSlots == 16), "wide node = two s_load_dwordx16 per four slots");

// Leaf references, duplicated per leaf in the reference's DFS-left-first order (objects.cpp:622-629): reference r of the
// mesh (r = Node::first + position in the leaf) is one entry of three parallel arrays, so that LANE i of a wave reads
// reference base + i with three fully coalesced vector loads (16 + 16 + 8 bytes per lane):
//   refA[r] = (v0.x, v0.y, v0.z, triangle index)    refB[r] = (e1.x, e1.y, e1.z, e2.x)    refC[r] = (e2.y, e2.z)
// e1 = b - a and e2 = c - a are the fp32 differences the reference recomputes per test (objects.cpp:70-71); they are
// ray-independent, so computing them once on upload is bit-identical.  The arrays are padded by 64 entries: a wave
// always loads 64 consecutive references, the ones past the end of the leaf are masked out.
//
// How a leaf is processed (DESIGN_HISTORY.md 3.3): the 64 lanes of the wave first act as 64 TRIANGLES -- each lane classifies its
// reference against the wave's whole ray BUNDLE (box of origins x box of directions) with division-free tests on the
// Moller-Trumbore numerators and rigorous rounding-error margins; a reference survives unless the reference renderer
// is CERTAIN to reject it for every ray of the bundle.  Then the lanes act as 64 RAYS again and run the reference's
// exact arithmetic on the few survivors (operands broadcast through the LDS crossbar, ds_bpermute), in the reference's order.
struct RefA { float v0x, v0y, v0z; uint32_t tri; };
struct RefB { float e1x, e1y, e1z, e2x; };
struct RefC { float e2y, e2z; };
static_assert(sizeof(RefA) == 16 && sizeof(RefB) == 16 && sizeof(RefC) == 8, "reference records = dwordx4 + dwordx4 + dwordx2");

// Prune record of one WideNode slot (DESIGN_HISTORY.md 3.1c): the TRUE box of every triangle referenced in the slot's subtree, as
// centre / half-extent, and P = the largest |e1|_1 |e2|_1 among them.  The reference finds a triangle through the loose
// cell of its leaf (objects.cpp:587-631) but what it ACCEPTS lies near the triangle: with det_c >= 1e-8 (objects.cpp:75-79)
// the point orig + t_c dir is within 36 u dmax |orig - v0|_inf P / 1e-8 of it, whatever the conditioning.  A slot whose
// inflated box no ray of the bundle can meet within [0, limit] cannot contribute and is not walked.  h < 0: no triangles.
//
// Source records (round 4, DESIGN_HISTORY.md 3.1d): the bound above takes the worst conditioning the reference lets through (det_c = 1e-8).
// When every ray of a walk passes within sigma of ONE point S -- the camera (primary rays: sigma = 0) or a point light (shadow
// rays: sigma ~ bias) -- the conditioning of an accepted pair is bounded by geometry instead: det = H |dir| / |S' - X*| with H the
// height of S' above the triangle's plane and X* the exact plane intersection, so the accepted point lies within
//     216 dmax |orig - v0|_inf P_S,    P_S ~ 0.0152 E u lsum D / (H |m|_2)                          (rtx_source.hip, sourceP)
// of the triangle -- the same formula with a per-source P, two to three orders of magnitude below s1 s2 for all but the few
// triangles whose plane passes close to S.  Every mesh therefore has 2 + nSrcLights copies of its prune blocks: copy 0 holds
// P = Pgen = max s1 s2 (any ray), copy 1 the camera's P_S (rebuilt with the view), copy 2 + l point light l's.  Pgen is the
// fallback inside a source copy where the certificate's side condition fails at run time (origin further than kSrcAinfMax from the box).
struct PruneRec { float c[3]; float P; float h[3]; float Pgen; };
static_assert(sizeof(PruneRec) == 32, "prune record = two dwordx4");
constexpr uint32_t kMaxSrcLights = 6;        // point lights with a source copy (lights beyond use copy 0)
constexpr float kSrcAinfMax = 32.0f;         // the source certificate assumes |orig - v0|_inf <= this (sourceP)
// Its companion: box (centre qc, half-extent qr) of the scaled plane normals q = (e2 x e1) / (|e1|_1 |e2|_1) of those triangles
// (|q|_inf <= 1) and the range [wlo, whi] of their plane offsets v0 . q.  The first stage of the bundle filter -- every ray
// certainly sees the back / starts beyond the plane / ends before it -- holds for ALL triangles of the slot when it holds for
// the box (interval arithmetic): det / (s1 s2) = dir . q,  Nt / (s1 s2) = v0 . q - orig . q.   No usable bound: q = 0 +- 0, [wlo, whi] = [-inf, +inf] (never rejected).
struct PlaneRec { float qc[3]; float wlo; float qr[3]; float whi; };
static_assert(sizeof(PlaneRec) == 32, "plane record = two dwordx4");
// per wide node: PruneRec[kWideSlots] then PlaneRec[kWideSlots] (slot order) = 64 bytes per slot.  Eight slots: lane 4 r + a of a wave reads words a and 4 + a of
// record r (r < 8 boxes, r >= 8 planes; a = 3: the records' fourth words) -- rtx_kernels.hip, pruneEval8; other widths: lane k of the first 2 kWideSlots lanes reads record k
struct PruneBlock { PruneRec box[kWideSlots]; PlaneRec plane[kWideSlots]; };
static_assert(sizeof(PruneBlock) == 64 * kWideSlots, "prune block");

struct Mesh {
	const Node* nodes;
	const RefA* refA;      // leaf references (see above), n_refs + 64 entries each
	const RefB* refB;
	const RefC* refC;
	const float* nrm;      // n_tris x 9
	const float* uv;       // n_tris x 6
	const float* tb;       // n_tris x 6 (tangent, bitangent) or null
	const float* diffuse;  // w*h x 3 or null
	const float* normal;   // w*h x 3 or null
	const float* specular; // w*h or null
	uint32_t nNodes, nRefs, nTris;
	uint32_t dW, dH, nW, nH, sW, sH;
	uint32_t boxesRegular;   // every node box is finite with lo <= hi on all axes (lets the walk use the min / max form of the box test)
	// The same tree with every other level skipped (`wide`, see WideNode): usable when no NaN can arise and every box
	// lies inside its parent's (nWide = 0 otherwise).
	const struct WideNode* wide;
	uint32_t nWide, padw;
	const PruneBlock* prune;   // one per wide node, or null
	float vmax, padv;          // largest |vertex coordinate| of the mesh
	PruneRec rootRec;          // the PruneRec of the whole mesh (h = +inf: none); rtx_scene_create decides from its P whether the box test can prune at all (Object::pruneBoxes)
	// bundle splitting (rtx_kernels.hip, traceWave): a wave's rays are walked as ONE bundle unless the bundle is wider than
	// fatRadius at this mesh (a few mean triangle edges); centre / radius = bounding sphere of the root box
	float fatRadius, centre[3], radius;
};

// 192 bytes = three s_load_dwordx16 issued together; everything Render::trace needs of a sphere or a plane is in the first.
struct Object {
	int32_t type, material;
	float pos[3];
	float r2;
	float normal[3];
	int32_t mesh;
	float color[3];
	float ior, ambient, diffuse, specular, nSpecular;
	// Meshes: what Render::trace needs before it decides to walk (a copy of the Mesh's fields: the second s_load_dwordx16 of the
	// record instead of a chain of dependent loads -- object -> mesh -> node array -> root node)
	float rootBox[6];          // lo.x hi.x | lo.y hi.y | lo.z hi.z of node 0
	float fatRadius, centre[3], radius;
	uint32_t meshFlags;        // bit 0: the mesh has nodes, 1: boxesRegular, 2: the wide walk is available
	uint32_t pad[2];
	// third line (meshes): what the walk itself needs
	const Node* nodes; const RefA* refA; const RefB* refB; const RefC* refC; const struct WideNode* wide; const struct PruneBlock* prune;
	uint32_t nNodes; float vmax;
	uint32_t pruneBoxes;       // the PruneRec test can prune for this mesh (small triangles: P well under 1 / 216, see rtx_scene_create)
	uint32_t srcStride;        // prune blocks per source copy (= the number of wide nodes), 0: the mesh has only copy 0
};
static_assert(sizeof(Object) == 192, "object record = three 64-byte lines");

// 64 bytes: one s_load_dwordx16 when the lanes of a wave are at the same light (the usual case).
struct Light {
	int32_t type;
	float color[3];
	float intensity;
	float dir[3];
	float pos[3];
	uint32_t nPoints;
	const float* points;
	uint32_t pad[2];
};
static_assert(sizeof(Light) == 64, "light record = one 64-byte line");

struct View {
	uint32_t width, height;
	float bias;
	int32_t maxDepth;
	float bg[3];
	uint32_t flags;
	float camPos[3];
	float camM[16];
	float scale, aspect;
};

// Kernel argument block (lives in the kernarg segment -> scalar loads).
struct Params {
	View view;
	const Object* objects;
	const Mesh* meshes;
	const Light* lights;
	const float* const* sky;        // device array of the six skybox faces (or null)
	uint32_t nObjects, nLights, skyW, skyH;
	// work distribution
	uint32_t rowBegin, rowEnd;      // pass 1: image rows [rowBegin,rowEnd)
	uint32_t tilesX, tileRow0, nTiles;
	// multi-GPU row ownership: row y belongs to this device iff (y / bandH) % nParts == part (bandH == 0: all rows)
	uint32_t bandH, nParts, part, halo;
	uint32_t* workCounter;          // persistent-wave work queue head (pass 1: eight heads, one per XCD, 64 B apart)
	const uint32_t* tileList;       // pass 1: [0,8) first entry of each XCD's queue, [8,16) its length, then the tiles (ty << 16 | tx)
	uint32_t tilesY, pad3;          // pass 1: tile rows covered by this launch
	const uint8_t* ssaaMask;        // Sobel mask consumed by the SSAA kernels
	const uint32_t* ssaaPixels;     // the flagged pixels (x | y << 16; ~0 = padding), tiles that were expensive in pass 1 first, tile by tile
	const uint32_t* ssaaScan;       // [t]: first list slot of heavy tile t, [nTiles + t]: of normal tile t, [2 nTiles]: number of flagged pixels
	uint32_t* tileCost;             // pass 1: wall-clock ticks (100 MHz) spent on each 8x8 tile of the frame
	uint32_t tilesXFull, pad2;      // tiles per row of the whole frame (tileCost indexing)
	// recursion frames: [slot][field][lane]
	float* frames;
	uint32_t totalLanes;
	uint32_t nSrcLights;            // point lights 0 .. nSrcLights - 1 have a source copy of the prune blocks (rtxd::PruneRec)
	float* fb;
	// probe rays (rtx_cast_rays)
	const float* probeRays;
	float* probeHits;
	float* probeColours;
	uint32_t nProbe;
	float srcNmax2;                 // (1 + 1e-5) x the square of the longest shading normal the source copies of the point lights were built for (rtx_api.hip, buildSources: sigma ~ |N| bias)
	unsigned long long* counters;   // rays, boxTests, triTests
	// whole frame in one launch (rtxFrameKernel): per-tile dependency counters, the SSAA item queue
	uint32_t* tileReady;            // [tile] pass-1 completions seen among the listed tiles of the tile's 3x3 neighbourhood
	uint32_t* tileSobel;            // [tile] Sobel completions seen among them
	const uint8_t* tileNeed;        // [tile] number of listed tiles in the 3x3 neighbourhood; 0 = the tile itself is not listed
	unsigned long long* tileFlags;  // [tile] Sobel-flagged pixels of the tile (bit r * 8 + c)
	unsigned long long* ssaaQueue;  // 64 queues of queueCap SSAA items: epoch << 32 | tile << 8 | group << 2 | (0: 16, 1: 4, 2: 1 pixels per group)
	unsigned long long* frameCtl;   // control block (rtx_kernels.hip, FC_*)
	const uint32_t* countExpect;    // [k], k < 64: listed tiles with index % 64 == k; [64]: how many of those are not zero
	const uint32_t* splitLimits;    // [0], [1]: pass-1 cost (ticks) above which a tile is rendered / re-sampled in 4, in 16 parts (rtxTileOrderKernel)
	uint8_t* maskOut;               // Sobel mask written by the frame kernel
	uint32_t listedTiles, epoch, queueCap, veryBudget, tilesYFull, heavyTicks, stripBit, padF1;      // stripBit: pass 1, 0x10000000 when the tile list may hold 64 x 1 strips
};

constexpr int kFrameFields = 14;

} // namespace rtxd
