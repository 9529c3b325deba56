// Device-side data layout of the flattened scene (gfx950).  Shared by the kernels and the upload code.
//
// Everything the traversal touches is laid out so that a WAVE reads it through the scalar unit: one BVH
// node is one 32-byte s_load_dwordx8, two leaf triangles are one 80-byte s_load_dwordx16 + s_load_dwordx4.
// The 64 lanes of a wave hold 64 different rays (an 8x8 pixel tile or 16 SSAA pixels x 4 samples); node and
// triangle operands sit in SGPRs, so the VALU does nothing but the reference's own arithmetic.
#pragma once
#include <stdint.h>

namespace rtxd {

// Pre-order BVH node (reference: AccelerationStructure, objects.h:125-164).
//   link  > 0 : inner node; link = pre-order index of the first node after this subtree ("skip")
//   link  < 0 : leaf with ~link triangles; `first` = index of its LeafHeader in the pair array, pairs follow
struct Node {
	float b[6];      // lo.x hi.x | lo.y hi.y | lo.z hi.z : (lo_i, hi_i) pairs are the operands of the packed slab test
	int32_t link;
	int32_t first;
};
static_assert(sizeof(Node) == 32, "node must be one s_load_dwordx8");

// Leaf references, duplicated per leaf in the reference's DFS-left-first order (objects.cpp:622-629), stored in
// PAIRS: 80 bytes = two triangles INTERLEAVED field by field, fetched with one s_load_dwordx16 + one s_load_dwordx4.
// Every field is an even-aligned SGPR pair (A, B), i.e. directly an operand of the packed-f32 VALU ops
// (v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 operations per instruction at the issue cost of one -- measured,
// tools/ubench/pk_rate.hip), so the Moeller-Trumbore arithmetic of both triangles is issued once.
// A leaf with an odd number of references is padded with a degenerate triangle (all zero: det = 0 is rejected by
// the reference's own epsilon test, objects.cpp:76-79), so a leaf always starts on a pair boundary.
// Inside a leaf the references are stored in SPATIAL order (Morton code of the centroids), not in the reference's
// vector order, so that a chunk of consecutive references is a compact patch with a tight box and a narrow normal
// range.  The reference keeps the FIRST of several accepted triangles with equal t (strict `<`, objects.cpp:623), so
// every record carries its position in the reference's order and the walk breaks exact ties with it -- the winner
// is the same triangle, bit for bit.
// e1 = b - a and e2 = c - a are the fp32 differences the reference recomputes per test (objects.cpp:70-71);
// they are ray-independent, so computing them once on upload is bit-identical.
struct LeafPair {
	float e2x[2], e2y[2], e2z[2];
	float e1x[2], e1y[2], e1z[2];
	float v0x[2], v0y[2];
	float v0z[2];
	uint32_t tri[2];     // triangle index (into the per-triangle shading arrays) | position in the reference's leaf order << Mesh::triBits
};
static_assert(sizeof(LeafPair) == 80, "pair record = s_load_dwordx16 + s_load_dwordx4");

// Certificate headers (one pair slot, fetched with one s_load_dwordx16).  Leaf layout in the pair array:
//   n <= kChunkTris:  [leaf header] pairs
//   n <= kGroupTris:  [leaf header] { [chunk header] kChunkTris/2 pairs }...
//   larger:           [leaf header] { [group header] { [chunk header] kChunkTris/2 pairs }... }...   (pad[0] of a group
//                     header = pair slots of the group, what a wave jumps over)
// A header lets a wave skip all its references for a ray when the reference is CERTAIN to reject every one of them
// -- skipping is then exact.  Three certificates, all with rigorous fp32 rounding-error bounds (derivation:
// DESIGN.md section 3.3):
//   (1) back-face:  det = v0v1 . (dir x v0v2) = dir . m with m = v0v2 x v0v1.  [mlo, mhi] bounds m component-wise
//       over the range, so U = sum_i max(dir_i*mlo_i, dir_i*mhi_i) >= det_exact and L = sum_i min(..) <= det_exact.
//       `err` (per unit of max|dir_i|) bounds the reference's rounding error of det plus the error of evaluating
//       U / L in fp32.  U < -err*dmax  =>  det_computed < 0 < 1e-8 for every triangle (objects.cpp:75-77, culling on).
//   (2) behind the origin:  if every triangle certainly faces the ray (L >= 4*err*dmax, so det >= g = L - 2*err*dmax)
//       and the range's true AABB [blo, bhi] lies behind the ray origin by more than the error budget
//       (-boxdot * g > dmax^2 * (Dinf*a1 + a2)), then any triangle that passes the reference's det / u / v tests gets
//       a computed t < 0 and is rejected by objects.cpp:91.  (The reference's box test has no t range, so e.g. every
//       shadow ray leaving the mesh walks all the leaves behind it.)
//   (3) missed:  same g; the exact plane hit of any accepted triangle lies within rho <= 2*dmax*(Dinf*a1 + a2)/g of the
//       triangle, so a ray LINE that misses [blo - rho, bhi + rho] cannot be accepted by any triangle of the range.
struct LeafHeader {
	float m[3][2];     // (mlo_i, mhi_i): interval of m = v0v2 x v0v1, an operand pair of the packed instructions
	float err;
	float a1;
	float b[3][2];     // (blo_i, bhi_i): the AABB
	float a2;
	uint32_t pad[5];   // pad[0]: group headers: pair slots of the group (what a wave jumps over)
};
static_assert(sizeof(LeafHeader) == sizeof(LeafPair), "header occupies one pair slot");

struct Mesh {
	const Node* nodes;
	const LeafPair* leaf;   // pairs of leaf references (+1 pair: the walk's prefetch may run one pair past a leaf)
	const float* nrm;      // n_tris x 9
	const float* uv;       // n_tris x 6
	const float* tb;       // n_tris x 6 (tangent, bitangent) or null
	const float* diffuse;  // w*h x 3 or null
	const float* normal;   // w*h x 3 or null
	const float* specular; // w*h or null
	uint32_t nNodes, nRefs, nTris;
	uint32_t dW, dH, nW, nH, sW, sH;
	uint32_t triBits;      // low bits of LeafPair::tri that hold the triangle index
};

struct Object {
	int32_t type, material;
	float pos[3];
	float color[3];
	float ior, ambient, diffuse, specular, nSpecular;
	float r2;
	float normal[3];
	int32_t mesh;
};

struct Light {
	int32_t type;
	float color[3];
	float intensity;
	float dir[3];
	float pos[3];
	uint32_t nPoints;
	const float* points;
};

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
	const float* sky[6];
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
	uint32_t pad0;
	float* fb;
	// probe rays (rtx_cast_rays)
	const float* probeRays;
	float* probeHits;
	float* probeColours;
	uint32_t nProbe;
	uint32_t pad1;
	unsigned long long* counters;   // rays, boxTests, triTests
};

constexpr int kFrameFields = 14;
constexpr uint32_t kGroupTris = 64; // leaves with more references also carry one LeafHeader per kGroupTris references (LeafHeader::pad[0] = slots of the group)
constexpr uint32_t kChunkTris = 8; // leaves with more references carry one extra LeafHeader per kChunkTris references

} // namespace rtxd
