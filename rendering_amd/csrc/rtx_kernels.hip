// gfx950 kernels of the ray-trace hot path: Render::castRay / Render::trace / BVH walk / Moller-Trumbore /
// Phong-Fresnel-texture-skybox shading / Sobel mask / SSAA (reference: src/scene.cpp:381-946,
// src/objects.cpp:59-175,534-631,766-824, src/lights.cpp:18-63).  Citations are file:line into the reference.
//
// Execution model (DESIGN_HISTORY.md section 3):
//   * persistent waves pull work items (8x8 pixel tiles, or 16 / 4 / 1 SSAA pixels x 4 samples) from atomic queues;
//   * the 64 lanes of a wave are 64 rays = one BUNDLE; every Render::trace of the wave is ONE cooperative walk
//     (meshWalk): the reference's tree three levels per fetch (rtxd::WideNode: eight slots, scalar loads, a wave-level stack
//     in LDS), every lane testing the slot boxes on its own ray with the reference's arithmetic; slots whose triangles no ray
//     of the bundle can hit are dropped beforehand (rtxd::PruneBlock; pruneEval8: the lanes acting as (record, axis) pairs --
//     the inequalities are those of pruneAlive / planeAlive, which remain for the other node widths);
//   * the references of the reached leaves are streamed 64 at a time with the lanes acting as TRIANGLES: the bundle
//     filter (bundleRejects1/2) keeps what some ray might hit, the survivors are tested exactly (the reference's
//     Moller-Trumbore, operation by operation) with the lanes acting as rays again, in the reference's order --
//     "strict <, first hit wins" (objects.cpp:587-631);
//   * what does not allow the wide walk (boxes not nested, a tree deeper than the stack, the instrumented variant; RAYS with
//     a zero direction component or a huge origin, walked apart from the rest of their wave) takes the stackless binary walk:
//     a lane that fails a box sleeps until the wave's cursor reaches the node's skip index.  Back-face culling off is a
//     kernel of its own (CULLK): the same wide walk with the culling-free forms of the filter and of the plane records;
//   * castRay's recursion is an explicit per-lane frame stack in HBM ([slot][field][lane], coalesced) so the
//     nested colour expressions keep the reference's association order (scene.cpp:858-940).
//
// Numerics contract: compiled with -ffp-contract=off, no fast-math; fp32 divide/sqrt are the
// correctly rounded expansions; fp64 islands where the reference has them; powf is the restated glibc 2.35
// (FMA variant) algorithm.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtx_device.h"
#include "rtx_tuning.h"

#pragma clang fp contract(off)

using namespace rtxd;

// the cheap half of a pass-1 queue (tiles from this fraction of the queue on) is popped RTX_POP_MANY tiles per atomic
constexpr uint32_t kPopNum = 1u, kPopDen = 2u;

namespace {

// diagnostics that exist only in the instrumented build (RTX_DBG, rtx_tuning.h)
#if RTX_DBG
#define RTX_DBG_ONLY(...) __VA_ARGS__
#else
#define RTX_DBG_ONLY(...)
#endif
#if RTX_DBG || RTX_WAVE_TRACE
#define RTX_TRACE_ONLY(...) __VA_ARGS__
#else
#define RTX_TRACE_ONLY(...)
#endif

#define RTX_AS4 __attribute__((address_space(4)))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Wave-uniform loads through the constant address space -> SMEM instructions.
__device__ __forceinline__ uint32_t sload1(const void* p) { return *(const RTX_AS4 uint32_t*)(uintptr_t)p; }
__device__ __forceinline__ u32x8 sload8(const void* p) { return *(const RTX_AS4 u32x8*)(uintptr_t)p; }
__device__ __forceinline__ u32x16 sload16(const void* p) { return *(const RTX_AS4 u32x16*)(uintptr_t)p; }
__device__ __forceinline__ u32x4 sload4(const void* p) { return *(const RTX_AS4 u32x4*)(uintptr_t)p; }
__device__ __forceinline__ float sloadf(const float* p) { return __uint_as_float(sload1(p)); }
__device__ __forceinline__ const void* sloadp(const void* p)
{
	u32x2 v = *(const RTX_AS4 u32x2*)(uintptr_t)p;
	return (const void*)(((uint64_t)v.y << 32) | v.x);
}
#define F(x) __uint_as_float(x)
// Pins a wave-uniform value into SGPRs (v_readfirstlane) so that addresses derived from it select SMEM loads
// even when register pressure made the compiler park it in a VGPR.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ const T* uni(const T* p)
{
	const uint64_t a = (uint64_t)p;
	return (const T*)(((uint64_t)uni((uint32_t)(a >> 32)) << 32) | uni((uint32_t)a));
}

__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// lane index, recomputed where it is needed (two VALU instructions) instead of being kept live / spilled across the walk
__device__ __forceinline__ uint32_t laneNow()
{
	uint32_t l;
	asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
	return l;
}
// lane `lane` (wave-uniform) of v := val (wave-uniform); the lane select goes through m0 (constant-bus limit of gfx9)
__device__ __forceinline__ uint32_t writeLane(uint32_t val, uint32_t lane, uint32_t v)
{
	val = __builtin_amdgcn_readfirstlane(val); lane = __builtin_amdgcn_readfirstlane(lane);      // (folded away when already scalar)
	asm("" : "+s"(val));      // a register, never a literal (v_writelane_b32 takes no 32-bit literal on gfx9)
	asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val), "s"(lane) : "m0");
	return v;
}
#define RTX_AS1 __attribute__((address_space(1)))
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
// leaf reference r of a mesh through global (address space 1) vector loads: dwordx4 + dwordx4 + dwordx2 per lane
__device__ __forceinline__ void loadRef(const RTX_AS1 f4v* A, const RTX_AS1 f4v* Bp, const RTX_AS1 f2v* Cp, uint32_t r, RefA& a, RefB& b, RefC& c)
{
	const f4v va = A[r], vb = Bp[r];
	const f2v vc = Cp[r];
	a.v0x = va.x; a.v0y = va.y; a.v0z = va.z; a.tri = __float_as_uint(va.w);
	b.e1x = vb.x; b.e1y = vb.y; b.e1z = vb.z; b.e2x = vb.w;
	c.e2y = vc.x; c.e2z = vc.y;
}
__device__ __forceinline__ uint64_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }

constexpr float kFltMax = 3.402823466e+38f;
// (double)x < 1e-8 for a float x  <=>  x < 0x322bcc78 (the smallest float >= the double 1e-8);
// objects.cpp:76,79,810 compare in double.
#define RTX_EPS8 __uint_as_float(0x322bcc78u)
constexpr uint32_t kNever = 0x7fffffffu;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // geometry.h:84-87
__device__ __forceinline__ float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// geometry.h:99-102: (float)sqrt((double)len2) == correctly rounded sqrtf (53 >= 2*24+2)
__device__ __forceinline__ float length(V3 a) { return __builtin_sqrtf(len2(a)); }
// geometry.h:104-112: factor = (float)(1 / sqrt((double)len2)).  As written that is ~40 fp64 instructions (an fp64 square root and an fp64 division: two quarter-rate
// transcendentals and two dozen half-rate FMAs), several times per shaded point (the primary direction, the normal, every light's L).  The fast path below
// reaches the SAME float in 16 fp32 instructions: v_rsq_f32, one Newton step whose residual 1 - l2 y^2 is formed exactly (FMA: l2 y = t + te), and the exact
// residual e1 of the candidate y1, which says how far y1 is from the true 1 / sqrt(l2) in units of its own last place: |y1 e1| / 2 < ulp / 2 (1 - 2^-16) means y1 is the
// correctly rounded float of the true value with room to spare, and the reference's value -- the true value rounded to double twice (relative 2^-52), then to
// float -- can only differ from that within 2^-28 ulp of a rounding boundary.  Otherwise (15 inputs in a million), and outside [2^-60, 2^60], the lane takes the
// fp64 path.  Proof by enumeration: tools/research/rsqrt_exhaustive.c tries every mantissa, both exponent parities and every starting value within 3 ulp
// of the truth (352 M cases: accepted => bit-identical; v_rsq_f32 is good to 1 ulp); on the device tests/test_gpu_parity.py runs all 2^24 mantissas.
__device__ __forceinline__ float invLenSlow(float l2) { return (float)(1.0 / __builtin_sqrt((double)l2)); }
__device__ __forceinline__ float invLenD(float l2)
{
#if RTX_FAST_INVLEN
	const float y0 = __builtin_amdgcn_rsqf(l2);
	const float t = l2 * y0, te = __builtin_fmaf(l2, y0, -t);
	float e = __builtin_fmaf(-t, y0, 1.0f); e = __builtin_fmaf(-te, y0, e);
	const float y1 = __builtin_fmaf(y0 * 0.5f, e, y0);
	const float t1 = l2 * y1, t1e = __builtin_fmaf(l2, y1, -t1);
	float e1 = __builtin_fmaf(-t1, y1, 1.0f); e1 = __builtin_fmaf(-t1e, y1, e1);
	const float ulp = __uint_as_float((__float_as_uint(y1) & 0x7f800000u) - (23u << 23));
	const bool sure = fabsf(e1 * y1) < ulp * 0.99998474f && l2 >= 0x1p-60f && l2 <= 0x1p60f;      // (NaN compares false: the fp64 path)
	float r = y1;
	if (!sure) r = invLenSlow(l2);
	return r;
#else
	return invLenSlow(l2);
#endif
}
__device__ __forceinline__ V3 normalized(V3 a)
{
	float l2 = len2(a);
	if (l2 > 0) {
		float f = invLenD(l2);
		a.x *= f; a.y *= f; a.z *= f;
	}
	return a;
}
__device__ __forceinline__ float fminRef(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b)
__device__ __forceinline__ float fmaxRef(float a, float b) { return (a < b) ? b : a; }   // std::max(a,b)
__device__ __forceinline__ float clampRef(float lo, float hi, float v) { return fmaxRef(lo, fminRef(hi, v)); }

// ------------------------------------------------------------------------------------------------
// powf -- glibc 2.35 e_powf.c (ARM optimized-routines) as executed by the x86-64 FMA ifunc variant.
// Tables: __powf_log2_data (16 x {invc, logc}, poly A[5]) and __exp2f_data (32 x u64, shift, poly C[3]).
// ------------------------------------------------------------------------------------------------
__device__ const double kLog2Tab[32] = {
	0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
	0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2,  0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
	0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3,
	0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
	0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1p+0, 0x0p+0,
	0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4,  0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3,
	0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3,  0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
	0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,  0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
};
__device__ const uint64_t kExp2Tab[32] = {
	0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
	0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
	0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
	0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
	0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
	0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
	0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

// The two tables in LDS (512 bytes per block, copied by fillPowTab at the start of every ray kernel): the lookups are per lane and
// DEPENDENT (the exp2 entry's index comes out of the logarithm) -- from global memory two round trips of a microsecond or two in
// every Phong / mirror / glass shading step, the longest part of a trace round of a scene of spheres (RTX_DBG=3: 12 000 of a
// round's 17 000 cycles).
__shared__ unsigned long long powTab[64];      // [0, 32): kLog2Tab as bits, [32, 64): kExp2Tab
__device__ __forceinline__ void fillPowTab()
{
	if (threadIdx.x < 32) powTab[threadIdx.x] = (unsigned long long)__double_as_longlong(kLog2Tab[threadIdx.x]);
	else if (threadIdx.x < 64) powTab[threadIdx.x] = kExp2Tab[threadIdx.x - 32];
	__syncthreads();
}

__device__ __forceinline__ int powfCheckInt(uint32_t iy)
{
	int e = iy >> 23 & 0xff;
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
__device__ __forceinline__ bool zeroInfNan(uint32_t i) { return 2 * i - 1 >= 2u * 0x7f800000 - 1; }

__device__ __noinline__ float powfRef(float x, float y)
{
	uint32_t signBias = 0;
	uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
	if (ix - 0x00800000 >= 0x7f800000 - 0x00800000 || zeroInfNan(iy)) {
		if (zeroInfNan(iy)) {
			if (2 * iy == 0) return 1.0f;
			if (ix == 0x3f800000) return 1.0f;
			if (2 * ix > 2u * 0x7f800000 || 2 * iy > 2u * 0x7f800000) return x + y;
			if (2 * ix == 2 * 0x3f800000) return 1.0f;
			if ((2 * ix < 2 * 0x3f800000) == !(iy & 0x80000000)) return 0.0f;
			return y * y;
		}
		if (zeroInfNan(ix)) {
			float x2 = x * x;
			if ((ix & 0x80000000) && powfCheckInt(iy) == 1) x2 = -x2;
			return (iy & 0x80000000) ? 1 / x2 : x2;
		}
		if (ix & 0x80000000) {
			int yint = powfCheckInt(iy);
			if (yint == 0) return __uint_as_float(0x7fc00000u);
			if (yint == 1) signBias = 1u << 16;
			ix &= 0x7fffffff;
		}
		if (ix < 0x00800000) {
			ix = __float_as_uint(x * 0x1p23f);
			ix &= 0x7fffffff;
			ix -= 23 << 23;
		}
	}
	uint32_t tmp = ix - 0x3f330000;
	int i = (tmp >> 19) % 16;
	uint32_t top = tmp & 0xff800000;
	uint32_t iz = ix - top;
	int k = (int32_t)top >> 23;
	double invc = __longlong_as_double((long long)powTab[2 * i]), logc = __longlong_as_double((long long)powTab[2 * i + 1]);
	double z = (double)__uint_as_float(iz);
	double r = __builtin_fma(z, invc, -1.0);
	double y0 = logc + (double)k;
	double r2 = r * r;
	double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
	double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
	double r4 = r2 * r2;
	double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
	q = __builtin_fma(p, r2, q);
	double logx = __builtin_fma(yy, r4, q);
	double ylogx = (double)y * logx;
	if ((__double_as_longlong(ylogx) >> 47 & 0xffff) >= (0x405f800000000000LL >> 47)) {
		if (ylogx > 0x1.fffffffd1d571p+6) return signBias ? -__builtin_inff() : __builtin_inff();
		if (ylogx <= -150.0) return signBias ? -0.0f : 0.0f;
		if (ylogx < -149.0) return signBias ? -0x1p-149f : 0x1p-149f;
	}
	double kd = ylogx + 0x1.8p+47;
	uint64_t ki = (uint64_t)__double_as_longlong(kd);
	kd -= 0x1.8p+47;
	double rr = ylogx - kd;
	uint64_t t = powTab[32 + ki % 32];
	t += (ki + signBias) << (52 - 5);
	double s = __longlong_as_double((long long)t);
	double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
	double rr2 = rr * rr;
	double yv = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
	yv = __builtin_fma(zz, rr2, yv);
	yv = yv * s;
	return (float)yv;
}

// ------------------------------------------------------------------------------------------------
// shading helpers (scene.cpp:672-722, 381-442; lights.cpp:18-38)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ V3 reflectDir(V3 d, V3 n) { float k = 2 * dot(d, n); return d - n * k; }   // scene.cpp:674

__device__ __forceinline__ V3 refractDir(V3 d, V3 n, float ior)      // scene.cpp:677-696
{
	float n1 = 1, n2 = ior;
	float cosi = clampRef(-1, 1, dot(d, n));
	V3 mn = n;
	if (cosi < 0) cosi = -cosi;
	else { float t = n1; n1 = n2; n2 = t; mn = -n; }
	float rri = n1 / n2;
	float k = 1 - rri * rri * (1 - cosi * cosi);
	if (k < 0) return mk(0, 0, 0);
	return d * rri + mn * (rri * cosi - __builtin_sqrtf(k));
}

__device__ __forceinline__ float fresnelKr(V3 d, V3 n, float ior)    // scene.cpp:698-722
{
	float n1 = 1, n2 = ior;
	float cosi = clampRef(-1, 1, dot(d, n));
	if (cosi > 0) { float t = n1; n1 = n2; n2 = t; }
	float sint = n1 / n2 * __builtin_sqrtf(fmaxRef(0.f, 1 - cosi * cosi));
	if (sint >= 1) return 1;
	float cost = __builtin_sqrtf(fmaxRef(0.f, 1 - sint * sint));
	cosi = fabsf(cosi);
	float rs = ((n2 * cosi) - (n1 * cost)) / ((n2 * cosi) + (n1 * cost));
	float rp = ((n1 * cosi) - (n2 * cost)) / ((n1 * cosi) + (n2 * cost));
	return (rs * rs + rp * rp) / 2;
}

__device__ __forceinline__ int toPixel(float v, int mx)               // scene.cpp:387-392
{
	int val = (int)((v + 1.0f) / 2.0f * mx);
	if (val >= mx) val = mx - 1;
	if (val < 0) val = 0;            // hardening only (NaN / degenerate directions index out of bounds in the reference)
	return val;
}
__device__ __forceinline__ V3 load3(const float* p) { return mk(p[0], p[1], p[2]); }
// A pointer that was itself loaded from memory (mesh arrays, texture maps, skybox faces, light sample points) has no
// known address space: the compiler reads through it with flat_load, which counts against the LDS / scalar counter as
// well as the vector-memory one.  These all point to device memory.
typedef const __attribute__((address_space(1))) float* GlobalFloats;
__device__ __forceinline__ GlobalFloats inGlobal(const float* p) { return (GlobalFloats)(uintptr_t)p; }
__device__ __forceinline__ V3 load3(GlobalFloats p) { return mk(p[0], p[1], p[2]); }

// (takes the few fields it needs by value: a kernel-argument struct whose address escapes into a call is copied to scratch)
__device__ __noinline__ V3 skyFetch(const float* const* sky, int W, int H, V3 dir)           // scene.cpp:394-441
{
	float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
	float m = fmaxRef(ax, fmaxRef(ay, az));
	V3 a; int face, i, j;
	if (m == az) {
		if (dir.z < 0) { a = dir * (1 / -dir.z); face = 1; i = toPixel(a.y, H); j = toPixel(a.x, W); }
		else { a = dir * (1 / dir.z); face = 3; i = toPixel(a.y, H); j = toPixel(-a.x, W); }
	}
	else if (m == ax) {
		if (dir.x < 0) { a = dir * (1 / -dir.x); face = 0; i = toPixel(a.y, H); j = toPixel(-a.z, W); }
		else { a = dir * (1 / dir.x); face = 2; i = toPixel(a.y, H); j = toPixel(a.z, W); }
	}
	else {
		if (dir.y < 0) { a = dir * (1 / -dir.y); face = 5; i = toPixel(a.z, H); j = toPixel(a.x, W); }
		else { a = dir * (1 / dir.y); face = 4; i = toPixel(a.z, H); j = toPixel(a.x, W); }
	}
	return load3(inGlobal(sky[face]) + ((size_t)i * W + j) * 3);
}

__device__ __forceinline__ V3 skyColor(const Params& P, V3 dir)           // scene.cpp:381-442
{
	if (!(P.view.flags & 2u)) return mk(P.view.bg[0], P.view.bg[1], P.view.bg[2]);
	return skyFetch(P.sky, (int)P.skyW, (int)P.skyH, dir);
}

// (int)(dim * coord), clamped on the high side like the reference (objects.cpp:144-147, 156-159).  The low-side clamp is
// hardening only: negative / NaN coordinates are out-of-bounds reads (UB) in the reference.
__device__ __forceinline__ int texel(int dim, float coord) { int v = (int)(dim * coord); if (v >= dim) v = dim - 1; if (v < 0) v = 0; return v; }

// 4*M_PI*len2/1000 attenuation in fp64 (lights.cpp:35; scene.cpp:796,832,875,925)
__device__ __forceinline__ float attenuation(float intensity, float l2)
{
	return fminRef(1.0f, (float)((double)intensity / (4 * 3.14159265358979323846 * (double)l2 / 1000)));
}

// ------------------------------------------------------------------------------------------------
// Render::trace for a whole wave (scene.cpp:724-756)
// ------------------------------------------------------------------------------------------------
struct Hit { int obj; float t; uint32_t tri; float u, v; };
RTX_TRACE_ONLY(__device__ unsigned long long gDbgWave[3 * 16384];)   // per wave of the last pass 1: first pop, last tile end, busy ticks
RTX_DBG_ONLY(
__device__ unsigned long long gDbgTimeline[3 * 8192 * 160];   // frame kernel: per wave up to 160 work items (start, duration, kind << 32 | item); start 0 = unused
__device__ unsigned long long gDbgHist[64];   // [0,8) certificate outcomes (one sampled lane per evaluation), [16,64) by log2(leaf size)
)
#if RTX_DBG >= 3      // per-block timers of the castRay state machine (their atomics disturb a full launch: use on small frames)
#define RTX_T0 const unsigned long long dbgB0 = __builtin_readcyclecounter();
#define RTX_ACC(k) { const unsigned long long dbgE = __builtin_readcyclecounter() - dbgB0; if (__lane_id() == (uint32_t)__ffsll((long long)ballot(true)) - 1u) { atomicAdd(&gDbgHist[16 + 2 * (k)], dbgE); atomicAdd(&gDbgHist[17 + 2 * (k)], 1ull); } }
#else
#define RTX_T0
#define RTX_ACC(k)
#endif
struct Counts { unsigned long long rays, box, tri, wNodes, wTri, wS2, wS3, wS4, wLeaves, wLeafSkips, wChunks, wChunkSkips, triLanes, moot, cNodes, cFilter, cExact,
                aWalks, sWalks, sVisits, sLeaves, sPasses, sExact; };      // (RTX_DBG: wide walks; s*: of shadow bundles only)

// Ordering of scalar loads.  SMEM returns out of order, so lgkmcnt can only be waited down to zero: a load issued
// before the first use of the previous one is covered by the same wait and nothing overlaps.  after(x, v) is an
// empty asm that makes x depend on the value v having ARRIVED; deriving the next record's address from it forces
//     wait(current) -> issue(next) -> compute(current)
// without emitting an instruction.  The loads stay ordinary IR loads: every s_waitcnt is placed by the compiler.
template <typename T> __device__ __forceinline__ T after(T x, uint32_t v)
{
	asm("" : "+s"(x) : "s"(v));
	return x;
}

// ------------------------------------------------------------------------------------------------
// The ray bundle of one Render::trace of the wave, and the bundle filter (DESIGN_HISTORY.md 3.3)
// ------------------------------------------------------------------------------------------------
// Wave-wide maximum over the lanes (all 64 lanes execute this; lanes that must not contribute pass -inf).  Four DPP
// steps inside each row of 16, two row broadcasts, the result is read from lane 63.
__device__ __forceinline__ float waveMax(float v)
{
	// (one v_max_f32 with a DPP source per step; s_nop 1 = the two wait states before a DPP read of a just-written VGPR)
#define RTX_STEP(ctrl) "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl "\n\t"
	asm volatile(RTX_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") RTX_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
	             RTX_STEP("row_half_mirror row_mask:0xf bank_mask:0xf") RTX_STEP("row_mirror row_mask:0xf bank_mask:0xf")
	             RTX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") RTX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
	             "s_nop 0" : "+v"(v));
#undef RTX_STEP
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Box of origins x box of directions of the active rays of the wave, as centre / half-width (wave-uniform values,
// kept in SGPRs).  The half-widths are inflated so that every lane's fp32 origin / direction really lies inside
// [c - r, c + r] despite the rounding of c and r themselves.
struct Bundle {
	float ocx, ocy, ocz, rox, roy, roz;
	float dcx, dcy, dcz, rdx, rdy, rdz;
	float kd;          // K * dmax            (K = 2^-18, dmax = max |dir_i| over the bundle)
	float roMax;       // max_i ro_i
	float kdRoSum;     // dmax * (rox + roy + roz)
	float roRd;        // (rox + roy + roz) * max_i rd_i: the second-order part of the u / v radii (bundleRejects2)
	bool sane;         // every active lane has finite, moderate coordinates (|orig_i| < 2^40, |dir_i| < 2^20)
};
constexpr float kFilterK = 0x1p-18f;     // 64 u: covers the reference's rounding errors AND the filter's own (see bundleRejects)
constexpr float kFilterEta = 1e-30f;     // absolute slack for underflow in either computation

__device__ __forceinline__ void centreRadius(float lo, float hi, float& c, float& r)
{
	c = unif(0.5f * (lo + hi));
	r = unif((0.5f * (hi - lo)) * (1.0f + 0x1p-19f) + 0x1p-21f * fmaxf(fabsf(lo), fabsf(hi)));
}

// max and min of one value over the wave, as two interleaved chains of v_max_f32 / v_min_f32 with a DPP source (one
// instruction per step and chain; s_nop 0 + the other chain's instruction = the two wait states a DPP read of a
// just-written VGPR needs).  Through __builtin_amdgcn_update_dpp every step is v_mov + v_mov_dpp + canonicalising v_max + v_max.
__device__ __forceinline__ void waveMaxMin(float& hi, float& lo)
{
#define RTX_STEP(ctrl) "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_min_f32_dpp %1, %1, %1 " ctrl "\n\ts_nop 0\n\t"
	asm volatile("s_nop 1\n\t"
		RTX_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") RTX_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
		RTX_STEP("row_half_mirror row_mask:0xf bank_mask:0xf") RTX_STEP("row_mirror row_mask:0xf bank_mask:0xf")
		RTX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") RTX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
		"s_nop 0" : "+v"(hi), "+v"(lo));
#undef RTX_STEP
	hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hi), 63));
	lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lo), 63));
}

// sameOrigin: every active lane's origin is the SAME point bit for bit (rays handed in at the camera: traceWave) -- the origin box is that point,
// radius 0 (three of the six wave reductions are not needed)
__device__ __forceinline__ Bundle makeBundle(bool active, const V3& o, const V3& d, bool sameOrigin = false)
{
	const float ninf = -__builtin_inff();
	Bundle B;
	float lo, hi;
	// (RTX_BUNDLE_PAIRS: max and min of a coordinate as one pair of interleaved DPP chains, waveMaxMin -- rtx_tuning.h)
#if RTX_BUNDLE_PAIRS
#define RTX_RANGE(x, c, r) hi = active ? (x) : ninf; lo = active ? (x) : -ninf; waveMaxMin(hi, lo); centreRadius(lo, hi, c, r)
#else
#define RTX_RANGE(x, c, r) hi = waveMax(active ? (x) : ninf); lo = -waveMax(active ? -(x) : ninf); centreRadius(lo, hi, c, r)
#endif
	if (RTX_SAME_ORIGIN && sameOrigin) {
		const int first = __builtin_ctzll(ballot(active) | (1ull << 63));
		B.ocx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), first)); B.ocy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), first));
		B.ocz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), first));
		B.rox = B.roy = B.roz = 0.0f;
	}
	else {
		RTX_RANGE(o.x, B.ocx, B.rox); RTX_RANGE(o.y, B.ocy, B.roy); RTX_RANGE(o.z, B.ocz, B.roz);
	}
	RTX_RANGE(d.x, B.dcx, B.rdx); RTX_RANGE(d.y, B.dcy, B.rdy); RTX_RANGE(d.z, B.dcz, B.rdz);
#undef RTX_RANGE
	const float dmax = fmaxf(fmaxf(fabsf(B.dcx) + B.rdx, fabsf(B.dcy) + B.rdy), fabsf(B.dcz) + B.rdz) * (1.0f + 0x1p-20f);
	B.kd = unif(kFilterK * dmax);
	B.roMax = unif(fmaxf(fmaxf(B.rox, B.roy), B.roz));
	B.kdRoSum = unif(dmax * (B.rox + B.roy + B.roz) * (1.0f + 0x1p-20f));
	B.roRd = unif((B.rox + B.roy + B.roz) * fmaxf(fmaxf(B.rdx, B.rdy), B.rdz) * (1.0f + 0x1p-20f));
	// NaN / inf / huge coordinates fail these compares: the filter then rejects nothing and the lanes run the exact path
	const bool tame = fabsf(o.x) < 0x1p40f && fabsf(o.y) < 0x1p40f && fabsf(o.z) < 0x1p40f && fabsf(d.x) < 0x1p20f && fabsf(d.y) < 0x1p20f && fabsf(d.z) < 0x1p20f;
	B.sane = ballot(active && !tame) == 0;
	return B;
}

// Bundle filter for ONE leaf reference (v0, e1, e2) -- this lane's -- against the whole bundle: true when
// Triangle::rayTriangleIntersect (objects.cpp:59-95) is CERTAIN to return false, or to return a t that is not below the
// ray's limit, for EVERY ray (o, d) with o in [oc - ro, oc + ro], d in [dc - rd, dc + rd].  Division-free: with
//     m = e2 x e1,  a = o - v0:    det = d . m,   Nu = tvec . pvec = d . (e2 x a),   Nv = d . qvec = d . (a x e1),
//     Nt = e2 . qvec = -(a . m)    (u = Nu / det, v = Nv / det, t = Nt / det)
// the four numerators are (bi)linear in (o, d), so their range over the bundle is centre +- radius.  Error budget, in units
// of the natural scales (s1 = |e1|_1, s2 = |e2|_1, ainf >= max_i |o_i - v0_i| over the bundle, dmax >= max |d_i|):
//     reference:  |det_c - det| <= 5.1 u dmax s1 s2,  |Nu_c - Nu| <= 12.2 u dmax ainf s2,  |Nv_c - Nv| <= 12.2 u dmax ainf s1,
//                 |Nt_c - Nt| <= 6.1 u ainf s1 s2      (u = 2^-24; two roundings per cross-product component, one per tvec
//                 component, three per dot product; fp32, no FMA, objects.cpp:70-90)
//     this filter: the same expressions evaluated once at the bundle centre, with FMAs: at most as much again.
// K = 64 u covers both with a margin of more than two; kFilterEta covers underflow.  With det_c >= 1e-8 > 0 (culling on, or
// the sign of det certain) inv = RN(1 / det_c) > 0 and every product below is rounded at most three times (relative
// 4 u), hence the factor (1 + 2^-18) on the right-hand sides:
//     det + Ed < 0                          =>  det_c < 1e-8                 (objects.cpp:75-77, culling on)
//     Nu + Eu < 0  /  Nv + Ev < 0           =>  u_c < 0  /  v_c < 0          (objects.cpp:84, 88; |Nu_c| > eta: no underflow to -0)
//     Nu - Eu > (det + Ed)(1 + 2^-18)       =>  u_c > 1                      (objects.cpp:84)
//     Nu + Nv - Eu - Ev > (det + Ed)(1+2^-18)  =>  u_c + v_c > 1             (objects.cpp:88; u_c, v_c >= 0 at that point)
//     Nt + Et < 0                           =>  t_c < 0                      (objects.cpp:91)
//     Nt - Et >= tmax (det + Ed)(1 + 2^-18) =>  t_c >= tmax >= the lane's limit  (objects.cpp:623 / scene.cpp:740: strict <)
// Magnitudes outside the assumed range (or NaN) make the compares false: nothing is rejected.
// Evaluated in two stages so that a wave whose 64 references all fail the cheap first stage (orientation and t range:
// the back of the mesh, everything behind a shadow ray's origin) skips the second (u, v).
// RTX_FILTER_MASKS: the verdicts as wave masks -- every compare feeds a ballot directly (a v_cmp into an SGPR pair) and the masks are combined on the scalar side.
// As bools, `ballot(valid && !rej1)` of values that went through the `||` chains' control flow compiled to v_cndmask + v_cmp per ballot.
#if RTX_FILTER_MASKS
typedef uint64_t FilterVerdict;
#else
typedef bool FilterVerdict;
#endif
struct FilterState { float kda, detHi1, sg; FilterVerdict ok; };      // (kept small: it is live across the ballot between the stages)

template <bool CULL>
__device__ __forceinline__ FilterVerdict bundleRejects1(const Bundle& B, float tmaxB, const RefA& ra, const RefB& rb, const RefC& rc, FilterState& f)
{
	const float e1x = rb.e1x, e1y = rb.e1y, e1z = rb.e1z, e2x = rb.e2x, e2y = rc.e2y, e2z = rc.e2z;
	const float ax = B.ocx - ra.v0x, ay = B.ocy - ra.v0y, az = B.ocz - ra.v0z;
	// m = e2 x e1
	const float mx = __builtin_fmaf(e2y, e1z, -(e2z * e1y)), my = __builtin_fmaf(e2z, e1x, -(e2x * e1z)), mz = __builtin_fmaf(e2x, e1y, -(e2y * e1x));
	const float s1 = fabsf(e1x) + fabsf(e1y) + fabsf(e1z), s2 = fabsf(e2x) + fabsf(e2y) + fabsf(e2z);
	const float s12 = s1 * s2;
	const float ainf = (fmaxf(fmaxf(fabsf(ax), fabsf(ay)), fabsf(az)) + B.roMax) * (1.0f + 0x1p-20f);
#if !RTX_FILTER_MASKS
	const bool tame = s1 < 0x1p20f && s2 < 0x1p20f && ainf < 0x1p41f;
#endif
	// det over the bundle
	float detc = __builtin_fmaf(B.dcz, mz, __builtin_fmaf(B.dcy, my, B.dcx * mx));
	const float detr = __builtin_fmaf(B.rdz, fabsf(mz), __builtin_fmaf(B.rdy, fabsf(my), B.rdx * fabsf(mx)));
	const float Ed = __builtin_fmaf(B.kd, s12, kFilterEta);
	float sg = 1.0f;
#if RTX_FILTER_MASKS
	uint64_t usable = ~0ull;
#else
	bool usable = true;
#endif
	if (!CULL) {
		// culling off: the tests need the sign of det_c; both signs are handled by mirroring, an uncertain sign by not testing
		const bool neg = detc + detr + Ed < 0;
#if RTX_FILTER_MASKS
		usable = ballot(neg) | ballot(detc - detr - Ed > 0);
#else
		usable = neg || detc - detr - Ed > 0;
#endif
		sg = neg ? -1.0f : 1.0f;
		detc *= sg;
	}
	const float detHi = detc + detr + Ed;
#if RTX_FILTER_MASKS
	f.ok = B.sane ? usable & ballot(s1 < 0x1p20f) & ballot(s2 < 0x1p20f) & ballot(ainf < 0x1p41f) : 0ull;
	const uint64_t rejO = CULL ? ballot(detHi < 0) & f.ok : 0ull;
	// (an exit here for passes whose open references all face away -- before the t range -- measured neutral: profiles/r06_ab_control_flow.txt)
#endif
	// t: Nt = -(a . m), radius from the origin box
	const float ntc = -sg * __builtin_fmaf(az, mz, __builtin_fmaf(ay, my, ax * mx));
	const float ntr = __builtin_fmaf(B.roz, fabsf(mz), __builtin_fmaf(B.roy, fabsf(my), B.rox * fabsf(mx)));
	const float Et = __builtin_fmaf(kFilterK * ainf, s12, kFilterEta);
	const float detHi1 = detHi * (1.0f + 0x1p-18f);
	f.kda = B.kd * ainf; f.detHi1 = detHi1; f.sg = sg;
#if RTX_FILTER_MASKS
	return rejO | ((ballot(ntc + ntr < -Et) | ballot(ntc - ntr - Et >= tmaxB * detHi1)) & f.ok);
#else
	const bool rej = (CULL && detHi < 0) || ntc + ntr < -Et || ntc - ntr - Et >= tmaxB * detHi1;
	f.ok = usable && tame && B.sane;
	return rej && f.ok;
#endif
}

template <bool CULL>
__device__ __forceinline__ FilterVerdict bundleRejects2(const Bundle& B, const RefA& ra, const RefB& rb, const RefC& rc, const FilterState& f, FilterVerdict open = FilterVerdict(0))      // open: the lanes still undecided (mask form)
{
	const float e1x = rb.e1x, e1y = rb.e1y, e1z = rb.e1z, e2x = rb.e2x, e2y = rc.e2y, e2z = rc.e2z;
	const float ax = B.ocx - ra.v0x, ay = B.ocy - ra.v0y, az = B.ocz - ra.v0z;      // (recomputed: cheaper than three live registers)
	const float s1 = fabsf(e1x) + fabsf(e1y) + fabsf(e1z), s2 = fabsf(e2x) + fabsf(e2y) + fabsf(e2z);
	// u: Nu = d . (e2 x a)
	const float wux = __builtin_fmaf(e2y, az, -(e2z * ay)), wuy = __builtin_fmaf(e2z, ax, -(e2x * az)), wuz = __builtin_fmaf(e2x, ay, -(e2y * ax));
	float nuc = __builtin_fmaf(B.dcz, wuz, __builtin_fmaf(B.dcy, wuy, B.dcx * wux));
	// The origin box's part of the radius.  Nu = (a + do) . ((dc + dd) x e2) = Nu(centre) + dd . wu + do . (dc x e2) + do . (dd x e2): the third
	// term is bounded by sum ro_k |(dc x e2)_k| -- not by dmax |e2|_1 sum ro_k, which is what a box of origins costs a bundle of
	// SHADOW rays (a tilted patch of surface points; primary rays share their origin: nothing changes for them): 41 % fewer
	// survivors for shadow bundles in tools/research/bundle_filter_sim.py, exact tests per headline launch 9.45 M -> 5.02 M -- and
	// the last by (sum ro_k) max rd |e2|_1.
	const float cux = __builtin_fmaf(B.dcy, e2z, -(B.dcz * e2y)), cuy = __builtin_fmaf(B.dcz, e2x, -(B.dcx * e2z)), cuz = __builtin_fmaf(B.dcx, e2y, -(B.dcy * e2x));
	const float nurO = __builtin_fmaf(B.roRd, s2, __builtin_fmaf(B.roz, fabsf(cuz), __builtin_fmaf(B.roy, fabsf(cuy), B.rox * fabsf(cux))));
	const float nur = (nurO + __builtin_fmaf(B.rdz, fabsf(wuz), __builtin_fmaf(B.rdy, fabsf(wuy), B.rdx * fabsf(wux)))) * (1.0f + 0x1p-19f);
	const float Eu = __builtin_fmaf(f.kda, s2, kFilterEta);
	if (!CULL) nuc *= f.sg;
	const float nuLo = nuc - nur - Eu;
#if RTX_FILTER_MASKS
	const uint64_t rejU = (ballot(nuc + nur < -Eu) | ballot(nuLo > f.detHi1)) & f.ok;
	if ((open & ~rejU) == 0) return rejU;      // (u decides every open lane: the wave skips v)
#endif
	// v: Nv = d . (a x e1)
	const float wvx = __builtin_fmaf(ay, e1z, -(az * e1y)), wvy = __builtin_fmaf(az, e1x, -(ax * e1z)), wvz = __builtin_fmaf(ax, e1y, -(ay * e1x));
	float nvc = __builtin_fmaf(B.dcz, wvz, __builtin_fmaf(B.dcy, wvy, B.dcx * wvx));
	// (Nv = (dc + dd) . ((a + do) x e1): do . (e1 x dc))
	const float cvx = __builtin_fmaf(e1y, B.dcz, -(e1z * B.dcy)), cvy = __builtin_fmaf(e1z, B.dcx, -(e1x * B.dcz)), cvz = __builtin_fmaf(e1x, B.dcy, -(e1y * B.dcx));
	const float nvrO = __builtin_fmaf(B.roRd, s1, __builtin_fmaf(B.roz, fabsf(cvz), __builtin_fmaf(B.roy, fabsf(cvy), B.rox * fabsf(cvx))));
	const float nvr = (nvrO + __builtin_fmaf(B.rdz, fabsf(wvz), __builtin_fmaf(B.rdy, fabsf(wvy), B.rdx * fabsf(wvx)))) * (1.0f + 0x1p-19f);
	const float Ev = __builtin_fmaf(f.kda, s1, kFilterEta);
	if (!CULL) nvc *= f.sg;
#if RTX_FILTER_MASKS
	return rejU | ((ballot(nvc + nvr < -Ev) | ballot(nuLo + (nvc - nvr - Ev) > f.detHi1)) & f.ok);
#else
	const bool rej = nuc + nur < -Eu || nuLo > f.detHi1 || nvc + nvr < -Ev || nuLo + (nvc - nvr - Ev) > f.detHi1;
	return rej && f.ok;
#endif
}

// The reference's test (objects.cpp:59-95) of the rays in exec against ONE triangle whose record is wave-uniform (read
// from the surviving lane through the LDS crossbar, ds_bpermute): the reference's own arithmetic, operation by operation.
template <bool CULL, bool STATS>
__device__ __forceinline__ void triTestOne(float v0x, float v0y, float v0z, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                                           uint32_t tri, const V3& o, const V3& d, float& bt, float& bu, float& bv, uint32_t& btri)
{
	// pvec = dir x v0v2 (objects.cpp:72), det = v0v1 . pvec (objects.cpp:73)
	const float px = d.y * e2z - d.z * e2y, py = d.z * e2x - d.x * e2z, pz = d.x * e2y - d.y * e2x;
	const float det = e1x * px + e1y * py + e1z * pz;
	// culling on: reject iff det < 1e-8 (then |det| < 1e-8 is implied); off: reject iff |det| < 1e-8.  Both compares
	// are false for NaN, exactly like the reference's two ifs (objects.cpp:75-79).
	if ((CULL ? det : fabsf(det)) < RTX_EPS8) return;
	const float tx = o.x - v0x, ty = o.y - v0y, tz = o.z - v0z;                    // tvec = orig - v0 (objects.cpp:82)
	const float nu = tx * px + ty * py + tz * pz;
	if (CULL) {
		// Exact-safe rejection before the IEEE division.  Here 1e-8 <= det, so inv = RN(1/det) > 0 with relative
		// error <= 2^-22 (2^-24 while 1/det is normal, <= 2^-22 in the denormal range det > 2^126), and u = RN(nu*inv):
		//   nu < -2^-20            =>  |nu*inv| >= 2^-20 * 2^-128 * (1 - 2^-3) > 2^-149: no underflow to -0  =>  u < 0;
		//   nu > RN(det*(1+2^-20)) =>  nu/det > 1+2^-21, and the roundings lose < 2^-21                    =>  u > 1.
		// Lanes outside these sure cases take the division, which re-derives the same verdict for the sure cases, so the
		// result is bit-identical either way; the wave skips the division when no lane needs it.
		if (nu < -0x1p-20f || nu > det * (1.0f + 0x1p-20f)) return;
	}
	const float inv = 1 / det;
	const float u = nu * inv;
	if (u < 0 || u > 1) return;
	const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;   // tvec x v0v1
	const float v = (d.x * qx + d.y * qy + d.z * qz) * inv;
	if (v < 0 || u + v > 1) return;
	const float t = (e2x * qx + e2y * qy + e2z * qz) * inv;
	if (t < 0) return;
	if (t < bt) { bt = t; bu = u; bv = v; btri = tri; }                             // objects.cpp:623: strict <, first wins
}

// The same test the other way round: the lane's OWN triangle against ONE ray whose origin / direction are wave-uniform.
// Operation for operation the arithmetic of triTestOne (objects.cpp:59-95); t stays as passed in where the reference
// returns without a hit.
template <bool CULL>
__device__ __forceinline__ void triTestLane(const RefA& ra, const RefB& rb, const RefC& rc, float ox, float oy, float oz, float dx, float dy, float dz,
                                            float& t, float& u, float& v)
{
	const float e1x = rb.e1x, e1y = rb.e1y, e1z = rb.e1z, e2x = rb.e2x, e2y = rc.e2y, e2z = rc.e2z;
	const float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
	const float det = e1x * px + e1y * py + e1z * pz;
	if ((CULL ? det : fabsf(det)) < RTX_EPS8) return;
	const float tx = ox - ra.v0x, ty = oy - ra.v0y, tz = oz - ra.v0z;
	const float nu = tx * px + ty * py + tz * pz;
	if (CULL) { if (nu < -0x1p-20f || nu > det * (1.0f + 0x1p-20f)) return; }      // (exact-safe: see triTestOne)
	const float inv = 1 / det;
	const float uu = nu * inv;
	if (uu < 0 || uu > 1) return;
	const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
	const float vv = (dx * qx + dy * qy + dz * qz) * inv;
	if (vv < 0 || uu + vv > 1) return;
	const float tt = (e2x * qx + e2y * qy + e2z * qz) * inv;
	if (tt < 0) return;
	t = tt; u = uu; v = vv;
}

// AccelerationStructure::intersectAccelStruct (objects.cpp:587-631) for a whole wave: stackless pre-order walk of the
// nodes (phase 1: scalar-fed box tests, reached leaves are noted), then the noted leaves are processed in the same
// order (phase 2: bundle filter with the lanes acting as triangles, exact test of the survivors with the lanes acting as
// rays).  The two phases alternate every RTX_LEAF_BATCH leaves so that any-hit shadow rays still stop early.
// A hit exists iff bt < FLT_MAX on return (the first accepted t is < FLT_MAX by objects.cpp:598,623).
// Leaves noted before their references are processed (meshWalk): ONE where a launch is bound by throughput (pass 1: a hit found in a
// leaf tightens the bundle's limit / ends a shadow ray before the next node is visited -- headline pass 1 3.275 -> 3.176 ms, cfg5 11.40 ->
// 11.01 against the 4 of round 3), TWO where it lasts as long as its slowest wave's chain of fetches (SSAA items, the frame kernel: FEWRAYS;
// cfg2 at 1080p 1.306 (4) / 1.292 (2) / 1.353 (1) ms).  profiles/r04_ab_leaf_batch.txt
#define RTX_LEAF_BATCH_MAX (RTX_LEAF_BATCH > RTX_LEAF_BATCH_FEW ? RTX_LEAF_BATCH : RTX_LEAF_BATCH_FEW)
// one reached leaf: first reference, number of references, the lanes (rays) that passed its box, start in the batch's stream
struct LeafEntry { uint32_t start, count, first, pad0, maskLo, maskHi, pad1, pad2; };      // (what the assignment of a pass needs arrives with one 16-byte read)
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__shared__ LeafEntry leafBatch[4][RTX_LEAF_BATCH_MAX];      // per wave of a 256-thread block; private to the wave (no barrier)
// WIDE walk: pending subtrees / leaves of the wave, top of the stack = next in the reference's order.  link > 0: wide node
// link - 1; link < 0: leaf with ~link references from `first`; mask = the rays that passed the item's own box.
struct WideItem { int32_t link; uint32_t first, maskLo, maskHi; };
__shared__ WideItem wideStack[4][kWideStackEntries];      // (kWideSlots - 1 entries per wide level + 4: rtx_scene_create checks the depth)
// WIDE walk with prune records (rtxd::PruneRec): what the slot test needs from the wave's bundle, per wave, written once per
// walk: [0..5] the range of 1 / dir over the rays per axis (lo, hi; mirrored so that it is positive), [6..11] the range of
// the origins per axis in the same mirrored coordinates (lo, hi), [12] 216 dmax, [13] the largest |origin| coordinate,
// [14] bits 0-2: axis mirrored, bits 3-5: axis usable (1 / dir of one sign over the wave).
#if RTX_PRUNE_AXIS
// RTX_PRUNE_AXIS: the same quantities laid out for pruneEval8 -- six records of eight floats per wave, one per (test, axis):
// [0..2] box test, axis x / y / z: { 1 / dir lo, hi (mirrored: positive), origin lo, hi (mirrored), the sign bit that mirrors a record's centre, +inf (axis usable) / -inf,
//                                    216 dmax, the largest |origin| coordinate };
// [3..5] plane test, axis x / y / z: { dc, rd, oc, ro (the bundle's direction / origin boxes as centre, radius), 2 K dmax, K (|orig| + |vertex|), -, - }
__shared__ float pruneUni[4][48];
#else
__shared__ float pruneUni[4][24];      // ([16..18] the sign bits that mirror a record's centre, [19..21] per axis +inf (usable) / -inf (not))
#endif
// the prune records of the root's slots are evaluated too: +-0 with four slots (two levels down: as good as never pruned), -1.6 % with eight
constexpr bool kPruneRoot = kWideLevels >= 3;
// 36 u / 1e-8 (u = 2^-24) = 214.6: see pruneSlots
constexpr float kPruneC = 216.0f;
// the reference's box test in its min / max form (exact when no NaN can arise, see meshWalk) against a box in SGPRs
__device__ __forceinline__ bool boxFailsRegular(float blox, float bhix, float bloy, float bhiy, float bloz, float bhiz, const V3& o, float ix, float iy, float iz)
{
	const float xlo = (blox - o.x) * ix, xhi = (bhix - o.x) * ix, ylo = (bloy - o.y) * iy, yhi = (bhiy - o.y) * iy;
	const float zlo = (bloz - o.z) * iz, zhi = (bhiz - o.z) * iz;
	float nx, ny, nz, fx, fy, fz, tn, tf;
	asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(xlo), "v"(xhi)); asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(xlo), "v"(xhi));
	asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ylo), "v"(yhi)); asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ylo), "v"(yhi));
	asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(zlo), "v"(zhi)); asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(zlo), "v"(zhi));
	asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(nz));
	asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(fz));
	return tn > tf;
}
// Can the subtree of a wide-node slot contribute to this walk?  r0 / r1 = the slot's rtxd::PruneRec (true box T of its
// triangles as centre c / half-extent h, P = max |e1|_1 |e2|_1), pu = the wave's pruneUni block, tmaxB = the largest limit
// of any open ray.  Returns false only when NO ray of the bundle can have a hit accepted by the reference
// (objects.cpp:59-95) in that subtree with t below its limit:
//   An accepted hit has det_c >= 1e-8, 0 <= u_c <= 1, 0 <= v_c, u_c + v_c <= 1 (+ one rounding), 0 <= t_c < limit.  With the
//   identity  det (orig - v0) = -Nt dir + Nu e1 + Nv e2  (Cramer; exact for the fp32 inputs) and the reference's rounding
//   errors (|det_c - det| <= 5.1 u dmax s1 s2, |Nu_c - Nu| <= 12.2 u dmax ainf s2, |Nv_c - Nv| <= 12.2 u dmax ainf s1,
//   |Nt_c - Nt| <= 6.1 u ainf s1 s2; s1 = |e1|_1, s2 = |e2|_1, ainf = |orig - v0|_inf, u = 2^-24, DESIGN_HISTORY.md 3.3):
//       orig + t' dir = v0 + u' e1 + v' e2 + R / det_c,   |R|_inf <= 35.6 u dmax ainf s1 s2,
//   where t', u', v' are the computed numerators over det_c -- within 2 roundings of t_c, u_c, v_c, so the right-hand point
//   is in the triangle up to 3 u (s1 + s2).  Hence orig + t' dir lies in T inflated by rho = 36 u dmax ainf P / 1e-8
//   (det_c >= 1e-8: the bound needs no assumption on the conditioning of the pair -- the reference does accept hits
//   tenths of a unit off an edge-on triangle; tools/research/rho_check.py), and t' in [0, limit (1 + 3 u)].
// Here: rho from ainf = the largest distance of any origin of the bundle from T's far side, plus 2^-17 of the coordinate
// scale for the roundings of this test itself; entry / exit of the inflated box over ALL origins and 1 / dir of the bundle
// (interval arithmetic, endpoints only: every product is monotone in each factor); alive unless the exit is certainly
// before the entry, behind the origin, or the entry certainly beyond tmaxB.  NaN compares false -> alive.
__device__ __forceinline__ bool pruneAlive(const f4v& r0, const f4v& r1, const float* pu, float tmaxB)
{
	const f4v ua = *(const f4v*)(pu + 0), ub = *(const f4v*)(pu + 4), uc = *(const f4v*)(pu + 8), ue = *(const f4v*)(pu + 12);
	// The walk's mirror and usability flags as operands instead of tests: the centre is mirrored by an XOR with the axis' sign bit, and an axis that is
	// not usable (1 / dir changes sign over the bundle) drops out through one more operand of the min / max: cap = -inf gives entry -inf, exit +inf --
	// whatever the products are, NaN included (v_min / v_max return the other operand); on a usable axis cap = +inf changes nothing (the products of
	// finite records with finite reciprocals are never NaN).  18 VALU instructions fewer per evaluation than the flag tests.
	const f4v uf = *(const f4v*)(pu + 16), ug = *(const f4v*)(pu + 20);
	const float cx = __uint_as_float(__float_as_uint(r0.x) ^ __float_as_uint(uf.x)), cy = __uint_as_float(__float_as_uint(r0.y) ^ __float_as_uint(uf.y));
	const float cz = __uint_as_float(__float_as_uint(r0.z) ^ __float_as_uint(uf.z));
	const float capX = uf.w, capY = ug.x, capZ = ug.y;
	// largest |orig - vertex| coordinate over the bundle and the box
	const float ainf = fmaxf(fmaxf(fmaxf(cx - ub.z, ub.w - cx) + r1.x, fmaxf(cy - uc.x, uc.y - cy) + r1.y), fmaxf(cz - uc.z, uc.w - cz) + r1.z);
	// P of the walk's source copy (rtxd::PruneRec: the camera's, a point light's) holds for origins within kSrcAinfMax of the box;
	// beyond, and in copy 0 (r0.w == r1.w), the unconditional Pgen
	const float Pn = ainf <= kSrcAinfMax ? r0.w : r1.w;
	const float rho = __builtin_fmaf(ue.x * ainf, Pn, 0x1p-17f * (ainf + ue.y)) * (1.0f + 0x1p-20f) + 1e-30f;
	const float hx = r1.x + rho, hy = r1.y + rho, hz = r1.z + rho;
	const float inf = __builtin_inff();
	// per axis (mirrored so that 1 / dir > 0): entry >= (lo' - o_hi) inv, exit <= (hi' - o_lo) inv over the ranges
	const float ax = (cx - hx) - ub.w, bx = (cx + hx) - ub.z;
	const float ay = (cy - hy) - uc.y, by = (cy + hy) - uc.x;
	const float az = (cz - hz) - uc.w, bz = (cz + hz) - uc.z;
	(void)inf;
	const float ex = fminf(fminf(ax * ua.x, ax * ua.y), capX), fx = fmaxf(fmaxf(bx * ua.x, bx * ua.y), -capX);
	const float ey = fminf(fminf(ay * ua.z, ay * ua.w), capY), fy = fmaxf(fmaxf(by * ua.z, by * ua.w), -capY);
	const float ez = fminf(fminf(az * ub.x, az * ub.y), capZ), fz = fmaxf(fmaxf(bz * ub.x, bz * ub.y), -capZ);
	const float ent = fmaxf(fmaxf(ex, ey), ez), ext = fminf(fminf(fx, fy), fz);
	return !(ent > ext || ext < 0.0f || ent > tmaxB * (1.0f + 0x1p-18f));
}


// The first stage of the bundle filter (bundleRejects1) for ALL triangles below a wide-node slot at once.  r0 / r1 = the slot's
// rtxd::PlaneRec: every triangle's scaled plane normal q = (e2 x e1) / (s1 s2) lies in the box qc +- qr and its plane offset
// v0 . q in [wlo, whi].  Dividing the filter's inequalities by s1 s2 > 0:
//     det / (s1 s2) = dir . q,   Nt / (s1 s2) = v0 . q - orig . q,   Ed / (s1 s2) = K dmax,   Et / (s1 s2) = K ainf,
// so with the range of dir . q and orig . q over the bundle's boxes x the normal box (exact for boxes: centre product +-
// |centre| radius + radius (|centre| + radius) per axis) the three rejections hold for every triangle when they hold for
// the ends of the ranges.  eB = K (|orig|_max + |vertex|_max) >= K ainf, and the 64 u it stands for cover the reference's
// 6.1 u ainf and the ~30 u (|orig| + |vertex|) this evaluation can be off by (w and orig . q are both of coordinate size);
// kd is doubled for the same reason (5.1 u dmax + ~24 u dmax of evaluation error against 128 u dmax).
//     (a) max dir . q + 2 kd < 0                                  => det_c < 1e-8 for every triangle and ray (objects.cpp:75-77)
//     (b) whi - min orig . q + eB < 0                             => t_c < 0                                  (objects.cpp:91)
//     (c) wlo - max orig . q - eB >= tmaxB (max dir . q + 2 kd) (1 + 2^-18)  => t_c >= every ray's limit      (scene.cpp:740)
// NaN compares false -> alive; qr < 0 (no bound for this slot) -> alive.
__device__ __forceinline__ bool planeAlive(const f4v& r0, const f4v& r1, const Bundle& B, float eB, float tmaxB)
{
	const float ax = fabsf(r0.x) + r1.x, ay = fabsf(r0.y) + r1.y, az = fabsf(r0.z) + r1.z;      // |qc| + qr
	const float dqHi = __builtin_fmaf(B.rdz, az, __builtin_fmaf(fabsf(B.dcz), r1.z, __builtin_fmaf(B.dcz, r0.z,
	                   __builtin_fmaf(B.rdy, ay, __builtin_fmaf(fabsf(B.dcy), r1.y, __builtin_fmaf(B.dcy, r0.y,
	                   __builtin_fmaf(B.rdx, ax, __builtin_fmaf(fabsf(B.dcx), r1.x, B.dcx * r0.x))))))));
	const float oqC = __builtin_fmaf(B.ocz, r0.z, __builtin_fmaf(B.ocy, r0.y, B.ocx * r0.x));
	const float oqR = __builtin_fmaf(B.roz, az, __builtin_fmaf(fabsf(B.ocz), r1.z, __builtin_fmaf(B.roy, ay, __builtin_fmaf(fabsf(B.ocy), r1.y,
	                  __builtin_fmaf(B.rox, ax, fabsf(B.ocx) * r1.x)))));
	const float detHi = dqHi + 2.0f * B.kd;
	const float ntHi = (r1.w - (oqC - oqR)) + eB, ntLo = (r0.w - (oqC + oqR)) - eB;
	const bool dead = detHi < 0.0f || ntHi < 0.0f || (detHi > 0.0f && ntLo >= tmaxB * (detHi * (1.0f + 0x1p-18f)));
	return !(dead && r1.x >= 0.0f);
}

#if RTX_PRUNE_AXIS
// pruneAlive and planeAlive of the eight slots of one wide node with the lanes acting as (record, AXIS) pairs instead of records: lane 4 r + a looks at axis a
// of record r (r < 8: slot r's PruneRec, r >= 8: slot r - 8's PlaneRec; a = 3: the record's fourth words -- P / Pgen, wlo / whi -- which it hands to its quad), so
// that the per-axis arithmetic of both tests is one instruction for all three axes and what joins the axes (the largest |orig - vertex|, entry / exit, the three
// dot products) is two DPP steps inside the quad: rotations among lanes 0-2 (lane 3 is never read).  ~70 VALU instructions per node visit against ~110 with one
// record per lane on 16 lanes ("the prune / plane record evaluation runs on 16 of 64 lanes").  Same inequalities, same margins as pruneAlive /
// planeAlive above -- only the order in which a dot product's three terms are added differs per lane, which the margins (kd doubled, eB: see planeAlive) cover
// whatever the order; lane 4 r's verdict is the record's.  Returns bit 4 k set when slot k may contribute.
// pu = the wave's six axis records (pruneUni); blk = the node's PruneBlock (wave-uniform).
// CULL = false (options::useBackfaceCulling off, objects.cpp:75-79: a pair is rejected iff |det_c| < 1e-8): the plane test cannot drop back faces, and
// which sign of Nt means "behind the origin" depends on the sign of det.  With min and max of dir . q over the bundle and the slot:
//     min dir . q - 2 kd > 0  (every pair faces the rays)    -> (b), (c) as with culling;
//     max dir . q + 2 kd < 0  (every pair shows its back: t = Nt / det = (-Nt) / |det|)  -> dead iff  wlo - max orig . q - eB > 0   (Nt > 0: t_c < 0)
//                                                               or  -(whi - min orig . q + eB) >= tmaxB (-(min dir . q - 2 kd)) (1 + 2^-18)   (t_c >= every limit);
//     sign uncertain -> alive.
// The box test rests on |det_c| >= 1e-8 and magnitudes only (pruneAlive; the source certificates likewise: rtx_source.hip works with |a . m|, and its
// error terms are symmetric in e1, e2 -- a back face is the front face (e2, e1) with u and v exchanged), so it is the same with and without culling.
#if RTX_PRUNE_LANEK
// What a lane of pruneEval8 reads, as ONE register kept across the walk: bits 0-15 the byte offset of its two words in a node's PruneBlock, bits 16-31 the LDS
// address of its axis record.  Deriving both from the lane index on every visit took 12 VALU instructions; kept as two or three separate registers they spilled.
__device__ __forceinline__ uint32_t pruneLaneK(const float* pu)
{
	const uint32_t lane = laneNow(), axis = lane & 3u;
	const uint32_t word = 2u * lane - axis;                   // record lane >> 2 (eight words each), word lane & 3
	uint32_t ui = (lane >> 5) * 3u + axis;
	ui = ui < 5u ? ui : 5u;                      // (the fourth lane of a plane quad reads a record that exists; what it computes is never used)
	const uint32_t lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)(pu + ui * 8u);
	return lds << 16 | word * 4u;
}
#endif
template <bool BOXES, bool CULL = true>
__device__ __forceinline__ uint32_t pruneEval8(const RTX_AS1 char* blk, const float* pu, float tmaxB, uint32_t laneK = 0)
{
	static_assert(kWideSlots == 8, "pruneEval8: lanes = 16 records x 4");
#if RTX_PRUNE_LANEK
	const uint32_t off = laneK & 0xffffu;
	const float f0 = *(const RTX_AS1 float*)(blk + off), f1 = *(const RTX_AS1 float*)(blk + off + 16u);      // (uniform base + a 32-bit lane offset)
	const __attribute__((address_space(3))) f4v* ur = (const __attribute__((address_space(3))) f4v*)(uintptr_t)(laneK >> 16);
	const f4v ua = ur[0], ub = ur[1];
#else
	const uint32_t lane = laneNow();      // (recomputed here: the addresses derived from it are otherwise hoisted out of the node loop and live -- spilled -- across the walk)
	const uint32_t axis = lane & 3u;
	const RTX_AS1 float* pf = (const RTX_AS1 float*)blk;      // (uniform base + a 32-bit lane offset: one scalar-base load, no 64-bit address arithmetic per lane)
	const uint32_t word = 2u * lane - axis;                   // record lane >> 2 (eight words each), word lane & 3
	const float f0 = pf[word], f1 = pf[word + 4u];            // box: c_a, h_a (lane 3: P, Pgen); plane: qc_a, qr_a (lane 3: wlo, whi)
	uint32_t ui = (lane >> 5) * 3u + axis;
	ui = ui < 5u ? ui : 5u;                      // (the fourth lane of a plane quad reads a record that exists; what it computes is never used)
	const f4v ua = *(const f4v*)(pu + ui * 8u), ub = *(const f4v*)(pu + ui * 8u + 4u);
#endif
	// ---- per axis, both tests (each meaningful on its own half of the wave)
	// box (pruneAlive): centre mirrored, the largest |orig - vertex| along this axis
	const float c = __uint_as_float(__float_as_uint(f0) ^ __float_as_uint(ub.x));
	const float ai = fmaxf(c - ua.z, ua.w - c) + f1;
	// plane (planeAlive): this axis' terms of  max dir . q,  orig . q -+ its radius
	const float ax = fabsf(f0) + f1;
	const float dq = __builtin_fmaf(ua.y, ax, __builtin_fmaf(fabsf(ua.x), f1, ua.x * f0));
	const float oq = ua.z * f0;
	const float orr = __builtin_fmaf(ua.w, ax, fabsf(ua.z) * f1);
	const float pm = oq - orr, pp = oq + orr;
	float ainf, dqHi, oqLo, oqHi, w0, w1, t0, t1, t2, t3;
	// (a DPP source must not have been written by the two preceding VALU instructions: s_nop 1; inside the block the DPP sources are the block's inputs)
	asm volatile("s_nop 1\n\t"
	             "v_max_f32_dpp %6, %10, %10 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_add_f32_dpp %7, %11, %11 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_add_f32_dpp %8, %12, %12 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_add_f32_dpp %9, %13, %13 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_mov_b32_dpp %4, %14 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_mov_b32_dpp %5, %15 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_max_f32_dpp %0, %10, %6 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_add_f32_dpp %1, %11, %7 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_add_f32_dpp %2, %12, %8 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n\t"
	             "v_add_f32_dpp %3, %13, %9 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf"
	             : "=&v"(ainf), "=&v"(dqHi), "=&v"(oqLo), "=&v"(oqHi), "=&v"(w0), "=&v"(w1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
	             : "v"(ai), "v"(dq), "v"(pm), "v"(pp), "v"(f0), "v"(f1));
	// ---- plane: (a) every ray sees the back, (b) starts beyond, (c) ends before every plane of the slot (planeAlive)
	const float detHi = dqHi + ub.x;                                    // + 2 K dmax
	const float ntHi = (w1 - oqLo) + ub.y, ntLo = (w0 - oqHi) - ub.y;   // whi - min orig . q + eB,  wlo - max orig . q - eB
	// (a ballot per compare, joined on the scalar side: straight-line code, and a ballot of anything but a compare goes through a VGPR)
	uint64_t deadPlane;
	if (CULL) deadPlane = ballot(detHi < 0.0f) | ballot(ntHi < 0.0f) | (ballot(detHi > 0.0f) & ballot(ntLo >= tmaxB * (detHi * (1.0f + 0x1p-18f))));
	else {
		// min dir . q: the same three terms with the radii subtracted
		const float dql = __builtin_fmaf(-ua.y, ax, __builtin_fmaf(-fabsf(ua.x), f1, ua.x * f0));
		float dqLo;
		asm volatile("s_nop 1\n\t"
		             "v_add_f32_dpp %1, %2, %2 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
		             "s_nop 1\n\t"
		             "v_add_f32_dpp %0, %2, %1 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf"
		             : "=&v"(dqLo), "=&v"(t0) : "v"(dql));
		const float detLo = dqLo - ub.x;
		const uint64_t front = ballot(detLo > 0.0f), back = ballot(detHi < 0.0f);
		deadPlane = (front & (ballot(ntHi < 0.0f) | ballot(ntLo >= tmaxB * (detHi * (1.0f + 0x1p-18f))))) |
		            (back & (ballot(ntLo > 0.0f) | ballot(-ntHi >= tmaxB * (-detLo * (1.0f + 0x1p-18f)))));
	}
	uint32_t deadBits = (uint32_t)(deadPlane >> 32);      // slot k: lane 32 + 4 k
	if (BOXES) {
		// ---- box: the slot's true box inflated by rho against the bundle's segment [0, tmaxB] (pruneAlive)
		const float Pn = ainf <= kSrcAinfMax ? w0 : w1;
		const float rho = __builtin_fmaf(ub.z * ainf, Pn, 0x1p-17f * (ainf + ub.w)) * (1.0f + 0x1p-20f) + 1e-30f;
		const float hh = f1 + rho;
		const float a = (c - hh) - ua.w, b = (c + hh) - ua.z;
		const float e = fminf(fminf(a * ua.x, a * ua.y), ub.y), f = fmaxf(fmaxf(b * ua.x, b * ua.y), -ub.y);
		float ent, ext;
		asm volatile("s_nop 1\n\t"
		             "v_max_f32_dpp %2, %4, %4 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
		             "v_min_f32_dpp %3, %5, %5 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"
		             "s_nop 0\n\t"
		             "v_max_f32_dpp %0, %4, %2 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n\t"
		             "v_min_f32_dpp %1, %5, %3 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf"
		             : "=&v"(ent), "=&v"(ext), "=&v"(t0), "=&v"(t1) : "v"(e), "v"(f));
		const uint64_t deadBox = ballot(ent > ext) | ballot(ext < 0.0f) | ballot(ent > tmaxB * (1.0f + 0x1p-18f));
		deadBits |= (uint32_t)deadBox;                             // slot k: lane 4 k
	}
	return ~deadBits & 0x11111111u;
}
#endif

template <bool STATS, bool CULL, bool REGULAR, bool WIDE = false, bool FEWRAYS = false, bool BOXES = true>
__device__ __forceinline__ void meshWalk(const u32x16& mp, const Bundle& B, bool consider, bool shadow, const V3& o, const V3& d,
                                         float ix, float iy, float iz, bool sx, bool sy, bool sz, float tLimit,
                                         float& bt, float& bu, float& bv, uint32_t& btri, Counts& cnt, uint32_t srcSel = 0)
{
	// mp = the third line of the object's record (rtxd::Object): nodes, refA, refB, refC, wide, prune, nNodes, vmax
#define RTX_MP(k) (((uint64_t)mp[2 * (k) + 1] << 32) | mp[2 * (k)])
	const Node* nodes = (const Node*)RTX_MP(0);
	const RTX_AS1 char* refA = (const RTX_AS1 char*)(uintptr_t)RTX_MP(1);
	const RTX_AS1 char* refB = (const RTX_AS1 char*)(uintptr_t)RTX_MP(2);
	const RTX_AS1 char* refC = (const RTX_AS1 char*)(uintptr_t)RTX_MP(3);
	const uint32_t nN = mp[12];
	bt = kFltMax; bu = 0; bv = 0; btri = 0;
	if (nN == 0) return;
	constexpr uint32_t kLeafBatch = FEWRAYS ? RTX_LEAF_BATCH_FEW : RTX_LEAF_BATCH;
	// the largest limit of any ray of the wave: a triangle whose t is certainly not below it cannot be recorded by any
	// lane (tightened whenever a lane finds a closer hit)
	float tmaxB = unif(waveMax(consider ? tLimit : -__builtin_inff()));
	uint32_t resume = consider ? 0u : kNever;
	const uint32_t nRays4 = 4u * (uint32_t)__popcll(ballot(consider));      // (4 x the rays of this walk: see the exact tests)
	uint32_t i = 0;
	const uint32_t last = nN - 1;
	u32x8 nd = sload8(nodes);
	LeafEntry* entries = leafBatch[threadIdx.x >> 6];
	// WIDE: the tree with every other level skipped (rtxd::WideNode), walked with a wave-level stack in LDS; one dependent
	// fetch per two levels.  The items carry the rays that passed their own box, so nothing per lane has to remember where
	// it is in the tree; `resume` only marks the rays that are done.
	WideItem* stack = wideStack[threadIdx.x >> 6];
	const WideNode* wideNodes = WIDE ? (const WideNode*)RTX_MP(4) : nullptr;
	uint32_t sp = 0;
	const RTX_AS1 char* pruneRecs = nullptr;
	float* pu = pruneUni[threadIdx.x >> 6];
	if (WIDE) {
		pruneRecs = (const RTX_AS1 char*)(uintptr_t)RTX_MP(5);
#undef RTX_MP
		if (pruneRecs != nullptr) {
			// the copy of the prune blocks that belongs to the source all rays of this walk pass through (0: none -- traceWave)
			pruneRecs += (size_t)srcSel * mp[15] * sizeof(PruneBlock);
			// Range of 1 / dir over the rays of this walk, from the bundle's direction box (every ray's direction lies in dc +- rd):
			// 1 / x is monotone on either side of 0, so every lane's RN(1 / d) lies between the reciprocals of the box's ends --
			// v_rcp_f32 (1 ulp) widened by 2^-21 covers its own error, the rounding of the ends and the lane's own rounding.
			// An axis whose box touches 0 is not usable (flag), one with an end beyond 2^-100 neither (the reciprocal may be inf).
			const float dlx = B.dcx - B.rdx, dhx = B.dcx + B.rdx, dly = B.dcy - B.rdy, dhy = B.dcy + B.rdy, dlz = B.dcz - B.rdz, dhz = B.dcz + B.rdz;
			const float wide = 1.0f + 0x1p-21f, narrow = 1.0f - 0x1p-21f;
			// (1 / x falls on either side of 0: the smaller reciprocal belongs to the upper end of the box, whatever the sign)
			float lx = __builtin_amdgcn_rcpf(dhx), hx = __builtin_amdgcn_rcpf(dlx);
			float ly = __builtin_amdgcn_rcpf(dhy), hy = __builtin_amdgcn_rcpf(dly), lz = __builtin_amdgcn_rcpf(dhz), hz = __builtin_amdgcn_rcpf(dlz);
			lx *= lx > 0 ? narrow : wide; hx *= hx > 0 ? wide : narrow;
			ly *= ly > 0 ? narrow : wide; hy *= hy > 0 ? wide : narrow;
			lz *= lz > 0 ? narrow : wide; hz *= hz > 0 ? wide : narrow;
			const bool usx = (dlx > 0x1p-100f || dhx < -0x1p-100f), usy = (dly > 0x1p-100f || dhy < -0x1p-100f), usz = (dlz > 0x1p-100f || dhz < -0x1p-100f);
			if (laneNow() == 0) {
				const bool nx = hx < 0, ny = hy < 0, nz = hz < 0;
				const bool okx = usx && (lx > 0 || nx), oky = usy && (ly > 0 || ny), okz = usz && (lz > 0 || nz);
#if RTX_PRUNE_AXIS
				// (plain scalars, each record built in one go: assembled from the components of other vectors, the records went through 32 bytes of scratch per lane --
				// the kernel's only scratch)
				const float ilx = nx ? -hx : lx, ihx = nx ? -lx : hx, ily = ny ? -hy : ly, ihy = ny ? -ly : hy, ilz = nz ? -hz : lz, ihz = nz ? -lz : hz;
				const float ocx = nx ? -B.ocx : B.ocx, ocy = ny ? -B.ocy : B.ocy, ocz = nz ? -B.ocz : B.ocz;
				const float cap = B.kd * (kPruneC / kFilterK);      // 216 dmax (kd = K dmax, rounded up)
				const float omax = fmaxf(fmaxf(fabsf(B.ocx) + B.rox, fabsf(B.ocy) + B.roy), fabsf(B.ocz) + B.roz);
				const float eB = kFilterK * (omax + F(mp[13])) * (1.0f + 0x1p-20f) + 1e-30f;      // K (|orig| + |vertex|): see planeAlive
				const float inf = __builtin_inff(), kd2 = 2.0f * B.kd;
				*(f4v*)(pu + 0) = f4v{ ilx, ihx, ocx - B.rox, ocx + B.rox };  *(f4v*)(pu + 4) = f4v{ __uint_as_float(nx ? 0x80000000u : 0u), okx ? inf : -inf, cap, omax };
				*(f4v*)(pu + 8) = f4v{ ily, ihy, ocy - B.roy, ocy + B.roy };  *(f4v*)(pu + 12) = f4v{ __uint_as_float(ny ? 0x80000000u : 0u), oky ? inf : -inf, cap, omax };
				*(f4v*)(pu + 16) = f4v{ ilz, ihz, ocz - B.roz, ocz + B.roz }; *(f4v*)(pu + 20) = f4v{ __uint_as_float(nz ? 0x80000000u : 0u), okz ? inf : -inf, cap, omax };
				// (the bundle's fields as opaque scalars: the compiler keeps the struct in four-float slices and gathered these records from them through scratch)
				float dcx = B.dcx, dcy = B.dcy, dcz = B.dcz, rdx = B.rdx, rdy = B.rdy, rdz = B.rdz;
				asm volatile("" : "+v"(dcx), "+v"(dcy), "+v"(dcz), "+v"(rdx), "+v"(rdy), "+v"(rdz));
				*(f4v*)(pu + 24) = f4v{ dcx, rdx, B.ocx, B.rox }; *(f4v*)(pu + 28) = f4v{ kd2, eB, 0.0f, 0.0f };
				*(f4v*)(pu + 32) = f4v{ dcy, rdy, B.ocy, B.roy }; *(f4v*)(pu + 36) = f4v{ kd2, eB, 0.0f, 0.0f };
				*(f4v*)(pu + 40) = f4v{ dcz, rdz, B.ocz, B.roz }; *(f4v*)(pu + 44) = f4v{ kd2, eB, 0.0f, 0.0f };
#else
				f4v a, b, c, e;
				a.x = nx ? -hx : lx; a.y = nx ? -lx : hx; a.z = ny ? -hy : ly; a.w = ny ? -ly : hy;
				b.x = nz ? -hz : lz; b.y = nz ? -lz : hz;
				const float ocx = nx ? -B.ocx : B.ocx, ocy = ny ? -B.ocy : B.ocy, ocz = nz ? -B.ocz : B.ocz;
				b.z = ocx - B.rox; b.w = ocx + B.rox;
				c.x = ocy - B.roy; c.y = ocy + B.roy; c.z = ocz - B.roz; c.w = ocz + B.roz;
				e.x = B.kd * (kPruneC / kFilterK);      // 216 dmax (kd = K dmax, rounded up)
				e.y = fmaxf(fmaxf(fabsf(B.ocx) + B.rox, fabsf(B.ocy) + B.roy), fabsf(B.ocz) + B.roz);
				e.z = __uint_as_float((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u) | (okx ? 8u : 0u) | (oky ? 16u : 0u) | (okz ? 32u : 0u));
				e.w = kFilterK * (e.y + F(mp[13])) * (1.0f + 0x1p-20f) + 1e-30f;      // K (|orig| + |vertex|): see planeAlive
				f4v f, g;
				f.x = __uint_as_float(nx ? 0x80000000u : 0u); f.y = __uint_as_float(ny ? 0x80000000u : 0u); f.z = __uint_as_float(nz ? 0x80000000u : 0u);
				f.w = okx ? __builtin_inff() : -__builtin_inff();
				g.x = oky ? __builtin_inff() : -__builtin_inff(); g.y = okz ? __builtin_inff() : -__builtin_inff(); g.z = 0; g.w = 0;
				*(f4v*)(pu + 0) = a; *(f4v*)(pu + 4) = b; *(f4v*)(pu + 8) = c; *(f4v*)(pu + 12) = e;
				*(f4v*)(pu + 16) = f; *(f4v*)(pu + 20) = g;
#endif
			}
			if (!B.sane) pruneRecs = nullptr;      // NaN / inf / huge coordinates somewhere in the bundle: nothing is pruned
		}
	}
	if (WIDE && RTX_DBG) { cnt.aWalks++; if (shadow) cnt.sWalks++; }
#if RTX_PRUNE_LANEK
	uint32_t laneK = WIDE ? pruneLaneK(pu) : 0u;
	asm volatile("" : "+v"(laneK));      // (one register, computed once per walk: not re-derived inside the node loop)
#else
	const uint32_t laneK = 0;
#endif
	if (WIDE) {
		// (the rays in `consider` have passed the root box: traceWave)
		const uint64_t m0 = ballot(consider);
		if (laneNow() == 0) { WideItem it; it.link = 1; it.first = 0; it.maskLo = (uint32_t)m0; it.maskHi = (uint32_t)(m0 >> 32); stack[0] = it; }
		sp = 1;
	}
	for (;;) {
		RTX_DBG_ONLY(const unsigned long long dbgP1 = __builtin_readcyclecounter();)
		// ---- phase 1: nodes.  The reached leaves are noted in a small per-wave table in LDS.
		uint32_t batch = 0, total = 0;      // total = references of the batch: its leaves form ONE stream, entry k starts at entries[k].start
		uint32_t soleFirst = 0;             // kLeafBatch == 1: the first reference of the batch's only leaf
		uint32_t myReach = 0;               // bit k: this lane's ray reached leaf k of the batch (the lane's own column of the table's masks)
		if (WIDE) {
			const uint32_t lane = laneNow();
			// which rays an item concerns is wave-level bookkeeping (masks in SGPRs); only the box tests are per lane
			const uint64_t openM = ballot(resume != kNever);      // (rays close in phase 2 only)
			auto noteLeaf = [&](int32_t link, uint32_t first, uint64_t m) {
				const bool in = ((((lane & 32u) ? (uint32_t)(m >> 32) : (uint32_t)m) >> (lane & 31u)) & 1u) != 0;
				const uint32_t n = (uint32_t)~link;
				if (RTX_DBG) { cnt.wLeaves++; if (shadow) cnt.sLeaves++; }
				if (m != 0 && n != 0) {
					// (one leaf per batch: what the pass needs of it stays in SGPRs -- no trip through the table in LDS)
					if (kLeafBatch == 1 && !FEWRAYS) soleFirst = uni(first);
					else if (lane == 0) {
						LeafEntry en;
						en.first = first; en.count = n; en.maskLo = (uint32_t)m; en.maskHi = (uint32_t)(m >> 32); en.start = total; en.pad0 = en.pad1 = en.pad2 = 0;
						entries[batch] = en;
					}
					myReach |= in ? 1u << batch : 0u;
					batch = uni(batch + 1);
					total = uni(total + n);
				}
			};
#define RTX_PUSH(lnk, fst, m)                                                                                                  \
			{                                                                                                                  \
				if (lane == 0) { WideItem ni; ni.link = (int32_t)(lnk); ni.first = (fst); ni.maskLo = (uint32_t)(m); ni.maskHi = (uint32_t)((m) >> 32); stack[sp] = ni; } \
				sp = uni(sp + 1);                                                                                              \
			}
			while (sp != 0 && batch < kLeafBatch) {
				sp = uni(sp - 1);
				const WideItem it = stack[sp];
				const int32_t link = (int32_t)uni((uint32_t)it.link);
				const uint32_t itFirst = it.first;
				const uint32_t mlo = uni(it.maskLo), mhi = uni(it.maskHi);
				const uint64_t inM = ((uint64_t)mhi << 32 | mlo) & openM;
				if (link < 0) {
					// a leaf, in the reference's order: note it with the rays that reached it and are still open
					noteLeaf(link, itFirst, inM);
					continue;
				}
				if (inM == 0) continue;
				if (RTX_DBG) { cnt.wNodes++; if (shadow) cnt.sVisits++; }
				const WideNode* w = wideNodes + (uint32_t)(link - 1);
				// slots 3..0 are requested at once (their fetch runs while the prune records are evaluated); the SGPR file does not hold eight slots: see below
				u32x16 wa = sload16(w), wb = sload16((const char*)w + 64);
				// Which slots can contribute at all: of the first 2 kWideSlots lanes, lane k looks at record k (boxes, then planes: rtxd::PruneBlock).
				uint32_t aliveM = (1u << kWideSlots) - 1u;
				// (four slots: not at the root, where they are as good as never pruned -- 7 of 259 in tools/research/pruned_walk_sim.py; eight: everywhere)
				const bool evalPrune = pruneRecs != nullptr && (kPruneRoot || link != 1);
#if RTX_PRUNE_AXIS
				// (aliveM in the form pruneEval8 returns: slot k = bit 4 k)
				aliveM = 0x11111111u;
				if (evalPrune) {
					aliveM = pruneEval8<BOXES, CULL>(pruneRecs + ((size_t)(uint32_t)(link - 1) << 9), pu, tmaxB, laneK);
					if (RTX_DBG) { cnt.wS4++; cnt.wLeafSkips += (uint32_t)kWideSlots - (uint32_t)__popc(aliveM); }
				}
#define RTX_ALIVE(k) ((aliveM >> (4 * (k))) & 1u)
#define RTX_ALIVE4(q) ((aliveM >> (16 * (q))) & 0x1111u)
#else
#define RTX_ALIVE(k) ((aliveM >> (k)) & 1u)
#define RTX_ALIVE4(q) ((aliveM >> (4 * (q))) & 0xfu)
				if (evalPrune) {
					// lanes [0, kWideSlots): the slots' boxes (PruneRec), [kWideSlots, 2 kWideSlots): their planes (PlaneRec); both tests run on every lane's record
					const RTX_AS1 f4v* pr = (const RTX_AS1 f4v*)(pruneRecs + (((uint32_t)(link - 1) * (2u * kWideSlots) + (lane & (2u * kWideSlots - 1u))) << 5));
					const f4v r0 = pr[0], r1 = pr[1];
					const bool aliveBox = !BOXES || pruneAlive(r0, r1, pu, tmaxB);      // (BOXES: see the kernels' template parameter)
					const bool alivePlane = !CULL || planeAlive(r0, r1, B, pu[15], tmaxB);      // (planeAlive's rejections assume culling; pruneEval8 has the other form)
					const uint32_t bal = (uint32_t)ballot((lane & (uint32_t)kWideSlots) ? alivePlane : aliveBox);
					aliveM = bal & (bal >> kWideSlots) & ((1u << kWideSlots) - 1u);
					if (RTX_DBG) { cnt.wS4++; cnt.wLeafSkips += (uint32_t)kWideSlots - (uint32_t)__popc(aliveM); }
				}
#endif
				// The node's slots four at a time (two s_load_dwordx16: the SGPR file holds no more), the last four first, slots 3..0 of every four in
				// that order, so that slot 0 ends up on top of the stack; four slots none of which is alive are not fetched.
#define RTX_SLOT(rec, base, k)                                                                                                     \
				if ((int32_t)rec[base + 6] != 0 && RTX_ALIVE(k)) {                                                         \
					const bool fail = boxFailsRegular(F(rec[base]), F(rec[base + 1]), F(rec[base + 2]), F(rec[base + 3]), F(rec[base + 4]), F(rec[base + 5]), o, ix, iy, iz); \
					const uint64_t mk_ = ballot(!fail) & inM;                                                                       \
					if (RTX_DBG) cnt.wS3++;                                                                                         \
					if (mk_ != 0) RTX_PUSH(rec[base + 6], rec[base + 7], mk_)                                                       \
				}
				if (kWideSlots > 4 && RTX_ALIVE4(1) != 0) {
					// the upper slots first, four at a time from the top (they are pushed first): the registers of slots 3..0 are given up for them and loaded
					// again afterwards (a hit in the scalar cache; touching the lines of the upper slots when the node is popped made no difference: profiles/r04_wide8.txt)
#pragma unroll
					for (int q4 = kWideSlots / 4 - 1; q4 >= 1; --q4) {
						if (RTX_ALIVE4(q4) == 0) continue;
						const u32x16 wc = sload16((const char*)w + 128 * q4), wd = sload16((const char*)w + 128 * q4 + 64);
						RTX_SLOT(wd, 8, 4 * q4 + 3) RTX_SLOT(wd, 0, 4 * q4 + 2) RTX_SLOT(wc, 8, 4 * q4 + 1) RTX_SLOT(wc, 0, 4 * q4)
					}
					const char* w0 = (const char*)w;
					asm volatile("" : "+s"(w0));      // (a fresh load, not the value from before kept in 32 more SGPRs)
					wa = sload16(w0); wb = sload16(w0 + 64);
				}
				if (RTX_ALIVE4(0) != 0) {
					RTX_SLOT(wb, 8, 3) RTX_SLOT(wb, 0, 2) RTX_SLOT(wa, 8, 1)
					// slot 0 would be popped next: a leaf there is noted right away (no trip through the stack) while the batch has room
					if ((int32_t)wa[6] != 0 && RTX_ALIVE(0)) {
						const bool fail = boxFailsRegular(F(wa[0]), F(wa[1]), F(wa[2]), F(wa[3]), F(wa[4]), F(wa[5]), o, ix, iy, iz);
						const uint64_t mk_ = ballot(!fail) & inM;
						if (RTX_DBG) cnt.wS3++;
						if (mk_ != 0) {
							if ((int32_t)wa[6] < 0 && batch < kLeafBatch) noteLeaf((int32_t)wa[6], wa[7], mk_);
							else RTX_PUSH(wa[6], wa[7], mk_)
						}
					}
				}
#undef RTX_SLOT
#undef RTX_PUSH
#undef RTX_ALIVE
#undef RTX_ALIVE4
			}
		}
		else {
			while (i < nN && batch < kLeafBatch) {
				const int32_t link = (int32_t)nd[6];
				const uint32_t next = uni(i + 1);
				const uint32_t nxt = link > 0 ? (uint32_t)link : next;
				// speculative prefetch of both possible successors (clamped to the array), issued after nd has arrived
				// (written as two unconditional loads; the compiler sinks each into the branch that consumes it, so one
				// node record is fetched per visit -- measured faster than keeping both speculative loads in flight)
				const Node* nb = after(nodes, nd[7]);
				const u32x8 nxA = sload8(nb + (next < last ? next : last));
				const u32x8 nxB = sload8(nb + (nxt < last ? nxt : last));
				const bool act = i >= resume;
				// slab test, objects.cpp:546-567: (bounds[sign] - orig) * invdir per axis, sequential compares.
				// ((lo_i, hi_i) - orig_i) * invdir_i as one packed subtract + one packed multiply per axis
				const float xlo = (F(nd[0]) - o.x) * ix, xhi = (F(nd[1]) - o.x) * ix, ylo = (F(nd[2]) - o.y) * iy, yhi = (F(nd[3]) - o.y) * iy;
				const float zlo = (F(nd[4]) - o.z) * iz, zhi = (F(nd[5]) - o.z) * iz;
				bool fail;
				if (REGULAR) {
					// no NaN can arise (finite boxes with lo <= hi, finite origin, finite 1/dir): the sign-selected entry / exit values
					// are the smaller / larger product of each axis and the reference's sequential compares (objects.cpp:553-567) are
					// exactly "largest entry > smallest exit" (the same-axis pairs it never compares cannot fail)
					// (written as instructions: the compiler would first canonicalise all six operands for a possible signalling NaN)
					float nx, ny, nz, fx, fy, fz, tn, tf;
					asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(xlo), "v"(xhi)); asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(xlo), "v"(xhi));
					asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ylo), "v"(yhi)); asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ylo), "v"(yhi));
					asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(zlo), "v"(zhi)); asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(zlo), "v"(zhi));
					asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(nz));
					asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(fz));
					fail = tn > tf;
				}
				else {
					float tmin = sx ? xhi : xlo, tmx = sx ? xlo : xhi;
					const float tymin = sy ? yhi : ylo, tymax = sy ? ylo : yhi;
					fail = (tmin > tymax) || (tymin > tmx);
					if (tymin > tmin) tmin = tymin;
					if (tymax < tmx) tmx = tymax;
					const float tzmin = sz ? zhi : zlo, tzmax = sz ? zlo : zhi;
					fail = fail || (tmin > tzmax) || (tzmin > tmx);
				}
				const bool pass = act && !fail;
				if (act && fail) resume = nxt;
				if (STATS) cnt.box += __popcll(ballot(act));
				if (RTX_DBG) cnt.wNodes++;
				const uint64_t m = ballot(pass);
				if (m == 0) {
					nd = nxB;
					i = uni(nxt);
					continue;
				}
				if (link < 0) {
					const uint32_t n = (uint32_t)~link;
					if (STATS) cnt.tri += (unsigned long long)__popcll(m) * n;
					if (RTX_DBG) cnt.wLeaves++;
					if (n != 0) {
						if (kLeafBatch == 1 && !FEWRAYS) soleFirst = nd[7];
						else if (laneNow() == 0) {
							LeafEntry en;
							en.first = nd[7]; en.count = n; en.maskLo = (uint32_t)m; en.maskHi = (uint32_t)(m >> 32); en.start = total; en.pad0 = en.pad1 = en.pad2 = 0;
							entries[batch] = en;
						}
						myReach |= pass ? 1u << batch : 0u;
						batch = uni(batch + 1);
						total = uni(total + n);
					}
				}
				nd = nxA;
				i = next;
			}
		}
		// ---- phase 2: the references of the noted leaves, as one stream in the reference's order, 64 at a time (a pass may
		// hold the end of one leaf and several whole small ones).  ALL 64 lanes take part (uniform control flow): lane k
		// classifies reference k of the pass against the bundle; the survivors are then tested exactly, in stream order,
		// by the lanes that passed the box of the survivor's leaf.
		const uint32_t lane = laneNow();
		RTX_DBG_ONLY(
		const unsigned long long dbgP2 = __builtin_readcyclecounter();
		cnt.cNodes += dbgP2 - dbgP1;
		unsigned long long dbgExact = 0;
		)
		uint32_t ecur = 0;                 // first entry that is not finished yet
		// which reference, of which entry, a lane holds in the pass that starts at p0: later entries overwrite earlier ones
		// from their start on
		auto assign = [&](uint32_t p0, uint32_t& r, uint32_t& myEnt) {
			r = 0; myEnt = 0;
			if (kLeafBatch == 1 && !FEWRAYS) { r = soleFirst + p0 + lane; return; }
			for (uint32_t e = ecur; e < batch; e = uni(e + 1)) {
				const u32x4v head = *(const u32x4v*)&entries[e];          // one LDS round trip per entry: start, count, first
				const uint32_t st = uni(head.x);
				if (st >= p0 + 64) break;
				const uint32_t n = uni(head.y), first = uni(head.z);
				const int32_t rel = (int32_t)(st - p0);                   // (negative: the leaf began in an earlier pass)
				const bool here = (int32_t)lane >= rel;
				r = here ? first + (lane - (uint32_t)rel) : r;
				myEnt = here ? e : myEnt;
				if (st + n <= p0 + 64) ecur = uni(e + 1);              // finished within this pass
			}
		};
		// classification of one pass + exact tests of its survivors; true when every ray of the wave is done
		auto process = [&](uint32_t p0, const f4v& va, const f4v& vb, const f2v& vc, uint32_t myEnt) -> bool {
			RefA ra; RefB rb; RefC rc;
			ra.v0x = va.x; ra.v0y = va.y; ra.v0z = va.z; ra.tri = __float_as_uint(va.w);
			rb.e1x = vb.x; rb.e1y = vb.y; rb.e1z = vb.z; rb.e2x = vb.w; rc.e2y = vc.x; rc.e2z = vc.y;
			FilterState fs;
			const bool valid = p0 + lane < total;
#if RTX_FILTER_MASKS
			const uint64_t open1 = ballot(valid) & ~bundleRejects1<CULL>(B, tmaxB, ra, rb, rc, fs);
			if (RTX_DBG) { cnt.wChunks++; if (shadow) cnt.sPasses++; }
			if (open1 == 0) { if (RTX_DBG) cnt.wChunkSkips++; return false; }
			uint64_t cand = open1 & ~bundleRejects2<CULL>(B, ra, rb, rc, fs, open1);
#else
			const bool rej1 = bundleRejects1<CULL>(B, tmaxB, ra, rb, rc, fs);
			if (RTX_DBG) { cnt.wChunks++; if (shadow) cnt.sPasses++; }
			if (ballot(valid && !rej1) == 0) { if (RTX_DBG) cnt.wChunkSkips++; return false; }
			const bool rej2 = bundleRejects2<CULL>(B, ra, rb, rc, fs);
			uint64_t cand = ballot(valid && !rej1 && !rej2);
#endif
			if (RTX_DBG) { cnt.wTri += __popcll(cand); if (shadow) cnt.sExact += __popcll(cand); if (cand == 0) cnt.wS2++; }
			if (cand == 0) return false;
			bool improved = false;
			RTX_DBG_ONLY(const unsigned long long dbgE0 = __builtin_readcyclecounter();)
			// Few rays, many survivors (a part of a slow tile, an SSAA item of one pixel, at a pole where hundreds of sliver
			// triangles pass every filter): one step per RAY instead of one per survivor -- the ray is broadcast, every
			// surviving lane tests its own triangle against it, and the nearest hit is the wave minimum; among equal t the
			// lowest lane = the first in the reference's order, as its strict "<" has it (objects.cpp:623).
			if (FEWRAYS && !STATS && nRays4 <= (uint32_t)__popcll(cand) * 3u) {
				const uint64_t raysOpen = ballot(resume != kNever);
				const uint32_t mLo = entries[myEnt].maskLo, mHi = entries[myEnt].maskHi;      // the rays that reached this lane's leaf
				const bool isCand = ((cand >> lane) & 1ull) != 0;
				uint64_t rays = raysOpen;
				while (rays != 0) {
					const int r = __builtin_ctzll(rays);
					rays &= rays - 1;
#define RTX_RL(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), r))      // (through the crossbar: no measurable difference here)
					const float rox = RTX_RL(o.x), roy = RTX_RL(o.y), roz = RTX_RL(o.z), rdx = RTX_RL(d.x), rdy = RTX_RL(d.y), rdz = RTX_RL(d.z);
					const float rbt = RTX_RL(bt);
#undef RTX_RL
					const bool reached = ((((uint32_t)r & 32u) ? mHi : mLo) >> ((uint32_t)r & 31u)) & 1u;
					float t = __builtin_inff(), u = 0, v = 0;
					if (isCand && reached) triTestLane<CULL>(ra, rb, rc, rox, roy, roz, rdx, rdy, rdz, t, u, v);
					const bool better = t < rbt;
					const float tmin = -waveMax(better ? -t : -__builtin_inff());
					const uint64_t at = ballot(better && t == tmin);
					if (at != 0) {
						const int w = __builtin_ctzll(at);
						const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), w)), wu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), w));
						const float wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), w));
						const uint32_t wtri = (uint32_t)__builtin_amdgcn_readlane((int)ra.tri, w);
						if (lane == (uint32_t)r) { bt = wt; bu = wu; bv = wv; btri = wtri; improved = true; }
					}
				}
				cand = 0;
			}
#if RTX_EXACT_HOIST
			const float btBefore = bt;      // (bt only falls: one compare after the survivors instead of one per survivor, and no copy of bt carried round the loop)
#endif
			while (cand != 0) {
				const int c = __builtin_ctzll(cand);
				cand &= cand - 1;
				// The survivor's record for all lanes.  Through v_readlane it would arrive in SGPRs, and every instruction with an
				// SGPR source issues in ~4.2 cycles instead of ~2.4 (tools/ubench/valu_rate.hip): ~40 such instructions per test.
				// Through the LDS crossbar (ds_bpermute) it arrives in VGPRs without a VALU instruction: pass 1 at 4096^2
				// 5.77 -> 5.58 ms, cfg2 at 1080p 2.29 -> 2.17 (RTX_TRI_BPERMUTE=0: the v_readlane form).
#define RTX_RL(x) __int_as_float(__builtin_amdgcn_ds_bpermute(c << 2, __float_as_int(x)))
				const float v0x = RTX_RL(ra.v0x), v0y = RTX_RL(ra.v0y), v0z = RTX_RL(ra.v0z);
				const float e1x = RTX_RL(rb.e1x), e1y = RTX_RL(rb.e1y), e1z = RTX_RL(rb.e1z);
				const float e2x = RTX_RL(rb.e2x), e2y = RTX_RL(rc.e2y), e2z = RTX_RL(rc.e2z);
#undef RTX_RL
				const uint32_t tri = (uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)ra.tri);
				// the rays that reached the survivor's leaf
				const uint32_t ent = (kLeafBatch == 1 && !FEWRAYS) ? 0u : (uint32_t)__builtin_amdgcn_ds_bpermute(c << 2, (int)myEnt);
				const bool pass = ((myReach >> ent) & 1u) != 0;      // (no dependent LDS read of the table's mask per survivor)
#if !RTX_EXACT_HOIST
				const float before = bt;
#endif
				if (pass) triTestOne<CULL, STATS>(v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z, tri, o, d, bt, bu, bv, btri);
#if !RTX_EXACT_HOIST
				improved = improved || bt < before;
#endif
			}
#if RTX_EXACT_HOIST
			improved = improved || bt < btBefore;
#endif
			RTX_DBG_ONLY(dbgExact += __builtin_readcyclecounter() - dbgE0;)
			if (ballot(improved) != 0) {
				// any-hit: a shadow ray only asks "is some t < light distance" (scene.cpp:787); once that is true
				// for this lane no later triangle or object can change the answer.
				if (!STATS) { if (shadow && bt < tLimit) resume = kNever; }
				// no lane can record a t that is not below its own current limit any more
				const bool open = resume != kNever;
				tmaxB = unif(waveMax(open ? fminf(bt, tLimit) : -__builtin_inff()));
				if (!STATS && ballot(open) == 0) return true;
			}
			return false;
		};
		// one leaf per batch: a second pass is rare (a leaf of more than 64 references) -- one pass per round trip, no idle loads of a pass B
		if (kLeafBatch == 1 && !FEWRAYS) {
			for (uint32_t p0 = 0; p0 < total; p0 = uni(p0 + 64)) {
				uint32_t rA, entA;
				assign(p0, rA, entA);
				const f4v vaA = *(const RTX_AS1 f4v*)(refA + (rA << 4)), vbA = *(const RTX_AS1 f4v*)(refB + (rA << 4));
				const f2v vcA = *(const RTX_AS1 f2v*)(refC + (rA << 3));
				if (process(p0, vaA, vbA, vcA, entA)) return;
			}
		}
		else
		// two passes are requested together: one memory round trip per 128 references
		for (uint32_t p0 = 0; p0 < total; p0 = uni(p0 + 128)) {
			uint32_t rA, entA, rB = 0, entB = 0;
			assign(p0, rA, entA);
			const f4v vaA = *(const RTX_AS1 f4v*)(refA + (rA << 4)), vbA = *(const RTX_AS1 f4v*)(refB + (rA << 4));
			const f2v vcA = *(const RTX_AS1 f2v*)(refC + (rA << 3));
			const bool two = p0 + 64 < total;
			if (two) assign(p0 + 64, rB, entB);
			const f4v vaB = *(const RTX_AS1 f4v*)(refA + (rB << 4)), vbB = *(const RTX_AS1 f4v*)(refB + (rB << 4));
			const f2v vcB = *(const RTX_AS1 f2v*)(refC + (rB << 3));
			if (process(p0, vaA, vbA, vcA, entA)) return;
			if (two && process(p0 + 64, vaB, vbB, vcB, entB)) return;
		}
		RTX_DBG_ONLY(cnt.cExact += dbgExact; cnt.cFilter += __builtin_readcyclecounter() - dbgP2 - dbgExact;)
		if (WIDE ? sp == 0 : i >= nN) break;
	}
}

// MESH = false: the variant for scenes without triangle meshes (spheres and planes only).  Without the walk the castRay
// state machine fits the register file, and a small frame lasts as long as its slowest wave's chain of dependent rays.
// CULLK: options::useBackfaceCulling as the kernel knows it -- 1 on, 0 off (the product kernels: a per-view constant, so every kernel exists in both forms and
// neither carries the other's walks), -1 read from the view at run time (the instrumented and the probe kernels)
template <bool STATS, bool MESH = true, bool FEWRAYS = false, bool BOXES = true, int CULLK = -1>
__device__ __forceinline__ void traceWave(const Params& P, bool active, bool shadow, V3 o, V3 d, float tmax,
                                          Hit& h, Counts& cnt, uint32_t src = 0)
{
	// src: what the lane's ray is known to pass through (rtxd::PruneRec): 0 nothing, 1 the camera (o == view.camPos exactly),
	// 2 + l point light l (o = P + N bias, d = -normalize(P - pos_l): castRayWave)
	h.obj = -1; h.t = tmax; h.tri = 0; h.u = 0; h.v = 0;
	if (STATS) cnt.rays += __popcll(ballot(active));
	const bool cull = CULLK < 0 ? (uni(P.view.flags) & 1u) != 0 : CULLK != 0;
	// AccelerationStructure::intersectBox's per-ray part (objects.cpp:543-544), hoisted out of the walk
	const float ix = 1 / d.x, iy = 1 / d.y, iz = 1 / d.z;
	const bool sx = ix < 0, sy = iy < 0, sz = iz < 0;
	bool live = active;
	const uint32_t nObj = uni(P.nObjects);
	for (uint32_t oi = 0; oi < nObj; oi = uni(oi + 1)) {
		const Object* ob = uni(P.objects + oi);
		const u32x16 rec = sload16(ob);         // type, material, pos[3], r2, normal[3], mesh, ...
		const int type = (int)rec[0];
		const int mat = (int)rec[1];
		// transparent objects do not cast shadows (scene.cpp:733)
		const bool consider = live && !(shadow && mat == 2);
		if (ballot(consider) == 0) continue;
		if (MESH && type == 3) {
			const Mesh* M = uni(P.meshes + (int)rec[9]);
			const u32x16 rec2 = sload16((const char*)ob + 64);      // specular, nSpecular, rootBox[6], fatRadius, centre[3], radius, meshFlags
			const u32x16 rec3 = sload16((const char*)ob + 128);     // nodes, refA, refB, refC, wide, prune, nNodes, vmax
			// The rays are walked as one bundle (meshWalk) -- unless the bundle is too wide at this mesh for the bundle
			// filter to reject much (rays of a silhouette tile that hit different objects, a grazing strip of shadow-ray
			// origins, coarse frames): then the lanes on one side of the middle of the widest axis go first, the others
			// later, halving until the bundle is narrow.  Which lanes walk together changes the amount of work only.
#if !RTX_REC2_RELOAD
			const float fat = F(rec2[8]), mrad = F(rec2[12]);
			const float mcx = F(rec2[9]), mcy = F(rec2[10]), mcz = F(rec2[11]);
#endif
			const uint32_t mflags = rec2[13];
			// Rays that fail the root box (objects.cpp:590) take no further part: the bundles are formed by the others.
			bool pending = consider;
			// the min / max form of the box test (meshWalk<.., REGULAR>) is exact when no NaN can arise: regular boxes, finite
			// origins and finite 1 / dir for every ray of the wave
			// -- per RAY: a ray with a zero direction component (1 / 0 = inf: the central column / row of an unrotated camera) or a huge origin is walked
			// apart from the others, in the binary form; the rest of the wave keeps the wide walk (one such lane used to demote all 64)
			const bool laneRegular = fabsf(ix) < __builtin_inff() && fabsf(iy) < __builtin_inff() && fabsf(iz) < __builtin_inff() &&
			                         fabsf(o.x) < 0x1p100f && fabsf(o.y) < 0x1p100f && fabsf(o.z) < 0x1p100f;
			const bool meshRegular = (mflags & 2u) != 0;
			const bool wideOk = (mflags & 4u) != 0;
			if ((mflags & 1u) != 0) {
				const float xlo = (F(rec2[2]) - o.x) * ix, xhi = (F(rec2[3]) - o.x) * ix, ylo = (F(rec2[4]) - o.y) * iy, yhi = (F(rec2[5]) - o.y) * iy;
				const float zlo = (F(rec2[6]) - o.z) * iz, zhi = (F(rec2[7]) - o.z) * iz;
				float tmin = sx ? xhi : xlo, tmx = sx ? xlo : xhi;
				const float tymin = sy ? yhi : ylo, tymax = sy ? ylo : yhi;
				bool fail = (tmin > tymax) || (tymin > tmx);
				if (tymin > tmin) tmin = tymin;
				if (tymax < tmx) tmx = tymax;
				const float tzmin = sz ? zhi : zlo, tzmax = sz ? zlo : zhi;
				fail = fail || (tmin > tzmax) || (tzmin > tmx);
				if (STATS) cnt.box += __popcll(ballot(consider && fail));       // (their root-box test is still a test of the reference)
				pending = consider && !fail;
			}
			// (src == 1: the ray starts at the camera, o == view.camPos bit for bit -- castRayWave)
			const bool oneOrigin = RTX_SAME_ORIGIN && !STATS && ballot(pending && src != 1u) == 0;
			while (ballot(pending) != 0) {
				bool cl = pending;
				// the regular rays first, as one bundle (or more: the split below); what is left afterwards are the irregular ones
				bool regular = false;
				if (meshRegular) {
					const uint64_t regM = ballot(pending && laneRegular);
					if (regM != 0) { cl = pending && laneRegular; regular = true; }
				}
#if RTX_REC2_RELOAD
				// what the split rule needs of the object's second line, fetched again per bundle (a hit in the scalar cache, requested ahead of the reductions below)
				// instead of being kept in -- i.e. spilled from -- sixteen SGPRs across the walk of the previous bundle
				const char* ob2 = (const char*)ob + 96;
				asm volatile("" : "+s"(ob2));
				const u32x8 geo = sload8(ob2);      // fatRadius, centre[3], radius, meshFlags, pad[2]
				const float fat = F(geo[0]), mcx = F(geo[1]), mcy = F(geo[2]), mcz = F(geo[3]), mrad = F(geo[4]);
#endif
				Bundle B = makeBundle(cl, o, d, oneOrigin);
				for (int split = 0; split < RTX_MAX_SPLITS; ++split) {
					// width of the bundle where it can meet the mesh: origin box, and direction box times the distance to the far side
					const float dist = fmaxf(fmaxf(fabsf(B.ocx - mcx), fabsf(B.ocy - mcy)), fabsf(B.ocz - mcz)) + mrad;
					const float wo = B.roMax, wd = fmaxf(fmaxf(B.rdx, B.rdy), B.rdz) * dist;
					if (!(wo > fat || wd > fat)) break;
					bool lo;
					if (wo >= wd) lo = B.rox >= B.roy && B.rox >= B.roz ? o.x < B.ocx : (B.roy >= B.roz ? o.y < B.ocy : o.z < B.ocz);
					else lo = B.rdx >= B.rdy && B.rdx >= B.rdz ? d.x < B.dcx : (B.rdy >= B.rdz ? d.y < B.dcy : d.z < B.dcz);
					if (ballot(cl && lo) == 0 || ballot(cl && !lo) == 0) break;
					cl = cl && lo;
					B = makeBundle(cl, o, d, oneOrigin);
				}
				float bt, bu, bv; uint32_t btri;
				// The walk may use a source copy of the prune records when ALL its rays pass through that source and have a direction
				// of length 0.99 .. 1.001 (what sourceP assumes; origins further than kSrcAinfMax from a box fall back per record)
				uint32_t srcSel = 0;
				if (!STATS) {
					const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)src, __builtin_ctzll(ballot(cl)));
					if (s0 != 0) {
						const float l2 = len2(d);
						if (ballot(cl && !(src == s0 && l2 >= 0.9802f && l2 <= 1.002f)) == 0) srcSel = s0;
					}
				}
				if (!STATS && cull && regular && wideOk) meshWalk<STATS, true, true, true, FEWRAYS, BOXES>(rec3, B, cl, shadow, o, d, ix, iy, iz, sx, sy, sz, h.t, bt, bu, bv, btri, cnt, srcSel);
#if RTX_WIDE_NOCULL
				// culling off (options.h:27, objects.cpp:75-79): the same wide walk; its filter mirrors by the sign of det (bundleRejects1), its plane records
				// keep what may show either face (pruneEval8<.., false>)
				else if (!STATS && regular && wideOk) meshWalk<STATS, false, true, true, FEWRAYS, BOXES>(rec3, B, cl, shadow, o, d, ix, iy, iz, sx, sy, sz, h.t, bt, bu, bv, btri, cnt, srcSel);
#endif
				else if (cull && regular) meshWalk<STATS, true, true, false, FEWRAYS>(rec3, B, cl, shadow, o, d, ix, iy, iz, sx, sy, sz, h.t, bt, bu, bv, btri, cnt);
				else if (cull) meshWalk<STATS, true, false, false, FEWRAYS>(rec3, B, cl, shadow, o, d, ix, iy, iz, sx, sy, sz, h.t, bt, bu, bv, btri, cnt);
				else meshWalk<STATS, false, false, false, FEWRAYS>(rec3, B, cl, shadow, o, d, ix, iy, iz, sx, sy, sz, h.t, bt, bu, bv, btri, cnt);
				if (cl && bt < kFltMax && bt < h.t) { h.obj = (int)oi; h.t = bt; h.tri = btri; h.u = bu; h.v = bv; }   // scene.cpp:740-745
				pending = pending && !cl;
			}
		}
		else {
			const V3 c = mk(F(rec[2]), F(rec[3]), F(rec[4]));
			float t0 = kFltMax; bool hit;
			if (type == 1) {                       // Sphere::intersectObject, objects.cpp:774-786
				const float r2 = F(rec[5]);
				const V3 L = c - o;
				const float tca = dot(L, d);
				const float d2 = dot(L, L) - tca * tca;
				hit = !(d2 > r2);
				const float thc = __builtin_sqrtf(r2 - d2);
				t0 = tca - thc;
				const float t1 = tca + thc;
				if (t0 < 0) t0 = t1;
				if (t0 < 0) hit = false;
			}
			else {                                 // Plane::intersectObject, objects.cpp:807-814
				const V3 n = mk(F(rec[6]), F(rec[7]), F(rec[8]));
				const float denom = dot(d, n);
				hit = !(fabsf(denom) < RTX_EPS8);
				t0 = dot(c - o, n) / denom;
				hit = hit && (t0 >= 0);
			}
			if (consider && hit && t0 < h.t) { h.obj = (int)oi; h.t = t0; h.u = 0; h.v = 0; }   // scene.cpp:748-752
		}
		if (!STATS) live = live && !(shadow && h.obj >= 0);
	}
}

// ------------------------------------------------------------------------------------------------
// castRay as a per-lane state machine (scene.cpp:758-946)
// ------------------------------------------------------------------------------------------------
enum : int {
	ST_DONE = 0, ST_NEWRAY, ST_WAIT_PRIMARY, ST_WAIT_SHADOW, ST_NEXT_LIGHT, ST_LIGHTS_DONE, ST_RETURN
};
enum : int { FR_REFL = 1, FR_TRANS1 = 2, FR_TRANS2 = 3 };

struct Lane {
	int state;
	int sp;                       // recursion depth == number of live frames
	V3 ro, rd;                    // current ray
	V3 col;                       // value being returned
	// hit being shaded
	int obj, mat;
	V3 P, N, objColor, diff, spec, L, I;
	float specCoef, nSpec, dsum, ssum;
	uint32_t li, si;
	// pending trace request
	float qtmax;                  // (origin / direction of the pending request are derived from the state: see castRayWave)
	bool qmoot;                   // the pending shadow ray cannot influence the pixel (see advance)
	bool qarea;                   // the pending shadow ray goes to a sample point of an area light (consume: no second look at the light's record)
	uint32_t qsrc;                // the pending shadow ray's source class: 2 + l for point light l with a source copy of the prune records, else 0
};

__device__ __forceinline__ float& frameAt(const Params& P, uint32_t gl, int slot, int field)
{
	return P.frames[((size_t)slot * kFrameFields + field) * P.totalLanes + gl];
}

// One object's part of shadePrimary: the object's record `a` / `b` (its two 64-byte halves) is wave-uniform.
__device__ __forceinline__ void shadeObject(const Params& P, Lane& s, const Hit& h, const u32x16& a, const u32x16& b)
{
	s.mat = (int)a[1];
	s.objColor = mk(F(a[10]), F(a[11]), F(a[12]));
	s.specCoef = F(b[0]);
	s.nSpec = F(b[1]);
	const int type = (int)a[0];
	if (type == 1) s.N = normalized(s.P - mk(F(a[2]), F(a[3]), F(a[4])));          // objects.cpp:788-796
	else if (type == 2) s.N = mk(F(a[6]), F(a[7]), F(a[8]));                       // objects.cpp:816-824
	else {
		// Mesh::getSurfaceData, objects.cpp:121-151
		const Mesh* M = uni(P.meshes + (int)a[9]);
		const GlobalFloats uvp = inGlobal((const float*)sloadp(&M->uv)) + (size_t)h.tri * 6;
		const GlobalFloats np = inGlobal((const float*)sloadp(&M->nrm)) + (size_t)h.tri * 9;
		const float* normalMap = (const float*)sloadp(&M->normal);
		const float* diffuseMap = (const float*)sloadp(&M->diffuse);
		const float* specularMap = (const float*)sloadp(&M->specular);
		const float u = h.u, v = h.v;
		const float w = 1 - u - v;
		const float texx = uvp[2] * u + uvp[4] * v + uvp[0] * w;
		const float texy = uvp[3] * u + uvp[5] * v + uvp[1] * w;
		V3 n = (load3(np + 3) * u + load3(np + 6) * v + load3(np) * (1 - u - v)) / 3;
		n = normalized(n);
		if (normalMap) {
			const GlobalFloats tbp = inGlobal((const float*)sloadp(&M->tb)) + (size_t)h.tri * 6;
			const uint32_t nW = sload1(&M->nW), nH = sload1(&M->nH);
			const int x = texel((int)nW, texx), y = texel((int)nH, texy);
			// normalise(texel as loaded): the reference's in-place re-normalisation race is resolved this way
			const V3 tn = normalized(load3(inGlobal(normalMap) + ((size_t)y * nW + x) * 3));
			V3 r;
			r.x = tn.x * tbp[0] + tn.y * tbp[3] + tn.z * n.x + 0.0f;
			r.y = tn.x * tbp[1] + tn.y * tbp[4] + tn.z * n.y + 0.0f;
			r.z = tn.x * tbp[2] + tn.y * tbp[5] + tn.z * n.z + 0.0f;
			n = normalized(r);
		}
		s.N = n;
		if (diffuseMap) {                                   // Mesh::getDiffuseColor, objects.cpp:153-163
			const uint32_t dW = sload1(&M->dW), dH = sload1(&M->dH);
			s.objColor = load3(inGlobal(diffuseMap) + ((size_t)texel((int)dH, texy) * dW + texel((int)dW, texx)) * 3);
		}
		if (specularMap) {                                  // Mesh::getSpecularValue, objects.cpp:165-175
			const uint32_t sW = sload1(&M->sW), sH = sload1(&M->sH);
			s.specCoef = inGlobal(specularMap)[(size_t)texel((int)sH, texy) * sW + texel((int)sW, texx)];
		}
	}
}

__device__ __forceinline__ void shadePrimary(const Params& P, Lane& s, const Hit& h)
{
	// scene.cpp:763-775 + Object::getSurfaceData overrides.  The lanes that are here hit one object, sometimes two: the object's
	// record (and a mesh's array pointers) come through the scalar unit, once per distinct object (instead of a chain of
	// dependent per-lane loads: object -> mesh -> arrays).
	s.obj = h.obj;
	s.P = s.ro + s.rd * h.t;
	for (uint64_t rem = ballot(true); rem != 0;) {
		const int o0 = __builtin_amdgcn_readlane(h.obj, __builtin_ctzll(rem));
		const Object* ob = uni(P.objects + o0);
		const u32x16 a = sload16(ob), b = sload16((const char*)ob + 64);
		const bool mine = h.obj == o0;
		if (mine) shadeObject(P, s, h, a, b);
		rem &= ~ballot(mine);
	}
	s.diff = mk(0, 0, 0); s.spec = mk(0, 0, 0);
	s.li = 0; s.si = 0; s.dsum = 0; s.ssum = 0;
}

struct LightRec { int type; V3 color; float intensity; V3 dir, pos; uint32_t nPoints; const float* points; };

// Runs the lane's castRay state machine until it needs a Render::trace (returns true) or is finished.
// PLAIN (round 6): the kernel was chosen for a scene whose objects are all Diffuse and whose lights are all point / distant lights (rtx_scene_create checks; nothing can change
// either afterwards) -- the mirror / glass recursion with its frame stack, Phong's powf (a real call) and the area-light sums cannot be reached, and are not compiled in: a third of
// the state machine's code and half of the kernel's scratch go with them (pass 1 of the headline -2.7 %: profiles/r06_ab_plain.txt).
#if RTX_ADVANCE_UNIFORM
template <bool PLAIN = false>
__device__ __forceinline__ void advance(const Params& P, Lane& s, uint32_t gl)
{
	const int maxDepth = P.view.maxDepth;
	const float bias = P.view.bias;
	// One loop for the whole wave with a UNIFORM exit: a lane that has reached a waiting state (or ST_DONE) sits out the remaining steps (run = false) instead
	// of leaving the loop.  With per-lane `return`s the compiler kept every field of the lane state twice -- the value a lane left with and the value the others go on
	// changing -- and copied 16-20 registers at the head of every step and at every exit (RTX_ADVANCE_UNIFORM = 0: that form).
	bool run = true;
	do {
		if (!run) continue;
		if (s.state == ST_NEWRAY) {
			RTX_T0
			if (s.sp > maxDepth) { s.col = skyColor(P, s.rd); s.state = ST_RETURN; RTX_ACC(0) continue; }   // scene.cpp:760
			s.qtmax = kFltMax;
			s.state = ST_WAIT_PRIMARY;
			run = false; continue;
		}
		if (s.state == ST_NEXT_LIGHT) {
			RTX_T0
			if (s.li >= P.nLights) { s.state = ST_LIGHTS_DONE; continue; }
			// The light's record comes through the scalar unit in one load, for the lanes that are at the light of the first lane here -- nearly always all of
			// them; the others stay in this state and take their turn in a later step of the loop (the per-lane form was a chain of dependent vector loads --
			// type, then the fields of that type -- and merging it with the scalar form cost 13 copies per step).
			const uint32_t li0 = __builtin_amdgcn_readfirstlane(s.li);
			if (s.li != li0) continue;
			LightRec lr;
			{
				const u32x16 w = sload16(P.lights + li0);
				lr.type = (int)w[0]; lr.color = mk(F(w[1]), F(w[2]), F(w[3])); lr.intensity = F(w[4]);
				lr.dir = mk(F(w[5]), F(w[6]), F(w[7])); lr.pos = mk(F(w[8]), F(w[9]), F(w[10])); lr.nPoints = w[11];
				lr.points = (const float*)(((uint64_t)w[13] << 32) | w[12]);
			}
			const LightRec* l = &lr;
			const int lt = l->type;
			if (PLAIN && lt != 1 && lt != 2) __builtin_unreachable();      // (rtx_scene_create: light types are 1 .. 3, and a PLAIN scene has no area light)
			float dist;
			if (lt == 1) {                 // DistantLight::illuminate, lights.cpp:18-23
				s.L = l->dir;
				s.I = l->color * l->intensity;
				dist = kFltMax;
			}
			else if (lt == 2) {            // PointLight::illuminate, lights.cpp:32-38
				const V3 lp = l->pos;
				V3 L = s.P - lp;
				s.I = l->color * attenuation(l->intensity, len2(L));
				s.L = normalized(L);
				dist = length(s.P - lp);
			}
			else {                         // area light sample loop, scene.cpp:790-806 etc.
				const uint32_t np = l->nPoints;
				if (s.si == 0) {
					const V3 lp = l->pos;
					s.I = l->color * attenuation(l->intensity, len2(s.P - lp));
					s.dsum = 0; s.ssum = 0;
				}
				if (s.si >= np) {
					const float fn = (float)np;
					if (s.mat == 0) s.diff = s.diff + s.I * (s.dsum / fn);                       // scene.cpp:805
					else if (s.mat == 3) {                                                      // scene.cpp:845-846
						s.diff = s.diff + s.I * (s.dsum / fn);
						s.spec = s.spec + s.I * powfRef(s.ssum / fn, s.nSpec);
					}
					else s.spec = s.spec + s.I * powfRef(s.ssum / fn, s.nSpec);                  // scene.cpp:887, 937
					s.li++; s.si = 0;
					continue;
				}
				V3 L = s.P - load3(inGlobal(l->points) + (size_t)s.si * 3);
				dist = length(L);
				s.L = normalized(L);
			}
			// Diffuse (scene.cpp:780-809): the only use of the shadow ray is  vis * max(0, N . -L)  with vis in {0, 1}.  When
			// the max is +0 (surface turned away from the light, or NaN) the product is the same +0 for either answer: the
			// ray cannot influence the pixel ("moot"), and the product kernels do not walk it (castRayWave).
			s.qmoot = s.mat == 0 && fmaxRef(0.f, dot(s.N, -s.L)) == 0.f;
			s.qarea = lt == 3;
			// (a light's source copy was derived for shadow-ray origins within srcNmax |bias| of the surface: a shading normal longer than the host
			// looked at -- a caller's un-normalised tri_nrm, a future object type -- or NaN falls back to copy 0 instead of pruning with too small a sigma)
			s.qsrc = (lt == 2 && s.li < P.nSrcLights && len2(s.N) <= P.srcNmax2) ? 2u + s.li : 0u;
			s.qtmax = dist;               // the ray itself: Ray{P + N*bias, -L, ShadowRay} (scene.cpp:787), built in castRayWave
			s.state = ST_WAIT_SHADOW;
			RTX_ACC(1)
			run = false; continue;
		}
		if (s.state == ST_LIGHTS_DONE) {
			RTX_T0
			const Object* ob = P.objects + s.obj;
			if (PLAIN && s.mat != 0) __builtin_unreachable();
			if (s.mat == 0) { s.col = s.objColor * s.diff; s.state = ST_RETURN; continue; }       // scene.cpp:808
			if (s.mat == 3) {                                                                   // scene.cpp:852
				s.col = s.objColor * ob->ambient + s.diff * ob->diffuse + s.spec * s.specCoef;
				s.state = ST_RETURN; continue;
			}
			if (s.mat == 1) {                                                                   // scene.cpp:856-858, 890
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_REFL);
				frameAt(P, gl, s.sp, 2) = s.spec.x; frameAt(P, gl, s.sp, 3) = s.spec.y; frameAt(P, gl, s.sp, 4) = s.spec.z;
				const V3 nd = s.rd - s.N * (2 * dot(s.rd, s.N));
				s.ro = s.P + s.N * bias; s.rd = nd;
				s.sp++; s.state = ST_NEWRAY; RTX_ACC(2) continue;
			}
			// Transparent, scene.cpp:893-907
			const float ior = ob->ior;
			const float kr = fresnelKr(s.rd, s.N, ior);
			const bool outside = dot(s.rd, s.N) < 0;
			const V3 biasVec = s.N * bias;
			const V3 fd = normalized(reflectDir(s.rd, s.N));
			const V3 fo = outside ? s.P + biasVec : s.P - biasVec;
			frameAt(P, gl, s.sp, 1) = kr;
			frameAt(P, gl, s.sp, 2) = s.spec.x; frameAt(P, gl, s.sp, 3) = s.spec.y; frameAt(P, gl, s.sp, 4) = s.spec.z;
			if (kr < 1) {
				const V3 rd = normalized(refractDir(s.rd, s.N, ior));
				const V3 ro = outside ? s.P - biasVec : s.P + biasVec;
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_TRANS1);
				frameAt(P, gl, s.sp, 8) = fo.x; frameAt(P, gl, s.sp, 9) = fo.y; frameAt(P, gl, s.sp, 10) = fo.z;
				frameAt(P, gl, s.sp, 11) = fd.x; frameAt(P, gl, s.sp, 12) = fd.y; frameAt(P, gl, s.sp, 13) = fd.z;
				s.ro = ro; s.rd = rd;
			}
			else {
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_TRANS2);
				frameAt(P, gl, s.sp, 5) = 0.f; frameAt(P, gl, s.sp, 6) = 0.f; frameAt(P, gl, s.sp, 7) = 0.f;
				s.ro = fo; s.rd = fd;
			}
			s.sp++; s.state = ST_NEWRAY; RTX_ACC(3) continue;
		}
		if (s.state == ST_RETURN) {
			RTX_T0
			if (s.sp == 0) { s.state = ST_DONE; run = false; continue; }
			if (PLAIN) __builtin_unreachable();      // (nothing ever pushed a frame)
			s.sp--;
			// The whole frame is requested at once (14 coalesced loads, one round trip) instead of the kind first and then the fields of
			// that kind: a deep reflect / refract tree is a chain of these, and a small frame lasts as long as its deepest pixel.
			float fr[kFrameFields];
			for (int k = 0; k < kFrameFields; ++k) fr[k] = frameAt(P, gl, s.sp, k);
			asm volatile("" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5]), "+v"(fr[6]));
			asm volatile("" : "+v"(fr[7]), "+v"(fr[8]), "+v"(fr[9]), "+v"(fr[10]), "+v"(fr[11]), "+v"(fr[12]), "+v"(fr[13]));
			const int kind = __float_as_int(fr[0]);
			const V3 spec = mk(fr[2], fr[3], fr[4]);
			if (kind == FR_REFL) { s.col = s.col * 0.8f + spec; RTX_ACC(4) continue; }                      // scene.cpp:858, 890
			const float kr = fr[1];
			if (kind == FR_TRANS1) {                                                             // scene.cpp:896-902
				const V3 acc = mk(0, 0, 0) + s.col * (1 - kr);
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_TRANS2);
				frameAt(P, gl, s.sp, 5) = acc.x; frameAt(P, gl, s.sp, 6) = acc.y; frameAt(P, gl, s.sp, 7) = acc.z;
				s.ro = mk(fr[8], fr[9], fr[10]);
				s.rd = mk(fr[11], fr[12], fr[13]);
				s.sp++; s.state = ST_NEWRAY; RTX_ACC(5) continue;
			}
			V3 acc = mk(fr[5], fr[6], fr[7]);
			acc = acc + s.col * kr;                                                              // scene.cpp:908
			s.col = acc + spec * kr;                                                             // scene.cpp:940
			RTX_ACC(6)
			continue;
		}
		run = false;   // ST_DONE / waiting states
	} while (ballot(run) != 0);
}
#else
template <bool PLAIN = false>
__device__ __forceinline__ void advance(const Params& P, Lane& s, uint32_t gl)
{
	const int maxDepth = P.view.maxDepth;
	const float bias = P.view.bias;
	for (;;) {
		if (s.state == ST_NEWRAY) {
			RTX_T0
			if (s.sp > maxDepth) { s.col = skyColor(P, s.rd); s.state = ST_RETURN; RTX_ACC(0) continue; }   // scene.cpp:760
			s.qtmax = kFltMax;
			s.state = ST_WAIT_PRIMARY;
			return;
		}
		if (s.state == ST_NEXT_LIGHT) {
			RTX_T0
			if (s.li >= P.nLights) { s.state = ST_LIGHTS_DONE; continue; }
			// The light's record: the lanes that are here are nearly always at the same light -- then it comes through the scalar
			// unit in one load (the per-lane form is a chain of dependent vector loads: type, then the fields of that type).
			LightRec lr;
			const uint32_t li0 = __builtin_amdgcn_readfirstlane(s.li);
			if (ballot(s.li != li0) == 0) {
				const u32x16 w = sload16(P.lights + li0);
				lr.type = (int)w[0]; lr.color = mk(F(w[1]), F(w[2]), F(w[3])); lr.intensity = F(w[4]);
				lr.dir = mk(F(w[5]), F(w[6]), F(w[7])); lr.pos = mk(F(w[8]), F(w[9]), F(w[10])); lr.nPoints = w[11];
				lr.points = (const float*)(((uint64_t)w[13] << 32) | w[12]);
			}
			else {
				const Light* lp = P.lights + s.li;
				lr.type = lp->type; lr.color = mk(lp->color[0], lp->color[1], lp->color[2]); lr.intensity = lp->intensity;
				lr.dir = mk(lp->dir[0], lp->dir[1], lp->dir[2]); lr.pos = mk(lp->pos[0], lp->pos[1], lp->pos[2]); lr.nPoints = lp->nPoints; lr.points = lp->points;
			}
			const LightRec* l = &lr;
			const int lt = l->type;
			if (PLAIN && lt == 3) __builtin_unreachable();
			float dist;
			if (lt == 1) {                 // DistantLight::illuminate, lights.cpp:18-23
				s.L = l->dir;
				s.I = l->color * l->intensity;
				dist = kFltMax;
			}
			else if (lt == 2) {            // PointLight::illuminate, lights.cpp:32-38
				const V3 lp = l->pos;
				V3 L = s.P - lp;
				s.I = l->color * attenuation(l->intensity, len2(L));
				s.L = normalized(L);
				dist = length(s.P - lp);
			}
			else {                         // area light sample loop, scene.cpp:790-806 etc.
				const uint32_t np = l->nPoints;
				if (s.si == 0) {
					const V3 lp = l->pos;
					s.I = l->color * attenuation(l->intensity, len2(s.P - lp));
					s.dsum = 0; s.ssum = 0;
				}
				if (s.si >= np) {
					const float fn = (float)np;
					if (s.mat == 0) s.diff = s.diff + s.I * (s.dsum / fn);                       // scene.cpp:805
					else if (s.mat == 3) {                                                      // scene.cpp:845-846
						s.diff = s.diff + s.I * (s.dsum / fn);
						s.spec = s.spec + s.I * powfRef(s.ssum / fn, s.nSpec);
					}
					else s.spec = s.spec + s.I * powfRef(s.ssum / fn, s.nSpec);                  // scene.cpp:887, 937
					s.li++; s.si = 0;
					continue;
				}
				V3 L = s.P - load3(inGlobal(l->points) + (size_t)s.si * 3);
				dist = length(L);
				s.L = normalized(L);
			}
			// Diffuse (scene.cpp:780-809): the only use of the shadow ray is  vis * max(0, N . -L)  with vis in {0, 1}.  When
			// the max is +0 (surface turned away from the light, or NaN) the product is the same +0 for either answer: the
			// ray cannot influence the pixel ("moot"), and the product kernels do not walk it (castRayWave).
			s.qmoot = s.mat == 0 && fmaxRef(0.f, dot(s.N, -s.L)) == 0.f;
			s.qarea = lt == 3;
			// (a light's source copy was derived for shadow-ray origins within srcNmax |bias| of the surface: a shading normal longer than the host
			// looked at -- a caller's un-normalised tri_nrm, a future object type -- or NaN falls back to copy 0 instead of pruning with too small a sigma)
			s.qsrc = (lt == 2 && s.li < P.nSrcLights && len2(s.N) <= P.srcNmax2) ? 2u + s.li : 0u;
			s.qtmax = dist;               // the ray itself: Ray{P + N*bias, -L, ShadowRay} (scene.cpp:787), built in castRayWave
			s.state = ST_WAIT_SHADOW;
			RTX_ACC(1)
			return;
		}
		if (s.state == ST_LIGHTS_DONE) {
			RTX_T0
			const Object* ob = P.objects + s.obj;
			if (PLAIN && s.mat != 0) __builtin_unreachable();
			if (s.mat == 0) { s.col = s.objColor * s.diff; s.state = ST_RETURN; continue; }       // scene.cpp:808
			if (s.mat == 3) {                                                                   // scene.cpp:852
				s.col = s.objColor * ob->ambient + s.diff * ob->diffuse + s.spec * s.specCoef;
				s.state = ST_RETURN; continue;
			}
			if (s.mat == 1) {                                                                   // scene.cpp:856-858, 890
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_REFL);
				frameAt(P, gl, s.sp, 2) = s.spec.x; frameAt(P, gl, s.sp, 3) = s.spec.y; frameAt(P, gl, s.sp, 4) = s.spec.z;
				const V3 nd = s.rd - s.N * (2 * dot(s.rd, s.N));
				s.ro = s.P + s.N * bias; s.rd = nd;
				s.sp++; s.state = ST_NEWRAY; RTX_ACC(2) continue;
			}
			// Transparent, scene.cpp:893-907
			const float ior = ob->ior;
			const float kr = fresnelKr(s.rd, s.N, ior);
			const bool outside = dot(s.rd, s.N) < 0;
			const V3 biasVec = s.N * bias;
			const V3 fd = normalized(reflectDir(s.rd, s.N));
			const V3 fo = outside ? s.P + biasVec : s.P - biasVec;
			frameAt(P, gl, s.sp, 1) = kr;
			frameAt(P, gl, s.sp, 2) = s.spec.x; frameAt(P, gl, s.sp, 3) = s.spec.y; frameAt(P, gl, s.sp, 4) = s.spec.z;
			if (kr < 1) {
				const V3 rd = normalized(refractDir(s.rd, s.N, ior));
				const V3 ro = outside ? s.P - biasVec : s.P + biasVec;
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_TRANS1);
				frameAt(P, gl, s.sp, 8) = fo.x; frameAt(P, gl, s.sp, 9) = fo.y; frameAt(P, gl, s.sp, 10) = fo.z;
				frameAt(P, gl, s.sp, 11) = fd.x; frameAt(P, gl, s.sp, 12) = fd.y; frameAt(P, gl, s.sp, 13) = fd.z;
				s.ro = ro; s.rd = rd;
			}
			else {
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_TRANS2);
				frameAt(P, gl, s.sp, 5) = 0.f; frameAt(P, gl, s.sp, 6) = 0.f; frameAt(P, gl, s.sp, 7) = 0.f;
				s.ro = fo; s.rd = fd;
			}
			s.sp++; s.state = ST_NEWRAY; RTX_ACC(3) continue;
		}
		if (s.state == ST_RETURN) {
			RTX_T0
			if (s.sp == 0) { s.state = ST_DONE; return; }
			if (PLAIN) __builtin_unreachable();      // (nothing ever pushed a frame)
			s.sp--;
			// The whole frame is requested at once (14 coalesced loads, one round trip) instead of the kind first and then the fields of
			// that kind: a deep reflect / refract tree is a chain of these, and a small frame lasts as long as its deepest pixel.
			float fr[kFrameFields];
			for (int k = 0; k < kFrameFields; ++k) fr[k] = frameAt(P, gl, s.sp, k);
			asm volatile("" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5]), "+v"(fr[6]));
			asm volatile("" : "+v"(fr[7]), "+v"(fr[8]), "+v"(fr[9]), "+v"(fr[10]), "+v"(fr[11]), "+v"(fr[12]), "+v"(fr[13]));
			const int kind = __float_as_int(fr[0]);
			const V3 spec = mk(fr[2], fr[3], fr[4]);
			if (kind == FR_REFL) { s.col = s.col * 0.8f + spec; RTX_ACC(4) continue; }                      // scene.cpp:858, 890
			const float kr = fr[1];
			if (kind == FR_TRANS1) {                                                             // scene.cpp:896-902
				const V3 acc = mk(0, 0, 0) + s.col * (1 - kr);
				frameAt(P, gl, s.sp, 0) = __int_as_float(FR_TRANS2);
				frameAt(P, gl, s.sp, 5) = acc.x; frameAt(P, gl, s.sp, 6) = acc.y; frameAt(P, gl, s.sp, 7) = acc.z;
				s.ro = mk(fr[8], fr[9], fr[10]);
				s.rd = mk(fr[11], fr[12], fr[13]);
				s.sp++; s.state = ST_NEWRAY; RTX_ACC(5) continue;
			}
			V3 acc = mk(fr[5], fr[6], fr[7]);
			acc = acc + s.col * kr;                                                              // scene.cpp:908
			s.col = acc + spec * kr;                                                             // scene.cpp:940
			RTX_ACC(6)
			continue;
		}
		return;   // ST_DONE / waiting states
	}
}
#endif

// Consumes a finished Render::trace for this lane.
template <bool PLAIN = false>
__device__ __forceinline__ void consume(const Params& P, Lane& s, const Hit& h)
{
	if (s.state == ST_WAIT_PRIMARY) {
		RTX_T0
		if (h.obj < 0) { s.col = skyColor(P, s.rd); s.state = ST_RETURN; RTX_ACC(7) return; }             // scene.cpp:945
		shadePrimary(P, s, h);
		s.state = ST_NEXT_LIGHT;
		RTX_ACC(8)
		return;
	}
	if (s.state == ST_WAIT_SHADOW) {
		RTX_T0
		const float vis = (h.obj < 0) ? 1.0f : 0.0f;       // bool vis = !trace(...)
		if (PLAIN && (s.mat != 0 || s.qarea)) __builtin_unreachable();
		const bool area = s.qarea;
		const V3 nL = -s.L;
		if (s.mat == 0) {
			const float c = vis * fmaxRef(0.f, dot(s.N, nL));
			if (!area) s.diff = s.diff + s.I * c;                                              // scene.cpp:788
			else s.dsum += c;                                                                  // scene.cpp:803
		}
		else {
			const V3 R = reflectDir(s.L, s.N);
			const float sd = fmaxRef(0.f, dot(R, -s.rd));
			if (!area) {
				const V3 vI = s.I * vis;
				if (s.mat == 3) s.diff = s.diff + vI * fmaxRef(0.f, dot(s.N, nL));             // scene.cpp:820
				s.spec = s.spec + vI * powfRef(sd, s.nSpec);                                   // scene.cpp:824,867,917
			}
			else {
				if (s.mat == 3) s.dsum += vis * fmaxRef(0.f, dot(s.N, nL));                    // scene.cpp:841
				s.ssum += vis * sd;                                                            // scene.cpp:843,885,935
			}
		}
		if (area) s.si++; else s.li++;
		s.state = ST_NEXT_LIGHT;
		RTX_ACC(9)
	}
}

__device__ __forceinline__ void primaryRay(const Params& P, float x, float y, V3& o, V3& d)
{
	// scene.cpp:453-457 (getPixels adds another 0.5) and Camera::getRay, scene.cpp:52-53
	const float w = (float)P.view.width, hgt = (float)P.view.height;
	const float xp = (2 * (x + 0.5f) / w - 1) * P.view.scale * P.view.aspect;
	const float yp = -(2 * (y + 0.5f) / hgt - 1) * P.view.scale;
	const V3 s = normalized(mk(xp, yp, -1));
	const float* M = P.view.camM;
	V3 r;
	r.x = s.x * M[0] + s.y * M[4] + s.z * M[8] + M[12];
	r.y = s.x * M[1] + s.y * M[5] + s.z * M[9] + M[13];
	r.z = s.x * M[2] + s.y * M[6] + s.z * M[10] + M[14];
	const float ww = s.x * M[3] + s.y * M[7] + s.z * M[11] + M[15];
	if (ww != 0.0f && ww != 1.0f) { const float wi = 1.0f / ww; r.x *= wi; r.y *= wi; r.z *= wi; }   // geometry.h:300-305
	o = mk(P.view.camPos[0], P.view.camPos[1], P.view.camPos[2]);
	d = r;
}

// The shading state of a lane that nothing reads while its ray is traced (the hit being shaded, the light, the sums):
// parked in LDS around every trace so that the walk has the registers.  Left to the compiler it goes to scratch, i.e.
// through L2 into HBM and back (round 1 / 2: 5.9 GB of writes per launch for a 0.2-GB framebuffer); the kernels use
// 5-10 KB of the 32-40 KB of LDS their occupancy leaves a block.
constexpr bool kParkColor = kWideSlots < 16;      // (sixteen-slot nodes: the stack takes the 3 KB of objColor's three fields)
constexpr int kParkFields = 25 - (kParkColor ? 0 : 3);
constexpr int kPk = kParkColor ? 0 : -3;          // index shift of the fields behind objColor      // (26 with nSpec until the eight-slot walk's stack needed the kilobyte: five blocks per CU hold 31 744 B each)
// (PLAIN kernels: no specular sum and no specular coefficient to park -- four kilobytes less, which with the pow table they do not touch is what a SIXTH block per CU needs)
template <bool PLAIN> struct ParkLayout { static constexpr int kFields = kParkFields - (PLAIN ? 4 : 0), kSp = PLAIN ? -3 : 0; };
template <bool PLAIN>
__device__ __forceinline__ float (*parkArea())[256]
{
	__shared__ float area[ParkLayout<PLAIN>::kFields][256];
	return area;
}
// Five blocks per CU hold 31 744 B of LDS each (160 KB / 5, rounded down to the allocation granule): one array over the edge and the pass-1 kernel silently
// runs four blocks per CU (-3 ... -20 %, DESIGN_HISTORY.md 3.1e).  sobelStage belongs to the frame kernel only (four blocks per CU: 40 960 B each).
static_assert(sizeof(powTab) + sizeof(leafBatch) + sizeof(wideStack) + sizeof(pruneUni) + ParkLayout<false>::kFields * 1024 <= (RTX_WAVES >= 5 ? 31744 : 40960),
              "the ray kernels' LDS no longer fits the blocks per CU that RTX_WAVES asks for");
static_assert(RTX_WAVES_PLAIN < 6 || sizeof(leafBatch) + sizeof(wideStack) + sizeof(pruneUni) + ParkLayout<true>::kFields * 1024 <= 27136,
              "the PLAIN kernels' LDS no longer fits six blocks per CU (160 KB / 6, rounded down to the allocation granule)");

// The kernel's argument block, read afresh from the kernarg segment.  Every ray kernel takes `const Params P` as its only
// argument, so the segment starts with it.  The empty asm makes the pointer opaque: fields read through it are loaded where
// they are used (s_load from the kernarg segment: scalar memory, no VALU slot) instead of being kept in SGPRs across the
// walk -- where the register allocator parks them in lanes of a VGPR (v_writelane / v_readlane: ~100 VALU instructions per
// trace round of a kernel that is bound by VALU issue).
__device__ __forceinline__ const Params& freshParams(const Params& P)
{
	uint64_t p = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(p));
	return *(const Params*)(const RTX_AS4 Params*)p;      // (the loads are still selected as s_load: the address space is inferred back)
}

// CAM: the rays handed in start at the camera (o == view.camPos bit for bit: pass 1, SSAA, the frame kernel -- not the probe rays)
template <bool STATS, bool MESH = true, bool FEWRAYS = false, bool BOXES = true, bool CAM = true, int CULLK = -1, bool PLAIN = false>
__device__ __forceinline__ V3 castRayWave(const Params& P0, bool valid, V3 o, V3 d, uint32_t gl, Counts& cnt)
{
	const Params& P = freshParams(P0);
	Lane s;
	s.state = valid ? ST_NEWRAY : ST_DONE;
	s.sp = 0; s.ro = o; s.rd = d; s.col = mk(0, 0, 0);
	s.obj = 0; s.mat = 0; s.li = 0; s.si = 0;
	s.P = s.N = s.objColor = s.diff = s.spec = s.L = s.I = mk(0, 0, 0);
	s.specCoef = s.nSpec = s.dsum = s.ssum = 0;
	s.qtmax = kFltMax; s.qmoot = false; s.qarea = false; s.qsrc = 0;
	RTX_DBG_ONLY(unsigned long long dbgRounds = 0, dbgTrace = 0, dbgState = 0;)
#if RTX_ONE_ADVANCE
	// (one copy of advance() in the kernel: at the head of the round instead of before the loop and at its end -- the same sequence of steps)
	for (;;) {
		advance<PLAIN>(freshParams(P0), s, gl);
		if (ballot(s.state != ST_DONE) == 0) break;
#else
	advance<PLAIN>(P, s, gl);
	while (ballot(s.state != ST_DONE) != 0) {
#endif
		Hit h;
		RTX_DBG_ONLY(const unsigned long long dbgT0 = __builtin_readcyclecounter();)
		// moot shadow rays (see advance): only the instrumented variant walks them -- the reference's statistics count
		// them; here the lane just sits the trace out and consumes "not occluded", which gives the same +0 product
		const bool moot = s.state == ST_WAIT_SHADOW && s.qmoot;
		if (STATS) cnt.moot += __popcll(ballot(moot));
		// the pending request of every lane: a shadow ray from the point being shaded, or the lane's current ray
		const bool qshadow = s.state == ST_WAIT_SHADOW;
		const V3 qo = qshadow ? s.P + s.N * P.view.bias : s.ro, qd = qshadow ? -s.L : s.rd;
		const bool qactive = s.state != ST_DONE && (STATS || !moot);
		const float qtmax = s.qtmax;
		// (a ray of recursion depth 0 is the one handed in; reflected / refracted rays pass through no known point)
		const uint32_t qsrc = qshadow ? s.qsrc : ((CAM && s.sp == 0) ? 1u : 0u);
		if (MESH) {
			const uint32_t t = threadIdx.x;
			float (*parkedState)[256] = parkArea<PLAIN>();
			constexpr int kSp = ParkLayout<PLAIN>::kSp;
			parkedState[0][t] = s.P.x; parkedState[1][t] = s.P.y; parkedState[2][t] = s.P.z;
			parkedState[3][t] = s.N.x; parkedState[4][t] = s.N.y; parkedState[5][t] = s.N.z;
			if (kParkColor) { parkedState[6][t] = s.objColor.x; parkedState[7][t] = s.objColor.y; parkedState[8][t] = s.objColor.z; }
			parkedState[9 + kPk][t] = s.diff.x; parkedState[10 + kPk][t] = s.diff.y; parkedState[11 + kPk][t] = s.diff.z;
			if (!PLAIN) { parkedState[12 + kPk][t] = s.spec.x; parkedState[13 + kPk][t] = s.spec.y; parkedState[14 + kPk][t] = s.spec.z; }
			parkedState[15 + kPk + kSp][t] = s.L.x; parkedState[16 + kPk + kSp][t] = s.L.y; parkedState[17 + kPk + kSp][t] = s.L.z;
			parkedState[18 + kPk + kSp][t] = s.I.x; parkedState[19 + kPk + kSp][t] = s.I.y; parkedState[20 + kPk + kSp][t] = s.I.z;
			parkedState[21 + kPk + kSp][t] = s.rd.x; parkedState[22 + kPk + kSp][t] = s.rd.y; parkedState[23 + kPk + kSp][t] = s.rd.z;
			if (!PLAIN) parkedState[24 + kPk][t] = s.specCoef;
			asm volatile("" ::: "memory");
		}
		traceWave<STATS, MESH, FEWRAYS, BOXES, CULLK>(freshParams(P0), qactive, qshadow, qo, qd, qtmax, h, cnt, qsrc);
		if (MESH) {
			asm volatile("" ::: "memory");
			const uint32_t t = threadIdx.x;
			float (*parkedState)[256] = parkArea<PLAIN>();
			constexpr int kSp = ParkLayout<PLAIN>::kSp;
			s.P = mk(parkedState[0][t], parkedState[1][t], parkedState[2][t]);
			s.N = mk(parkedState[3][t], parkedState[4][t], parkedState[5][t]);
			if (kParkColor) s.objColor = mk(parkedState[6][t], parkedState[7][t], parkedState[8][t]);
			s.diff = mk(parkedState[9 + kPk][t], parkedState[10 + kPk][t], parkedState[11 + kPk][t]);
			if (!PLAIN) s.spec = mk(parkedState[12 + kPk][t], parkedState[13 + kPk][t], parkedState[14 + kPk][t]);
			s.L = mk(parkedState[15 + kPk + kSp][t], parkedState[16 + kPk + kSp][t], parkedState[17 + kPk + kSp][t]);
			s.I = mk(parkedState[18 + kPk + kSp][t], parkedState[19 + kPk + kSp][t], parkedState[20 + kPk + kSp][t]);
			s.rd = mk(parkedState[21 + kPk + kSp][t], parkedState[22 + kPk + kSp][t], parkedState[23 + kPk + kSp][t]);
			if (!PLAIN) s.specCoef = parkedState[24 + kPk][t];
		}
		RTX_DBG_ONLY(const unsigned long long dbgT1 = __builtin_readcyclecounter();)
		if (s.state != ST_DONE) {
			const Params& Pa = freshParams(P0);
			consume<PLAIN>(Pa, s, h);
#if !RTX_ONE_ADVANCE
			advance<PLAIN>(Pa, s, gl);
#endif
		}
		RTX_DBG_ONLY(dbgRounds++; dbgTrace += dbgT1 - dbgT0; dbgState += __builtin_readcyclecounter() - dbgT1;)
	}
	RTX_DBG_ONLY(
	// trace rounds of one work item, and where its cycles went (s_memtime): slowest item and sums
	if (__lane_id() == 0) {
		const unsigned long long before = atomicMax(&gDbgHist[8], dbgTrace + dbgState);
		if (dbgTrace + dbgState > before) { gDbgHist[9] = dbgRounds; gDbgHist[10] = dbgTrace; gDbgHist[11] = dbgState; }
		atomicAdd(&gDbgHist[12], dbgRounds); atomicAdd(&gDbgHist[13], dbgTrace); atomicAdd(&gDbgHist[14], dbgState); atomicAdd(&gDbgHist[15], 1ull);
	}
	)
	return s.col;
}

// Row ownership for the pixel-sharded multi-GPU path (bands of bandH rows dealt round-robin to nParts devices).
__device__ __forceinline__ bool rowOwned(uint32_t bandH, uint32_t nParts, uint32_t part, uint32_t y)
{
	return bandH == 0 || (y / bandH) % nParts == part;
}
// Pass 1 also renders `halo` rows either side of every owned band so the Sobel mask of the owned rows can be
// computed without exchanging pass-1 results between devices (scene.cpp:554-568 reads a 3x3 neighbourhood).
__device__ __forceinline__ bool rowRendered(const Params& P, uint32_t y)
{
	if (rowOwned(P.bandH, P.nParts, P.part, y)) return true;
	if (!P.halo) return false;
	return (y > 0 && rowOwned(P.bandH, P.nParts, P.part, y - 1)) || rowOwned(P.bandH, P.nParts, P.part, y + 1);
}

__device__ __forceinline__ uint32_t nextWork(uint32_t* counter)
{
	uint32_t w = 0;
	if (__lane_id() == 0) w = atomicAdd(counter, 1u);
	return __builtin_amdgcn_readfirstlane(w);
}

__device__ __forceinline__ void flushCounts(const Params& P, const Counts& c)
{
	if (__lane_id() == 0) {
		atomicAdd(P.counters + 0, c.rays);
		atomicAdd(P.counters + 1, c.box);
		atomicAdd(P.counters + 2, c.tri);
		atomicAdd(P.counters + 5, c.wNodes); atomicAdd(P.counters + 6, c.wTri); atomicAdd(P.counters + 7, c.wS2);
		atomicAdd(P.counters + 8, c.wS3); atomicAdd(P.counters + 9, c.wS4);
		atomicAdd(P.counters + 10, c.wLeaves); atomicAdd(P.counters + 11, c.wLeafSkips);
		atomicAdd(P.counters + 12, c.wChunks); atomicAdd(P.counters + 13, c.wChunkSkips); atomicAdd(P.counters + 14, c.triLanes);
		atomicAdd(P.counters + 15, c.moot);
		RTX_DBG_ONLY(
		atomicAdd(&gDbgHist[32], c.cNodes); atomicAdd(&gDbgHist[33], c.cFilter); atomicAdd(&gDbgHist[34], c.cExact);
		atomicAdd(&gDbgHist[40], c.aWalks); atomicAdd(&gDbgHist[41], c.sWalks); atomicAdd(&gDbgHist[42], c.sVisits); atomicAdd(&gDbgHist[43], c.sLeaves); atomicAdd(&gDbgHist[44], c.sPasses); atomicAdd(&gDbgHist[45], c.sExact);
		)
	}
}

} // namespace

// ------------------------------------------------------------------------------------------------
// Pass 1: Scene::renderWorker over 8x8 pixel tiles (scene.cpp:444-468)
// ------------------------------------------------------------------------------------------------
// BOXES = false: the variant for scenes whose meshes all have triangles too large for the box test of the prune records to prune
// anything (rtxd::Object::pruneBoxes; cfg4): the same kernel without that test (the plane test stays).
template <bool STATS, bool MESH = true, bool BOXES = true, int CULLK = (STATS || !MESH) ? -1 : 1, bool PLAIN = false>
__global__ void __launch_bounds__(256, MESH ? (PLAIN ? RTX_WAVES_PLAIN : RTX_WAVES) : RTX_WAVES_ANALYTIC) rtxPass1Kernel(const Params P)
{
	if (!PLAIN) fillPowTab();      // (Phong's powf: not reachable in a PLAIN kernel, whose LDS then does not hold the table)
	const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = __lane_id();
	const uint32_t W = P.view.width, H = P.view.height;
	Counts cnt = {};
	// XCD-affine work distribution.  Each XCD has its own 4 MB L2; if consecutive tiles went to different XCDs
	// (one global queue) every L2 would have to hold the triangles of the whole sweep.  Instead the frame is cut
	// into bands of 8 tile rows (64 pixel rows), band b belongs to queue b % 8, and a wave first drains the queue
	// of the XCD it runs on (s_getreg XCC_ID), then helps the others (work stealing, for load balance only --
	// any wave may render any tile).  The queues are explicit tile lists built by the host (rtx_api.hip,
	// buildTileList): tiles that can see a mesh come first, so the tail of the launch consists of cheap tiles.
	const uint32_t xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;   // HW_REG_XCC_ID[3:0]
	RTX_TRACE_ONLY(
	const unsigned long long dbgStart = wall_clock64();
	unsigned long long dbgEnd = dbgStart, dbgBusy = 0;
	)
	for (uint32_t attempt = 0; attempt < 8; attempt = uni(attempt + 1)) {
		const uint32_t q = (xcd + attempt) & 7u;
		const uint32_t qBase = sload1(P.tileList + q), qSize = sload1(P.tileList + 8 + q);
		// The queues are ordered heaviest-first: from the middle of a queue on the tiles are the cheap ones (sky, floor: 10-20 us), where the
		// returning atomic and the dependent read of the list are a sizeable part of a tile -- there a wave takes RTX_POP_MANY tiles per atomic.
		uint32_t take = 1, jEnd = 0, j = 0;
		for (;;) {
			if (j >= jEnd) {
				uint32_t w = 0;
				if (__lane_id() == 0) w = atomicAdd(P.workCounter + q * 16, take);
				j = __builtin_amdgcn_readfirstlane(w);
				if (j >= qSize) break;
				jEnd = j + take < qSize ? j + take : qSize;
				if (kPopDen * j >= kPopNum * qSize) take = RTX_POP_MANY;
			}
			const uint32_t tile = sload1(P.tileList + qBase + j);
			j = uni(j + 1);
			RTX_DBG_ONLY(if (P.pad3 != 0 && tile != P.pad3 - 1) continue;)   // RTX_DBG_TILE=tx,ty: only this tile (counters of one work item)
			// a tile (ty << 16 | tx), or a 64 x 1 strip of a halo row (0x10000000 | strip << 16 | y: rtx_api.hip, buildTileList),
			// which is accounted to the first of the eight tiles it runs through
#if RTX_TILE_RECOMPUTE
			// What the tile's pixels are is derived from the list entry twice -- before the rays are cast and again afterwards, from fresh copies of the parameters --
			// instead of being kept (x, y and a dozen scalars: spilled) across castRayWave.
			uint32_t tx, ty, x, y; bool strip, valid;
			auto place = [&](const Params& Pt, uint32_t ln) {
				strip = (tile & Pt.stripBit) != 0;      // (stripBit = 0 when the list holds no strips: tile rows from 4096 on use bit 28 themselves)
				tx = strip ? ((tile >> 16) & 0xfffu) * 8 : tile & 0xffffu; ty = strip ? (tile & 0x7fffu) >> 3 : tile >> 16;
				x = strip ? tx * 8 + ln : tx * 8 + (ln & 7); y = strip ? tile & 0x7fffu : ty * 8 + (ln >> 3);
				// x1/y1 are clamped to W-1/H-1: the last column and row are never rendered (scene.cpp:369-372)
				valid = x < Pt.view.width - 1 && y < Pt.view.height - 1 && y >= Pt.rowBegin && y < Pt.rowEnd && rowRendered(Pt, y);
			};
			const Params& Pa = freshParams(P);
			place(Pa, laneNow());
			if (ballot(valid) == 0) continue;
			V3 o, d;
			primaryRay(Pa, (float)x + 0.5f, (float)y + 0.5f, o, d);
			if (take > 1) __builtin_amdgcn_s_setprio(0);      // (the cheap part of the queue: no look at the tile's cost)
			else
			if (sload1(Pa.tileCost + ty * Pa.tilesXFull + tx) > RTX_PRIO_TICKS) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
			const unsigned long long t0 = wall_clock64();
			const V3 c = castRayWave<STATS, MESH, false, BOXES, true, CULLK, PLAIN>(Pa, valid, o, d, gl, cnt);
			const unsigned long long dt = wall_clock64() - t0;
			RTX_TRACE_ONLY(dbgEnd = t0 + dt; dbgBusy += dt;)
			uint32_t tileWas = tile;
			asm volatile("" : "+s"(tileWas));      // (opaque: the expressions below are evaluated again, not carried over)
			const Params& Pb = freshParams(P);
			{
				const uint32_t ln = laneNow();
				const bool strip2 = (tileWas & Pb.stripBit) != 0;
				const uint32_t tx2 = strip2 ? ((tileWas >> 16) & 0xfffu) * 8 : tileWas & 0xffffu, ty2 = strip2 ? (tileWas & 0x7fffu) >> 3 : tileWas >> 16;
				const uint32_t x2 = strip2 ? tx2 * 8 + ln : tx2 * 8 + (ln & 7), y2 = strip2 ? tileWas & 0x7fffu : ty2 * 8 + (ln >> 3);
				const bool valid2 = x2 < Pb.view.width - 1 && y2 < Pb.view.height - 1 && y2 >= Pb.rowBegin && y2 < Pb.rowEnd && rowRendered(Pb, y2);
				if (ln == 0) {
					// remembered per tile: the SSAA pass starts with the tiles that were expensive here (longest job first)
					if (!strip2) Pb.tileCost[ty2 * Pb.tilesXFull + tx2] = dt > 0xffffffffull ? 0xffffffffu : (uint32_t)dt;
					if (STATS) { atomicMax(Pb.counters + 3, dt); atomicAdd(Pb.counters + 4, dt); }
				}
				if (strip2 && ln < 8 && tx2 + ln < Pb.tilesX) Pb.tileCost[ty2 * Pb.tilesXFull + tx2 + ln] = (uint32_t)((dt > 0xffffffffull ? 0xffffffffull : dt) / 8);
				if (valid2) {
					float* px = Pb.fb + ((size_t)y2 * Pb.view.width + x2) * 3;
					px[0] = c.x; px[1] = c.y; px[2] = c.z;
				}
			}
		}
	}
#else
			const bool strip = (tile & P.stripBit) != 0;      // (stripBit = 0 when the list holds no strips: tile rows from 4096 on use bit 28 themselves)
			const uint32_t tx = strip ? ((tile >> 16) & 0xfffu) * 8 : tile & 0xffffu, ty = strip ? (tile & 0x7fffu) >> 3 : tile >> 16;
			const uint32_t x = strip ? tx * 8 + lane : tx * 8 + (lane & 7), y = strip ? tile & 0x7fffu : ty * 8 + (lane >> 3);
			// x1/y1 are clamped to W-1/H-1: the last column and row are never rendered (scene.cpp:369-372)
			const bool valid = x < W - 1 && y < H - 1 && y >= P.rowBegin && y < P.rowEnd && rowRendered(P, y);
			if (ballot(valid) == 0) continue;
			V3 o, d;
			primaryRay(P, (float)x + 0.5f, (float)y + 0.5f, o, d);
			// a tile that was slow in the previous launch of this view (a pole, a silhouette) runs at raised priority: the
			// launch ends when its slowest wave does, and such a wave otherwise gets one issue slot in RTX_WAVES
			if (take > 1) __builtin_amdgcn_s_setprio(0);      // (the cheap part of the queue: no look at the tile's cost)
			else
			if (sload1(P.tileCost + ty * P.tilesXFull + tx) > RTX_PRIO_TICKS) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
			const unsigned long long t0 = wall_clock64();
			const V3 c = castRayWave<STATS, MESH, false, BOXES, true, CULLK, PLAIN>(P, valid, o, d, gl, cnt);
			const unsigned long long dt = wall_clock64() - t0;
			RTX_TRACE_ONLY(dbgEnd = t0 + dt; dbgBusy += dt;)
			if (lane == 0) {
				// remembered per tile: the SSAA pass starts with the tiles that were expensive here (longest job first)
				if (!strip) P.tileCost[ty * P.tilesXFull + tx] = dt > 0xffffffffull ? 0xffffffffu : (uint32_t)dt;
				if (STATS) { atomicMax(P.counters + 3, dt); atomicAdd(P.counters + 4, dt); }
			}
			if (strip && lane < 8 && tx + lane < P.tilesX) P.tileCost[ty * P.tilesXFull + tx + lane] = (uint32_t)((dt > 0xffffffffull ? 0xffffffffull : dt) / 8);
			if (valid) {
				float* px = P.fb + ((size_t)y * W + x) * 3;
				px[0] = c.x; px[1] = c.y; px[2] = c.z;
			}
		}
	}
#endif
	RTX_TRACE_ONLY(if (lane == 0 && (gl >> 6) < 16384) { gDbgWave[3 * (gl >> 6)] = dbgStart; gDbgWave[3 * (gl >> 6) + 1] = dbgEnd; gDbgWave[3 * (gl >> 6) + 2] = dbgBusy; })
	if (STATS || RTX_DBG) flushCounts(P, cnt);
}

// ------------------------------------------------------------------------------------------------
// Pass 2: SSAAworker (scene.cpp:523-537).  Work item = one 8x8 pixel tile; the flagged pixels of the tile are
// re-rendered 16 at a time, 4 lanes (= the 4 sub-samples) per pixel, so the 64 rays of a trace stay coherent.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t nthSetBit(uint64_t m, uint32_t n)   // position of the n-th (0-based) set bit
{
	uint32_t pos = 0;
	uint32_t lo = (uint32_t)m, c = __popc(lo);
	if (n >= c) { pos = 32; n -= c; lo = (uint32_t)(m >> 32); }
	c = __popc(lo & 0xffffu); if (n >= c) { pos += 16; n -= c; lo >>= 16; }
	c = __popc(lo & 0xffu); if (n >= c) { pos += 8; n -= c; lo >>= 8; }
	c = __popc(lo & 0xfu); if (n >= c) { pos += 4; n -= c; lo >>= 4; }
	c = __popc(lo & 0x3u); if (n >= c) { pos += 2; n -= c; lo >>= 2; }
	if (n >= (lo & 1u)) pos += 1;
	return pos;
}

template <bool STATS, bool MESH = true, bool BOXES = true, int CULLK = (STATS || !MESH) ? -1 : 1, bool PLAIN = false>
__global__ void __launch_bounds__(256, MESH ? RTX_WAVES_SSAA : RTX_WAVES_ANALYTIC) rtxSsaaKernel(const Params P)
{
	if (!PLAIN) fillPowTab();      // (Phong's powf: not reachable in a PLAIN kernel, whose LDS then does not hold the table)
	const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = __lane_id();
	const uint32_t W = P.view.width, H = P.view.height;
	Counts cnt = {};
	RTX_DBG_ONLY(uint32_t dbgItems = 0;)
	for (;;) {
		// work item = 16 consecutive entries of the flagged-pixel list (rtxSsaaCountKernel / rtxSsaaScatterKernel):
		// full waves even where a tile has only a few flagged pixels; the pixels of tiles that were expensive in pass 1
		// come first (longest-job-first), tile by tile, so the rays of a wave stay close together
		const uint32_t work = nextWork(P.workCounter);
		const uint32_t total = sload1(P.ssaaScan + 2 * (size_t)P.nTiles);
		const uint32_t first = work * 16;
		if (first >= total) break;
		// the items of the tiles that were slow in pass 1 come first in the list: they run at raised priority
		if (first < sload1(P.ssaaScan + P.nTiles)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
		const uint32_t g = first + (lane >> 2), sub = lane & 3;
		const uint32_t pxy = P.ssaaPixels[g < total ? g : first];
		const bool valid = g < total && pxy != 0xffffffffu;
		const uint32_t x = pxy & 0xffffu, y = pxy >> 16;
		// offsets in the reference's order: (.25,.25) (.25,.75) (.75,.25) (.75,.75)  (scene.cpp:527-534)
		const float fx = (float)x + ((sub & 2) ? 0.75f : 0.25f), fy = (float)y + ((sub & 1) ? 0.75f : 0.25f);
		V3 o, d;
		primaryRay(P, fx, fy, o, d);
		const unsigned long long t0 = wall_clock64();
		const V3 c = castRayWave<STATS, MESH, true, BOXES, true, CULLK, PLAIN>(P, valid, o, d, gl, cnt);      // (work items of 16, 4 or 1 pixels: see the exact tests of meshWalk)
		if (!STATS) {
			// what the item cost, as the time of a 16-pixel item (a 4-pixel item takes at least a quarter of it), kept per tile
			// in the second half of tileCost: a profiling aid (rtx_tile_cost_read, tools/ssaa_items.py).  Ordering and sizing
			// the next frame's items by it was measured and lost to the pass-1 costs that do it now (DESIGN_HISTORY.md 6c)
			const uint32_t npx = (uint32_t)__popcll(ballot(valid)) >> 2;
			const unsigned long long dt16 = (wall_clock64() - t0) * (npx <= 4u ? 4u : (npx <= 8u ? 2u : 1u));
			if (lane == 0 && pxy != 0xffffffffu) atomicMax(P.tileCost + P.nTiles + (y >> 3) * P.tilesXFull + (x >> 3), (uint32_t)(dt16 > 0xffffffffull ? 0xffffffffull : dt16));
		}
		RTX_DBG_ONLY(
		if (!STATS && lane == 0 && (gl >> 6) < 8192 && dbgItems < 160) {      // (tools/ssaa_timeline.py)
			const size_t e = (size_t)(gl >> 6) * 160 + dbgItems;
			gDbgTimeline[3 * e] = t0; gDbgTimeline[3 * e + 1] = wall_clock64() - t0; gDbgTimeline[3 * e + 2] = 2ull << 32 | pxy;
		}
		dbgItems++;
		)
		if (STATS && lane == 0) { atomicMax(P.counters + 3, wall_clock64() - t0); atomicAdd(P.counters + 4, wall_clock64() - t0); }
		// color = 0; color += c0; += c1; += c2; += c3; fb = color / 4
		const int base = (int)(lane & ~3u);
		V3 sum = mk(0, 0, 0);
		for (int k = 0; k < 4; ++k)
			sum = sum + mk(__shfl(c.x, base + k), __shfl(c.y, base + k), __shfl(c.z, base + k));
		if (valid && sub == 0) {
			float* px = P.fb + ((size_t)y * W + x) * 3;
			px[0] = sum.x / 4; px[1] = sum.y / 4; px[2] = sum.z / 4;
		}
	}
	if (STATS || RTX_DBG) flushCounts(P, cnt);
}

// Re-orders the eight pass-1 queues by the cost the tiles had in the previous launch of the same view (heaviest first:
// longest-job-first keeps the end of the launch free of long tiles).  One block per queue: histogram of the cost
// classes (log2 of the ticks), offsets in descending class order, scatter.  The order inside a class is arbitrary --
// the picture does not depend on the order tiles are rendered in.
// rtx_render_frame, once per tile list: which tiles are listed (mark), then for every listed tile the number of listed
// tiles in its 3x3 neighbourhood, itself included (need; 0 = not listed), the number of listed tiles by index % 64
// (expect[k]) and how many of those are not zero (expect[64]; the last block to finish counts them).
// The eight pass-1 queues of a view, written on the device from a per-tile-row PLAN (rtx_api.hip, buildTileList: what a tile row is listed as, its
// queue and where its entries start -- O(rows) on the host, read here straight from pinned host memory): words [0, 16) of the plan are the list's
// header, then four words per tile row -- kind (0 not listed, 1 tiles, 2 the 64 x 1 strips of one pixel row; bit 8: the row crosses the meshes' tile
// rectangle), ty or y, first entry of the row's tiles inside the rectangle, first entry of those outside (the queues list the tiles that can see
// a mesh first).  in0 / in1: the columns (tiles, or strips for kind 2: sIn0 / sIn1) inside the rectangle.  One thread per tile or strip.
__global__ void __launch_bounds__(256) rtxTileListKernel(const uint32_t* __restrict__ plan, uint32_t tilesX, uint32_t nStrips, uint32_t in0, uint32_t in1,
                                                         uint32_t sIn0, uint32_t sIn1, uint32_t* __restrict__ list)
{
	const uint32_t row = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
	if (row == 0 && c < 16) list[c] = plan[c];
	const uint32_t kind = plan[16 + 4 * row], val = plan[17 + 4 * row], base0 = plan[18 + 4 * row], base1 = plan[19 + 4 * row];
	const bool inY = (kind & 0x100u) != 0;
	if ((kind & 0xffu) == 1u) {
		if (c >= tilesX) return;
		const uint32_t nIn = inY && in1 > in0 ? in1 - in0 : 0u;
		const bool in = nIn != 0 && c >= in0 && c < in1;
		list[in ? base0 + (c - in0) : base1 + (nIn != 0 && c >= in1 ? c - nIn : c)] = val << 16 | c;
	}
	else if ((kind & 0xffu) == 2u) {
		if (c >= nStrips) return;
		const uint32_t nIn = inY && sIn1 > sIn0 ? sIn1 - sIn0 : 0u;
		const bool in = nIn != 0 && c >= sIn0 && c < sIn1;
		list[in ? base0 + (c - sIn0) : base1 + (nIn != 0 && c >= sIn1 ? c - nIn : c)] = 0x10000000u | c << 16 | val;
	}
}

__global__ void __launch_bounds__(256) rtxTileMarkKernel(const uint32_t* __restrict__ list, uint32_t tilesXFull, uint8_t* __restrict__ mark)
{
	const uint32_t q = blockIdx.y, base = list[q], n = list[8 + q];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t t = list[base + i];
		mark[(t >> 16) * tilesXFull + (t & 0xffffu)] = 1;
	}
}

__global__ void __launch_bounds__(256) rtxTileNeedKernel(const uint8_t* __restrict__ mark, uint32_t tilesXFull, uint32_t tilesYFull,
                                                         uint8_t* __restrict__ need, uint32_t* __restrict__ expect, uint32_t* __restrict__ blocksDone)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < tilesXFull * tilesYFull) {
		uint32_t n = 0;
		if (mark[t]) {
			const int ty = (int)(t / tilesXFull), tx = (int)(t - (uint32_t)ty * tilesXFull);
			for (int dy = -1; dy <= 1; ++dy)
				for (int dx = -1; dx <= 1; ++dx) {
					const int x = tx + dx, y = ty + dy;
					if (x >= 0 && y >= 0 && x < (int)tilesXFull && y < (int)tilesYFull) n += mark[(uint32_t)y * tilesXFull + (uint32_t)x];
				}
			atomicAdd(expect + (t & 63u), 1u);
		}
		need[t] = (uint8_t)n;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		if (atomicAdd(blocksDone, 1u) + 1 == gridDim.x) {
			uint32_t groups = 0;
			for (int k = 0; k < 64; ++k) groups += atomicAdd(expect + k, 0u) != 0;
			expect[64] = groups;
		}
	}
}

// rtx_render_frame, one launch: everything the frame kernel expects to find zeroed (the per-tile counters, its control
// block, the tile-queue heads, the cost sum) and, for a small frame, the mask -- five memsets were five launch gaps.
__global__ void __launch_bounds__(256) rtxFrameClearKernel(uint32_t* __restrict__ work, uint32_t* __restrict__ deps, size_t depWords,
                                                           uint32_t* __restrict__ ctl, uint32_t ctlWords, uint32_t errWord, uint8_t* __restrict__ mask, size_t maskBytes)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
	if (t < 2) work[16 + t] = 0;
	if (t < 128) work[128 + t] = 0;
	// (an error word left by an earlier frame is kept for rtx_frame_status: work[24])
	for (size_t i = t; i < ctlWords; i += stride) { if (i == errWord && ctl[i]) work[24] = ctl[i]; ctl[i] = 0; }
	for (size_t i = t; i < depWords; i += stride) deps[i] = 0;
	for (size_t i = t; i < maskBytes; i += stride) mask[i] = 0;
}

// First-frame cost estimate (the reference renders ONE frame per process, main.cpp:15): what a tile will cost is not known
// before it has been rendered once, so the first frame could neither start with its slow tiles nor split them.  A tile's
// cost is mostly the leaves its rays' lines run through and the references in them (objects.cpp:587-631): every leaf's box
// is projected through the camera and its reference count added to the cells (2 x 2 tiles) its bounding rectangle touches
// (the TRUE box of the leaf's triangles: the cells of the reference's builder are several times larger).
// rtxCostFillKernel turns cells into ticks per tile (coefficients fitted to measured tile costs: tools/cost_fit.py).
// The estimate only orders and splits work; no pixel depends on it.  Measured costs replace it tile by tile.
// blockIdx.y = 0: the camera's splat; y >= 1: the shadow of the leaves cast by light / plane pair y - 1 (below).  One launch for all of them (they only add to the
// grid), and one 64-bit atomic per cell: (references, leaves) are neighbouring words.
struct SplatSources { uint32_t n; int32_t kind[16]; uint32_t weight[16]; float l[16][3]; float p[16][6]; };      // weight: shadow rays per shaded point towards this light (an area light's n_points)      // kind 1: distant light (l = the direction it travels in), 2: point light (l = position); p = plane (position, normal)
__global__ void __launch_bounds__(256) rtxCostSplatKernel(const float* __restrict__ boxes, uint32_t nLeaves, const View view, uint32_t gridW, uint32_t gridH,
                                                          uint32_t* __restrict__ grid, const SplatSources src)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nLeaves) return;
	const float* b = boxes + (size_t)i * 8;      // true box of the leaf's triangles (lo, hi), reference count
	const uint32_t sIdx = blockIdx.y;
	const uint32_t wgt = sIdx != 0 ? src.weight[sIdx - 1] : 1u;
	const uint32_t n = (uint32_t)b[6] * wgt;
	const float* M = view.camM;
	float x0 = 1e30f, x1 = -1e30f, y0 = 1e30f, y1 = -1e30f;
	for (int c = 0; c < 8; ++c) {
		float cx = b[(c & 1) ? 3 : 0], cy = b[(c & 2) ? 4 : 1], cz = b[(c & 4) ? 5 : 2];
		if (sIdx != 0) {
			// The lights' side of the estimate (round 5): a tile of a PLANE that lies in the shadow of a leaf -- as seen from a point or distant light -- sends its
			// shadow rays through that leaf (scene.cpp:787), and where they graze the mesh's silhouette such floor tiles cost as much as mesh tiles (up to 0.5 ms at
			// the headline) while the camera's splat gives them 0: they ran last, four per atomic, and were the first frame's tail.  The corner is projected from the
			// light onto the plane (beyond the corner, or the leaf casts no bounded shadow on it) and from there through the camera.
			const uint32_t k = sIdx - 1;
			const bool point = src.kind[k] == 2;
			const float dx = point ? cx - src.l[k][0] : src.l[k][0], dy = point ? cy - src.l[k][1] : src.l[k][1], dz = point ? cz - src.l[k][2] : src.l[k][2];
			const float nx = src.p[k][3], ny = src.p[k][4], nz = src.p[k][5];
			const float den = dx * nx + dy * ny + dz * nz;
			if (!(fabsf(den) > 1e-6f)) return;
			const float t = ((src.p[k][0] - cx) * nx + (src.p[k][1] - cy) * ny + (src.p[k][2] - cz) * nz) / den;
			if (!(t > 0.0f) || !(t < 1e4f)) return;
			cx += t * dx; cy += t * dy; cz += t * dz;
		}
		const float px = cx - view.camPos[0], py = cy - view.camPos[1], pz = cz - view.camPos[2];
		// camera space: the inverse of primaryRay()'s rotation (orthonormal rMatrix, scene.cpp:22-49)
		const float sx = px * M[0] + py * M[1] + pz * M[2], sy = px * M[4] + py * M[5] + pz * M[6], sz = px * M[8] + py * M[9] + pz * M[10];
		if (!(sz < -1e-4f)) return;                        // reaches behind the camera: no estimate from this leaf
		const float xp = sx / -sz, yp = sy / -sz;
		const float fx = (xp / (view.scale * view.aspect) + 1.0f) * 0.5f * (float)view.width - 1.0f, fy = (-yp / view.scale + 1.0f) * 0.5f * (float)view.height - 1.0f;
		x0 = fminf(x0, fx); x1 = fmaxf(x1, fx); y0 = fminf(y0, fy); y1 = fmaxf(y1, fy);
	}
	if (!(x1 >= 0.0f && y1 >= 0.0f && x0 < (float)view.width && y0 < (float)view.height)) return;
	const int cx0 = max(0, (int)floorf(x0 / 16.0f)), cx1 = min((int)gridW - 1, (int)floorf(x1 / 16.0f));
	const int cy0 = max(0, (int)floorf(y0 / 16.0f)), cy1 = min((int)gridH - 1, (int)floorf(y1 / 16.0f));
	if ((long long)(cx1 - cx0 + 1) * (cy1 - cy0 + 1) > 4096) return;      // (a leaf that fills the screen says nothing about where the work is)
	for (int cy = cy0; cy <= cy1; ++cy)
		for (int cx = cx0; cx <= cx1; ++cx)
			atomicAdd((unsigned long long*)(grid + 2 * ((size_t)cy * gridW + cx)), ((unsigned long long)wgt << 32) | n);      // (+ n references, + 1 leaf -- times the shadow rays per point)
}

// farPlanes (up to four planes: position, normal) + farTicks: a tile through which the camera sees a plane FAR away (the horizon of a floor) shades points
// thousands of units from the meshes; their shadow rays come back through the scene as one huge, badly conditioned bundle that no record can prune (the
// reference's own rounding error grows with |orig - v0|: the rigorous margins are metres wide) -- 1.5 ms for such a tile at 8192^2, and with an estimate of 0
// they ran last, four per atomic: the first frame of a view took 13.7 ms of pass 1 against 9.9.  They are given farTicks, i.e. they start first.
struct FarPlanes { float p[4][6]; uint32_t n; float farDist; uint32_t farTicks; };
__global__ void __launch_bounds__(256) rtxCostFillKernel(const uint32_t* __restrict__ grid, uint32_t gridW, uint32_t tilesXFull, uint32_t tilesYFull,
                                                         uint32_t* __restrict__ tileCost, float perRef, float perLeaf, float base, const View view, const FarPlanes far)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= tilesXFull * tilesYFull) return;
	const uint32_t ty = t / tilesXFull, tx = t - ty * tilesXFull;
	const size_t cell = (size_t)(ty / 2) * gridW + tx / 2;
	const float refs = (float)grid[2 * cell], leaves = (float)grid[2 * cell + 1];
	uint32_t c = leaves > 0 ? (uint32_t)fminf(base + perRef * refs + perLeaf * leaves, 4.0e9f) : 0u;
	if (far.n) {
		// the rays through the tile's top and bottom rows (primaryRay's formula; no need for its bits here)
		const float* M = view.camM;
		for (int k = 0; k < 2; ++k) {
			const float x = (float)(tx * 8 + 4) + 0.5f, y = (float)(ty * 8 + (k ? 7 : 0)) + 0.5f;
			const float xp = (2 * (x + 0.5f) / (float)view.width - 1) * view.scale * view.aspect, yp = -(2 * (y + 0.5f) / (float)view.height - 1) * view.scale;
			const float il = rsqrtf(xp * xp + yp * yp + 1.0f);
			const float sx = xp * il, sy = yp * il, sz = -il;
			const float dx = sx * M[0] + sy * M[4] + sz * M[8], dy = sx * M[1] + sy * M[5] + sz * M[9], dz = sx * M[2] + sy * M[6] + sz * M[10];
			for (uint32_t q = 0; q < far.n; ++q) {
				const float den = dx * far.p[q][3] + dy * far.p[q][4] + dz * far.p[q][5];
				if (!(fabsf(den) > 1e-12f)) continue;
				const float tt = ((far.p[q][0] - view.camPos[0]) * far.p[q][3] + (far.p[q][1] - view.camPos[1]) * far.p[q][4] + (far.p[q][2] - view.camPos[2]) * far.p[q][5]) / den;
				if (tt > far.farDist) c = max(c, far.farTicks);
			}
		}
	}
	tileCost[t] = c;
}

// klass != null (rtx_render_frame): the class of a tile is the highest one within two tiles of it (rtxTileClassKernel) --
// the SSAA items of a slow tile can only be queued once the 5 x 5 tiles around it have been rendered.
__global__ void __launch_bounds__(256) rtxTileClassKernel(const uint32_t* __restrict__ cost, uint32_t tilesXFull, uint32_t tilesYFull,
                                                          uint8_t* __restrict__ klass, unsigned long long* __restrict__ costSum)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	{
		// sum of the tile costs = wave-ticks of pass 1 in the previous frame (rtxTileOrderKernel: which tiles to split)
		unsigned long long mine = t < tilesXFull * tilesYFull ? cost[t] : 0;
		for (int sh = 32; sh > 0; sh >>= 1) mine += __shfl_xor(mine, sh);
		if ((threadIdx.x & 63) == 0 && mine) atomicAdd(costSum, mine);
	}
	if (t >= tilesXFull * tilesYFull) return;
	const int ty = (int)(t / tilesXFull), tx = (int)(t - (uint32_t)ty * tilesXFull);
	uint32_t c = 0;
	for (int dy = -2; dy <= 2; ++dy)
		for (int dx = -2; dx <= 2; ++dx) {
			const int x = tx + dx, y = ty + dy;
			if (x >= 0 && y >= 0 && x < (int)tilesXFull && y < (int)tilesYFull) c = max(c, cost[(uint32_t)y * tilesXFull + (uint32_t)x]);
		}
	klass[t] = (uint8_t)(c ? 31u - (uint32_t)__builtin_clz(c) : 0u);
}

// splitPercent != 0 (rtx_render_frame): a slow tile is listed as four 4x4-pixel quarters (tile | 0x8000 | part << 13) or
// sixteen 2x2-pixel parts (| 0x80000000, the upper two bits of the part in bits 29-30), rendered by as many waves --
// the frame cannot end before its slowest tile has been rendered AND re-sampled, and a wave's time on a tile is mostly
// the walk of its ray bundle, which a smaller part shortens.  The output queue q starts at 16 + 16 (base_q - 16).
// Splitting multiplies the work on a tile (each part walks its own bundle), so it is for the tiles that decide when
// the frame ends: those that took a good part of the time pass 1 would need if its work were spread evenly over the
// waves (cost sum / waves).  In a small frame that is every tile -- and there are idle waves to take the parts.
// thresholds[0..1] = the two limits (ticks), also used for the size of the SSAA items (rtxFrameKernel).
template <bool PLACE>
__global__ void __launch_bounds__(256) rtxTileOrderKernel(uint32_t* __restrict__ global, const uint32_t* __restrict__ list, const uint32_t* __restrict__ cost,
                                                           uint32_t tilesXFull, uint32_t* __restrict__ out, const uint8_t* __restrict__ klassIn = nullptr,
                                                           const unsigned long long* __restrict__ costSum = nullptr, uint32_t nWaves = 1,
                                                           uint32_t splitPercent = 0, uint32_t splitFloor = 0, uint32_t* __restrict__ thresholds = nullptr,
                                                           uint32_t tilesX = 0, uint32_t stripLimit = 0xffffffffu, uint32_t* __restrict__ heads = nullptr,
                                                           uint32_t stripBit = 0)      // 0x10000000 when the list may hold strips (buildTileList), else 0
{
	// (the eight queue heads of the launch that follows, 64 bytes apart: saves a memset per frame)
	if (PLACE && heads && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 128) heads[threadIdx.x] = 0;
	uint32_t split4 = 0xffffffffu, split16 = 0xffffffffu;
	if (costSum && splitPercent) {
		const unsigned long long even = *costSum / nWaves * splitPercent / 100;
		split4 = even > 0x0fffffffull ? 0x0fffffffu : (uint32_t)even;
		if (split4 < splitFloor) split4 = splitFloor;
		split16 = split4 * 4;
	}
	// Two launches over 32 blocks per queue (one block per queue took 70 us: 32 dependent rounds of loads per thread), each
	// block a contiguous piece of the queue: PLACE = false counts the entries of every class of the piece into
	// global[(q * 32 + block) * 32 + class]; PLACE = true places the piece's entries of a class behind those of the higher
	// classes of the queue and of the same class in the pieces before it -- the list order (= neighbouring tiles, which share
	// their nodes in L2) survives within a class, and nothing depends on the order blocks run in.
	__shared__ uint32_t hist[32], cursor[32];
	const uint32_t q = blockIdx.y;
	const uint32_t base = list[q], n = list[8 + q];
	const uint32_t obase = 16 + 16 * (base - 16);
	if (threadIdx.x < 32) hist[threadIdx.x] = 0;
	if (thresholds && q == 0 && blockIdx.x == 0 && threadIdx.x == 0) { thresholds[0] = split4; thresholds[1] = split16; }
	__syncthreads();
	auto entry = [&](uint32_t tile) {      // (a strip: the first tile it runs through)
		return (tile & stripBit) ? ((tile & 0x7fffu) >> 3) * tilesXFull + ((tile >> 16) & 0xfffu) * 8 : (tile >> 16) * tilesXFull + (tile & 0xffffu);
	};
	auto klass = [&](uint32_t tile) {
		if (klassIn) return (uint32_t)klassIn[entry(tile)];
		uint32_t c = cost[entry(tile)];
		if (tile & stripBit) c = c > 0x1fffffffu ? 0xffffffffu : c * 8;      // (a strip's time is spread over eight entries)
		return c ? 31u - (uint32_t)__builtin_clz(c) : 0u;
	};
	// A strip that was slow (it runs through dense geometry: its wide bundle is halved again and again, and ONE wave does
	// all of it) goes back to being up to eight tiles for eight waves.  Its cost is spread over the entries of those tiles
	// (rtxPass1Kernel), so the sum is the strip's time in either form.
	auto parts = [&](uint32_t tile) {
		if (tile & stripBit) {
			const uint32_t tx0 = ((tile >> 16) & 0xfffu) * 8, n = tilesX - tx0 < 8u ? tilesX - tx0 : 8u;
			unsigned long long sum = 0;
			for (uint32_t e = 0; e < n; ++e) sum += cost[entry(tile) + e];
			return sum > stripLimit ? n : 1u;
		}
		const uint32_t c = cost[entry(tile)];
		return c > split16 ? 16u : (c > split4 ? 4u : 1u);
	};
	const uint32_t piece = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * piece, i1 = i0 + piece < n ? i0 + piece : n;
	if (!PLACE) {
		for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) { const uint32_t tile = list[base + i]; atomicAdd(&hist[klass(tile)], parts(tile)); }
		__syncthreads();
		if (threadIdx.x < 32) global[(q * 32 + blockIdx.x) * 32 + threadIdx.x] = hist[threadIdx.x];
		return;
	}
	if (gridDim.x == 1) {
		// a short queue: one block does both steps in one launch
		for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) { const uint32_t tile = list[base + i]; atomicAdd(&hist[klass(tile)], parts(tile)); }
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t run = 0;
			for (int k = 31; k >= 0; k--) { cursor[k] = run; run += hist[k]; }
			out[q] = obase; out[8 + q] = run;
		}
	}
	else {
		// the counts of the queue's pieces (up to 32 x 32 words) through LDS: read once by the whole block -- a thread that
		// summed a thousand of them straight from memory made this launch last 70 us
		__shared__ uint32_t counts[32 * 32], classTotal[32];
		for (uint32_t w = threadIdx.x; w < gridDim.x * 32; w += blockDim.x) counts[w] = global[q * 32 * 32 + w];
		__syncthreads();
		if (threadIdx.x < 32) {
			const uint32_t k = threadIdx.x;
			uint32_t all = 0, before = 0;
			for (uint32_t b = 0; b < gridDim.x; ++b) { const uint32_t c = counts[b * 32 + k]; all += c; if (b < blockIdx.x) before += c; }
			classTotal[k] = all;
			cursor[k] = before;      // (+ the higher classes of the whole queue, below)
		}
		__syncthreads();
		if (threadIdx.x < 32) {
			const uint32_t k = threadIdx.x;
			uint32_t higher = 0;
			for (uint32_t kk = k + 1; kk < 32; ++kk) higher += classTotal[kk];
			cursor[k] += higher;
			if (blockIdx.x == 0 && k == 0) { out[q] = obase; out[8 + q] = higher + classTotal[0]; }
		}
	}
	__syncthreads();
	for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
		const uint32_t tile = list[base + i], np = parts(tile);
		const uint32_t at = obase + atomicAdd(&cursor[klass(tile)], np);
		if (np == 1) out[at] = tile;
		else if (tile & stripBit) for (uint32_t e = 0; e < np; ++e) out[at + e] = ((tile & 0x7fffu) >> 3) << 16 | (((tile >> 16) & 0xfffu) * 8 + e);
		else if (np == 4) for (uint32_t e = 0; e < 4; ++e) out[at + e] = tile | 0x8000u | e << 13;
		else for (uint32_t e = 0; e < 16; ++e) out[at + e] = tile | 0x80008000u | (e & 3u) << 13 | (e >> 2) << 29;
	}
}

// SSAA work list, step 1: one thread per 8x8 tile counts the tile's flagged pixels (of the rows this launch re-renders)
// into the heavy or the normal half of `scan` (heavy = pass 1 spent more than `heavyTicks` on the tile).  After an
// exclusive scan over the 2 nTiles + 1 entries, step 2 writes the pixels to their slots.
__device__ __forceinline__ uint64_t ssaaFlagged(const Params& P, uint32_t tx, uint32_t ty)
{
	const uint32_t W = P.view.width, H = P.view.height;
	uint64_t m = 0;
	for (uint32_t r = 0; r < 8; ++r) {
		const uint32_t y = ty * 8 + r;
		// the workers only visit x < W-1, y < H-1 (scene.cpp:369-372, 523-525)
		if (!(y < H - 1 && y >= P.rowBegin && y < P.rowEnd && rowOwned(P.bandH, P.nParts, P.part, y))) continue;
		for (uint32_t c = 0; c < 8; ++c) {
			const uint32_t x = tx * 8 + c;
			if (x < W - 1 && P.ssaaMask[(size_t)y * W + x] != 0) m |= 1ull << (r * 8 + c);
		}
	}
	return m;
}

// mode[0] != 0 ("local"): every tile's pixels are padded to a multiple of 16 slots, so a wave never mixes tiles.
// Chosen on the device (decide != 0: from the total of the previous, unpadded count): with few flagged pixels the
// launch is bounded by its slowest wave and coherent, tile-local waves are shorter; with many, full waves win.
__global__ void __launch_bounds__(256) rtxSsaaCountKernel(const Params P, uint32_t* __restrict__ scan, uint32_t* __restrict__ mode,
                                                          uint32_t heavyTicks, uint32_t decide, uint32_t localBelow, uint32_t spreadSlots, uint32_t sparseBelow = 0)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t > 2 * P.nTiles) return;
	// mode[1] = number of flagged pixels found by the previous (unpadded) count
	// decide: 0 = the plain count (for the decision), 1 = decide from it, 2 = the layout decided for an earlier frame of this view
	const uint32_t local = decide == 2 ? mode[0] : (decide ? (mode[1] < localBelow ? 1u : 0u) : 0u);
	// mode[3] ("sparse"): fewer 16-pixel items than half the waves of the launch (a device's share of a frame, a small frame): every wave
	// gets at most one item anyway and the launch lasts as long as its slowest item, so the items are cut one step finer -- 4 pixels for
	// the tiles that were slow in pass 1, ONE pixel for the very slow ones.
	const uint32_t sparse = decide == 2 ? mode[3] : (decide ? (mode[1] < sparseBelow ? 1u : 0u) : 0u);
	if (t >= P.nTiles) { if (t == 2 * P.nTiles) { scan[t] = 0; if (decide == 1) { mode[0] = local; mode[3] = sparse; } } return; }
	const uint32_t ty = t / P.tilesXFull, tx = t - ty * P.tilesXFull;
	uint32_t nf = (uint32_t)__popcll(ssaaFlagged(P, tx, ty));
	// local mode: a wave holds 16 pixels of one tile -- or only 4 of a tile that was VERY slow in pass 1 (a silhouette),
	// because the launch lasts as long as its slowest wave.  The 4-pixel layout needs up to 4x the slots; the list holds
	// spreadSlots extra ones (mode[2] = handed out so far), a tile that does not get its share is packed normally.  The
	// scatter kernel recognises the layout from the tile's slot count.
	if (local) {
		const uint32_t cost = P.tileCost[t];
		uint32_t per = 16u;
		if (cost > RTX_SSAA_VERY * heavyTicks) per = sparse ? 1u : RTX_SSAA_SPREAD_PX;
		else if (sparse && cost > heavyTicks) per = RTX_SSAA_SPREAD_PX;
		const uint32_t packed = (nf + 15u) & ~15u, spread = ((nf + per - 1u) / per) * 16u;
		nf = packed;
		if (per < 16u && spread > packed && atomicAdd(mode + 2, spread - packed) + (spread - packed) <= spreadSlots) nf = spread;
	}
	const bool heavy = P.tileCost[t] > heavyTicks;
	scan[t] = heavy ? nf : 0u;
	scan[P.nTiles + t] = heavy ? 0u : nf;
}

__global__ void __launch_bounds__(256) rtxSsaaScatterKernel(const Params P, const uint32_t* __restrict__ scan, uint32_t* __restrict__ mode,
                                                            uint32_t* __restrict__ pixels, uint32_t heavyTicks)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	// (instead of two memsets per frame: the queue head of the SSAA launch that follows, and the slot budget the NEXT count
	// kernel hands out -- nothing uses either while this kernel runs; both start at zero, rtx_api.hip ensureWork)
	if (t == 0) { P.workCounter[0] = 0; mode[2] = 0; }
	if (t >= P.nTiles) return;
	const uint32_t ty = t / P.tilesXFull, tx = t - ty * P.tilesXFull;
	uint64_t m = ssaaFlagged(P, tx, ty);
	const uint32_t idx = P.tileCost[t] > heavyTicks ? t : P.nTiles + t;
	P.tileCost[P.nTiles + t] = 0;      // (the SSAA item costs of this frame are collected from here on: rtxSsaaKernel)
	uint32_t slot = scan[idx];
	const uint32_t slots = scan[idx + 1] - slot, nf = (uint32_t)__popcll(m);      // (the other half's entry of a tile is 0 slots wide)
	const bool spread = mode[0] && slots > ((nf + 15u) & ~15u);      // 4 pixels (or one: the count kernel's "sparse" layout) per group of 16 slots
	const uint32_t per = !spread ? 16u : (slots == ((nf + RTX_SSAA_SPREAD_PX - 1u) / RTX_SSAA_SPREAD_PX) * 16u ? RTX_SSAA_SPREAD_PX : 1u);
	uint32_t n = 0, px = 0;
	while (m) {
		const uint32_t pos = (uint32_t)__builtin_ctzll(m);
		m &= m - 1;
		pixels[slot + n++] = (tx * 8 + (pos & 7)) | (ty * 8 + (pos >> 3)) << 16;
		if (spread && ++px % per == 0) for (uint32_t k = 0; k < 16u - per; k++) pixels[slot + n++] = 0xffffffffu;
	}
	if (mode[0]) for (; n < slots; n++) pixels[slot + n] = 0xffffffffu;
}

// ------------------------------------------------------------------------------------------------
// Probe rays: Render::trace + Render::castRay for arbitrary rays (64 per wave), used by the parity tests
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rtxProbeKernel(const Params P)
{
	fillPowTab();
	const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = __lane_id();
	const uint32_t nWork = (P.nProbe + 63) / 64;
	Counts cnt = {};
	for (;;) {
		const uint32_t work = nextWork(P.workCounter);
		if (work >= nWork) break;
		const uint32_t i = work * 64 + lane;
		const bool valid = i < P.nProbe;
		V3 o = mk(0, 0, 0), d = mk(0, 0, -1);
		if (valid) { o = load3(P.probeRays + (size_t)i * 6); d = load3(P.probeRays + (size_t)i * 6 + 3); }
		Hit h;
		traceWave<false>(P, valid, false, o, d, kFltMax, h, cnt);
		if (valid) {
			float* out = P.probeHits + (size_t)i * 8;
			const bool hit = h.obj >= 0;
			const bool mesh = hit && P.objects[hit ? h.obj : 0].type == 3;
			out[0] = hit ? 1.f : 0.f; out[1] = hit ? (float)h.obj : -1.f; out[2] = mesh ? (float)h.tri : -1.f;
			out[3] = h.t; out[4] = hit ? h.u : -1.f; out[5] = hit ? h.v : -1.f; out[6] = 0; out[7] = 0;
		}
		const V3 c = castRayWave<false, true, false, true, false>(P, valid, o, d, gl, cnt);
		if (valid) { float* pc = P.probeColours + (size_t)i * 3; pc[0] = c.x; pc[1] = c.y; pc[2] = c.z; }
	}
}

// ------------------------------------------------------------------------------------------------
// Sobel mask (scene.cpp:547-568)
// ------------------------------------------------------------------------------------------------
// The operator on one interior pixel; fetch(a, b) returns the pixel at (x - 1 + b, y - 1 + a).  One body for both
// kernels that compute the mask, so the two cannot differ in a rounding.
template <typename Fetch>
__device__ __forceinline__ bool sobelFlag(Fetch fetch)
{
	V3 gx = mk(0, 0, 0), gy = mk(0, 0, 0);
	const float op[3][3] = { { -1, 0, 1 }, { -2, 0, 2 }, { -1, 0, 1 } };
#pragma unroll
	for (int a = 0; a < 3; ++a)
#pragma unroll
		for (int b = 0; b < 3; ++b) {
			const V3 p = fetch(a, b);
			gx = gx + p * op[a][b];
			gy = gy + p * op[b][a];
		}
	const float lx = length(gx), ly = length(gy);
	const float val = __builtin_sqrtf(lx * lx + ly * ly);       // powf(.,2) == x*x (g++ folds it)
	return val > 0.5f;
}

constexpr uint32_t kSobelRows = 8;
__global__ void __launch_bounds__(256) rtxSobelKernel(const float* __restrict__ fb, uint8_t* __restrict__ mask,
                                                      uint32_t W, uint32_t H, uint32_t rowBegin, uint32_t rowEnd,
                                                      uint32_t bandH, uint32_t nParts, uint32_t part)
{
	// A wave = 62 columns (its 64 lanes hold the columns x0 - 1 .. x0 + 62; the outer two only feed their neighbours) x
	// kSobelRows rows, walked downwards with the last three rows of pixels in registers: every pixel is loaded ONCE per wave
	// (one 12-byte load per lane and row) and reaches the lanes either side through the crossbar -- instead of nine
	// pixels = 27 scalar loads per output pixel.  HBM-bound: the framebuffer is read once.
	const uint32_t lane = threadIdx.x & 63;
	const int x = (int)(blockIdx.x * 62 + lane) - 1;
	const uint32_t y0 = rowBegin + (blockIdx.y * 4 + (threadIdx.x >> 6)) * kSobelRows;
	if (y0 >= rowEnd || y0 >= H) return;
	// (a wave none of whose rows this device owns has nothing to write: one part of an 8-way sharded frame spent as long
	// here as the whole frame does, reading rows it then did not flag)
	bool anyOwned = false;
	for (uint32_t r = 0; r < kSobelRows; ++r) anyOwned = anyOwned || rowOwned(bandH, nParts, part, y0 + r);
	if (!anyOwned) return;
	const bool xin = x >= 0 && x < (int)W;
	const bool interiorX = x >= 1 && x + 1 < (int)W && lane >= 1 && lane <= 62;
	// (all the rows requested before the first is used: walked one load at a time a wave waited for 18 memory round trips)
	V3 own[kSobelRows + 2];
#pragma unroll
	for (uint32_t r = 0; r < kSobelRows + 2; ++r) {
		const int y = (int)(y0 + r) - 1;
		own[r] = mk(0, 0, 0);
		if (xin && y >= 0 && y < (int)H && y <= (int)rowEnd) own[r] = load3(fb + ((size_t)y * W + (size_t)x) * 3);
	}
	V3 rows[3][3];      // [row above, row, row below][left, centre, right]
	auto spread = [&](const V3& c, V3* out) {
		out[1] = c;
		out[0] = mk(__shfl_up(c.x, 1), __shfl_up(c.y, 1), __shfl_up(c.z, 1));
		out[2] = mk(__shfl_down(c.x, 1), __shfl_down(c.y, 1), __shfl_down(c.z, 1));
	};
	spread(own[0], rows[0]);
	spread(own[1], rows[1]);
#pragma unroll
	for (uint32_t r = 0; r < kSobelRows; ++r) {
		const uint32_t y = y0 + r;
		if (y >= rowEnd || y >= H) break;
		spread(own[r + 2], rows[2]);
		bool flag = false;
		if (interiorX && y >= 1 && y + 1 < H) flag = sobelFlag([&](int a, int b) { return rows[a][b]; });
		// borders are defined as 0 (reference: uninitialised)
		if (xin && lane >= 1 && lane <= 62 && rowOwned(bandH, nParts, part, y)) mask[(size_t)y * W + (size_t)x] = flag ? 1 : 0;
		for (int b = 0; b < 3; ++b) { rows[0][b] = rows[1][b]; rows[1][b] = rows[2][b]; }
	}
}

// ------------------------------------------------------------------------------------------------
// The whole frame in one launch: pass 1, Sobel mask and SSAA (scene.cpp:444-568) driven by per-tile dependencies.
//
// Three separate launches cost the frame its two tails: every launch lasts as long as its slowest wave, and the SSAA
// launch consists of little else (its slow items are the silhouette and pole tiles that were slow in pass 1).  Here a
// wave that finishes a pass-1 tile counts it in at the 3x3 tiles around it; whoever completes a tile's neighbourhood
// computes that tile's Sobel flags, counts THAT in at the 3x3 around, and whoever completes the second count queues the
// tile's SSAA items (16 flagged pixels x 4 sub-samples, or 4 x 4 for a tile that was very slow).  SSAA may overwrite a
// pixel only when no Sobel will read its pass-1 value any more -- that is the second count.  Waves take SSAA items
// before pass-1 tiles, and pass-1 tiles are ordered by the cost of their surroundings in the previous frame, so the
// slow SSAA items start while most of pass 1 is still to be done.
//
// Coherence between XCDs (each has its own L2): everything one wave writes for another to read -- pass-1 pixels, tile
// costs, flags, queue entries -- is stored with agent-scope atomic stores (write-through, sc1), read with agent-scope
// atomic loads, and the writer waits for its stores to be acknowledged (s_waitcnt vmcnt(0)) before the atomic that
// publishes them.  No cache-wide write-back or invalidate is involved.
// The picture does not depend on any of the scheduling: every pixel is a pure function of the scene.
// ------------------------------------------------------------------------------------------------
#define RTX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void storesAcknowledged() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__shared__ float sobelStage[4][304];      // per wave: the 10 x 10 pixels around a tile

// Sobel flags of tile (nx, ny); writes the tile's part of the mask.  All the listed tiles around it are complete.
__device__ __forceinline__ uint64_t frameSobelTile(const Params& P, uint32_t nx, uint32_t ny, uint32_t lane)
{
	const uint32_t W = P.view.width, H = P.view.height;
	float* stage = sobelStage[threadIdx.x >> 6];
	for (uint32_t i = lane; i < 300; i += 64) {
		const uint32_t r = i / 30, c3 = i - r * 30, c = c3 / 3, ch = c3 - c * 3;
		// pixels outside the image are never used (only interior pixels get a flag): any valid address will do
		int yy = (int)(ny * 8 + r) - 1, xx = (int)(nx * 8 + c) - 1;
		yy = yy < 0 ? 0 : (yy > (int)H - 1 ? (int)H - 1 : yy);
		xx = xx < 0 ? 0 : (xx > (int)W - 1 ? (int)W - 1 : xx);
		stage[i] = __hip_atomic_load(P.fb + ((size_t)yy * W + xx) * 3 + ch, RTX_AGENT);
	}
	const uint32_t cx = lane & 7, cy = lane >> 3;
	const uint32_t x = nx * 8 + cx, y = ny * 8 + cy;
	bool flag = false;
	const bool inImage = x < W && y < P.rowEnd && y < H && y >= P.rowBegin && rowOwned(P.bandH, P.nParts, P.part, y);
	if (inImage && x >= 1 && x + 1 < W && y >= 1 && y + 1 < H)
		flag = sobelFlag([&](int a, int b) { return load3(stage + (cy + a) * 30 + (cx + b) * 3); });
	if (inImage) P.maskOut[(size_t)y * W + x] = flag ? 1 : 0;
	return ballot(flag);
}

// Counts tile (tx, ty) in at the listed tiles of its 3x3 neighbourhood; lanes whose neighbour is thereby complete
// return true and its index in nIdx.
__device__ __forceinline__ bool frameCountIn(const Params& P, uint32_t* counters, uint32_t tx, uint32_t ty, uint32_t lane, uint32_t& nIdx)
{
	bool mine = false;
	nIdx = 0;
	if (lane < 9) {
		const uint32_t nx = tx + lane % 3 - 1, ny = ty + lane / 3 - 1;      // (unsigned wrap: -1 is out of range)
		if (nx < P.tilesXFull && ny < P.tilesYFull) {
			nIdx = ny * P.tilesXFull + nx;
			const uint32_t need = P.tileNeed[nIdx];
			if (need) mine = atomicAdd(counters + nIdx, 1u) + 1 == need;
		}
	}
	return mine;
}

// Control block (P.frameCtl, 32-bit words, every item on its own 64-byte line: one address takes about 30 M atomic
// operations or coherent reads per second, whoever issues them):
enum : uint32_t {
	FC_QUEUE = 0,                 // + 16 q, q < 64: head, tail of SSAA item queue q
	FC_COUNT = 16 * 64,           // + 16 k, k < 64: tiles with index % 64 == k whose SSAA items are queued
	FC_GROUPS = 16 * 128,         // number of k whose count is complete
	FC_ERROR = 16 * 129,
	FC_VERY = 16 * 130,           // extra items handed to tiles that get 4-pixel items
	FC_ALL = 16 * 131,            // + 16 r, r < 64: copies of "every tile has queued its items"
	FC_WORDS = 16 * 195
};
#define RTX_FRAME_QUEUES 64u

template <bool MESH, bool BOXES = true, int CULLK = MESH ? 1 : -1, bool PLAIN = false>
__global__ void __launch_bounds__(256, MESH ? RTX_WAVES_FRAME : RTX_WAVES_ANALYTIC) rtxFrameKernel(const Params P)
{
	if (!PLAIN) fillPowTab();      // (Phong's powf: not reachable in a PLAIN kernel, whose LDS then does not hold the table)
	const uint32_t gl = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = __lane_id();
	const uint32_t W = P.view.width, H = P.view.height;
	Counts cnt = {};
	const uint32_t xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;   // HW_REG_XCC_ID[3:0]
	uint32_t* const ctl = (uint32_t*)P.frameCtl;
	const uint32_t wave = gl >> 6;
	uint32_t attempt = 0;                       // pass-1 queues found empty so far (see rtxPass1Kernel)
	uint32_t rot = wave * 5u;                   // where this wave's next SSAA item goes
	RTX_DBG_ONLY(uint32_t dbgItems = 0;)
	const unsigned long long started = wall_clock64();
	// a wave never waits for another one except on a queue entry that is being written; the watchdog (10 s) only turns a
	// bug into an error code instead of a hung device
	auto expired = [&]() { return wall_clock64() - started > 1000000000ull; };
	// SSAA items travel through 64 queues.  A working wave looks at ONE of them (wave % 64) between two work items; the
	// wave that queues items deals them round; an idle wave watches four.  So no address is read by more than 1/64 of the
	// waves, and an item is seen by about a hundred waves as soon as one of them finishes what it is doing.
	// (known: the head / tail word of the queue as read a moment ago by lane 0, or ~0 = read it now)
	auto popFrom = [&](uint32_t q, uint32_t& item, unsigned long long known = ~0ull) -> int {       // 1: got an item, 0: queue empty, -1: gave up on an entry (error set)
		uint32_t got = 0xffffffffu;
		if (lane == 0) {
			const unsigned long long v = known != ~0ull ? known : __hip_atomic_load((const unsigned long long*)(ctl + FC_QUEUE + 16 * q), RTX_AGENT);
			uint32_t head = (uint32_t)v;
			const uint32_t tail = (uint32_t)(v >> 32);
			for (int tries = 0; tries < 4 && head < tail; ++tries) {
				const uint32_t old = atomicCAS(ctl + FC_QUEUE + 16 * q, head, head + 1);
				if (old == head) { got = head; break; }
				head = old;
			}
		}
		got = __builtin_amdgcn_readfirstlane(got);
		if (got == 0xffffffffu) return 0;
		const unsigned long long* entry = P.ssaaQueue + (size_t)q * P.queueCap + got;
		unsigned long long e;
		bool late = false;
		do {                              // the tail is advanced before the entry is written
			e = __hip_atomic_load(entry, RTX_AGENT);
			// (an entry that is never written: its producer found the queue full and set the error word -- nobody waits 10 s for that)
			if ((uint32_t)(e >> 32) != P.epoch && __hip_atomic_load(ctl + FC_ERROR, RTX_AGENT) != 0) return -1;
			late = (uint32_t)(e >> 32) != P.epoch && expired();
		} while ((uint32_t)(e >> 32) != P.epoch && !late);
		if (late) { if (lane == 0) atomicExch(ctl + FC_ERROR, 1u); return -1; }
		item = __builtin_amdgcn_readfirstlane((uint32_t)e);
		return 1;
	};
	// the first non-empty queue among `n` starting at q0 (idle waves), or ~0
	auto scan = [&](uint32_t q0, uint32_t n) -> uint32_t {
		unsigned long long v = 0;
		if (lane < n) v = __hip_atomic_load((const unsigned long long*)(ctl + FC_QUEUE + 16 * ((q0 + lane) & (RTX_FRAME_QUEUES - 1))), RTX_AGENT);
		const uint64_t nonEmpty = ballot((uint32_t)v < (uint32_t)(v >> 32));
		return nonEmpty ? (q0 + (uint32_t)__builtin_ctzll(nonEmpty)) & (RTX_FRAME_QUEUES - 1) : 0xffffffffu;
	};
	// The look into this wave's item queue is a coherent read of its own; after a pass-1 tile it is issued BEFORE waiting
	// for that tile's stores to be acknowledged, so that the two round trips are one.  (Taking the next tile at the same
	// time was measured too: a wave then sits on a tile -- the slowest ones come first -- while it works through items.)
	unsigned long long reqV = 0;             // lane 0: head / tail of the item queue
	bool reqPending = false;
	auto request = [&]() {
		if (lane == 0) reqV = __hip_atomic_load((const unsigned long long*)(ctl + FC_QUEUE + 16 * (wave & (RTX_FRAME_QUEUES - 1))), RTX_AGENT);
		reqPending = true;
	};
	for (;;) {
		uint32_t kind = 0, item = 0;
		if (!reqPending) request();
		reqPending = false;
		// 1. an SSAA item from this wave's queue
		{
			const uint32_t vlo = __builtin_amdgcn_readfirstlane((uint32_t)reqV), vhi = __builtin_amdgcn_readfirstlane((uint32_t)(reqV >> 32));
			if (vlo < vhi) {
				const int r = popFrom(wave & (RTX_FRAME_QUEUES - 1), item, (unsigned long long)vhi << 32 | vlo);
				if (r < 0) break;
				if (r > 0) kind = 2;
			}
		}
		// 2. a pass-1 tile: the queue of this XCD, then the others
		while (kind == 0 && attempt < 8) {
			const uint32_t q = (xcd + attempt) & 7u;
			const uint32_t qBase = sload1(P.tileList + q), qSize = sload1(P.tileList + 8 + q);
			const uint32_t j = nextWork(P.workCounter + q * 16);
			if (j < qSize) { item = sload1(P.tileList + qBase + j); kind = 1; }
			else attempt = uni(attempt + 1);
		}
		// 3. no tiles left, nothing in this wave's queue.  Items may still come until every tile has queued its own (the
		// FC_ALL words); then the frame is finished when the queues are empty.  A wave waits for items in its own queue;
		// the first wave of every block also looks into four others (all of them at the end), so that nothing is left
		// behind in a queue whose own waves are busy or gone.
		if (kind == 0) {
			const bool helper = (threadIdx.x >> 6) == 0;
			const uint32_t all = __hip_atomic_load(ctl + FC_ALL + 16 * (wave & 63u), RTX_AGENT);
			if (!helper && all) {
				// nothing more will be queued: leave when this queue and the seven after it are empty (the rest is looked after
				// by their own waves and by the helpers; everybody reading all 64 words at the end would take longer)
				const uint32_t q = scan(wave & (RTX_FRAME_QUEUES - 1), 8);
				if (q == 0xffffffffu) break;
				const int r = popFrom(q, item);
				if (r < 0) break;
				if (r > 0) kind = 2;
			}
			if (helper) {
				const uint32_t q = all ? scan(0, RTX_FRAME_QUEUES) : scan((blockIdx.x * 4u) & (RTX_FRAME_QUEUES - 1), 4);
				if (q != 0xffffffffu) {
					const int r = popFrom(q, item);
					if (r < 0) break;
					if (r > 0) kind = 2;
				}
				else if (all) break;
			}
			if (kind == 0) {
				uint32_t stop = 0;
				if (lane == 0) {
					stop = __hip_atomic_load(ctl + FC_ERROR, RTX_AGENT) != 0;
					if (!stop && expired()) { atomicExch(ctl + FC_ERROR, 2u); stop = 1; }
				}
				if (__builtin_amdgcn_readfirstlane(stop)) break;
				for (int k = 0; k < 4; ++k) __builtin_amdgcn_s_sleep(127);      // about 14 us
				continue;
			}
		}
		// the 64 rays of the work item
		uint32_t tx, ty, x, y;
		bool valid;
		float fx, fy;
		bool slow;
		if (kind == 1) {
			tx = item & 0x1fffu; ty = (item >> 16) & 0x1fffu;
			x = tx * 8 + (lane & 7); y = ty * 8 + (lane >> 3);
			uint32_t lanes = 64;
			if (item & 0x80000000u) {               // one of sixteen parts of a very slow tile (rtxTileOrderKernel): 2 x 2 pixels
				const uint32_t e = ((item >> 13) & 3u) | ((item >> 29) & 3u) << 2;
				x = tx * 8 + (e & 3u) * 2 + (lane & 1); y = ty * 8 + (e >> 2) * 2 + ((lane >> 1) & 1);
				lanes = 4;
			}
			else if (item & 0x8000u) {              // a quarter of a slow tile: 4 x 4 pixels
				const uint32_t e = (item >> 13) & 3u;
				x = tx * 8 + (e & 1u) * 4 + (lane & 3); y = ty * 8 + (e >> 1) * 4 + ((lane >> 2) & 3);
				lanes = 16;
			}
			// x1/y1 are clamped to W-1/H-1: the last column and row are never rendered (scene.cpp:369-372)
			valid = x < W - 1 && y < H - 1 && y >= P.rowBegin && y < P.rowEnd && rowRendered(P, y) && lane < lanes;
			fx = (float)x + 0.5f; fy = (float)y + 0.5f;
			slow = sload1(P.tileCost + ty * P.tilesXFull + tx) > RTX_PRIO_TICKS;
		}
		else {
			const uint32_t m = item >> 8, k = (item >> 2) & 63u, per = 16u >> (2 * (item & 3u));      // 16, 4 or 1 pixels
			const unsigned long long flags = __hip_atomic_load(P.tileFlags + m, RTX_AGENT);
			ty = m / P.tilesXFull; tx = m - ty * P.tilesXFull;
			const uint32_t n = k * per + (lane >> 2), sub = lane & 3;
			valid = (lane >> 2) < per && n < (uint32_t)__popcll(flags);
			const uint32_t pos = nthSetBit(flags, valid ? n : 0u);
			x = tx * 8 + (pos & 7); y = ty * 8 + (pos >> 3);
			// offsets in the reference's order: (.25,.25) (.25,.75) (.75,.25) (.75,.75)  (scene.cpp:527-534)
			fx = (float)x + ((sub & 2) ? 0.75f : 0.25f); fy = (float)y + ((sub & 1) ? 0.75f : 0.25f);
			slow = __hip_atomic_load(P.tileCost + m, RTX_AGENT) > P.heavyTicks;
		}
		V3 o, d;
		primaryRay(P, fx, fy, o, d);
		if (slow) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
		const unsigned long long t0 = wall_clock64();
		const V3 c = castRayWave<false, MESH, true, BOXES, true, CULLK, PLAIN>(P, valid, o, d, gl, cnt);
		const unsigned long long dt = wall_clock64() - t0;
		RTX_DBG_ONLY(
		if (lane == 0 && wave < 8192 && dbgItems < 160) {
			const size_t e = (size_t)wave * 160 + dbgItems;
			gDbgTimeline[3 * e] = t0; gDbgTimeline[3 * e + 1] = dt; gDbgTimeline[3 * e + 2] = (unsigned long long)kind << 32 | item;
		}
		dbgItems++;
		)
		if (kind == 2) {
			// color = 0; color += c0; += c1; += c2; += c3; fb = color / 4
			const int base = (int)(lane & ~3u);
			V3 sum = mk(0, 0, 0);
			for (int k = 0; k < 4; ++k)
				sum = sum + mk(__shfl(c.x, base + k), __shfl(c.y, base + k), __shfl(c.z, base + k));
			if (valid && (lane & 3) == 0) {
				float* px = P.fb + ((size_t)y * W + x) * 3;
				__hip_atomic_store(px, sum.x / 4, RTX_AGENT); __hip_atomic_store(px + 1, sum.y / 4, RTX_AGENT); __hip_atomic_store(px + 2, sum.z / 4, RTX_AGENT);
			}
			continue;
		}
		if (valid) {
			float* px = P.fb + ((size_t)y * W + x) * 3;
			__hip_atomic_store(px, c.x, RTX_AGENT); __hip_atomic_store(px + 1, c.y, RTX_AGENT); __hip_atomic_store(px + 2, c.z, RTX_AGENT);
		}
		__builtin_amdgcn_s_setprio(0);
		request();              // (its round trip overlaps the acknowledgement of the stores above)
		uint32_t cost = dt > 0x3fffffffull ? 0x3fffffffu : (uint32_t)dt;
		if (item & 0x8000u) {
			// the wave that finishes the last part speaks for the tile; its cost is the sum of the parts' 
			storesAcknowledged();
			uint32_t last = 0;
			if (lane == 0) {
				const uint32_t t = ty * P.tilesXFull + tx;
				cost += atomicAdd(P.tileSobel + P.tilesXFull * P.tilesYFull + t, cost);      // (third array of the counters: cost so far)
				last = atomicAdd(P.tileSobel + 2 * P.tilesXFull * P.tilesYFull + t, 1u) == ((item & 0x80000000u) ? 15u : 3u);
				if (last) cost = __hip_atomic_load(P.tileSobel + P.tilesXFull * P.tilesYFull + t, RTX_AGENT);
			}
			if (!__builtin_amdgcn_readfirstlane(last)) continue;
		}
		if (lane == 0) __hip_atomic_store(P.tileCost + ty * P.tilesXFull + tx, cost, RTX_AGENT);
		storesAcknowledged();
		// count the tile in; Sobel for the tiles whose surroundings are now complete
		uint32_t nIdx;
		uint64_t todo = ballot(frameCountIn(P, P.tileReady, tx, ty, lane, nIdx));
		while (todo) {
			const uint32_t l = (uint32_t)__builtin_ctzll(todo);
			todo &= todo - 1;
			const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)nIdx, (int)l);
			const uint32_t ny = n / P.tilesXFull, nx = n - ny * P.tilesXFull;
			const uint64_t flags = frameSobelTile(P, nx, ny, lane);
			if (lane == 0) __hip_atomic_store(P.tileFlags + n, (unsigned long long)flags, RTX_AGENT);
			storesAcknowledged();
			// second count; the tiles it completes may now be overwritten by SSAA: queue their items
			uint32_t mIdx;
			uint64_t todo2 = ballot(frameCountIn(P, P.tileSobel, nx, ny, lane, mIdx));
			while (todo2) {
				const uint32_t l2 = (uint32_t)__builtin_ctzll(todo2);
				todo2 &= todo2 - 1;
				const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)mIdx, (int)l2);
				const uint32_t nf = (uint32_t)__popcll(__hip_atomic_load(P.tileFlags + m, RTX_AGENT));
				uint32_t mode = 0, items = (nf + 15u) >> 4;
				// A tile that was slow in pass 1 (a silhouette, a pole) gets 4-pixel or 1-pixel items: what matters is when its
				// last ray returns.  Limits: as for splitting the tile in pass 1, and 4 x heavyTicks whatever the frame.
				if (items) {
					const uint32_t c = __hip_atomic_load(P.tileCost + m, RTX_AGENT);
					const uint32_t want = c > sload1(P.splitLimits + 1) ? 2u : (c > sload1(P.splitLimits) || c > RTX_SSAA_VERY * P.heavyTicks ? 1u : 0u);
					if (want) {
						const uint32_t spread = want == 2 ? nf : (nf + 3u) >> 2;
						uint32_t ok = 0;
						if (lane == 0) ok = atomicAdd(ctl + FC_VERY, spread - items) + (spread - items) <= P.veryBudget;
						if (__builtin_amdgcn_readfirstlane(ok)) { mode = want; items = spread; }
					}
				}
				if (items) {
					if (lane < items) {
						const uint32_t q = (rot + lane) & (RTX_FRAME_QUEUES - 1);
						const uint32_t slot = atomicAdd(ctl + FC_QUEUE + 16 * q + 1, 1u);
						if (slot >= P.queueCap) atomicExch(ctl + FC_ERROR, 3u);
						else __hip_atomic_store(P.ssaaQueue + (size_t)q * P.queueCap + slot, (unsigned long long)P.epoch << 32 | (m << 8 | lane << 2 | mode), RTX_AGENT);
					}
					rot = uni(rot + items);
					storesAcknowledged();
				}
				// the tile is accounted for; the last one tells the idle waves that nothing more will be queued
				if (lane == 0) {
					const uint32_t k = m & 63u;
					if (atomicAdd(ctl + FC_COUNT + 16 * k, 1u) + 1 == P.countExpect[k] && atomicAdd(ctl + FC_GROUPS, 1u) + 1 == sload1(P.countExpect + 64))
						for (uint32_t r = 0; r < 64; ++r) __hip_atomic_store(ctl + FC_ALL + 16 * r, 1u, RTX_AGENT);
				}
			}
		}
	}
}

// saveImage's quantiser (util.cpp:46-58): bottom-up rows, BGR, (uint8)(clamp(0, 1, v) * 255).  A thread converts four
// pixels (W % 4 == 0, util.cpp:28-29): 48 bytes in as three 16-byte loads, 12 bytes out as three dwords.  Rows of the
// image that belong to another device (rtx_set_row_ownership) are left alone -- rtx_gather fills them on the root.
__global__ void __launch_bounds__(256) rtxQuantizeKernel(const float* __restrict__ fb, uint8_t* __restrict__ out,
                                                         uint32_t W, uint32_t H, uint32_t bandH, uint32_t nParts, uint32_t part)
{
	const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of four output pixels, bottom-up rows
	const uint32_t perRow = W / 4;
	if (q >= (size_t)perRow * H) return;
	const uint32_t row = (uint32_t)(q / perRow), x = (uint32_t)(q - (size_t)row * perRow) * 4;
	const uint32_t y = H - 1 - row;
	if (!rowOwned(bandH, nParts, part, y)) return;
	const float4* in = (const float4*)(fb + ((size_t)y * W + x) * 3);
	const float4 a = in[0], b = in[1], c = in[2];
	const float v[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
	uint32_t o[3] = { 0, 0, 0 };
	for (int px = 0; px < 4; ++px)
		for (int k = 0; k < 3; ++k) {
			const uint32_t byte = (uint32_t)(uint8_t)(int)(clampRef(0.0f, 1.0f, v[px * 3 + (2 - k)]) * 255);
			const int at = px * 3 + k;
			o[at >> 2] |= byte << ((at & 3) * 8);
		}
	uint32_t* dst = (uint32_t*)(out + ((size_t)row * W + x) * 3);
	dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
}

// Device-math self-check (rtx_math_probe)
__global__ void rtxMathProbeKernel(int op, uint32_t n, const float* x, const float* y, float* out)
{
	fillPowTab();
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float r = 0;
	if (op == 0) r = powfRef(x[i], y[i]);
	else if (op == 1) r = 1 / x[i];
	else if (op == 2) r = __builtin_sqrtf(x[i]);
	else if (op == 3) r = invLenD(x[i]);
	else if (op == 4) r = x[i] / y[i];
	out[i] = r;
}

// Device-side vector helpers of the shading path against the reference's unit vectors (rtx_vec_probe)
__global__ void rtxVecProbeKernel(int op, uint32_t n, const float* a, const float* b, float ior, float* out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const V3 d = load3(a + (size_t)i * 3), nn = b ? load3(b + (size_t)i * 3) : mk(0, 0, 0);
	V3 r = mk(0, 0, 0);
	if (op == 0) r = reflectDir(d, nn);
	else if (op == 1) r = refractDir(d, nn, ior);
	else if (op == 2) r.x = fresnelKr(d, nn, ior);
	else if (op == 3) r = normalized(d);
	out[(size_t)i * 3] = r.x; out[(size_t)i * 3 + 1] = r.y; out[(size_t)i * 3 + 2] = r.z;
}

// explicit instantiations used by rtx_api.hip
template __global__ void rtxPass1Kernel<false>(const Params);
template __global__ void rtxPass1Kernel<true>(const Params);
template __global__ void rtxSsaaKernel<false>(const Params);
template __global__ void rtxSsaaKernel<true>(const Params);
template __global__ void rtxPass1Kernel<false, false>(const Params);
template __global__ void rtxSsaaKernel<false, false>(const Params);
template __global__ void rtxFrameKernel<true>(const Params);
template __global__ void rtxFrameKernel<false>(const Params);
template __global__ void rtxPass1Kernel<false, true, false>(const Params);
template __global__ void rtxSsaaKernel<false, true, false>(const Params);
template __global__ void rtxFrameKernel<true, false>(const Params);
// the same kernels for options::useBackfaceCulling = 0
template __global__ void rtxPass1Kernel<false, true, true, 0>(const Params);
template __global__ void rtxPass1Kernel<false, true, false, 0>(const Params);
template __global__ void rtxSsaaKernel<false, true, true, 0>(const Params);
template __global__ void rtxSsaaKernel<false, true, false, 0>(const Params);
template __global__ void rtxFrameKernel<true, true, 0>(const Params);
template __global__ void rtxFrameKernel<true, false, 0>(const Params);
// ... and all of them again for scenes of Diffuse objects under point / distant lights only (PLAIN: see advance)
#define RTX_PLAIN_INSTANCES(B, C)                                                  \
template __global__ void rtxPass1Kernel<false, true, B, C, true>(const Params);    \
template __global__ void rtxSsaaKernel<false, true, B, C, true>(const Params);     \
template __global__ void rtxFrameKernel<true, B, C, true>(const Params);
RTX_PLAIN_INSTANCES(true, 1) RTX_PLAIN_INSTANCES(false, 1) RTX_PLAIN_INSTANCES(true, 0) RTX_PLAIN_INSTANCES(false, 0)
#undef RTX_PLAIN_INSTANCES
