// TEST INFRASTRUCTURE -- CPU oracle (see rt_oracle.h for the contract and the parity-pinning statement).
//
// Own restatement of the holoskii/Rendering hot path; citations are file:line into /root/reference.
// Numerics contract: fp32 IEEE, no FMA contraction (built with -ffp-contract=off, no -march),
// left-to-right association as written in the reference, fp64 islands where the reference has them.
#include "rt_oracle.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------
// math (include/geometry.h)
// ------------------------------------------------------------------------------------------------
struct V2 { float x = 0, y = 0; };
struct V3 { float x = 0, y = 0, z = 0; };

inline V3 v3(float a, float b, float c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }  // geometry.h:171-174: v.x * r
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // geometry.h:84-87
inline V3 cross(V3 a, V3 b)                                                       // geometry.h:89-92
{
	return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// geometry.h:99-102: unqualified sqrt in the template binds to ::sqrt(double); result narrowed to float
inline float length(V3 a) { return (float)std::sqrt((double)len2(a)); }
// geometry.h:104-112: factor = (float)(1 / sqrt((double)len2)), then three fp32 multiplies
inline V3 normalized(V3 a)
{
	float l2 = len2(a);
	if (l2 > 0) {
		float f = (float)(1 / std::sqrt((double)l2));
		a.x *= f; a.y *= f; a.z *= f;
	}
	return a;
}
inline float fmin_ref(float a, float b) { return (b < a) ? b : a; }   // std::min(a,b)
inline float fmax_ref(float a, float b) { return (a < b) ? b : a; }   // std::max(a,b)
inline float clampf(float lo, float hi, float v) { return fmax_ref(lo, fmin_ref(hi, v)); } // util.h:26-29

struct M44 { float m[4][4]; };
M44 matmul(const M44& a, const M44& b)                                 // geometry.h:248-257
{
	M44 c;
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++)
			c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
	return c;
}
// row-vector * matrix incl. the w logic (geometry.h:289-307)
V3 xform(const M44& M, V3 s)
{
	V3 d;
	d.x = s.x * M.m[0][0] + s.y * M.m[1][0] + s.z * M.m[2][0] + M.m[3][0];
	d.y = s.x * M.m[0][1] + s.y * M.m[1][1] + s.z * M.m[2][1] + M.m[3][1];
	d.z = s.x * M.m[0][2] + s.y * M.m[1][2] + s.z * M.m[2][2] + M.m[3][2];
	float w = s.x * M.m[0][3] + s.y * M.m[1][3] + s.z * M.m[2][3] + M.m[3][3];
	if (w != 0.0f && w != -0.0f && w != 1.0f) {
		const float wi = 1.0f / w;
		d.x *= wi; d.y *= wi; d.z *= wi;
	}
	return d;
}
inline float deg2rad(float f) { return f * (float)(M_PI) / 180.0f; }   // util.h:31-34

// Euler rotation rMatrix = mz * my * mx (scene.cpp:24-48, objects.cpp:180-204)
M44 eulerMatrix(V3 rot)
{
	const float x = deg2rad(rot.x), y = deg2rad(rot.y), z = deg2rad(rot.z);
	M44 mx = { { { 1, 0, 0, 0 }, { 0, cosf(x), -sinf(x), 0 }, { 0, sinf(x), cosf(x), 0 }, { 0, 0, 0, 1 } } };
	M44 my = { { { cosf(y), 0, sinf(y), 0 }, { 0, 1, 0, 0 }, { -sinf(y), 0, cosf(y), 0 }, { 0, 0, 0, 1 } } };
	M44 mz = { { { cosf(z), -sinf(z), 0, 0 }, { sinf(z), cosf(z), 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } } };
	return matmul(matmul(mz, my), mx);
}

// ------------------------------------------------------------------------------------------------
// powf: glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c (= ARM optimized-routines powf, tables
// __powf_log2_data / __exp2f_data), in the form the x86-64 FMA ifunc variant executes it (every a*b+c
// of the two polynomial kernels is one fused multiply-add; verified against libm's disassembly and
// exhaustively-sampled outputs in tests/test_oracle_powf.py).  The reference calls std::pow(float,float)
// -> glibc powf (scene.cpp:824,846,867,887,917,937); restating it makes the oracle independent of the
// host's libm/CPU and gives the HIP kernel an exact specification.
// ------------------------------------------------------------------------------------------------
const double kLog2Tab[16][2] = {
	{ 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 }, { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 },
	{ 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2 },  { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 },
	{ 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 }, { 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3 },
	{ 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 }, { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 },
	{ 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 }, { 0x1p+0, 0x0p+0 },
	{ 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 },  { 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3 },
	{ 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 },
	{ 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 },  { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 },
};
const double kLog2Poly[5] = { 0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2,
	-0x1.7154748bef6c8p-1, 0x1.71547652ab82bp+0 };
const uint64_t kExp2Tab[32] = {
	0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
	0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
	0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
	0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
	0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
	0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
	0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
const double kExp2Shift = 0x1.8p+47;   // 0x1.8p52 / 32
const double kExp2Poly[3] = { 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1 };

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

inline int powfCheckInt(uint32_t iy)   // 0: not int, 1: odd, 2: even
{
	int e = iy >> 23 & 0xff;
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
inline bool zeroInfNan(uint32_t i) { return 2 * i - 1 >= 2u * 0x7f800000 - 1; }

float powfRestated(float x, float y)
{
	uint32_t signBias = 0;
	uint32_t ix = f2u(x), iy = f2u(y);
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
			if (yint == 0) return std::numeric_limits<float>::quiet_NaN();
			if (yint == 1) signBias = 1u << 16;   // SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11)
			ix &= 0x7fffffff;
		}
		if (ix < 0x00800000) {
			ix = f2u(x * 0x1p23f);
			ix &= 0x7fffffff;
			ix -= 23 << 23;
		}
	}
	// log2_inline
	uint32_t tmp = ix - 0x3f330000;
	int i = (tmp >> 19) % 16;
	uint32_t top = tmp & 0xff800000;
	uint32_t iz = ix - top;
	int k = (int32_t)top >> 23;
	double invc = kLog2Tab[i][0], logc = kLog2Tab[i][1];
	double z = (double)u2f(iz);
	double r = std::fma(z, invc, -1.0);
	double y0 = logc + (double)k;
	double r2 = r * r;
	double yy = std::fma(kLog2Poly[0], r, kLog2Poly[1]);
	double p = std::fma(kLog2Poly[2], r, kLog2Poly[3]);
	double r4 = r2 * r2;
	double q = std::fma(kLog2Poly[4], r, y0);
	q = std::fma(p, r2, q);
	double logx = std::fma(yy, r4, q);
	double ylogx = (double)y * logx;
	if ((d2u(ylogx) >> 47 & 0xffff) >= (d2u(126.0) >> 47)) {
		if (ylogx > 0x1.fffffffd1d571p+6) return signBias ? -INFINITY : INFINITY;
		if (ylogx <= -150.0) return signBias ? -0.0f : 0.0f;
		if (ylogx < -149.0) return signBias ? -0x1p-149f : 0x1p-149f;   // __math_may_uflowf
	}
	// exp2_inline
	double kd = ylogx + kExp2Shift;
	uint64_t ki = d2u(kd);
	kd -= kExp2Shift;
	double rr = ylogx - kd;
	uint64_t t = kExp2Tab[ki % 32];
	t += (ki + signBias) << (52 - 5);
	double s = u2d(t);
	double zz = std::fma(kExp2Poly[0], rr, kExp2Poly[1]);
	double rr2 = rr * rr;
	double yv = std::fma(kExp2Poly[2], rr, 1.0);
	yv = std::fma(zz, rr2, yv);
	yv = yv * s;
	return (float)yv;
}

// ------------------------------------------------------------------------------------------------
// scene model (include/objects.h, lights.h, scene.h, options.h)
// ------------------------------------------------------------------------------------------------
enum { OBJ_SPHERE = 1, OBJ_PLANE = 2, OBJ_MESH = 3 };
enum { MAT_DIFFUSE = 0, MAT_REFLECTIVE = 1, MAT_TRANSPARENT = 2, MAT_PHONG = 3 };
enum { LIGHT_DISTANT = 1, LIGHT_POINT = 2, LIGHT_AREA = 3 };

struct Tri {
	V3 a, b, c, na, nb, nc;
	V2 ta, tb, tc;
	V3 tangent, bitangent;
};

struct BvhNode {
	V3 lo, hi;
	std::unique_ptr<BvhNode> left, right;
	std::vector<uint32_t> tris;
};

struct Object {
	int type = 0;
	int material = MAT_DIFFUSE;
	V3 pos = v3(1, 1, 1);                 // objects.h:27 (Object default centre = 1)
	V3 color = v3(1, 1, 1);
	float ior = 1.4f, ambient = 0.1f, diffuse = 0.1f, specular = 1.0f, nSpecular = 5.0f;  // objects.h:42-46
	float r = 1, r2 = 1;                  // sphere
	V3 normal = v3(0, 1, 0);              // plane
	// mesh
	V3 size, rot;
	std::vector<Tri> tris;
	std::unique_ptr<BvhNode> root;
	int dW = 0, dH = 0, nW = 0, nH = 0, sW = 0, sH = 0;
	std::vector<V3> diffuseMap, normalMap;
	std::vector<float> specularMap;
	bool hasDiffuse = false, hasNormal = false, hasSpecular = false;
};

struct Light {
	int type = 0;
	V3 color = v3(1, 1, 1);
	float intensity = 1;
	V3 dir = v3(0, 0, -1);   // distant (lights.h:39; ctor-normalised, already unit)
	V3 pos;                  // point / area centre
	V3 ai, aj;               // area base vectors
	int samples = 1;
	std::vector<V3> points;  // area sample points (lights.cpp:46-63)
};

struct Stats { int64_t rays = 0, boxTests = 0, triTests = 0; };

struct Hit {
	int obj = -1;
	float t = std::numeric_limits<float>::max();
	int tri = -1;
	V2 uv = { -1, -1 };
};

struct Ray { V3 o, d; bool shadow = false; };

} // namespace

struct orc_scene {
	// Options (options.h:9-20)
	size_t width = 800, height = 600;
	float bias = 0.0001f;
	int maxRayDepth = 10;
	int nWorkers = 32;
	V3 background;
	int acPenalty = 1;
	std::string skyNames[6];
	std::string imageName = "out";
	// options:: flags that matter on the hot path (options.h:23-37)
	bool useBackfaceCulling = true, useSkybox = false, useTextures = true, collectStatistics = false;
	// camera (scene.h:52-66)
	V3 camPos, camRot;
	float fov = 60.0f;
	M44 camMatrix;
	bool camReady = false;
	std::vector<std::unique_ptr<Object>> objects;
	std::vector<std::unique_ptr<Light>> lights;
	int skyW = 0, skyH = 0;
	std::vector<V3> sky[6];
	// statistics
	std::atomic<int64_t> sRays{ 0 }, sBox{ 0 }, sTri{ 0 };
};

namespace {

thread_local Stats* tlStats = nullptr;
std::string gLastError;

struct LoadError { std::string msg; };
[[noreturn]] void fail(const std::string& m) { throw LoadError{ m }; }

// ------------------------------------------------------------------------------------------------
// BMP I/O (util.cpp:78-113, 15-76)
// ------------------------------------------------------------------------------------------------
// Returns RGB bytes in file row order (bottom-up kept, util.cpp:98-110); assumes 54-byte header, 24 bpp.
std::vector<unsigned char> loadBmp(const std::string& path, int& w, int& h)
{
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) fail("Could not open .bmp file: " + path);
	unsigned char info[54];
	if (fread(info, 1, 54, f) != 54) { fclose(f); fail("short bmp header: " + path); }
	memcpy(&w, info + 18, 4);
	memcpy(&h, info + 22, 4);
	size_t size = (size_t)3 * w * h;
	std::vector<unsigned char> data(size);
	size_t got = fread(data.data(), 1, size, f);
	(void)got;
	fclose(f);
	for (size_t i = 0; i + 2 < size; i += 3) std::swap(data[i], data[i + 2]);   // BGR -> RGB
	return data;
}

// ------------------------------------------------------------------------------------------------
// BVH build (objects.cpp:470-526, 633-763)
// ------------------------------------------------------------------------------------------------
inline float axisOf(const V3& v, int ax) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); }

float sahCost(int ax, const std::vector<uint32_t>& ids, const std::vector<Tri>& T, const BvhNode& n, float s)
{
	// objects.cpp:633-674: nLeft*(s-min) + nRight*(max-s), int counts promoted to float
	int nl = 0, nr = 0;
	for (uint32_t id : ids) {
		const Tri& t = T[id];
		float a = axisOf(t.a, ax), b = axisOf(t.b, ax), c = axisOf(t.c, ax);
		if (a <= s || b <= s || c <= s) nl++;
		if (a >= s || b >= s || c >= s) nr++;
	}
	return nl * (s - axisOf(n.lo, ax)) + nr * (axisOf(n.hi, ax) - s);
}

float sahSearch(int ax, const std::vector<uint32_t>& ids, const std::vector<Tri>& T, const BvhNode& n, float lo, float hi)
{
	// objects.cpp:676-689 (the recursion is a plain loop)
	for (;;) {
		float mid = hi - (hi - lo) / 2;
		if (hi - lo < 0.1f) return mid;
		if (sahCost(ax, ids, T, n, mid - 0.05f) < sahCost(ax, ids, T, n, mid + 0.05f)) hi = mid;
		else lo = mid;
	}
}

void buildNode(BvhNode& n, std::vector<uint32_t>& ids, int depth, const std::vector<Tri>& T, int acPenalty)
{
	// objects.cpp:476-483
	if (ids.size() <= depth * (size_t)acPenalty) { n.tris = ids; return; }
	// objects.cpp:485-490
	V3 dim = n.hi - n.lo;
	int ax;
	if (dim.x > dim.y && dim.x > dim.z) ax = 0;
	else if (dim.y > dim.z) ax = 1;
	else ax = 2;
	// objects.cpp:691-763
	float s = sahSearch(ax, ids, T, n, axisOf(n.lo, ax), axisOf(n.hi, ax));
	std::vector<uint32_t> L, R;
	for (uint32_t id : ids) {
		const Tri& t = T[id];
		float a = axisOf(t.a, ax), b = axisOf(t.b, ax), c = axisOf(t.c, ax);
		if (a <= s || b <= s || c <= s) L.push_back(id);
		if (a >= s || b >= s || c >= s) R.push_back(id);
	}
	// objects.cpp:498-504
	if ((L.size() == 0 || R.size() == 0) || (L.size() + R.size() >= ids.size() * 1.5)) { n.tris = ids; return; }
	n.left.reset(new BvhNode);
	n.right.reset(new BvhNode);
	// objects.cpp:510-521
	n.left->lo = n.lo; n.left->hi = n.hi; n.right->lo = n.lo; n.right->hi = n.hi;
	if (ax == 0) { n.left->hi.x = s; n.right->lo.x = s; }
	else if (ax == 1) { n.left->hi.y = s; n.right->lo.y = s; }
	else { n.left->hi.z = s; n.right->lo.z = s; }
	// objects.cpp:524-525 (order irrelevant to the result)
	buildNode(*n.right, R, depth + 1, T, acPenalty);
	buildNode(*n.left, L, depth + 1, T, acPenalty);
}

// ------------------------------------------------------------------------------------------------
// OBJ loader (objects.cpp:177-394)
// ------------------------------------------------------------------------------------------------
size_t objUInt(const char*& p)      // objects.cpp:207-215
{
	size_t v = 0;
	while (*p == ' ') p++;
	if (*p == '/') p++;
	while (*p && *p != ' ' && *p != '/') v = v * 10 + *p++ - '0';
	return v;
}

Tri makeTri(V3 a, V3 b, V3 c)       // objects.cpp:17-21
{
	Tri t; t.a = a; t.b = b; t.c = c;
	t.na = t.nb = t.nc = cross(b - a, c - a);
	return t;
}
Tri makeTriN(V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc)   // objects.cpp:23-30
{
	Tri t = makeTri(a, b, c); t.na = na; t.nb = nb; t.nc = nc; return t;
}
Tri makeTriNT(V3 a, V3 b, V3 c, V3 na, V3 nb, V3 nc, V2 ta, V2 tb, V2 tc)   // objects.cpp:32-56
{
	Tri t = makeTriN(a, b, c, na, nb, nc);
	t.ta = ta; t.tb = tb; t.tc = tc;
	V3 e1 = b - a, e2 = c - a;
	V2 d1 = { tb.x - ta.x, tb.y - ta.y }, d2 = { tc.x - ta.x, tc.y - ta.y };
	float f = 1.0f / (d1.x * d2.y - d2.x * d1.y);
	t.tangent.x = f * (d2.y * e1.x - d1.y * e2.x);
	t.tangent.y = f * (d2.y * e1.y - d1.y * e2.y);
	t.tangent.z = f * (d2.y * e1.z - d1.y * e2.z);
	t.bitangent.x = f * (-d2.x * e1.x + d1.x * e2.x);
	t.bitangent.y = f * (-d2.x * e1.y + d1.x * e2.y);
	t.bitangent.z = f * (-d2.x * e1.z + d1.x * e2.z);
	return t;
}

bool loadObj(Object& m, const std::string& filename, const orc_scene& sc)
{
	const M44 R = eulerMatrix(m.rot);
	std::ifstream ifs(filename, std::ios::in);
	if (!ifs.good()) return false;     // objects.cpp:219-222: message + return false, mesh stays without AC
	m.root.reset(new BvhNode);
	std::string line;
	bool fitted = false;
	std::vector<V3> V, N;
	std::vector<V2> TX;
	V3 mn = v3(std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max());
	V3 mx = v3(std::numeric_limits<float>::min(), std::numeric_limits<float>::min(), std::numeric_limits<float>::min()); // objects.cpp:231
	do {
		std::getline(ifs, line);
		if (line.find('#') != std::string::npos) line.erase(line.find('#'));
		if (line.length() <= 0) continue;
		const char* c = line.c_str();
		char hdr[32] = { 0 };
		int res = sscanf(c, "%31s", hdr);
		if (res == 0) return false;
		c += strlen(hdr) + 1;
		if (strcmp(hdr, "v") == 0) {
			float x, y, z;
			if (sscanf(c, "%f %f %f", &x, &y, &z) != 3) fail("bad v line in " + filename);
			mn.x = fmin_ref(x, mn.x); mn.y = fmin_ref(y, mn.y); mn.z = fmin_ref(z, mn.z);
			mx.x = fmax_ref(x, mx.x); mx.y = fmax_ref(y, mx.y); mx.z = fmax_ref(z, mx.z);
			V.push_back(v3(x, y, z));
		}
		else if (strcmp(hdr, "vn") == 0) {
			float x, y, z;
			if (sscanf(c, "%f %f %f", &x, &y, &z) != 3) fail("bad vn line in " + filename);
			N.push_back(normalized(v3(x, y, z)));          // objects.cpp:270
		}
		else if (strcmp(hdr, "vt") == 0) {
			float x, y;
			if (sscanf(c, "%f %f", &x, &y) != 2) fail("bad vt line in " + filename);
			V2 t; t.x = x; t.y = y; TX.push_back(t);
		}
		else if (strcmp(hdr, "f") == 0) {
			if (!fitted) {
				fitted = true;
				// objects.cpp:285-303
				V3 range = mx - mn;
				V3 ns = m.size;
				if (!(range.x < sc.bias || range.y < sc.bias || range.z < sc.bias)) {
					V3 st = m.size / range;
					float ms = fmin_ref(st.x, fmin_ref(st.y, st.z));
					if (ms == st.x) { ns.y = ns.x / (range.x / range.y); ns.z = ns.x / (range.x / range.z); }
					else if (ms == st.y) { ns.x = ns.y / (range.y / range.x); ns.z = ns.y / (range.y / range.z); }
					else { ns.x = ns.z / (range.z / range.x); ns.y = ns.z / (range.z / range.y); }
				}
				// objects.cpp:306-320
				for (auto& v : V) {
					v.x = ns.x * ((v.x - mn.x) / range.x - 0.5f);
					v.y = ns.y * ((v.y - mn.y) / range.y - 0.5f);
					v.z = ns.z * ((v.z - mn.z) / range.z - 0.5f);
					v = xform(R, v);
					v.x += m.pos.x; v.y += m.pos.y; v.z += m.pos.z;
					if (range.x < sc.bias) v.x = m.pos.x;
					if (range.y < sc.bias) v.y = m.pos.y;
					if (range.z < sc.bias) v.z = m.pos.z;
				}
				for (auto& n : N) n = xform(R, n);                // objects.cpp:323-325
				// objects.cpp:328-330 (root box from the rotated size vector -- wrong for rotated meshes, kept)
				ns = xform(R, ns);
				ns = v3((float)fabs(ns.x), (float)fabs(ns.y), (float)fabs(ns.z));
				m.root->lo = m.pos - ns / 2;
				m.root->hi = m.pos + ns / 2;
			}
			int slashes = 0;
			for (const char* p = c; *p; p++) if (*p == '/') slashes++;
			const char* p = c;
			if (slashes == 0) {
				std::vector<size_t> vi;
				size_t v;
				while ((v = objUInt(p)) > 0) vi.push_back(v);
				for (size_t i = 1; i + 1 < vi.size(); i++)
					m.tris.push_back(makeTri(V.at(vi[0] - 1), V.at(vi[i] - 1), V.at(vi[i + 1] - 1)));
			}
			else if (slashes % 2 == 0) {
				std::vector<size_t> vi, ti, ni;
				size_t v, t, n;
				while ((v = objUInt(p)) > 0) {
					t = objUInt(p);
					n = objUInt(p);
					vi.push_back(v);
					if (t > 0) ti.push_back(t);
					if (n > 0) ni.push_back(n);
				}
				for (size_t i = 1; i + 1 < vi.size(); i++) {
					if (ni.size() == 0)
						m.tris.push_back(makeTri(V.at(vi[0] - 1), V.at(vi[i] - 1), V.at(vi[i + 1] - 1)));
					else if (ti.size() == 0)
						m.tris.push_back(makeTriN(V.at(vi[0] - 1), V.at(vi[i] - 1), V.at(vi[i + 1] - 1),
							N.at(ni.at(0) - 1), N.at(ni.at(i) - 1), N.at(ni.at(i + 1) - 1)));
					else
						m.tris.push_back(makeTriNT(V.at(vi[0] - 1), V.at(vi[i] - 1), V.at(vi[i + 1] - 1),
							N.at(ni.at(0) - 1), N.at(ni.at(i) - 1), N.at(ni.at(i + 1) - 1),
							TX.at(ti.at(0) - 1), TX.at(ti.at(i) - 1), TX.at(ti.at(i + 1) - 1)));
				}
			}
			// odd slash counts ("a/t") are reported and skipped by the reference (objects.cpp:376-378)
		}
	} while (ifs.good());
	std::vector<uint32_t> ids(m.tris.size());
	for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint32_t)i;
	buildNode(*m.root, ids, 1, m.tris, sc.acPenalty);      // objects.cpp:389
	return true;
}

// texture maps (objects.cpp:396-458)
bool loadDiffuse(Object& m, const std::string& fn, const orc_scene& sc)
{
	if (!sc.useTextures) return false;
	std::vector<unsigned char> d = loadBmp(fn, m.dW, m.dH);
	m.diffuseMap.resize((size_t)m.dW * m.dH);
	for (size_t i = 0; i < m.diffuseMap.size(); i++) {
		float x = d[i * 3], y = d[i * 3 + 1], z = d[i * 3 + 2];
		x /= 256; y /= 256; z /= 256;
		m.diffuseMap[i] = v3(x, y, z);
	}
	return true;
}
bool loadNormal(Object& m, const std::string& fn, const orc_scene& sc)
{
	if (!sc.useTextures) return false;
	std::vector<unsigned char> d = loadBmp(fn, m.nW, m.nH);
	m.normalMap.resize((size_t)m.nW * m.nH);
	for (size_t i = 0; i < m.normalMap.size(); i++) {
		float x = d[i * 3], y = d[i * 3 + 1], z = d[i * 3 + 2];
		x /= 256; y /= 256; z /= 256;
		m.normalMap[i] = normalized(v3(x * 2 - 1, -(y * 2 - 1), z));   // objects.cpp:433
	}
	return true;
}
bool loadSpecular(Object& m, const std::string& fn, const orc_scene& sc)
{
	if (!sc.useTextures) return false;
	std::vector<unsigned char> d = loadBmp(fn, m.sW, m.sH);
	m.specularMap.resize((size_t)m.sW * m.sH);
	for (size_t i = 0; i < m.specularMap.size(); i++) {
		float x = d[i * 3], y = d[i * 3 + 1], z = d[i * 3 + 2];
		x /= 256; y /= 256; z /= 256;
		m.specularMap[i] = (x + y + z) / 3.0f;
	}
	return true;
}

// ------------------------------------------------------------------------------------------------
// .scene loader (scene.cpp:62-360, util.h:36-90)
// ------------------------------------------------------------------------------------------------
template <typename T> T parseNum(const std::string& s)
{
	T r = 0;
	std::stringstream ss{ s };
	ss >> r;
	if (!ss.eof() && !ss.good()) fail("cannot parse value '" + s + "'");
	return r;
}
std::vector<std::string> split(const std::string& s, char delim)
{
	std::vector<std::string> out;
	std::stringstream ls{ s };
	std::string cell;
	while (std::getline(ls, cell, delim)) out.push_back(cell);
	return out;
}
V3 parse3(const std::string& s)
{
	auto p = split(s, ',');
	if (p.size() != 3) fail("expected 3 comma-separated values in '" + s + "'");
	return v3(parseNum<float>(p[0]), parseNum<float>(p[1]), parseNum<float>(p[2]));
}
bool has(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

void loadSkybox(orc_scene& sc)          // scene.cpp:336-360
{
	for (int k = 0; k < 6; k++) {
		int w = 0, h = 0;
		std::vector<unsigned char> d = loadBmp(sc.skyNames[k], w, h);
		sc.skyW = w; sc.skyH = h;
		sc.sky[k].resize((size_t)w * h);
		for (size_t i = 0; i < sc.sky[k].size(); i++) {
			float x = d[i * 3], y = d[i * 3 + 1], z = d[i * 3 + 2];
			x /= 256; y /= 256; z /= 256;
			sc.sky[k][i] = v3(x, y, z);
		}
	}
}

void loadScene(orc_scene& sc, const std::string& path)
{
	unsigned hw = std::thread::hardware_concurrency();
	if (hw != 0) sc.nWorkers = (int)hw;            // scene.cpp:68-70
	std::ifstream ifs(path, std::ifstream::in);
	if (!ifs.good()) fail("Could not open scene file: " + path);
	enum { B_NONE, B_OPTIONS, B_LIGHT, B_OBJECT } block = B_NONE;
	Light* light = nullptr; Object* object = nullptr;
	bool lightOwned = true, objectOwned = true;
	std::string s;
	auto finish = [&]() {
		if (block == B_LIGHT) {
			if (!light || lightOwned) fail("light block without type");
			sc.lights.emplace_back(light); lightOwned = true;
		}
		else if (block == B_OBJECT) {
			if (!object || objectOwned) fail("object block without type");
			sc.objects.emplace_back(object); objectOwned = true;
		}
	};
	while (ifs.good()) {
		std::getline(ifs, s);
		if (s.length() == 0) continue;
		if (has(s, "[")) finish();                                     // scene.cpp:96-107
		if (has(s, "#[")) {                                            // scene.cpp:110-116
			do { std::getline(ifs, s); } while ((!has(s, "[") || has(s, "#[")) && ifs.good());
			if (!ifs.good()) break;
		}
		if (has(s, "#")) s.erase(s.find('#'));                          // scene.cpp:119-122
		if (s.length() == 0) continue;
		if (s[0] == '[') {                                             // scene.cpp:125-132
			if (s == "[options]") block = B_OPTIONS;
			else if (s == "[light]") block = B_LIGHT;
			else if (s == "[object]") block = B_OBJECT;
			else if (s == "[end]") { block = B_NONE; break; }
			else fail("unknown block " + s);
			continue;
		}
		if (block == B_NONE) continue;
		if (!has(s, "=")) fail("line without '=': " + s);
		std::string key = s.substr(0, s.find('='));
		std::string val = s.substr(s.find('=') + 1);
		if (block == B_OPTIONS) {                                      // scene.cpp:135-198
			std::string k;
			for (char ch : key) if (ch != ' ' && ch != '\t') k.push_back(ch);
			if (k == "useBackfaceCulling") sc.useBackfaceCulling = parseNum<bool>(val);
			else if (k == "collectStatistics") sc.collectStatistics = parseNum<bool>(val);
			else if (k == "useSkybox") sc.useSkybox = parseNum<bool>(val);
			else if (k == "useTextures") sc.useTextures = parseNum<bool>(val);
			else if (k == "outputProgress" || k == "enableOutput" || k == "imageOutput" || k == "useAC" ||
				k == "showAC" || k == "showNormals") (void)parseNum<bool>(val);   // not on the measured path
			else if (k == "width") sc.width = parseNum<int>(val);
			else if (k == "height") sc.height = parseNum<int>(val);
			else if (k == "fov") sc.fov = parseNum<float>(val);
			else if (k == "image_name") sc.imageName = val;
			else if (k == "n_workers") sc.nWorkers = parseNum<int>(val);
			else if (k == "max_ray_depth") sc.maxRayDepth = parseNum<int>(val);
			else if (k == "ac_penalty") sc.acPenalty = parseNum<int>(val);
			else if (k == "background_color") sc.background = parse3(val);
			else if (k == "position") sc.camPos = parse3(val);
			else if (k == "rotation") sc.camRot = parse3(val);
			else if (k == "skyboxes") {
				auto r = split(val, ',');
				if (r.size() < 6) fail("skyboxes needs 6 names");
				for (int i = 0; i < 6; i++) sc.skyNames[i] = r[i].substr(0, 63);
				sc.useSkybox = true;
			}
		}
		else if (block == B_LIGHT) {                                   // scene.cpp:199-249
			if (key == "type") {
				Light* l = new Light;
				if (val == "distant") l->type = LIGHT_DISTANT;
				else if (val == "point") l->type = LIGHT_POINT;
				else if (val == "area") l->type = LIGHT_AREA;
				else { delete l; l = nullptr; }
				if (l) { light = l; lightOwned = false; }
			}
			else if (!light) continue;
			else if (key == "color") light->color = parse3(val);
			else if (key == "intensity") light->intensity = parseNum<float>(val);
			if (key == "direction") { if (light->type != LIGHT_DISTANT) fail("direction on non-distant light"); light->dir = parse3(val); }
			else if (key == "position") { if (light->type != LIGHT_POINT) fail("position on non-point light"); light->pos = parse3(val); }
			else if (key == "pos") { if (light->type != LIGHT_AREA) fail("pos on non-area light"); light->pos = parse3(val); }
			else if (key == "i") { if (light->type != LIGHT_AREA) fail("i on non-area light"); light->ai = parse3(val); }
			else if (key == "j") { if (light->type != LIGHT_AREA) fail("j on non-area light"); light->aj = parse3(val); }
			else if (key == "samples") { if (light->type != LIGHT_AREA) fail("samples on non-area light"); light->samples = parseNum<int>(val); }
		}
		else if (block == B_OBJECT) {                                  // scene.cpp:250-324
			if (key == "type") {
				Object* o = new Object;
				if (val == "plane") o->type = OBJ_PLANE;
				else if (val == "sphere") { o->type = OBJ_SPHERE; o->pos = v3(0, 0, 0); }   // objects.h:169
				else if (val == "mesh") o->type = OBJ_MESH;
				else { delete o; o = nullptr; }
				if (o) { object = o; objectOwned = false; }
			}
			else if (!object) continue;
			else if (key == "color") object->color = parse3(val);
			else if (key == "pos") object->pos = parse3(val);
			else if (key == "material") {
				auto r = split(val, ',');
				if (r.empty()) fail("empty material");
				if (r[0] == "transparent") { object->material = MAT_TRANSPARENT; object->ior = parseNum<float>(r.at(1)); }
				else if (r[0] == "reflective") object->material = MAT_REFLECTIVE;
				if (r[0] == "phong") {
					object->material = MAT_PHONG;
					object->ambient = parseNum<float>(r.at(1)); object->diffuse = parseNum<float>(r.at(2));
					object->specular = parseNum<float>(r.at(3)); object->nSpecular = parseNum<float>(r.at(4));
				}
			}
			else if (object->type == OBJ_SPHERE) {
				if (key == "radius") { object->r = parseNum<float>(val); object->r2 = object->r * object->r; }  // powf(r,2) folds to r*r
			}
			else if (object->type == OBJ_PLANE) {
				if (key == "normal") object->normal = parse3(val);     // NOT re-normalised (scene.cpp:300)
			}
			else if (object->type == OBJ_MESH) {
				if (key == "size") object->size = parse3(val);
				else if (key == "rot") object->rot = parse3(val);
				else if (key == "name") loadObj(*object, val, sc);
				else if (key == "diffuse_map") object->hasDiffuse = loadDiffuse(*object, val, sc);
				else if (key == "normal_map") object->hasNormal = loadNormal(*object, val, sc);
				else if (key == "specular_map") object->hasSpecular = loadSpecular(*object, val, sc);
			}
		}
	}
	if (!lightOwned) delete light;
	if (!objectOwned) delete object;
	if (sc.useSkybox) loadSkybox(sc);
}

// ------------------------------------------------------------------------------------------------
// intersection primitives (objects.cpp:59-95, 534-631, 774-824)
// ------------------------------------------------------------------------------------------------
inline bool hitTriangle(const orc_scene& sc, const Ray& ray, const Tri& tr, float& t, V2& uv)
{
	if (tlStats) tlStats->triTests++;
	V3 e1 = tr.b - tr.a, e2 = tr.c - tr.a;
	V3 pvec = cross(ray.d, e2);
	float det = dot(e1, pvec);
	if (sc.useBackfaceCulling) { if ((double)det < 1e-8) return false; }
	if (fabs((double)det) < 1e-8) return false;
	float inv = 1 / det;
	V3 tvec = ray.o - tr.a;
	float u = dot(tvec, pvec) * inv;
	if (u < 0 || u > 1) return false;
	V3 qvec = cross(tvec, e1);
	float v = dot(ray.d, qvec) * inv;
	if (v < 0 || u + v > 1) return false;
	t = dot(e2, qvec) * inv;
	if (t < 0) return false;
	uv.x = u; uv.y = v;
	return true;
}

inline bool hitBox(const Ray& ray, const BvhNode& n)      // objects.cpp:534-570
{
	if (tlStats) tlStats->boxTests++;
	const V3 inv = v3(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
	const V3* b[2] = { &n.lo, &n.hi };
	const int sx = inv.x < 0, sy = inv.y < 0, sz = inv.z < 0;
	float tmin = (b[sx]->x - ray.o.x) * inv.x;
	float tmax = (b[1 - sx]->x - ray.o.x) * inv.x;
	float tymin = (b[sy]->y - ray.o.y) * inv.y;
	float tymax = (b[1 - sy]->y - ray.o.y) * inv.y;
	if ((tmin > tymax) || (tymin > tmax)) return false;
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (b[sz]->z - ray.o.z) * inv.z;
	float tzmax = (b[1 - sz]->z - ray.o.z) * inv.z;
	if ((tmin > tzmax) || (tzmin > tmax)) return false;
	return true;
}

bool walkBvh(const orc_scene& sc, const Object& m, const BvhNode& n, const Ray& ray, float& t0, int& tri, V2& uv)
{
	// objects.cpp:587-631: exhaustive DFS, left then right, strict < keeps the first of equal hits
	if (!hitBox(ray, n)) return false;
	bool inter = false;
	float tt; V2 tuv; int ttri = -1;
	t0 = std::numeric_limits<float>::max();
	if (n.left) {
		if (walkBvh(sc, m, *n.left, ray, tt, ttri, tuv) && tt < t0) { inter = true; t0 = tt; uv = tuv; tri = ttri; }
		if (walkBvh(sc, m, *n.right, ray, tt, ttri, tuv) && tt < t0) { inter = true; t0 = tt; uv = tuv; tri = ttri; }
		return inter;
	}
	for (uint32_t id : n.tris) {
		if (hitTriangle(sc, ray, m.tris[id], tt, tuv) && tt < t0) { inter = true; t0 = tt; uv = tuv; tri = (int)id; }
	}
	return inter;
}

inline bool hitSphere(const Object& s, const Ray& ray, float& t0)     // objects.cpp:774-786
{
	V3 L = s.pos - ray.o;
	float tca = dot(L, ray.d);
	float d2 = dot(L, L) - tca * tca;
	if (d2 > s.r2) return false;
	float thc = sqrtf(s.r2 - d2);
	t0 = tca - thc;
	float t1 = tca + thc;
	if (t0 < 0) t0 = t1;
	if (t0 < 0) return false;
	return true;
}

inline bool hitPlane(const Object& p, const Ray& ray, float& t0)      // objects.cpp:807-814
{
	float denom = dot(ray.d, p.normal);
	if (fabs((double)denom) < 1e-8) return false;
	t0 = dot(p.pos - ray.o, p.normal) / denom;
	return (t0 >= 0);
}

// Render::trace (scene.cpp:724-756)
bool trace(const orc_scene& sc, const Ray& ray, Hit& h)
{
	if (tlStats) tlStats->rays++;
	h.obj = -1;
	for (size_t i = 0; i < sc.objects.size(); i++) {
		const Object& o = *sc.objects[i];
		if (ray.shadow && o.material == MAT_TRANSPARENT) continue;
		float tn = std::numeric_limits<float>::max();
		if (o.type == OBJ_MESH) {
			int tri = -1; V2 uv;
			if (!o.root) { fprintf(stderr, "oracle: mesh without BVH\n"); abort(); }
			if (walkBvh(sc, o, *o.root, ray, tn, tri, uv) && tn < h.t) { h.obj = (int)i; h.t = tn; h.tri = tri; h.uv = uv; }
		}
		else {
			bool hit = (o.type == OBJ_SPHERE) ? hitSphere(o, ray, tn) : hitPlane(o, ray, tn);
			// the analytic intersectors leave the caller's uv untouched (objects.cpp:774,807), so trace copies
			// its default-constructed (0,0) (scene.cpp:737,751)
			if (hit && tn < h.t) { h.obj = (int)i; h.t = tn; h.uv = V2(); }
		}
	}
	return h.obj >= 0;
}

// ------------------------------------------------------------------------------------------------
// shading (scene.cpp:672-722, 381-442; objects.cpp:121-175, 788-824; lights.cpp:18-63)
// ------------------------------------------------------------------------------------------------
inline V3 reflectDir(V3 d, V3 n) { return d - 2 * dot(d, n) * n; }       // scene.cpp:672-675: ((2*dot)*n)

V3 refractDir(V3 d, V3 n, float ior)     // scene.cpp:677-696
{
	float n1 = 1, n2 = ior;
	float cosi = clampf(-1, 1, dot(d, n));
	V3 mn = n;
	if (cosi < 0) cosi = -cosi;
	else { std::swap(n1, n2); mn = -n; }
	float rri = n1 / n2;
	float k = 1 - rri * rri * (1 - cosi * cosi);
	if (k < 0) return v3(0, 0, 0);
	return rri * d + (rri * cosi - sqrtf(k)) * mn;
}

float fresnelKr(V3 d, V3 n, float ior)   // scene.cpp:698-722
{
	float n1 = 1, n2 = ior;
	float cosi = clampf(-1, 1, dot(d, n));
	if (cosi > 0) std::swap(n1, n2);
	float sint = n1 / n2 * sqrtf(fmax_ref(0.f, 1 - cosi * cosi));
	if (sint >= 1) return 1;
	float cost = sqrtf(fmax_ref(0.f, 1 - sint * sint));
	cosi = fabsf(cosi);
	float rs = ((n2 * cosi) - (n1 * cost)) / ((n2 * cosi) + (n1 * cost));
	float rp = ((n1 * cosi) - (n2 * cost)) / ((n1 * cosi) + (n2 * cost));
	return (rs * rs + rp * rp) / 2;
}

V3 skyColor(const orc_scene& sc, V3 dir)  // scene.cpp:381-442
{
	if (!sc.useSkybox) return sc.background;
	auto toPixel = [](float v, int mx) { int val = (int)((v + 1.0f) / 2.0f * mx); if (val >= mx) val = mx - 1; return val; };
	const int W = sc.skyW, H = sc.skyH;
	double ax = fabs((double)dir.x), ay = fabs((double)dir.y), az = fabs((double)dir.z);
	double mx = std::max(ax, std::max(ay, az));       // fabs() on float promotes to double (exact); scene.cpp:397
	// std::max(a, b) = (a < b) ? b : a on doubles -- same selection as on the floats
	V3 a;
	if (mx == az) {
		if (dir.z < 0) { a = dir * (1 / -dir.z); return sc.sky[1][(size_t)toPixel(a.y, H) * W + toPixel(a.x, W)]; }
		a = dir * (1 / dir.z); return sc.sky[3][(size_t)toPixel(a.y, H) * W + toPixel(-a.x, W)];
	}
	else if (mx == ax) {
		if (dir.x < 0) { a = dir * (1 / -dir.x); return sc.sky[0][(size_t)toPixel(a.y, H) * W + toPixel(-a.z, W)]; }
		a = dir * (1 / dir.x); return sc.sky[2][(size_t)toPixel(a.y, H) * W + toPixel(a.z, W)];
	}
	else {
		if (dir.y < 0) { a = dir * (1 / -dir.y); return sc.sky[5][(size_t)toPixel(a.z, H) * W + toPixel(a.x, W)]; }
		a = dir * (1 / dir.y); return sc.sky[4][(size_t)toPixel(a.z, H) * W + toPixel(a.x, W)];
	}
}

// objects.cpp:144-147, 156-159: (int)(dim * coord), clamped on the high side only.  For 0 <= dim*coord < 2^31 this is the
// reference bit for bit.  Outside that range the reference is undefined (negative index = out-of-bounds read, NaN / huge
// = UB conversion); the oracle DEFINES those cases: negative or NaN -> texel 0, too large -> dim-1.
inline int texel(int dim, float coord)
{
	const float f = dim * coord;
	if (f >= (float)dim) return dim - 1;
	if (!(f >= 0)) return 0;
	return (int)f;
}

void surfaceData(const Object& o, V3 P, int tri, V2 uv, V3& N, V2& tex)
{
	if (o.type == OBJ_SPHERE) { N = normalized(P - o.pos); return; }       // objects.cpp:788-796 (uv dead)
	if (o.type == OBJ_PLANE) { N = o.normal; return; }                      // objects.cpp:816-824 (uv dead)
	const Tri& t = o.tris[tri];                                             // objects.cpp:121-151
	V2 a = { t.tb.x * uv.x, t.tb.y * uv.x }, b = { t.tc.x * uv.y, t.tc.y * uv.y };
	float w = 1 - uv.x - uv.y;
	tex.x = a.x + b.x + t.ta.x * w;
	tex.y = a.y + b.y + t.ta.y * w;
	N = normalized((t.nb * uv.x + t.nc * uv.y + t.na * (1 - uv.x - uv.y)) / 3);
	if (o.hasNormal) {
		int x = texel(o.nW, tex.x), y = texel(o.nH, tex.y);
		// the reference normalises the stored texel in place on every lookup (objects.cpp:148, a benign race);
		// the oracle defines the lookup as normalise(texel as loaded) -- SURVEY.md 5.
		V3 tn = normalized(o.normalMap[(size_t)y * o.nW + x]);
		// row-vector * [T;B;N;0] with w = 0 (objects.cpp:135-149): the +x[3][j] terms are + 0
		V3 r;
		r.x = tn.x * t.tangent.x + tn.y * t.bitangent.x + tn.z * N.x + 0.0f;
		r.y = tn.x * t.tangent.y + tn.y * t.bitangent.y + tn.z * N.y + 0.0f;
		r.z = tn.x * t.tangent.z + tn.y * t.bitangent.z + tn.z * N.z + 0.0f;
		N = normalized(r);
	}
}

void illuminate(const Light& l, V3 P, V3& L, V3& I, float& dist)   // lights.cpp:18-38
{
	if (l.type == LIGHT_DISTANT) {
		L = l.dir; I = l.color * l.intensity; dist = std::numeric_limits<float>::max();
		return;
	}
	L = P - l.pos;
	I = l.color * fmin_ref(1.0f, (float)(l.intensity / (4 * M_PI * len2(L) / 1000)));
	L = normalized(L);
	dist = length(P - l.pos);
}

void areaPoints(Light& l)                // lights.cpp:46-63
{
	if (!l.points.empty()) return;
	V3 corner = l.pos - (l.ai / 2.0f) - (l.aj / 2.0f);
	if (l.samples > 1) {
		for (int ii = 0; ii < l.samples; ii++)
			for (int jj = 0; jj < l.samples; jj++)
				l.points.push_back(corner + (l.ai * (((float)ii) / (l.samples - 1))) + (l.aj * (((float)jj) / (l.samples - 1))));
	}
	else l.points.push_back(l.pos);
}

V3 castRay(const orc_scene& sc, const Ray& ray, int depth)          // scene.cpp:758-946
{
	if (depth > sc.maxRayDepth) return skyColor(sc, ray.d);
	Hit h;
	if (!trace(sc, ray, h)) return skyColor(sc, ray.d);
	const Object& o = *sc.objects[h.obj];
	V3 objColor = o.color;
	V2 tex; V3 N, hitColor;
	V3 P = ray.o + ray.d * h.t;
	surfaceData(o, P, h.tri, h.uv, N, tex);
	if (o.type == OBJ_MESH && o.hasDiffuse)                            // objects.cpp:153-163
		objColor = o.diffuseMap[(size_t)texel(o.dH, tex.y) * o.dW + texel(o.dW, tex.x)];
	V3 diff, spec;
	V3 L, I;
	const V3 shadowOrig = P + N * sc.bias;
	auto areaIntensity = [&](const Light& l) {
		return l.color * fmin_ref(1.0f, (float)(l.intensity / (4 * M_PI * len2(P - l.pos) / 1000)));
	};
	if (o.material == MAT_DIFFUSE) {                                  // scene.cpp:780-809
		for (auto& lp : sc.lights) {
			Light& l = *lp;
			if (l.type != LIGHT_AREA) {
				Hit sh; illuminate(l, P, L, I, sh.t);
				bool vis = !trace(sc, Ray{ shadowOrig, -L, true }, sh);
				diff = diff + I * (vis * fmax_ref(0.f, dot(N, -L)));
			}
			else {
				float sum = 0;
				I = areaIntensity(l);
				for (const V3& p : l.points) {
					L = P - p;
					Hit sh; sh.t = length(L);
					V3 Ln = normalized(L);        // .normalize() mutates lightDir in place (scene.cpp:802)
					L = Ln;
					bool vis = !trace(sc, Ray{ shadowOrig, -L, true }, sh);
					sum += vis * fmax_ref(0.f, dot(N, -L));
				}
				diff = diff + sum / l.points.size() * I;
			}
		}
		hitColor = objColor * diff;
	}
	else if (o.material == MAT_PHONG) {                               // scene.cpp:810-853
		for (auto& lp : sc.lights) {
			Light& l = *lp;
			if (l.type != LIGHT_AREA) {
				Hit sh; illuminate(l, P, L, I, sh.t);
				bool vis = !trace(sc, Ray{ shadowOrig, -L, true }, sh);
				diff = diff + (float)vis * I * fmax_ref(0.f, dot(N, -L));
				V3 R = reflectDir(L, N);
				spec = spec + (float)vis * I * powfRestated(fmax_ref(0.f, dot(R, -ray.d)), o.nSpecular);
			}
			else {
				I = areaIntensity(l);
				float ssum = 0, dsum = 0;
				for (const V3& p : l.points) {
					L = P - p;
					Hit sh; sh.t = length(L);
					L = normalized(L);
					bool vis = !trace(sc, Ray{ shadowOrig, -L, true }, sh);
					dsum += vis * fmax_ref(0.f, dot(N, -L));
					V3 R = reflectDir(L, N);
					ssum += vis * fmax_ref(0.f, dot(R, -ray.d));
				}
				diff = diff + dsum / l.points.size() * I;
				spec = spec + powfRestated(ssum / l.points.size(), o.nSpecular) * I;
			}
		}
		float sc_ = o.specular;
		if (o.type == OBJ_MESH && o.hasSpecular)                       // objects.cpp:165-175
			sc_ = o.specularMap[(size_t)texel(o.sH, tex.y) * o.sW + texel(o.sW, tex.x)];
		hitColor = objColor * o.ambient + diff * o.diffuse + spec * sc_;
	}
	else {
		// Reflective (scene.cpp:854-891) and Transparent (892-941) share the specular-highlight loop
		auto highlights = [&]() {
			V3 s;
			for (auto& lp : sc.lights) {
				Light& l = *lp;
				if (l.type != LIGHT_AREA) {
					Hit sh; illuminate(l, P, L, I, sh.t);
					bool vis = !trace(sc, Ray{ shadowOrig, -L, true }, sh);
					V3 R = reflectDir(L, N);
					s = s + (float)vis * I * powfRestated(fmax_ref(0.f, dot(R, -ray.d)), o.nSpecular);
				}
				else {
					I = areaIntensity(l);
					float ssum = 0;
					for (const V3& p : l.points) {
						L = P - p;
						Hit sh; sh.t = length(L);
						L = normalized(L);
						bool vis = !trace(sc, Ray{ shadowOrig, -L, true }, sh);
						V3 R = reflectDir(L, N);
						ssum += vis * fmax_ref(0.f, dot(R, -ray.d));
					}
					s = s + powfRestated(ssum / l.points.size(), o.nSpecular) * I;
				}
			}
			return s;
		};
		if (o.material == MAT_REFLECTIVE) {
			Ray rr{ P + sc.bias * N, ray.d - 2 * dot(ray.d, N) * N, false };
			hitColor = 0.8f * castRay(sc, rr, depth + 1);
			spec = highlights();
			hitColor = hitColor + spec;
		}
		else {
			float kr = fresnelKr(ray.d, N, o.ior);
			bool outside = dot(ray.d, N) < 0;
			V3 biasVec = sc.bias * N;
			hitColor = v3(0, 0, 0);
			if (kr < 1) {
				V3 rd = normalized(refractDir(ray.d, N, o.ior));
				V3 ro = outside ? P - biasVec : P + biasVec;
				V3 c = castRay(sc, Ray{ ro, rd, false }, depth + 1);
				hitColor = hitColor + c * (1 - kr);
			}
			V3 fd = normalized(reflectDir(ray.d, N));
			V3 fo = outside ? P + biasVec : P - biasVec;
			V3 c = castRay(sc, Ray{ fo, fd, false }, depth + 1);
			hitColor = hitColor + c * kr;
			spec = highlights();
			hitColor = hitColor + spec * kr;
		}
	}
	return hitColor;
}

// ------------------------------------------------------------------------------------------------
// camera + frame driver (scene.cpp:19-54, 362-379, 444-593)
// ------------------------------------------------------------------------------------------------
void ensureCamera(orc_scene& sc)
{
	if (!sc.camReady) { sc.camMatrix = eulerMatrix(sc.camRot); sc.camReady = true; }
	for (auto& l : sc.lights) if (l->type == LIGHT_AREA) areaPoints(*l);
}

struct View { float scale, aspect, w, h; };
View viewOf(const orc_scene& sc)          // scene.cpp:447-450
{
	View v;
	v.scale = tanf(sc.fov * 0.5f / 180.0f * (float)(M_PI));
	v.aspect = (sc.width) / (float)sc.height;
	v.w = (float)sc.width; v.h = (float)sc.height;
	return v;
}
inline Ray primaryRay(const orc_scene& sc, const View& v, float x, float y)   // scene.cpp:453-457, 52-53
{
	float xp = (2 * (x + 0.5f) / v.w - 1) * v.scale * v.aspect;
	float yp = -(2 * (y + 0.5f) / v.h - 1) * v.scale;
	Ray r; r.o = sc.camPos; r.d = xform(sc.camMatrix, normalized(v3(xp, yp, -1)));
	return r;
}

struct Tile { size_t x0, x1, y0, y1; };
std::vector<Tile> tilesOf(const orc_scene& sc)     // scene.cpp:362-379
{
	const size_t ts = 128;
	std::vector<Tile> out;
	for (size_t i = 0; i < sc.width / ts + 1; i++)
		for (size_t j = 0; j < sc.height / ts + 1; j++) {
			Tile t{ i * ts, (i + 1) * ts, j * ts, (j + 1) * ts };
			if (t.y1 >= sc.height) t.y1 = sc.height - 1;
			if (t.x1 >= sc.width) t.x1 = sc.width - 1;
			if (t.x1 <= t.x0 || t.y1 <= t.y0) continue;
			out.push_back(t);
		}
	return out;
}

template <typename Fn> double runTiles(orc_scene& sc, const std::vector<Tile>& tiles, Fn fn)
{
	// scene.cpp:470-506: one std::thread per tile, at most nWorkers alive, 1 ms polling loop
	auto t0 = std::chrono::high_resolution_clock::now();
	std::atomic<int> running{ 0 };
	std::vector<std::thread> pool;
	size_t next = 0;
	const bool collect = sc.collectStatistics;
	do {
		std::this_thread::sleep_for(std::chrono::milliseconds(1));
		while (next < tiles.size() && running < sc.nWorkers) {
			Tile t = tiles[next++];
			running++;
			pool.emplace_back([&, t]() {
				Stats st;
				tlStats = collect ? &st : nullptr;
				fn(t);
				tlStats = nullptr;
				if (collect) { sc.sRays += st.rays; sc.sBox += st.boxTests; sc.sTri += st.triTests; }
				running--;
			});
		}
	} while (running > 0);
	for (auto& th : pool) th.join();
	return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
}

double renderPass1(orc_scene& sc, float* fbf, size_t yLo, size_t yHi)
{
	ensureCamera(sc);
	V3* fb = reinterpret_cast<V3*>(fbf);
	const View v = viewOf(sc);
	std::vector<Tile> tiles;
	for (Tile t : tilesOf(sc)) {
		if (t.y0 < yLo) t.y0 = yLo;
		if (t.y1 > yHi) t.y1 = yHi;
		if (t.y1 > t.y0) tiles.push_back(t);
	}
	return runTiles(sc, tiles, [&](const Tile& t) {            // scene.cpp:444-468
		for (size_t y = t.y0; y < t.y1; y++)
			for (size_t x = t.x0; x < t.x1; x++)
				fb[x + y * sc.width] = castRay(sc, primaryRay(sc, v, (float)x + 0.5f, (float)y + 0.5f), 0);
	});
}

void sobelMask(const orc_scene& sc, const float* fbf, uint8_t* mask)   // scene.cpp:547-568
{
	const V3* fb = reinterpret_cast<const V3*>(fbf);
	const int W = (int)sc.width, H = (int)sc.height;
	memset(mask, 0, (size_t)W * H);
	const float op[3][3] = { { -1, 0, 1 }, { -2, 0, 2 }, { -1, 0, 1 } };
	for (int i = 1; i < H - 1; i++)
		for (int j = 1; j < W - 1; j++) {
			V3 x, y;
			for (int a = 0; a < 3; a++)
				for (int b = 0; b < 3; b++) {
					const V3& p = fb[(size_t)(i - 1 + a) * W + j - 1 + b];
					x = x + p * op[a][b];
					y = y + p * op[b][a];
				}
			float lx = length(x), ly = length(y);
			float val = sqrtf(lx * lx + ly * ly);       // powf(.,2) is folded to a multiply by g++ -O2
			mask[(size_t)i * W + j] = val > 0.5f ? 1 : 0;
		}
}

double renderSsaa(orc_scene& sc, float* fbf, const uint8_t* mask)     // scene.cpp:508-540
{
	ensureCamera(sc);
	V3* fb = reinterpret_cast<V3*>(fbf);
	const View v = viewOf(sc);
	return runTiles(sc, tilesOf(sc), [&](const Tile& t) {
		for (size_t y = t.y0; y < t.y1; y++)
			for (size_t x = t.x0; x < t.x1; x++) {
				if (!mask[y * sc.width + x]) continue;
				V3 c;
				c = c + castRay(sc, primaryRay(sc, v, (float)x + 0.25f, (float)y + 0.25f), 0);
				c = c + castRay(sc, primaryRay(sc, v, (float)x + 0.25f, (float)y + 0.75f), 0);
				c = c + castRay(sc, primaryRay(sc, v, (float)x + 0.75f, (float)y + 0.25f), 0);
				c = c + castRay(sc, primaryRay(sc, v, (float)x + 0.75f, (float)y + 0.75f), 0);
				fb[x + y * sc.width] = c / 4;
			}
	});
}

struct Dump {
	float* bounds; int32_t* skip; int32_t* leafBegin; int32_t* leafCount; uint32_t* refs;
	int64_t nNodes = 0, nRefs = 0, nLeaves = 0, maxDepth = 0;
	void visit(const BvhNode& n, int depth)
	{
		int64_t me = nNodes++;
		if (depth > maxDepth) maxDepth = depth;
		if (bounds) { float* b = bounds + me * 6; b[0] = n.lo.x; b[1] = n.lo.y; b[2] = n.lo.z; b[3] = n.hi.x; b[4] = n.hi.y; b[5] = n.hi.z; }
		if (n.left) {
			if (leafCount) { leafCount[me] = -1; leafBegin[me] = -1; }
			visit(*n.left, depth + 1); visit(*n.right, depth + 1);
		}
		else {
			nLeaves++;
			if (leafCount) { leafCount[me] = (int32_t)n.tris.size(); leafBegin[me] = (int32_t)nRefs; }
			for (uint32_t id : n.tris) { if (refs) refs[nRefs] = id; nRefs++; }
		}
		if (skip) skip[me] = (int32_t)nNodes;
	}
};

const Object* meshAt(const orc_scene* s, int i)
{
	if (i < 0 || i >= (int)s->objects.size()) return nullptr;
	const Object* o = s->objects[i].get();
	return (o->type == OBJ_MESH && o->root) ? o : nullptr;
}

void bmpBytes(const float* fb, int W, int H, uint8_t* out)            // util.cpp:24-58
{
	uint8_t* hdr = out;
	memset(hdr, 0, 54);
	const uint64_t arraySize = (uint64_t)H * W * 3, total = 54 + arraySize;
	auto put64 = [&](int off, uint64_t v) { memcpy(hdr + off, &v, 8); };   // the overlapping size_t stores
	memcpy(hdr, "BM", 2);
	put64(0x2, total); put64(0xA, 54); put64(0xE, 40); put64(0x12, (uint64_t)W); put64(0x16, (uint64_t)H);
	hdr[0x1A] = 1; hdr[0x1C] = 24;
	put64(0x22, arraySize); put64(0x26, 2835); put64(0x2A, 2835);
	uint8_t* p = out + 54;
	for (int i = 0; i < H; i++)
		for (int j = 0; j < W; j++) {
			const float* px = fb + ((size_t)(H - 1 - i) * W + j) * 3;
			for (int k = 2; k >= 0; k--) *p++ = (uint8_t)(int)(clampf(0.0f, 1.0f, px[k]) * 255);
		}
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C API
// ------------------------------------------------------------------------------------------------
extern "C" {

orc_scene* orc_load(const char* cwd, const char* scene_path, int width, int height)
{
	if (cwd && cwd[0] && chdir(cwd) != 0) { gLastError = "chdir failed"; return nullptr; }
	orc_scene* sc = new orc_scene;
	try { loadScene(*sc, scene_path); }
	catch (const LoadError& e) { gLastError = e.msg; delete sc; return nullptr; }
	catch (const std::exception& e) { gLastError = e.what(); delete sc; return nullptr; }
	if (width > 0) sc->width = (size_t)width;
	if (height > 0) sc->height = (size_t)height;
	return sc;
}
void orc_free(orc_scene* s) { delete s; }
const char* orc_last_error(void) { return gLastError.c_str(); }

void orc_dims(const orc_scene* s, int* w, int* h, int* no, int* nl)
{
	*w = (int)s->width; *h = (int)s->height; *no = (int)s->objects.size(); *nl = (int)s->lights.size();
}
void orc_set_workers(orc_scene* s, int n) { s->nWorkers = n; }
void orc_set_flag(orc_scene* s, const char* name, int v)
{
	if (!strcmp(name, "useBackfaceCulling")) s->useBackfaceCulling = v;
	else if (!strcmp(name, "collectStatistics")) s->collectStatistics = v;
}
void orc_camera(orc_scene* s, float* scale, float* aspect, float* m16, float* pos3)
{
	ensureCamera(*s);
	View v = viewOf(*s);
	*scale = v.scale; *aspect = v.aspect;
	memcpy(m16, s->camMatrix.m, 64);
	pos3[0] = s->camPos.x; pos3[1] = s->camPos.y; pos3[2] = s->camPos.z;
}

double orc_pass1(orc_scene* s, float* fb) { return renderPass1(*s, fb, 0, s->height); }
double orc_pass1_rows(orc_scene* s, float* fb, int y0, int y1) { return renderPass1(*s, fb, (size_t)y0, (size_t)y1); }
void orc_sobel(const orc_scene* s, const float* fb, uint8_t* mask) { sobelMask(*s, fb, mask); }
double orc_ssaa(orc_scene* s, float* fb, const uint8_t* mask) { return renderSsaa(*s, fb, mask); }

void orc_stats_reset(orc_scene* s) { s->sRays = 0; s->sBox = 0; s->sTri = 0; }
void orc_stats(const orc_scene* s, int64_t out[3]) { out[0] = s->sRays; out[1] = s->sBox; out[2] = s->sTri; }

void orc_probe(orc_scene* s, int n, const float* rays, float* out, float* colour)
{
	ensureCamera(*s);
	for (int i = 0; i < n; i++) {
		Ray r; r.o = v3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]); r.d = v3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
		Hit h;
		bool hit = trace(*s, r, h);
		float* o = out + i * 8;
		o[0] = hit ? 1.f : 0.f; o[1] = hit ? (float)h.obj : -1.f;
		o[2] = (hit && s->objects[h.obj]->type == OBJ_MESH) ? (float)h.tri : -1.f;
		o[3] = h.t; o[4] = h.uv.x; o[5] = h.uv.y; o[6] = 0; o[7] = 0;
		V3 c = castRay(*s, r, 0);
		colour[i * 3] = c.x; colour[i * 3 + 1] = c.y; colour[i * 3 + 2] = c.z;
	}
}

void orc_reflect(const float* d, const float* n, float* out)
{
	V3 r = reflectDir(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float* d, const float* n, float ior, float* out)
{
	V3 r = refractDir(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]), ior); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_fresnel(const float* d, const float* n, float ior) { return fresnelKr(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]), ior); }
void orc_skybox(const orc_scene* s, const float* d, float* out)
{
	V3 r = skyColor(*s, v3(d[0], d[1], d[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_normalize(const float* v, float* out)
{
	V3 r = normalized(v3(v[0], v[1], v[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_illuminate(const orc_scene* s, int li, const float* p, float* o)
{
	V3 L, I; float dist = 0;
	// AreaLight::illuminate is not callable in the reference (lights.cpp:65-69): outputs stay zero
	if (s->lights[li]->type != LIGHT_AREA) illuminate(*s->lights[li], v3(p[0], p[1], p[2]), L, I, dist);
	o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = I.x; o[4] = I.y; o[5] = I.z; o[6] = dist; o[7] = 0;
}
float orc_powf(float x, float y) { return powfRestated(x, y); }

int orc_bvh_counts(const orc_scene* s, int obj, int64_t* c)
{
	const Object* m = meshAt(s, obj);
	if (!m) return -1;
	Dump d{ nullptr, nullptr, nullptr, nullptr, nullptr };
	d.visit(*m->root, 1);
	c[0] = d.nNodes; c[1] = d.nLeaves; c[2] = d.nRefs; c[3] = d.maxDepth; c[4] = (int64_t)m->tris.size();
	return 0;
}
int orc_bvh_dump(const orc_scene* s, int obj, float* bounds, int32_t* skip, int32_t* lb, int32_t* lc, uint32_t* refs)
{
	const Object* m = meshAt(s, obj);
	if (!m) return -1;
	Dump d{ bounds, skip, lb, lc, refs };
	d.visit(*m->root, 1);
	return 0;
}
int orc_tris(const orc_scene* s, int obj, float* out)
{
	const Object* m = meshAt(s, obj);
	if (!m) return -1;
	for (size_t i = 0; i < m->tris.size(); i++) {
		const Tri& t = m->tris[i];
		float* o = out + i * 30;
		const V3* v[6] = { &t.a, &t.b, &t.c, &t.na, &t.nb, &t.nc };
		for (int k = 0; k < 6; k++) { o[k * 3] = v[k]->x; o[k * 3 + 1] = v[k]->y; o[k * 3 + 2] = v[k]->z; }
		o[18] = t.ta.x; o[19] = t.ta.y; o[20] = t.tb.x; o[21] = t.tb.y; o[22] = t.tc.x; o[23] = t.tc.y;
		o[24] = t.tangent.x; o[25] = t.tangent.y; o[26] = t.tangent.z;
		o[27] = t.bitangent.x; o[28] = t.bitangent.y; o[29] = t.bitangent.z;
	}
	return 0;
}

void orc_encode_bmp(const float* fb, int W, int H, uint8_t* out) { bmpBytes(fb, W, H, out); }
int orc_save_bmp(const float* fb, int W, int H, const char* path)
{
	if (W % 4 != 0) return -2;     // the reference is only well-defined for W % 4 == 0 (util.cpp:28-29,55-57)
	std::vector<uint8_t> buf(54 + (size_t)3 * W * H);
	bmpBytes(fb, W, H, buf.data());
	FILE* f = fopen(path, "wb");
	if (!f) return -1;
	fwrite(buf.data(), 1, buf.size(), f);
	fclose(f);
	return 0;
}

} // extern "C"
