// Mesh loading and the spatial-split "SAH" acceleration-structure build (reference: src/objects.cpp:177-458,
// 470-526, 633-763), re-implemented over flat arrays.  The build must reproduce the reference's topology and
// bounds bit-for-bit because the kernels traverse exactly these boxes; tests compare it with
// the oracle and with the golden BVH digests of the real reference.
#include "objects.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>

#include "stats.h"
#include "timer.h"
#include "util.h"
#include "../../../include/rtx.h"

template <> Matrix44f Matrix44f::fromEulerDegrees(const Vec3f& rot)
{
	const float a = degToRad(rot.x), b = degToRad(rot.y), c = degToRad(rot.z);
	Matrix44f mx, my, mz;
	mx[1][1] = cosf(a); mx[1][2] = -sinf(a); mx[2][1] = sinf(a); mx[2][2] = cosf(a);
	my[0][0] = cosf(b); my[0][2] = sinf(b); my[2][0] = -sinf(b); my[2][2] = cosf(b);
	mz[0][0] = cosf(c); mz[0][1] = -sinf(c); mz[1][0] = sinf(c); mz[1][1] = cosf(c);
	return mz * my * mx;
}

// ------------------------------------------------------------------------------------------------
// acceleration structure
// ------------------------------------------------------------------------------------------------
namespace {

struct Builder {
	const std::vector<Triangle>& tris;
	const int penalty;
	AccelerationStructure& out;
	// per-axis extent of every triangle: "some vertex <= s" == lo <= s, "some vertex >= s" == hi >= s
	std::vector<float> lo[3], hi[3];

	Builder(const std::vector<Triangle>& t, int p, AccelerationStructure& o) : tris(t), penalty(p), out(o)
	{
		for (int ax = 0; ax < 3; ++ax) {
			lo[ax].resize(t.size()); hi[ax].resize(t.size());
			for (size_t i = 0; i < t.size(); ++i) {
				const float a = t[i].a[ax], b = t[i].b[ax], c = t[i].c[ax];
				lo[ax][i] = std::min(a, std::min(b, c));
				hi[ax][i] = std::max(a, std::max(b, c));
			}
		}
	}

	// nLeft*(s - min) + nRight*(max - s), counts promoted to float (objects.cpp:633-674)
	float cost(int ax, const std::vector<uint32_t>& ids, float mn, float mx, float s) const
	{
		int nl = 0, nr = 0;
		const float* l = lo[ax].data(); const float* h = hi[ax].data();
		for (uint32_t id : ids) { nl += l[id] <= s; nr += h[id] >= s; }
		return nl * (s - mn) + nr * (mx - s);
	}

	// bisection on [left,right] until narrower than 0.1, comparing the cost 0.05 either side of the midpoint
	// (objects.cpp:676-689)
	float split(int ax, const std::vector<uint32_t>& ids, float mn, float mx) const
	{
		float left = mn, right = mx;
		for (;;) {
			const float mid = right - (right - left) / 2;
			if (right - left < 0.1f) return mid;
			if (cost(ax, ids, mn, mx, mid - 0.05f) < cost(ax, ids, mn, mx, mid + 0.05f)) right = mid;
			else left = mid;
		}
	}

	void leaf(int32_t node, const std::vector<uint32_t>& ids)
	{
		out.nodes[node].leafBegin = (int32_t)out.refs.size();
		out.nodes[node].leafCount = (int32_t)ids.size();
		out.refs.insert(out.refs.end(), ids.begin(), ids.end());
		stats::triCopiesCount += ids.size();
	}

	void build(const Vec3f& bmin, const Vec3f& bmax, std::vector<uint32_t>& ids, int depth)
	{
		const int32_t me = (int32_t)out.nodes.size();
		out.nodes.emplace_back();
		out.nodes[me].bounds[0] = bmin; out.nodes[me].bounds[1] = bmax;
		if (depth > out.maxDepth) out.maxDepth = depth;
		stats::acCount++;
		bool isLeaf = ids.size() <= depth * (size_t)penalty;                 // objects.cpp:477
		std::vector<uint32_t> L, R;
		int ax = 2; float s = 0;
		if (!isLeaf) {
			const Vec3f dim = bmax - bmin;                                       // objects.cpp:486-490
			if (dim.x > dim.y && dim.x > dim.z) ax = 0;
			else if (dim.y > dim.z) ax = 1;
			s = split(ax, ids, bmin[ax], bmax[ax]);
			const float* l = lo[ax].data(); const float* h = hi[ax].data();
			for (uint32_t id : ids) {                                            // objects.cpp:737-760
				if (l[id] <= s) L.push_back(id);
				if (h[id] >= s) R.push_back(id);
			}
			isLeaf = L.empty() || R.empty() || (L.size() + R.size() >= ids.size() * 1.5);   // objects.cpp:498
		}
		if (isLeaf) {
			leaf(me, ids);
			out.nodes[me].skip = me + 1;
			return;
		}
		{ std::vector<uint32_t>().swap(ids); }
		Vec3f lmax = bmax, rmin = bmin;                                          // objects.cpp:510-521
		lmax[ax] = s; rmin[ax] = s;
		build(bmin, lmax, L, depth + 1);
		{ std::vector<uint32_t>().swap(L); }
		build(rmin, bmax, R, depth + 1);
		out.nodes[me].skip = (int32_t)out.nodes.size();
	}
};

} // namespace

// Where the structure is built.  options::acBuildOnDevice: 0 host, 1 device, -1 (default) = the faster one: the device
// build (rtx_bvh_build) wins from some tens of thousands of triangles on (250k: scene load 113 ms vs 133 ms with the host
// builder, profiles/r02_bvh_build_time.txt), below that its fixed cost of ~600 small launches makes it the slower one.
// Environment RENDERING_AMD_AC_BUILD=host|device overrides the default.  Both give the same structure bit for bit.
static bool wantDeviceBuild(size_t nTris)
{
	int mode = options::acBuildOnDevice;
	if (mode < 0) {
		const char* e = std::getenv("RENDERING_AMD_AC_BUILD");
		if (e && !strcmp(e, "host")) mode = 0;
		else if (e && !strcmp(e, "device")) mode = 1;
	}
	if (mode >= 0) return mode == 1;
	static int devices = -1;
	if (devices < 0) { int n = 0; devices = (rtx_device_count(&n) == RTX_OK && n > 0) ? n : 0; }
	return devices > 0 && nTris >= 50000;
}

bool AccelerationStructure::setup(const std::vector<Triangle>& tris, const Options& options)
{
	nodes.clear(); refs.clear(); maxDepth = 0; buildMs = 0; builtOnDevice = false;
	const bool onDevice = wantDeviceBuild(tris.size());
	Timer timer(onDevice ? "AC build (device)" : "AC build (host)");
	if (!onDevice) {
		Builder b(tris, options.acPenalty, *this);
		std::vector<uint32_t> ids(tris.size());
		for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)i;
		b.build(rootBounds[0], rootBounds[1], ids, 1);                               // root depth 1 (objects.cpp:389)
		return true;
	}
	// rtx_bvh_build: the same builder, level-synchronous on the GPU (include/rtx.h)
	std::vector<float> pos(tris.size() * 9);
	for (size_t i = 0; i < tris.size(); ++i) {
		const Vec3f* v[3] = { &tris[i].a, &tris[i].b, &tris[i].c };
		for (int k = 0; k < 3; ++k) { pos[i * 9 + k * 3] = v[k]->x; pos[i * 9 + k * 3 + 1] = v[k]->y; pos[i * 9 + k * 3 + 2] = v[k]->z; }
	}
	const float lo[3] = { rootBounds[0].x, rootBounds[0].y, rootBounds[0].z }, hi[3] = { rootBounds[1].x, rootBounds[1].y, rootBounds[1].z };
	rtx_bvh* b = nullptr;
	uint32_t nn = 0, nr = 0, md = 0;
	if (rtx_bvh_build(pos.data(), (uint32_t)tris.size(), lo, hi, options.acPenalty, options::acBuildDevice, &b) != RTX_OK ||
	    rtx_bvh_info(b, &nn, &nr, &md, &buildMs) != RTX_OK) {
		std::cout << "Error: acceleration-structure build on the device failed: " << rtx_last_error() << '\n';
		rtx_bvh_destroy(b);
		return false;
	}
	std::vector<float> bounds((size_t)nn * 6);
	std::vector<int32_t> skip(nn), lb(nn), lc(nn);
	refs.resize(nr);
	if (rtx_bvh_read(b, bounds.data(), skip.data(), lb.data(), lc.data(), refs.data()) != RTX_OK) {
		std::cout << "Error: acceleration-structure read-back failed: " << rtx_last_error() << '\n';
		rtx_bvh_destroy(b);
		return false;
	}
	rtx_bvh_destroy(b);
	nodes.resize(nn);
	for (uint32_t i = 0; i < nn; ++i) {
		nodes[i].bounds[0] = Vec3f(bounds[(size_t)i * 6], bounds[(size_t)i * 6 + 1], bounds[(size_t)i * 6 + 2]);
		nodes[i].bounds[1] = Vec3f(bounds[(size_t)i * 6 + 3], bounds[(size_t)i * 6 + 4], bounds[(size_t)i * 6 + 5]);
		nodes[i].skip = skip[i]; nodes[i].leafBegin = lb[i]; nodes[i].leafCount = lc[i];
	}
	maxDepth = (int)md;
	stats::acCount += nn;
	stats::triCopiesCount += nr;
	builtOnDevice = true;
	return true;
}

size_t AccelerationStructure::leafCount() const
{
	size_t n = 0;
	for (const Node& nd : nodes) n += nd.leafCount >= 0;
	return n;
}

// ------------------------------------------------------------------------------------------------
// OBJ loader
// ------------------------------------------------------------------------------------------------
namespace {

// index reader of the reference (objects.cpp:207-215): skips blanks and one '/', stops at blank, '/' or NUL
size_t readIndex(const char*& p)
{
	size_t v = 0;
	while (*p == ' ') ++p;
	if (*p == '/') ++p;
	for (; *p && *p != ' ' && *p != '/'; ++p) v = v * 10 + (size_t)(*p - '0');
	return v;
}

void faceNormal(Triangle& t) { t.n_a = t.n_b = t.n_c = (t.b - t.a).crossProduct(t.c - t.a); }   // objects.cpp:20

void tangentFrame(Triangle& t)                                                 // objects.cpp:43-55
{
	const Vec3f e1 = t.b - t.a, e2 = t.c - t.a;
	const Vec2f d1 = t.t_b - t.t_a, d2 = t.t_c - t.t_a;
	const float f = 1.0f / (d1.x * d2.y - d2.x * d1.y);
	for (int k = 0; k < 3; ++k) {
		t.tangent[k] = f * (d2.y * e1[k] - d1.y * e2[k]);
		t.bitangent[k] = f * (-d2.x * e1[k] + d1.x * e2[k]);
	}
}

} // namespace

bool Mesh::loadOBJ(const std::string& filename, const Options& options)
{
	const Matrix44f R = Matrix44f::fromEulerDegrees(rot);
	Timer timer("OBJ loading");
	std::ifstream in(filename);
	if (!in.good()) {
		std::cout << "Error, failed to load obj, filename: " << filename << '\n';
		return false;
	}
	if (options::enableOutput) std::cout << "Mesh: " << filename << '\n';
	ac = std::make_unique<AccelerationStructure>();
	std::vector<Vec3f> P, N;
	std::vector<Vec2f> T;
	const float big = std::numeric_limits<float>::max(), tiny = std::numeric_limits<float>::min();
	Vec3f lo(big), hi(tiny);              // `max` starts at the smallest positive float (objects.cpp:231)
	bool placed = false;

	// First face: fit the vertices into `size` (keeping proportions), centre, rotate, translate; pin flat axes;
	// derive the root box from the rotated size vector (objects.cpp:282-331).
	auto place = [&]() {
		const Vec3f range = hi - lo;
		Vec3f ns = size;
		if (!(range.x < options.bias || range.y < options.bias || range.z < options.bias)) {
			const Vec3f stretch = size / range;
			const float m = std::min(stretch.x, std::min(stretch.y, stretch.z));
			if (m == stretch.x) { ns.y = ns.x / (range.x / range.y); ns.z = ns.x / (range.x / range.z); }
			else if (m == stretch.y) { ns.x = ns.y / (range.y / range.x); ns.z = ns.y / (range.y / range.z); }
			else { ns.x = ns.z / (range.z / range.x); ns.y = ns.z / (range.z / range.y); }
		}
		for (Vec3f& v : P) {
			v.x = ns.x * ((v.x - lo.x) / range.x - 0.5f);
			v.y = ns.y * ((v.y - lo.y) / range.y - 0.5f);
			v.z = ns.z * ((v.z - lo.z) / range.z - 0.5f);
			v = R.multVecMatrix(v);
			v.x += pos.x; v.y += pos.y; v.z += pos.z;
			if (range.x < options.bias) v.x = pos.x;
			if (range.y < options.bias) v.y = pos.y;
			if (range.z < options.bias) v.z = pos.z;
		}
		for (Vec3f& n : N) n = R.multVecMatrix(n);
		Vec3f ext = R.multVecMatrix(ns);
		ext = Vec3f(std::fabs(ext.x), std::fabs(ext.y), std::fabs(ext.z));
		ac->setBounds(pos - ext / 2, pos + ext / 2);
	};

	// The file in one read, its lines in place (a 250 000-triangle OBJ is 18 MB and 630 000 lines: getline + sscanf per line took
	// twice as long as the rest of the scene load).  Same tokens as before: a line ends at '\n' (a '\r' stays part of it), a '#' ends
	// its content, the tag is the first blank-delimited word (at most 31 characters) and the fields start strlen(tag) + 1 characters
	// into the line (objects.cpp:236-249); numbers through strtof, which is what sscanf's %f runs.
	std::string text;
	{
		in.seekg(0, std::ios::end);
		const std::streamoff size = in.tellg();
		in.seekg(0, std::ios::beg);
		if (size > 0) { text.resize((size_t)size); in.read(&text[0], size); text.resize((size_t)in.gcount()); }
		text.push_back('\0');      // (the terminator of a last line without a newline)
	}
	auto floats = [](const char* p, float* out, int n) {      // sscanf(p, "%f %f ...") == n
		for (int k = 0; k < n; ++k) {
			char* end = nullptr;
			out[k] = strtof(p, &end);
			if (end == p) return false;
			p = end;
		}
		return true;
	};
	std::vector<size_t> vi, ti, ni;
	char* cur = &text[0];
	char* const fileEnd = cur + text.size() - 1;      // (the terminator is not part of the file)
	while (cur < fileEnd) {
		char* eol = (char*)memchr(cur, '\n', (size_t)(fileEnd - cur));
		if (!eol) eol = fileEnd;
		char* const next = eol < fileEnd ? eol + 1 : fileEnd;
		*eol = '\0';
		char* const lineStart = cur;
		cur = next;
		if (char* hash = strchr(lineStart, '#')) *hash = '\0';
		if (!*lineStart) continue;
		const char* w = lineStart;
		while (isspace((unsigned char)*w)) ++w;
		char tag[32] = { 0 };
		size_t tl = 0;
		while (w[tl] && !isspace((unsigned char)w[tl]) && tl < 31) { tag[tl] = w[tl]; ++tl; }
		const char* lineEnd = lineStart + strlen(lineStart);
		const char* rest = lineStart + tl + 1 <= lineEnd ? lineStart + tl + 1 : lineEnd;
		if (!strcmp(tag, "v")) {
			float c[3];
			if (!floats(rest, c, 3)) LOG_ERROR();
			const float x = c[0], y = c[1], z = c[2];
			lo.x = std::min(x, lo.x); lo.y = std::min(y, lo.y); lo.z = std::min(z, lo.z);
			hi.x = std::max(x, hi.x); hi.y = std::max(y, hi.y); hi.z = std::max(z, hi.z);
			P.emplace_back(x, y, z);
		}
		else if (!strcmp(tag, "vn")) {
			float c[3];
			if (!floats(rest, c, 3)) LOG_ERROR();
			N.push_back(Vec3f(c[0], c[1], c[2]).normalize());
		}
		else if (!strcmp(tag, "vt")) {
			float c[2];
			if (!floats(rest, c, 2)) LOG_ERROR();
			T.emplace_back(c[0], c[1]);
		}
		else if (!strcmp(tag, "f")) {
			if (!placed) { placed = true; place(); }
			int slashes = 0;
			for (const char* p = rest; *p; ++p) slashes += (*p == '/');
			if (slashes != 0 && slashes % 2 != 0) {
				std::cout << "Unhandled slash count: " << slashes << '\n';       // objects.cpp:376-378
				continue;
			}
			vi.clear(); ti.clear(); ni.clear();
			const char* p = rest;
			for (size_t v; (v = readIndex(p)) > 0;) {
				vi.push_back(v);
				if (slashes) {
					const size_t t = readIndex(p), n = readIndex(p);
					if (t > 0) ti.push_back(t);
					if (n > 0) ni.push_back(n);
				}
			}
			// triangle fan (objects.cpp:344-373)
			for (size_t k = 1; k + 1 < vi.size(); ++k) {
				Triangle t;
				t.a = P.at(vi[0] - 1); t.b = P.at(vi[k] - 1); t.c = P.at(vi[k + 1] - 1);
				faceNormal(t);
				if (!ni.empty()) {
					t.n_a = N.at(ni.at(0) - 1); t.n_b = N.at(ni.at(k) - 1); t.n_c = N.at(ni.at(k + 1) - 1);
					if (!ti.empty()) {
						t.t_a = T.at(ti.at(0) - 1); t.t_b = T.at(ti.at(k) - 1); t.t_c = T.at(ti.at(k + 1) - 1);
						tangentFrame(t);
					}
				}
				allTris.push_back(t);
			}
		}
	}
	if (!ac->setup(allTris, options)) return false;
	stats::meshCount += allTris.size();
	return true;
}

// ------------------------------------------------------------------------------------------------
// texture maps (objects.cpp:396-458): byte/256, bottom-up rows kept as loaded
// ------------------------------------------------------------------------------------------------
namespace {
struct Pixels {
	int w = 0, h = 0;
	std::unique_ptr<unsigned char[]> data;
	explicit Pixels(const std::string& fn) { data.reset(loadBMP(fn.c_str(), w, h)); }
	Vec3f at(size_t i) const
	{
		float r = data[i * 3], g = data[i * 3 + 1], b = data[i * 3 + 2];
		r /= 256; g /= 256; b /= 256;
		return Vec3f(r, g, b);
	}
	size_t count() const { return (size_t)w * h; }
};
}

bool Mesh::loadDiffuseMap(const std::string& filename)
{
	if (!options::useTextures) return false;
	Pixels px(filename);
	diffuseMapWidth = px.w; diffuseMapHeight = px.h;
	diffuseMap.resize(px.count());
	for (size_t i = 0; i < px.count(); ++i) diffuseMap[i] = px.at(i);
	return true;
}

bool Mesh::loadNormalMap(const std::string& filename)
{
	if (!options::useTextures) return false;
	Pixels px(filename);
	normalMapWidth = px.w; normalMapHeight = px.h;
	normalMap.resize(px.count());
	for (size_t i = 0; i < px.count(); ++i) {
		const Vec3f c = px.at(i);
		normalMap[i] = Vec3f(c.x * 2 - 1, -(c.y * 2 - 1), c.z).normalize();     // objects.cpp:433
	}
	return true;
}

bool Mesh::loadSpecularMap(const std::string& filename)
{
	if (!options::useTextures) return false;
	Pixels px(filename);
	specularMapWidth = px.w; specularMapHeight = px.h;
	specularMap.resize(px.count());
	for (size_t i = 0; i < px.count(); ++i) {
		const Vec3f c = px.at(i);
		specularMap[i] = (c.x + c.y + c.z) / 3.0f;
	}
	return true;
}
