// Thin C API over the host-side Scene for the Python tests / bench (ctypes).  Plain C types only.
#include <cstring>
#include <string>
#include <unistd.h>

#include "../../../include/rtx.h"
#include "scene.h"
#include "stats.h"
#include "util.h"

namespace {
thread_local std::string gLastError;
// Runs f with LOG_ERROR() throwing instead of exiting; a thrown error is kept for rah_last_error() and `bad` is returned.
template <typename R, typename F> R guarded(R bad, F&& f)
{
	ThrowErrorsScope scope;
	try { gLastError.clear(); return f(); }
	catch (const std::exception& e) { gLastError = e.what(); return bad; }
}
}

extern "C" {

const char* rah_last_error(void) { return gLastError.c_str(); }

// Loads a .scene file (relative asset paths resolve against cwd when given; the process's working directory is
// restored afterwards); w/h > 0 override the file's resolution.  The process-global options:: switches are reset to
// their defaults for the load and the ones the render reads are then pinned on the Scene (Scene::pinFlags), so that
// loading another scene later does not change this one.  NULL on failure (rah_last_error).
void* rah_scene_load(const char* cwd, const char* path, int width, int height)
{
	return guarded<void*>(nullptr, [&]() -> void* {
		options::reset();
		options::enableOutput = false;
		options::imageOutput = false;
		stats::reset();
		char back[4096];
		const bool moved = cwd && cwd[0];
		if (moved && (!getcwd(back, sizeof(back)) || chdir(cwd) != 0)) { gLastError = "cannot change to the scene's base directory"; return nullptr; }
		struct Restore { const char* dir; ~Restore() { if (dir && chdir(dir) != 0) {} } } restore{ moved ? back : nullptr };
		Scene* s = new Scene(path);
		if (!s->sceneLoadSuccess) { delete s; gLastError = std::string("could not load scene ") + path; return nullptr; }
		if (width > 0) s->options.width = (size_t)width;
		if (height > 0) s->options.height = (size_t)height;
		s->pinFlags();
		return s;
	});
}
void rah_scene_free(void* h) { delete (Scene*)h; }

void rah_scene_dims(void* h, int* w, int* ht, int* nObjects, int* nLights)
{
	Scene* s = (Scene*)h;
	*w = (int)s->options.width; *ht = (int)s->options.height; *nObjects = (int)s->objects.size(); *nLights = (int)s->lights.size();
}
void rah_scene_resize(void* h, int w, int ht)
{
	Scene* s = (Scene*)h;
	s->options.width = (size_t)w; s->options.height = (size_t)ht;
	s->invalidateView();
}
// Moves the camera (position, Euler angles in degrees: Camera::pos / Camera::rot, scene.h:52-66); the next use re-applies the view (rtx_scene_set_view).
void rah_camera_set(void* h, const float* pos3, const float* rot3)
{
	Scene* s = (Scene*)h;
	s->camera.pos = Vec3f(pos3[0], pos3[1], pos3[2]); s->camera.rot = Vec3f(rot3[0], rot3[1], rot3[2]);
	s->camera.cameraRotated = false;
	s->invalidateView();
}
void rah_camera_get(void* h, float* pos3, float* rot3)
{
	Scene* s = (Scene*)h;
	pos3[0] = s->camera.pos.x; pos3[1] = s->camera.pos.y; pos3[2] = s->camera.pos.z; rot3[0] = s->camera.rot.x; rot3[1] = s->camera.rot.y; rot3[2] = s->camera.rot.z;
}
void rah_scene_set_device(void* h, int device) { ((Scene*)h)->device = device; }
void rah_set_flag(void* h, const char* name, int v)
{
	Scene* s = (Scene*)h;
	if (!strcmp(name, "useBackfaceCulling")) s->useBackfaceCulling = v != 0;
	else if (!strcmp(name, "collectStatistics")) s->collectStatistics = v != 0;
	else if (!strcmp(name, "useSkybox")) s->useSkybox = v != 0;
	s->invalidateView();
}

// Flattened description (host arrays stay alive until rah_flat_free).
void* rah_flatten(void* h) { return flattenScene(*(Scene*)h); }
const rtx_scene_desc* rah_flat_desc(void* f) { return flatDesc((FlatScene*)f); }
void rah_flat_free(void* f) { freeFlatScene((FlatScene*)f); }

// Where the next rah_scene_load builds acceleration structures: -1 auto, 0 host builder, 1 device (rtx_bvh_build).
void rah_set_ac_build(int mode, int device) { options::acBuildOnDevice = mode; options::acBuildDevice = device; }
// Only the device (one process per GPU: every rank builds on the GPU it renders on, as render_amd does per rank).
void rah_set_ac_build_device(int device) { options::acBuildDevice = device; }

// {built on the device (0/1), device build time in ms}
int rah_bvh_build_info(void* h, int obj, int* onDevice, float* ms)
{
	Scene* s = (Scene*)h;
	if (obj < 0 || obj >= (int)s->objects.size() || s->objects[obj]->objectType != ObjectType::Mesh) return -1;
	const Mesh& m = static_cast<const Mesh&>(*s->objects[obj]);
	if (!m.ac) return -1;
	*onDevice = m.ac->builtOnDevice ? 1 : 0; *ms = m.ac->buildMs;
	return 0;
}

// BVH in the dump layout shared with the oracle / reference harness.
int rah_bvh_counts(void* h, int obj, long long* c)
{
	Scene* s = (Scene*)h;
	if (obj < 0 || obj >= (int)s->objects.size() || s->objects[obj]->objectType != ObjectType::Mesh) return -1;
	const Mesh& m = static_cast<const Mesh&>(*s->objects[obj]);
	if (!m.ac) return -1;
	c[0] = (long long)m.ac->nodes.size(); c[1] = (long long)m.ac->leafCount(); c[2] = (long long)m.ac->refs.size();
	c[3] = m.ac->maxDepth; c[4] = (long long)m.allTris.size();
	return 0;
}

int rah_bvh_dump(void* h, int obj, float* bounds, int32_t* skip, int32_t* leafBegin, int32_t* leafCount, uint32_t* refs)
{
	Scene* s = (Scene*)h;
	if (obj < 0 || obj >= (int)s->objects.size() || s->objects[obj]->objectType != ObjectType::Mesh) return -1;
	const Mesh& m = static_cast<const Mesh&>(*s->objects[obj]);
	if (!m.ac) return -1;
	for (size_t i = 0; i < m.ac->nodes.size(); ++i) {
		const auto& n = m.ac->nodes[i];
		float* b = bounds + i * 6;
		b[0] = n.bounds[0].x; b[1] = n.bounds[0].y; b[2] = n.bounds[0].z; b[3] = n.bounds[1].x; b[4] = n.bounds[1].y; b[5] = n.bounds[1].z;
		skip[i] = n.skip; leafBegin[i] = n.leafBegin; leafCount[i] = n.leafCount;
	}
	memcpy(refs, m.ac->refs.data(), m.ac->refs.size() * sizeof(uint32_t));
	return 0;
}

// The host builder (AccelerationStructure::setup, objects.cpp:385-392 + 470-526 + 633-763 of the reference restated) on a bare triangle array:
// pos = n x 9 floats (a, b, c), lo / hi = the root box, in the dump layout of rah_bvh_dump.  counts = {nodes, leaves, refs, maxDepth};
// the arrays may be NULL to ask for the counts only (the structure is built twice then).  For the parity tests on the reference's own models
// where the OBJ files are not at hand (tests/test_ref_models.py: the golden file keeps the triangles the reference's loader produced).
int rah_bvh_from_tris(const float* pos, int n, const float* lo, const float* hi, int acPenalty, long long* counts,
                      float* bounds, int32_t* skip, int32_t* leafBegin, int32_t* leafCount, uint32_t* refs)
{
	return guarded<int>(-1, [&]() -> int {
		std::vector<Triangle> tris((size_t)n);
		for (int i = 0; i < n; ++i) {
			tris[i].a = Vec3f(pos[i * 9], pos[i * 9 + 1], pos[i * 9 + 2]); tris[i].b = Vec3f(pos[i * 9 + 3], pos[i * 9 + 4], pos[i * 9 + 5]);
			tris[i].c = Vec3f(pos[i * 9 + 6], pos[i * 9 + 7], pos[i * 9 + 8]);
		}
		Options o; o.acPenalty = acPenalty;
		// the host builder, quietly; both process-wide options are put back on EVERY way out (setup may throw: guarded<> catches it -- ADVICE r5)
		struct Restore {
			decltype(options::acBuildOnDevice) keep = options::acBuildOnDevice; const bool quiet = options::enableOutput;
			Restore() { options::acBuildOnDevice = 0; options::enableOutput = false; }
			~Restore() { options::acBuildOnDevice = keep; options::enableOutput = quiet; }
		};
		AccelerationStructure ac;
		ac.setBounds(Vec3f(lo[0], lo[1], lo[2]), Vec3f(hi[0], hi[1], hi[2]));
		bool ok;
		{ Restore restore; ok = ac.setup(tris, o); }
		if (!ok) return -1;
		counts[0] = (long long)ac.nodes.size(); counts[1] = (long long)ac.leafCount(); counts[2] = (long long)ac.refs.size(); counts[3] = ac.maxDepth;
		if (!bounds) return 0;
		for (size_t i = 0; i < ac.nodes.size(); ++i) {
			const auto& nd = ac.nodes[i];
			float* b = bounds + i * 6;
			b[0] = nd.bounds[0].x; b[1] = nd.bounds[0].y; b[2] = nd.bounds[0].z; b[3] = nd.bounds[1].x; b[4] = nd.bounds[1].y; b[5] = nd.bounds[1].z;
			skip[i] = nd.skip; leafBegin[i] = nd.leafBegin; leafCount[i] = nd.leafCount;
		}
		memcpy(refs, ac.refs.data(), ac.refs.size() * sizeof(uint32_t));
		return 0;
	});
}

// 30 floats per triangle: a b c n_a n_b n_c | t_a t_b t_c | tangent bitangent
int rah_tris(void* h, int obj, float* out)
{
	Scene* s = (Scene*)h;
	if (obj < 0 || obj >= (int)s->objects.size() || s->objects[obj]->objectType != ObjectType::Mesh) return -1;
	const Mesh& m = static_cast<const Mesh&>(*s->objects[obj]);
	for (size_t i = 0; i < m.allTris.size(); ++i) {
		const Triangle& t = m.allTris[i];
		float* o = out + i * 30;
		const Vec3f* v[6] = { &t.a, &t.b, &t.c, &t.n_a, &t.n_b, &t.n_c };
		for (int k = 0; k < 6; ++k) { o[k * 3] = v[k]->x; o[k * 3 + 1] = v[k]->y; o[k * 3 + 2] = v[k]->z; }
		o[18] = t.t_a.x; o[19] = t.t_a.y; o[20] = t.t_b.x; o[21] = t.t_b.y; o[22] = t.t_c.x; o[23] = t.t_c.y;
		o[24] = t.tangent.x; o[25] = t.tangent.y; o[26] = t.tangent.z; o[27] = t.bitangent.x; o[28] = t.bitangent.y; o[29] = t.bitangent.z;
	}
	return 0;
}

// camera constants exactly as uploaded (rtx_view::scale/aspect/cam_matrix/cam_pos)
void rah_camera(void* h, float* scale, float* aspect, float* m16, float* pos3)
{
	guarded<int>(-1, [&] {
		FlatScene* f = flattenScene(*(Scene*)h);
		const rtx_view& v = flatDesc(f)->view;
		*scale = v.scale; *aspect = v.aspect;
		memcpy(m16, v.cam_matrix, 64); memcpy(pos3, v.cam_pos, 12);
		freeFlatScene(f);
		return 0;
	});
}

// rtx_view::flags this scene uploads (RTX_FLAG_*), -1 on error
int rah_view_flags(void* h)
{
	return guarded<int>(-1, [&] {
		FlatScene* f = flattenScene(*(Scene*)h);
		const int flags = (int)flatDesc(f)->view.flags;
		freeFlatScene(f);
		return flags;
	});
}

// loadBMP (util.h) into a caller buffer of `cap` bytes; returns the number of bytes the image needs (3*w*h).
// (-1: unreadable / unsupported file, rah_last_error; -2: the image needs more than INT_MAX bytes)
int rah_load_bmp(const char* path, int* w, int* h, unsigned char* out, int cap)
{
	return guarded<int>(-1, [&] {
		unsigned char* d = loadBMP(path, *w, *h);
		const size_t need = (size_t)3 * (size_t)*w * (size_t)*h;
		if (out && cap >= 0 && (size_t)cap >= need) memcpy(out, d, need);
		delete[] d;
		return need > 0x7fffffffu ? -2 : (int)need;
	});
}

// Numeric digest of objects and lights (for loader tests): per object 16 floats
// [type, material, pos3, color3, ior, ambient, diffuse, specular, nSpecular, r2 | normal.x, ...], per light 12 floats.
int rah_scene_digest(void* h, float* out, int maxFloats)
{
	Scene* s = (Scene*)h;
	FlatScene* f = flattenScene(*s);
	const rtx_scene_desc* d = flatDesc(f);
	int n = 0;
	auto put = [&](float v) { if (n < maxFloats) out[n] = v; n++; };
	for (uint32_t i = 0; i < d->n_objects; i++) {
		const rtx_object& o = d->objects[i];
		put((float)o.type); put((float)o.material);
		for (int k = 0; k < 3; k++) put(o.pos[k]);
		for (int k = 0; k < 3; k++) put(o.color[k]);
		put(o.ior); put(o.ambient); put(o.diffuse); put(o.specular); put(o.n_specular); put(o.radius2);
		for (int k = 0; k < 3; k++) put(o.normal[k]);
	}
	for (uint32_t i = 0; i < d->n_lights; i++) {
		const rtx_light& l = d->lights[i];
		put((float)l.type);
		for (int k = 0; k < 3; k++) put(l.color[k]);
		put(l.intensity);
		for (int k = 0; k < 3; k++) put(l.dir[k]);
		for (int k = 0; k < 3; k++) put(l.pos[k]);
		put((float)l.n_points);
	}
	freeFlatScene(f);
	return n;
}

// The uploaded GPU scene (created on first use); NULL when it cannot be created, e.g. without a GPU (rah_last_error).
rtx_scene* rah_scene_gpu(void* h) { return guarded<rtx_scene*>(nullptr, [&] { return ((Scene*)h)->gpu(); }); }

// Whole-frame convenience mirroring Scene::render() into a host buffer (H*W*3 floats, zero-initialised by caller).
int rah_render_host(void* h, float* fb, int withSsaa)
{
	return guarded<int>(-1, [&] {
		Scene* s = (Scene*)h;
		s->launchWorkers((Vec3f*)fb);
		if (withSsaa) s->launchSSAA((Vec3f*)fb);
		return 0;
	});
}

int rah_save_bmp(void* h, const float* fb, const char* nameNoExt)
{
	Scene* s = (Scene*)h;
	s->options.imageName = nameNoExt;
	return saveImage((const Vec3f*)fb, s->options);
}

} // extern "C"
