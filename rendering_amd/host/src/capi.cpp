// Thin C API over the host-side Scene for the Python tests / bench (ctypes).  Plain C types only.
#include <cstring>
#include <string>
#include <unistd.h>

#include "../../../include/rtx.h"
#include "scene.h"
#include "stats.h"
#include "util.h"

extern "C" {

// Loads a .scene file (relative asset paths resolve against cwd when given); w/h > 0 override the file's
// resolution.  Process-global options:: flags are reset to their defaults first.
void* rah_scene_load(const char* cwd, const char* path, int width, int height)
{
	options::reset();
	options::enableOutput = false;
	options::imageOutput = false;
	stats::reset();
	if (cwd && cwd[0] && chdir(cwd) != 0) return nullptr;
	Scene* s = new Scene(path);
	if (width > 0) s->options.width = (size_t)width;
	if (height > 0) s->options.height = (size_t)height;
	return s;
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
void rah_scene_set_device(void* h, int device) { ((Scene*)h)->device = device; }
void rah_set_flag(void* h, const char* name, int v)
{
	if (!strcmp(name, "useBackfaceCulling")) options::useBackfaceCulling = v;
	else if (!strcmp(name, "collectStatistics")) options::collectStatistics = v;
	((Scene*)h)->invalidateView();
}

// Flattened description (host arrays stay alive until rah_flat_free).
void* rah_flatten(void* h) { return flattenScene(*(Scene*)h); }
const rtx_scene_desc* rah_flat_desc(void* f) { return flatDesc((FlatScene*)f); }
void rah_flat_free(void* f) { freeFlatScene((FlatScene*)f); }

// Where the next rah_scene_load builds acceleration structures: -1 auto, 0 host builder, 1 device (rtx_bvh_build).
void rah_set_ac_build(int mode, int device) { options::acBuildOnDevice = mode; options::acBuildDevice = device; }

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
	FlatScene* f = flattenScene(*(Scene*)h);
	const rtx_view& v = flatDesc(f)->view;
	*scale = v.scale; *aspect = v.aspect;
	memcpy(m16, v.cam_matrix, 64); memcpy(pos3, v.cam_pos, 12);
	freeFlatScene(f);
}

// loadBMP (util.h) into a caller buffer of `cap` bytes; returns the number of bytes the image needs (3*w*h).
int rah_load_bmp(const char* path, int* w, int* h, unsigned char* out, int cap)
{
	unsigned char* d = loadBMP(path, *w, *h);
	const int need = 3 * *w * *h;
	if (out && cap >= need) memcpy(out, d, (size_t)need);
	delete[] d;
	return need;
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

// The uploaded GPU scene (created on first use; exits through LOG_ERROR when no GPU is available).
rtx_scene* rah_scene_gpu(void* h) { return ((Scene*)h)->gpu(); }

// Whole-frame convenience mirroring Scene::render() into a host buffer (H*W*3 floats, zero-initialised by caller).
void rah_render_host(void* h, float* fb, int withSsaa)
{
	Scene* s = (Scene*)h;
	s->launchWorkers((Vec3f*)fb);
	if (withSsaa) s->launchSSAA((Vec3f*)fb);
}

int rah_save_bmp(void* h, const float* fb, const char* nameNoExt)
{
	Scene* s = (Scene*)h;
	s->options.imageName = nameNoExt;
	return saveImage((const Vec3f*)fb, s->options);
}

} // extern "C"
