// TEST INFRASTRUCTURE -- built in the build container only (needs /root/reference), never on the product path.
//
// Thin C-ABI harness around the *real* holoskii/Rendering translation units.  It is compiled by
// oracle/Makefile against the sources where they lie (/root/reference/src/{scene,objects,lights,util}.cpp,
// headers from /root/reference/include) into oracle/_ref/libref_harness.so.  Nothing from the reference
// is copied: this file only calls its public API (scene.h:31-100, objects.h:69-164).
//
// What travels: the reference's SOURCES never leave /root/reference; the BUILT files under oracle/_ref/ (git-ignored, not
// gpurun-ignored) travel to the GPU box with the snapshot, as the task's rules for a compilable reference prescribe, and are used
// there for exactly two things -- (1) bench.py's cpu_baseline leg times this library (kind "reference") and compares the timed GPU
// frame with its framebuffers, (2) tests/test_gpu_ref_binding.py runs oracle/_ref/ref_binding (the reference's Scene class driving
// the GPU through the binding of INTEGRATION.md).  Here, in the build container, tools/make_golden.py uses it to (a) validate
// oracle/rt_oracle.cpp bit-for-bit and (b) emit the committed golden vectors under tests/golden/.  (SURVEY.md 8c planned for no
// reference binary on the GPU box; the compiled reference is the better baseline and checker, so it ships -- DESIGN.md section 4.)
#include "scene.h"
#include "stats.h"
#include "options.h"
#include "util.h"

#include <cstdint>
#include <cstring>
#include <unistd.h>
#include <unordered_map>

namespace {

void resetGlobals()
{
	// include/options.h:23-37 defaults; quiet + no image pop-up for harness use
	options::outputProgress = false;
	options::useBackfaceCulling = true;
	options::collectStatistics = false;
	options::enableOutput = false;
	options::imageOutput = false;
	options::useAC = true;
	options::showAC = false;
	options::useSkybox = false;
	options::useTextures = true;
	options::showNormals = false;
	options::enableSSAA = true;
}

struct Walk {
	std::unordered_map<const Triangle*, uint32_t> triIndex;
	float* bounds = nullptr;     // 6 per node, pre-order (left first)
	int32_t* skip = nullptr;     // pre-order index just past the subtree
	int32_t* leafBegin = nullptr;
	int32_t* leafCount = nullptr; // -1 for inner nodes
	uint32_t* refs = nullptr;
	int64_t nNodes = 0, nRefs = 0, nLeaves = 0, maxDepth = 0;

	void visit(const AccelerationStructure* n, int depth)
	{
		const int64_t me = nNodes++;
		if (depth > maxDepth) maxDepth = depth;
		if (bounds) {
			bounds[me * 6 + 0] = n->bounds[0].x; bounds[me * 6 + 1] = n->bounds[0].y; bounds[me * 6 + 2] = n->bounds[0].z;
			bounds[me * 6 + 3] = n->bounds[1].x; bounds[me * 6 + 4] = n->bounds[1].y; bounds[me * 6 + 5] = n->bounds[1].z;
		}
		if (n->left) {
			if (leafCount) { leafCount[me] = -1; leafBegin[me] = -1; }
			visit(n->left.get(), depth + 1);
			visit(n->right.get(), depth + 1);
		}
		else {
			nLeaves++;
			if (leafCount) { leafCount[me] = (int32_t)n->tris.size(); leafBegin[me] = (int32_t)nRefs; }
			for (const Triangle* t : n->tris) {
				if (refs) refs[nRefs] = triIndex.at(t);
				nRefs++;
			}
		}
		if (skip) skip[me] = (int32_t)nNodes;
	}
};

const Mesh* meshAt(const Scene* s, int objIdx)
{
	if (objIdx < 0 || objIdx >= (int)s->objects.size()) return nullptr;
	const Object* o = s->objects[objIdx].get();
	if (o->objectType != ObjectType::Mesh) return nullptr;
	return static_cast<const Mesh*>(o);
}

} // namespace

extern "C" {

// cwd: directory the scene's relative asset paths resolve against (may be NULL).
void* ref_load(const char* cwd, const char* scenePath, int width, int height)
{
	resetGlobals();
	if (cwd && cwd[0]) { if (chdir(cwd) != 0) return nullptr; }
	Scene* s = new Scene(scenePath);
	if (width > 0) s->options.width = (size_t)width;
	if (height > 0) s->options.height = (size_t)height;
	return s;
}

void ref_set_flag(const char* name, int v)
{
	if (!strcmp(name, "useBackfaceCulling")) options::useBackfaceCulling = v;
	else if (!strcmp(name, "collectStatistics")) options::collectStatistics = v;
	else if (!strcmp(name, "useTextures")) options::useTextures = v;
	else if (!strcmp(name, "useSkybox")) options::useSkybox = v;
}

void ref_dims(void* h, int* w, int* ht, int* nObjects, int* nLights)
{
	Scene* s = (Scene*)h;
	*w = (int)s->options.width; *ht = (int)s->options.height;
	*nObjects = (int)s->objects.size(); *nLights = (int)s->lights.size();
}

void ref_set_workers(void* h, int n) { ((Scene*)h)->options.nWorkers = n; }

// Camera constants exactly as the workers compute them (scene.cpp:447-448) + lazily built rMatrix.
void ref_camera(void* h, float* scale, float* aspect, float* rMatrix16, float* pos3)
{
	Scene* s = (Scene*)h;
	*scale = tanf(s->camera.fov * 0.5f / 180.0f * (float)(M_PI));
	*aspect = (s->options.width) / (float)s->options.height;
	s->camera.getRay(0.f, 0.f);
	for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) rMatrix16[i * 4 + j] = s->camera.rMatrix[i][j];
	pos3[0] = s->camera.pos.x; pos3[1] = s->camera.pos.y; pos3[2] = s->camera.pos.z;
}

// fb must be zero-initialised H*W*3 floats, like `new Vec3f[H*W]` (scene.cpp:599).
// The reference fills an area light's sample points lazily, from whichever worker thread shades with it first (scene.cpp:794 / 830 / 873 / 923 ->
// AreaLight::setPoints, lights.cpp:46-63: `pointsCreated = true` BEFORE the vector is filled) while the other workers already read them: with the 256 workers
// of a GPU box's host the first pixels of their tiles -- now and then tens of thousands of pixels -- come out wrong, differently in every run.  The harness
// makes the same call before the workers start: the same points, no race.
static void areaLightPointsFirst(Scene* s)
{
	for (auto& l : s->lights)
		if (l->type == LightType::AreaLight) static_cast<AreaLight*>(l.get())->setPoints();
}
void ref_pass1(void* h, float* fb) { areaLightPointsFirst((Scene*)h); ((Scene*)h)->launchWorkers((Vec3f*)fb); }
void ref_ssaa(void* h, float* fb) { areaLightPointsFirst((Scene*)h); ((Scene*)h)->launchSSAA((Vec3f*)fb); }

void ref_stats_reset()
{
	stats::rayTriTests = 0; stats::accelStructTests = 0; stats::raysCasted = 0;
}
void ref_stats(int64_t* out3)
{
	out3[0] = stats::raysCasted; out3[1] = stats::accelStructTests; out3[2] = stats::rayTriTests;
}

// Per-ray probes. rays: n x 6 (orig, dir). out: n x 8 floats:
// [hit, objIdx, triIdx(-1 if none), t, u, v, 0, 0]; colour: n x 3 (castRay at depth 0).
void ref_probe(void* h, int n, const float* rays, float* out, float* colour)
{
	Scene* s = (Scene*)h;
	std::unordered_map<const Object*, int> objIndex;
	for (size_t i = 0; i < s->objects.size(); i++) objIndex[s->objects[i].get()] = (int)i;
	std::unordered_map<const Triangle*, int> triIndex;
	for (auto& o : s->objects) {
		if (o->objectType != ObjectType::Mesh) continue;
		const Mesh* m = static_cast<const Mesh*>(o.get());
		for (size_t i = 0; i < m->allTris.size(); i++) triIndex[m->allTris[i]] = (int)i;
	}
	for (int i = 0; i < n; i++) {
		Ray ray{ Vec3f(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), Vec3f(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]) };
		IntersectInfo info;
		bool hit = Render::trace(ray, s->objects, info);
		float* o = out + i * 8;
		o[0] = hit ? 1.f : 0.f;
		o[1] = hit ? (float)objIndex[info.hitObject] : -1.f;
		o[2] = (hit && info.hitObject->objectType == ObjectType::Mesh) ? (float)triIndex[info.triPtr] : -1.f;
		o[3] = info.tNear; o[4] = info.uv.x; o[5] = info.uv.y; o[6] = 0; o[7] = 0;
		Vec3f c = Render::castRay(ray, *s, 0);
		colour[i * 3] = c.x; colour[i * 3 + 1] = c.y; colour[i * 3 + 2] = c.z;
	}
}

// Unit probes of the shading helpers (scene.cpp:672-722, 381-442).
void ref_reflect(const float* d, const float* n, float* out)
{
	Vec3f r = Render::reflect(Vec3f(d[0], d[1], d[2]), Vec3f(n[0], n[1], n[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_refract(const float* d, const float* n, float ior, float* out)
{
	Vec3f r = Render::refract(Vec3f(d[0], d[1], d[2]), Vec3f(n[0], n[1], n[2]), ior);
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_fresnel(const float* d, const float* n, float ior)
{
	return Render::fresnel(Vec3f(d[0], d[1], d[2]), Vec3f(n[0], n[1], n[2]), ior);
}
void ref_skybox(void* h, const float* d, float* out)
{
	Vec3f r = ((Scene*)h)->getSkybox(Vec3f(d[0], d[1], d[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_normalize(const float* v, float* out)
{
	Vec3f r = Vec3f(v[0], v[1], v[2]).normalize();
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_illuminate(void* h, int lightIdx, const float* p, float* out8)
{
	Scene* s = (Scene*)h;
	Vec3f L, I; float dist = 0;
	s->lights[lightIdx]->illuminate(Vec3f(p[0], p[1], p[2]), L, I, dist);
	out8[0] = L.x; out8[1] = L.y; out8[2] = L.z; out8[3] = I.x; out8[4] = I.y; out8[5] = I.z; out8[6] = dist; out8[7] = 0;
}

// BVH of the mesh at object index objIdx.  counts: [nNodes, nLeaves, nRefs, maxDepth, nTris].
int ref_bvh_counts(void* h, int objIdx, int64_t* counts)
{
	const Mesh* m = meshAt((Scene*)h, objIdx);
	if (!m) return -1;
	Walk w;
	for (size_t i = 0; i < m->allTris.size(); i++) w.triIndex[m->allTris[i]] = (uint32_t)i;
	w.visit(m->ac.get(), 1);
	counts[0] = w.nNodes; counts[1] = w.nLeaves; counts[2] = w.nRefs; counts[3] = w.maxDepth; counts[4] = (int64_t)m->allTris.size();
	return 0;
}

int ref_bvh_dump(void* h, int objIdx, float* bounds, int32_t* skip, int32_t* leafBegin, int32_t* leafCount, uint32_t* refs)
{
	const Mesh* m = meshAt((Scene*)h, objIdx);
	if (!m) return -1;
	Walk w;
	for (size_t i = 0; i < m->allTris.size(); i++) w.triIndex[m->allTris[i]] = (uint32_t)i;
	w.bounds = bounds; w.skip = skip; w.leafBegin = leafBegin; w.leafCount = leafCount; w.refs = refs;
	w.visit(m->ac.get(), 1);
	return 0;
}

// Triangle records, 30 floats each: a b c n_a n_b n_c (18) t_a t_b t_c (6) tangent bitangent (6).
int ref_tris(void* h, int objIdx, float* out)
{
	const Mesh* m = meshAt((Scene*)h, objIdx);
	if (!m) return -1;
	for (size_t i = 0; i < m->allTris.size(); i++) {
		const Triangle* t = m->allTris[i];
		float* o = out + i * 30;
		const Vec3f* v3[6] = { &t->a, &t->b, &t->c, &t->n_a, &t->n_b, &t->n_c };
		for (int k = 0; k < 6; k++) { o[k * 3] = v3[k]->x; o[k * 3 + 1] = v3[k]->y; o[k * 3 + 2] = v3[k]->z; }
		o[18] = t->t_a.x; o[19] = t->t_a.y; o[20] = t->t_b.x; o[21] = t->t_b.y; o[22] = t->t_c.x; o[23] = t->t_c.y;
		o[24] = t->tangent.x; o[25] = t->tangent.y; o[26] = t->tangent.z;
		o[27] = t->bitangent.x; o[28] = t->bitangent.y; o[29] = t->bitangent.z;
	}
	return 0;
}

// The reference BMP writer (util.cpp:15-76); imageName is taken without the ".bmp" suffix.
int ref_save(void* h, float* fb, const char* nameNoExt)
{
	Scene* s = (Scene*)h;
	s->options.imageName = nameNoExt;
	return saveImage((Vec3f*)fb, s->options);
}

float ref_powf(float x, float y) { return powf(x, y); }

} // extern "C"
