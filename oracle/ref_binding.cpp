// TEST INFRASTRUCTURE -- built in the build container only (needs /root/reference), never on the product path.  The built binary ships to the GPU box
// as a checker (see `render` below); the reference's sources never do.
//
// The binding of INTEGRATION.md, COMPILED AGAINST THE REAL REFERENCE: this file includes the reference's own headers
// (/root/reference/include/scene.h:68-100, objects.h:24-200, lights.h:21-73, options.h:9-37), is linked with the reference's own
// translation units (oracle/_ref/*.o, built by oracle/Makefile from the sources where they lie) and with this repo's
// librtx_hip.so, and fills an rtx_scene_desc from a Scene the REFERENCE's loader produced.  Nothing of the reference is copied:
// only its public members are read.  What a maintainer pastes into src/scene.cpp is the part between the two BINDING markers;
// INTEGRATION.md quotes it from here.
//
//   oracle/_ref/ref_binding dump <cwd> <scene> <width> <height> <out.bin>
//       loads the scene with the reference's Scene(path), fills the description exactly as uploadScene() does and writes its
//       canonical bytes (rtx_desc_serialize, include/rtx_debug.h).  tests/test_ref_binding.py compares them with the bytes of
//       the description this repo's own host builds for the same file (rendering_amd/host/src/scene.cpp, flattenScene).
//   oracle/_ref/ref_binding render <cwd> <scene> <width> <height> <out.bmp>      (needs a GPU)
//       the reference's Scene(path) -> uploadScene -> rtxLaunchWorkers + rtxLaunchSSAA -> the reference's own saveImage writes <out.bmp>;
//       then rtxRender (the one-call form), whose bytes must equal the file's.  The BINARY travels to the GPU box with the snapshot (oracle/_ref/ is
//       git-ignored, not gpurun-ignored; /root/reference itself never travels): tests/test_gpu_ref_binding.py runs it there -- the drop-in end to end.
#include "scene.h"
#include "stats.h"
#include "timer.h"
#include "options.h"
#include "util.h"

// ---- BINDING (begin) -----------------------------------------------------------------------------------------------------
#include "rtx.h"            // this repo: include/rtx.h ; link with -lrtx_hip
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <memory>
#include <unordered_map>
#include <vector>

// Walks the reference's pointer tree in ITS visiting order (left, then right: objects.cpp:601-616).
static void flattenAC(const AccelerationStructure* n, const std::unordered_map<const Triangle*, uint32_t>& id,
                      std::vector<float>& bounds, std::vector<int32_t>& skip, std::vector<int32_t>& leafBegin,
                      std::vector<int32_t>& leafCount, std::vector<uint32_t>& refs)
{
	const size_t me = skip.size();
	for (int k = 0; k < 2; k++) { bounds.push_back(n->bounds[k].x); bounds.push_back(n->bounds[k].y); bounds.push_back(n->bounds[k].z); }
	skip.push_back(0); leafBegin.push_back(-1); leafCount.push_back(-1);
	if (n->left) {
		flattenAC(n->left.get(), id, bounds, skip, leafBegin, leafCount, refs);
		flattenAC(n->right.get(), id, bounds, skip, leafBegin, leafCount, refs);
	}
	else {
		leafBegin[me] = (int32_t)refs.size(); leafCount[me] = (int32_t)n->tris.size();
		for (const Triangle* t : n->tris) refs.push_back(id.at(t));
	}
	skip[me] = (int32_t)skip.size();
}

// The host arrays an rtx_scene_desc points into (rtx_scene_create copies everything to the device and never reads them again).
struct RtxUpload {
	struct MeshArrays { std::vector<float> bounds, pos, nrm, uv, tb; std::vector<int32_t> skip, leafBegin, leafCount; std::vector<uint32_t> refs; };
	rtx_scene_desc d{};
	std::vector<rtx_object> objects;
	std::vector<rtx_mesh> meshes;
	std::vector<rtx_light> lights;
	std::vector<std::unique_ptr<MeshArrays>> arrays;
	std::vector<std::unique_ptr<std::vector<float>>> lightPoints;
};

static void put3(float* dst, const Vec3f& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

static void fillView(Scene& s, rtx_view& v)
{
	s.camera.getRay(0, 0);                                         // builds rMatrix (scene.cpp:22-49)
	v.width = (uint32_t)s.options.width; v.height = (uint32_t)s.options.height;
	v.bias = s.options.bias; v.max_ray_depth = s.options.maxRayDepth;
	put3(v.background, s.options.backgroundColor);
	v.flags = (options::useBackfaceCulling ? RTX_FLAG_BACKFACE_CULL : 0u) | (options::useSkybox ? RTX_FLAG_SKYBOX : 0u);
	put3(v.cam_pos, s.camera.pos);
	for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) v.cam_matrix[i * 4 + j] = s.camera.rMatrix[i][j];
	v.scale = tanf(s.camera.fov * 0.5f / 180.0f * (float)(M_PI));  // scene.cpp:447
	v.aspect = (s.options.width) / (float)s.options.height;        // scene.cpp:448
}

static void fillDesc(Scene& s, RtxUpload& u)
{
	fillView(s, u.d.view);
	for (auto& op : s.objects) {                                   // scene order = Render::trace's iteration order (scene.cpp:731)
		const Object& o = *op;
		rtx_object ro{};
		ro.type = o.objectType == ObjectType::Sphere ? RTX_OBJ_SPHERE : (o.objectType == ObjectType::Plane ? RTX_OBJ_PLANE : RTX_OBJ_MESH);
		ro.material = (int32_t)o.materialType;                     // Diffuse 0, Reflective 1, Transparent 2, Phong 3 (objects.h:19)
		put3(ro.pos, o.pos); put3(ro.color, o.color);
		ro.ior = o.indexOfRefraction; ro.ambient = o.ambient; ro.diffuse = o.diffuse; ro.specular = o.specular; ro.n_specular = o.nSpecular;
		ro.mesh = -1;
		if (o.objectType == ObjectType::Sphere) ro.radius2 = static_cast<const Sphere&>(o).r2;
		else if (o.objectType == ObjectType::Plane) put3(ro.normal, static_cast<const Plane&>(o).normal);      // as stored: not re-normalised (scene.cpp:300)
		else {
			const Mesh& m = static_cast<const Mesh&>(o);
			ro.mesh = (int32_t)u.meshes.size();
			u.arrays.push_back(std::make_unique<RtxUpload::MeshArrays>());
			RtxUpload::MeshArrays& A = *u.arrays.back();
			std::unordered_map<const Triangle*, uint32_t> id;
			const size_t nt = m.allTris.size();
			A.pos.resize(nt * 9); A.nrm.resize(nt * 9); A.uv.resize(nt * 6); A.tb.resize(nt * 6);
			for (size_t i = 0; i < nt; i++) {
				const Triangle& t = *m.allTris[i];
				id[&t] = (uint32_t)i;
				put3(&A.pos[i * 9], t.a); put3(&A.pos[i * 9 + 3], t.b); put3(&A.pos[i * 9 + 6], t.c);
				put3(&A.nrm[i * 9], t.n_a); put3(&A.nrm[i * 9 + 3], t.n_b); put3(&A.nrm[i * 9 + 6], t.n_c);
				A.uv[i * 6] = t.t_a.x; A.uv[i * 6 + 1] = t.t_a.y; A.uv[i * 6 + 2] = t.t_b.x; A.uv[i * 6 + 3] = t.t_b.y; A.uv[i * 6 + 4] = t.t_c.x; A.uv[i * 6 + 5] = t.t_c.y;
				put3(&A.tb[i * 6], t.tangent); put3(&A.tb[i * 6 + 3], t.bitangent);
			}
			flattenAC(m.ac.get(), id, A.bounds, A.skip, A.leafBegin, A.leafCount, A.refs);
			rtx_mesh rm{};
			rm.n_nodes = (uint32_t)A.skip.size(); rm.n_refs = (uint32_t)A.refs.size(); rm.n_tris = (uint32_t)nt;
			rm.node_bounds = A.bounds.data(); rm.node_skip = A.skip.data(); rm.leaf_begin = A.leafBegin.data(); rm.leaf_count = A.leafCount.data();
			rm.refs = A.refs.data();
			rm.tri_pos = A.pos.data(); rm.tri_nrm = A.nrm.data(); rm.tri_uv = A.uv.data(); rm.tri_tb = A.tb.data();
			// the maps as loaded: Vec3f[] / float[] of byte / 256 (objects.cpp:396-458); Vec3f is three packed floats
			if (m.diffuseMapLoaded) { rm.diffuse_w = (uint32_t)m.diffuseMapWidth; rm.diffuse_h = (uint32_t)m.diffuseMapHeight; rm.diffuse_map = &m.diffuseMap[0].x; }
			if (m.normalMapLoaded) { rm.normal_w = (uint32_t)m.normalMapWidth; rm.normal_h = (uint32_t)m.normalMapHeight; rm.normal_map = &m.normalMap[0].x; }
			if (m.specularMapLoaded) { rm.specular_w = (uint32_t)m.specularMapWidth; rm.specular_h = (uint32_t)m.specularMapHeight; rm.specular_map = m.specularMap; }
			u.meshes.push_back(rm);
		}
		u.objects.push_back(ro);
	}
	for (auto& lp : s.lights) {
		rtx_light rl{};
		put3(rl.color, lp->color); rl.intensity = lp->intensity;
		if (lp->type == LightType::DistantLight) { rl.type = RTX_LIGHT_DISTANT; put3(rl.dir, static_cast<DistantLight&>(*lp).dir); }      // as stored (scene.cpp:222)
		else if (lp->type == LightType::PointLight) { rl.type = RTX_LIGHT_POINT; put3(rl.pos, static_cast<PointLight&>(*lp).pos); }
		else if (lp->type == LightType::AreaLight) {
			AreaLight& al = static_cast<AreaLight&>(*lp);
			al.setPoints();                                        // what castRay does on first use (scene.cpp:794; lights.cpp:46-63)
			rl.type = RTX_LIGHT_AREA; put3(rl.pos, al.pos);
			u.lightPoints.push_back(std::make_unique<std::vector<float>>(al.points.size() * 3));
			for (size_t i = 0; i < al.points.size(); i++) put3(&(*u.lightPoints.back())[i * 3], al.points[i]);
			rl.n_points = (uint32_t)al.points.size(); rl.points = u.lightPoints.back()->data();
		}
		else LOG_ERROR();
		u.lights.push_back(rl);
	}
	u.d.n_objects = (uint32_t)u.objects.size(); u.d.objects = u.objects.data();
	u.d.n_meshes = (uint32_t)u.meshes.size(); u.d.meshes = u.meshes.data();
	u.d.n_lights = (uint32_t)u.lights.size(); u.d.lights = u.lights.data();
	if (options::useSkybox && s.skyboxes[0]) {                     // loadSkybox (scene.cpp:335-360): six faces of skyboxWidth x skyboxHeight
		u.d.sky_w = (uint32_t)s.skyboxWidth; u.d.sky_h = (uint32_t)s.skyboxHeight;
		for (int k = 0; k < 6; k++) u.d.sky[k] = &s.skyboxes[k][0].x;
	}
}

static rtx_scene* uploadScene(Scene& s)      // once, after loadScene()
{
	RtxUpload u;
	fillDesc(s, u);
	rtx_scene* g = nullptr;
	if (rtx_scene_create(&u.d, /*device*/0, &g) != RTX_OK) { std::cout << rtx_last_error() << '\n'; LOG_ERROR(); }
	return g;
}

// Scene::launchWorkers(Vec3f*) (scene.cpp:470-506): was a thread per 128 x 128 tile
static void rtxLaunchWorkers(Scene& s, rtx_scene* g, Vec3f* frameBuffer)
{
	Timer t("Render scene");
	const size_t bytes = s.options.width * s.options.height * sizeof(Vec3f);
	float* fb = nullptr;
	if (hipMalloc((void**)&fb, bytes) != hipSuccess || hipMemcpy(fb, frameBuffer, bytes, hipMemcpyHostToDevice) != hipSuccess) LOG_ERROR();
	if (rtx_render_pass1(g, 0, (uint32_t)s.options.height, fb, nullptr) != RTX_OK) { std::cout << rtx_last_error() << '\n'; LOG_ERROR(); }
	if (hipMemcpy(frameBuffer, fb, bytes, hipMemcpyDeviceToHost) != hipSuccess) LOG_ERROR();
	(void)hipFree(fb);
}

// Scene::launchSSAA(Vec3f*) (scene.cpp:542-593): was a single-threaded Sobel + a thread per tile
static void rtxLaunchSSAA(Scene& s, rtx_scene* g, Vec3f* frameBuffer)
{
	Timer t1("MSAA");
	const size_t px = s.options.width * s.options.height, bytes = px * sizeof(Vec3f);
	float* fb = nullptr; uint8_t* mask = nullptr;
	if (hipMalloc((void**)&fb, bytes) != hipSuccess || hipMalloc((void**)&mask, px) != hipSuccess || hipMemcpy(fb, frameBuffer, bytes, hipMemcpyHostToDevice) != hipSuccess) LOG_ERROR();
	if (rtx_sobel(g, fb, 0, (uint32_t)s.options.height, mask, nullptr) != RTX_OK || rtx_render_ssaa(g, mask, 0, (uint32_t)s.options.height, fb, nullptr) != RTX_OK) {
		std::cout << rtx_last_error() << '\n'; LOG_ERROR();
	}
	if (hipMemcpy(frameBuffer, fb, bytes, hipMemcpyDeviceToHost) != hipSuccess) LOG_ERROR();
	(void)hipFree(fb); (void)hipFree(mask);
}

// Scene::render() (scene.cpp:595-657) where it is under the maintainer's control as well: both launchers as ONE call, the frame never leaves the
// device between them, and saveImage's bytes (util.cpp:46-58: bottom-up BGR, truncating quantiser) come back at 3 bytes per pixel.
static void rtxRender(Scene& s, rtx_scene* g, std::vector<uint8_t>& bgrBottomUp)
{
	const uint32_t W = (uint32_t)s.options.width, H = (uint32_t)s.options.height;
	const size_t px = (size_t)W * H;
	float* fb = nullptr; uint8_t* mask = nullptr; uint8_t* bgr = nullptr;
	if (hipMalloc((void**)&fb, px * 12) != hipSuccess || hipMalloc((void**)&mask, px) != hipSuccess || hipMalloc((void**)&bgr, px * 3) != hipSuccess) LOG_ERROR();
	if (hipMemset(fb, 0, px * 12) != hipSuccess) LOG_ERROR();                                        // new Vec3f[H * W]() (scene.cpp:599)
	if (rtx_render_frame(g, 0, H, fb, mask, nullptr) != RTX_OK) { std::cout << rtx_last_error() << '\n'; LOG_ERROR(); }   // launchWorkers + launchSSAA
	uint32_t status = 0;
	if (rtx_frame_status(g, &status) != RTX_OK) { std::cout << rtx_last_error() << '\n'; LOG_ERROR(); }                   // synchronises
	if (rtx_quantize_bgr8(g, fb, bgr, nullptr) != RTX_OK) LOG_ERROR();
	bgrBottomUp.resize(px * 3);
	if (hipMemcpy(bgrBottomUp.data(), bgr, px * 3, hipMemcpyDeviceToHost) != hipSuccess) LOG_ERROR();
	(void)hipFree(fb); (void)hipFree(mask); (void)hipFree(bgr);
}
// ---- BINDING (end) -------------------------------------------------------------------------------------------------------

#include "rtx_debug.h"      // rtx_desc_serialize
#include <cstdio>
#include <cstring>
#include <unistd.h>

int main(int argc, char** argv)
{
	if (argc != 7 || (strcmp(argv[1], "dump") && strcmp(argv[1], "render"))) {
		fprintf(stderr, "usage: ref_binding dump|render <cwd> <scene> <width|0> <height|0> <out>\n");
		return 2;
	}
	// options.h:23-37 defaults, quiet, no image pop-up
	options::outputProgress = false; options::useBackfaceCulling = true; options::collectStatistics = false; options::enableOutput = false;
	options::imageOutput = false; options::useAC = true; options::showAC = false; options::useSkybox = false; options::useTextures = true;
	options::showNormals = false; options::enableSSAA = true;
	char back[4096];
	if (!getcwd(back, sizeof(back))) return 3;
	const std::string out = argv[6][0] == '/' ? std::string(argv[6]) : std::string(back) + "/" + argv[6];
	if (argv[2][0] && chdir(argv[2]) != 0) { fprintf(stderr, "cannot chdir to %s\n", argv[2]); return 3; }
	Scene scene(argv[3]);
	if (!scene.sceneLoadSuccess) { fprintf(stderr, "the reference's loader rejected %s\n", argv[3]); return 4; }
	if (atoi(argv[4]) > 0) scene.options.width = (size_t)atoi(argv[4]);
	if (atoi(argv[5]) > 0) scene.options.height = (size_t)atoi(argv[5]);
	if (!strcmp(argv[1], "dump")) {
		RtxUpload u;
		fillDesc(scene, u);
		size_t need = 0;
		if (rtx_desc_serialize(&u.d, nullptr, 0, &need) != RTX_OK) { fprintf(stderr, "%s\n", rtx_last_error()); return 5; }
		std::vector<char> buf(need);
		if (rtx_desc_serialize(&u.d, buf.data(), buf.size(), &need) != RTX_OK) return 5;
		FILE* f = fopen(out.c_str(), "wb");
		if (!f || fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 6; }
		fclose(f);
		return 0;
	}
	// render: the reference's own Scene::render() sequence (scene.cpp:595-657) with the two launchers replaced by the binding's -- a zeroed framebuffer,
	// rtxLaunchWorkers, rtxLaunchSSAA, then THE REFERENCE'S saveImage (util.cpp:15-76) writes <out> -- and, beside it, the one-call form rtxRender, whose
	// BGR bytes must be the pixel bytes of the file saveImage wrote (exit code 7 otherwise).
	if (out.size() < 5 || out.substr(out.size() - 4) != ".bmp") { fprintf(stderr, "render: <out> must end in .bmp\n"); return 2; }
	rtx_scene* g = uploadScene(scene);
	std::vector<Vec3f> fbHost(scene.options.width * scene.options.height, Vec3f(0, 0, 0));      // new Vec3f[H * W]() (scene.cpp:599)
	rtxLaunchWorkers(scene, g, fbHost.data());
	rtxLaunchSSAA(scene, g, fbHost.data());
	Options opts = scene.options;
	opts.imageName = out.substr(0, out.size() - 4);                // saveImage appends ".bmp" (util.cpp:17)
	if (saveImage(fbHost.data(), opts) != 0) return 6;
	std::vector<uint8_t> bgr;
	rtxRender(scene, g, bgr);
	{
		FILE* f = fopen(out.c_str(), "rb");
		if (!f) return 6;
		std::vector<uint8_t> file(54 + bgr.size());
		const size_t got = fread(file.data(), 1, file.size(), f);
		fclose(f);
		if (got != file.size() || memcmp(file.data() + 54, bgr.data(), bgr.size()) != 0) { fprintf(stderr, "rtxRender's bytes differ from the file saveImage wrote\n"); return 7; }
	}
	rtx_scene_destroy(g);
	return 0;
}
