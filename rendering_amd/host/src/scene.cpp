// Scene loading, flattening and frame orchestration of the MI355X renderer (reference: src/scene.cpp:57-360,
// 362-379, 470-657).  The reference's two worker loops -- Scene::launchWorkers / Scene::launchSSAA -- are
// replaced by launches of the gfx950 kernels through the C ABI (include/rtx.h); there is no CPU rendering
// path in this library.
#include "scene.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "../../../include/rtx.h"
#include "stats.h"
#include "timer.h"
#include "util.h"

void Camera::ensureMatrix()
{
	if (cameraRotated) return;
	rMatrix = Matrix44f::fromEulerDegrees(rot);
	cameraRotated = true;
}

// ------------------------------------------------------------------------------------------------
// .scene parser (scene.cpp:62-334): [options] / [light] / [object] blocks of key=value lines
// ------------------------------------------------------------------------------------------------
namespace {

bool contains(const std::string& s, const char* what) { return s.find(what) != std::string::npos; }

struct KeyValue { std::string key, value; };

KeyValue splitKeyValue(const std::string& line, bool stripBlanks)
{
	const size_t eq = line.find('=');
	if (eq == std::string::npos) LOG_ERROR();
	KeyValue kv{ line.substr(0, eq), line.substr(eq + 1) };
	if (stripBlanks) kv.key.erase(std::remove_if(kv.key.begin(), kv.key.end(), [](char c) { return c == ' ' || c == '\t'; }), kv.key.end());
	return kv;
}

void parseOption(Scene& sc, const KeyValue& kv)
{
	const std::string& k = kv.key; const std::string& v = kv.value;
	if (k == "outputProgress") options::outputProgress = strToBool(v);
	else if (k == "useBackfaceCulling") options::useBackfaceCulling = strToBool(v);
	else if (k == "collectStatistics") options::collectStatistics = strToBool(v);
	else if (k == "enableOutput") options::enableOutput = strToBool(v);
	else if (k == "imageOutput") options::imageOutput = strToBool(v);
	else if (k == "useAC") options::useAC = strToBool(v);
	else if (k == "showAC") options::showAC = strToBool(v);
	else if (k == "useSkybox") options::useSkybox = strToBool(v);
	else if (k == "useTextures") options::useTextures = strToBool(v);
	else if (k == "showNormals") options::showNormals = strToBool(v);
	else if (k == "width") sc.options.width = strToInt(v);
	else if (k == "height") sc.options.height = strToInt(v);
	else if (k == "fov") sc.camera.fov = strToFloat(v);
	else if (k == "image_name") {
		sc.options.imageName = v;
		// legacy files quote the name
		sc.options.imageName.erase(std::remove(sc.options.imageName.begin(), sc.options.imageName.end(), '"'), sc.options.imageName.end());
	}
	else if (k == "n_workers") sc.options.nWorkers = strToInt(v);
	else if (k == "max_ray_depth") sc.options.maxRayDepth = strToInt(v);
	else if (k == "ac_penalty") sc.options.acPenalty = strToInt(v);
	else if (k == "background_color") sc.options.backgroundColor = str3ToFloat(splitString(v, ','));
	else if (k == "position") sc.camera.pos = str3ToFloat(splitString(v, ','));
	else if (k == "rotation") sc.camera.rot = str3ToFloat(splitString(v, ','));
	else if (k == "skyboxes") {
		const auto names = splitString(v, ',');
		if (names.size() < 6) LOG_ERROR();
		for (int i = 0; i < 6; ++i) {
			strncpy(sc.options.skyboxNames[i], names[i].c_str(), 63);
			sc.options.skyboxNames[i][63] = 0;
		}
		options::useSkybox = true;
	}
	else std::cout << "Scene, unknown key: " << k << '\n';
}

void parseLight(std::unique_ptr<Light>& light, const KeyValue& kv)
{
	const std::string& k = kv.key; const std::string& v = kv.value;
	if (k == "type") {
		if (v == "distant") light = std::make_unique<DistantLight>();
		else if (v == "point") light = std::make_unique<PointLight>();
		else if (v == "area") light = std::make_unique<AreaLight>();
		return;
	}
	if (!light) { std::cout << "Error, light type missing\n"; return; }
	auto need = [&](LightType t) { if (light->type != t) LOG_ERROR(); };
	if (k == "color") light->color = str3ToFloat(splitString(v, ','));
	else if (k == "intensity") light->intensity = strToFloat(v);
	else if (k == "direction") { need(LightType::DistantLight); static_cast<DistantLight&>(*light).dir = str3ToFloat(splitString(v, ',')); }
	else if (k == "position") { need(LightType::PointLight); static_cast<PointLight&>(*light).pos = str3ToFloat(splitString(v, ',')); }
	else if (k == "pos") { need(LightType::AreaLight); static_cast<AreaLight&>(*light).pos = str3ToFloat(splitString(v, ',')); }
	else if (k == "i") { need(LightType::AreaLight); static_cast<AreaLight&>(*light).i = str3ToFloat(splitString(v, ',')); }
	else if (k == "j") { need(LightType::AreaLight); static_cast<AreaLight&>(*light).j = str3ToFloat(splitString(v, ',')); }
	else if (k == "samples") { need(LightType::AreaLight); static_cast<AreaLight&>(*light).samples = strToInt(v); }
}

void parseObject(Scene& sc, std::unique_ptr<Object>& object, const KeyValue& kv)
{
	const std::string& k = kv.key; const std::string& v = kv.value;
	if (k == "type") {
		if (v == "plane") object = std::make_unique<Plane>();
		else if (v == "sphere") object = std::make_unique<Sphere>();
		else if (v == "mesh") object = std::make_unique<Mesh>();
		return;
	}
	if (!object) { std::cout << "Error, object type missing\n"; return; }
	if (k == "color") object->color = str3ToFloat(splitString(v, ','));
	else if (k == "pos") object->pos = str3ToFloat(splitString(v, ','));
	else if (k == "material") {
		const auto parts = splitString(v, ',');
		if (parts.empty()) LOG_ERROR();
		if (parts[0] == "transparent") {
			object->materialType = MaterialType::Transparent;
			object->indexOfRefraction = strToFloat(parts.at(1));
		}
		else if (parts[0] == "reflective") object->materialType = MaterialType::Reflective;
		else if (parts[0] == "phong") {
			object->materialType = MaterialType::Phong;
			object->ambient = strToFloat(parts.at(1)); object->diffuse = strToFloat(parts.at(2));
			object->specular = strToFloat(parts.at(3)); object->nSpecular = strToFloat(parts.at(4));
		}
	}
	else if (object->objectType == ObjectType::Sphere) {
		if (k == "radius") {
			auto& s = static_cast<Sphere&>(*object);
			s.r = strToFloat(v);
			s.r2 = s.r * s.r;          // powf(r, 2) is folded to r*r by the reference build (scene.cpp:294)
		}
	}
	else if (object->objectType == ObjectType::Plane) {
		if (k == "normal") static_cast<Plane&>(*object).normal = str3ToFloat(splitString(v, ','));
	}
	else if (object->objectType == ObjectType::Mesh) {
		auto& m = static_cast<Mesh&>(*object);
		if (k == "size") m.size = str3ToFloat(splitString(v, ','));
		else if (k == "rot") m.rot = str3ToFloat(splitString(v, ','));
		else if (k == "name") m.loadOBJ(v, sc.options);
		else if (k == "diffuse_map") m.diffuseMapLoaded = m.loadDiffuseMap(v);
		else if (k == "normal_map") m.normalMapLoaded = m.loadNormalMap(v);
		else if (k == "specular_map") m.specularMapLoaded = m.loadSpecularMap(v);
	}
}

// ---- legacy dialect (hardening) ------------------------------------------------
// The reference ships input/smooth_shading.scene in an older one-line-per-entity form that its current loader
// rejects (LOG_ERROR at scene.cpp:200-201):
//     [light]   name, point_light|distant_light, x,y,z, r,g,b, intensity
//     [object]  name, plane,  px,py,pz, nx,ny,nz, r,g,b, material
//               name, sphere, px,py,pz, radius,   r,g,b, material
//               name, mesh, "file.obj", px,py,pz, sx,sy,sz [, r,g,b [, material]]
// material = none | diffuse | reflective | transparent[:ior] | phong:ambient:diffuse:specular:n
// A line without '=' inside [light]/[object] is read this way and becomes one entity.
std::vector<std::string> legacyFields(const std::string& line)
{
	std::vector<std::string> out;
	for (std::string cell : splitString(line, ',')) {
		cell.erase(std::remove_if(cell.begin(), cell.end(), [](char c) { return c == ' ' || c == '\t' || c == '"' || c == '\r'; }), cell.end());
		out.push_back(cell);
	}
	while (!out.empty() && out.back().empty()) out.pop_back();
	return out;
}

Vec3f legacy3(const std::vector<std::string>& f, size_t at)
{
	if (at + 3 > f.size()) LOG_ERROR();
	return Vec3f(strToFloat(f[at]), strToFloat(f[at + 1]), strToFloat(f[at + 2]));
}

void legacyMaterial(Object& o, const std::string& token)
{
	const auto parts = splitString(token, ':');
	if (parts.empty() || parts[0] == "none" || parts[0] == "diffuse") return;
	if (parts[0] == "reflective") o.materialType = MaterialType::Reflective;
	else if (parts[0] == "transparent") {
		o.materialType = MaterialType::Transparent;
		if (parts.size() > 1) o.indexOfRefraction = strToFloat(parts[1]);
	}
	else if (parts[0] == "phong") {
		if (parts.size() < 5) LOG_ERROR();
		o.materialType = MaterialType::Phong;
		o.ambient = strToFloat(parts[1]); o.diffuse = strToFloat(parts[2]); o.specular = strToFloat(parts[3]); o.nSpecular = strToFloat(parts[4]);
	}
	else LOG_ERROR();
}

void parseLegacyLight(Scene& sc, const std::string& line)
{
	const auto f = legacyFields(line);
	if (f.size() < 9) LOG_ERROR();
	std::unique_ptr<Light> l;
	if (f[1] == "point_light" || f[1] == "point") { auto p = std::make_unique<PointLight>(); p->pos = legacy3(f, 2); l = std::move(p); }
	else if (f[1] == "distant_light" || f[1] == "distant") { auto p = std::make_unique<DistantLight>(); p->dir = legacy3(f, 2); l = std::move(p); }
	else LOG_ERROR();
	l->color = legacy3(f, 5);
	l->intensity = strToFloat(f[8]);
	sc.lights.push_back(std::move(l));
}

void parseLegacyObject(Scene& sc, const std::string& line)
{
	const auto f = legacyFields(line);
	if (f.size() < 2) LOG_ERROR();
	if (f[1] == "plane") {
		if (f.size() < 11) LOG_ERROR();
		auto p = std::make_unique<Plane>();
		p->pos = legacy3(f, 2); p->normal = legacy3(f, 5); p->color = legacy3(f, 8);
		if (f.size() > 11) legacyMaterial(*p, f[11]);
		sc.objects.push_back(std::move(p));
	}
	else if (f[1] == "sphere") {
		if (f.size() < 9) LOG_ERROR();
		auto p = std::make_unique<Sphere>();
		p->pos = legacy3(f, 2); p->r = strToFloat(f[5]); p->r2 = p->r * p->r; p->color = legacy3(f, 6);
		if (f.size() > 9) legacyMaterial(*p, f[9]);
		sc.objects.push_back(std::move(p));
	}
	else if (f[1] == "mesh") {
		if (f.size() < 9) LOG_ERROR();
		auto p = std::make_unique<Mesh>();
		p->pos = legacy3(f, 3); p->size = legacy3(f, 6);
		if (f.size() >= 12) p->color = legacy3(f, 9);
		if (f.size() > 12) legacyMaterial(*p, f[12]);
		p->loadOBJ(f[2], sc.options);
		sc.objects.push_back(std::move(p));
	}
	else LOG_ERROR();
}

} // namespace

Scene::Scene(const std::string& sceneName) { sceneLoadSuccess = loadScene(sceneName); }


bool Scene::loadScene(const std::string& scenePath)
{
	if (options::enableOutput) std::cout << "Loading scene " << scenePath << '\n';
	const unsigned hw = std::thread::hardware_concurrency();
	if (hw != 0) options.nWorkers = (int)hw;
	std::ifstream in(scenePath);
	if (!in.good()) {
		std::cout << "Could not open scene file: " << scenePath << '\n';
		LOG_ERROR();
	}
	enum class Block { None, Options, Light, Object } block = Block::None;
	std::unique_ptr<Light> light;
	std::unique_ptr<Object> object;
	bool legacyBlock = false;     // the current [light]/[object] block held legacy one-line entities
	// a block is committed when the next line containing '[' is read (scene.cpp:96-107)
	auto commit = [&]() {
		if (block == Block::Light) { if (light) lights.push_back(std::move(light)); else if (!legacyBlock) LOG_ERROR(); }
		else if (block == Block::Object) { if (object) objects.push_back(std::move(object)); else if (!legacyBlock) LOG_ERROR(); }
		legacyBlock = false;
	};
	std::string line;
	while (in.good()) {
		std::getline(in, line);
		if (line.empty()) continue;
		if (contains(line, "[")) commit();
		if (contains(line, "#[")) {                        // commented-out block: skip to the next live header
			do { std::getline(in, line); } while (in.good() && (!contains(line, "[") || contains(line, "#[")));
			if (!in.good()) break;
		}
		if (contains(line, "#")) line.erase(line.find('#'));
		if (line.empty()) continue;
		if (line[0] == '[') {
			if (line == "[options]") block = Block::Options;
			else if (line == "[light]") block = Block::Light;
			else if (line == "[object]") block = Block::Object;
			else if (line == "[end]") break;
			else LOG_ERROR();
			continue;
		}
		switch (block) {
		case Block::Options: parseOption(*this, splitKeyValue(line, true)); break;
		case Block::Light:
			if (!contains(line, "=") && contains(line, ",")) { parseLegacyLight(*this, line); legacyBlock = true; }
			else parseLight(light, splitKeyValue(line, false));
			break;
		case Block::Object:
			if (!contains(line, "=") && contains(line, ",")) { parseLegacyObject(*this, line); legacyBlock = true; }
			else parseObject(*this, object, splitKeyValue(line, false));
			break;
		case Block::None: break;
		}
	}
	if (options::useSkybox) loadSkybox();
	return true;
}

void Scene::loadSkybox()
{
	if (!options::useSkybox) return;
	for (int k = 0; k < 6; ++k) {
		int w = 0, h = 0;
		std::unique_ptr<unsigned char[]> px(loadBMP(options.skyboxNames[k], w, h));
		if (k > 0 && (w != skyboxWidth || h != skyboxHeight)) {      // the lookup uses one size for all six faces (scene.cpp:383-385)
			std::cout << "Skybox faces differ in size: " << options.skyboxNames[k] << '\n';
			noteError("skybox faces differ in size");
			LOG_ERROR();
		}
		skyboxWidth = w; skyboxHeight = h;
		skyboxes[k].resize((size_t)w * h);
		for (size_t i = 0; i < skyboxes[k].size(); ++i) {
			float r = px[i * 3], g = px[i * 3 + 1], b = px[i * 3 + 2];
			r /= 256; g /= 256; b /= 256;                  // scene.cpp:354
			skyboxes[k][i] = Vec3f(r, g, b);
		}
	}
}

std::vector<tileInfo> Scene::getTiles()
{
	// 128x128 tiles, column-major, x1/y1 clamped to W-1/H-1 (scene.cpp:362-379).  The GPU path does not use
	// them for scheduling (it shards by rows and renders 8x8 wave tiles); they define which pixels exist.
	const size_t ts = 128;
	std::vector<tileInfo> tiles;
	for (size_t i = 0; i < options.width / ts + 1; ++i)
		for (size_t j = 0; j < options.height / ts + 1; ++j) {
			tileInfo t{ i * ts, std::min((i + 1) * ts, options.width - 1), j * ts, std::min((j + 1) * ts, options.height - 1) };
			if (t.x1 > t.x0 && t.y1 > t.y0) tiles.push_back(t);
		}
	return tiles;
}

// ------------------------------------------------------------------------------------------------
// flattening: Scene -> rtx_scene_desc (the arrays stay alive inside FlatScene)
// ------------------------------------------------------------------------------------------------
struct FlatScene {
	rtx_scene_desc desc{};
	std::vector<rtx_object> objects;
	std::vector<rtx_mesh> meshes;
	std::vector<rtx_light> lights;
	struct MeshArrays {
		std::vector<float> bounds, pos, nrm, uv, tb, diffuse, normal, specular;
		std::vector<int32_t> skip, leafBegin, leafCount;
	};
	std::vector<std::unique_ptr<MeshArrays>> meshArrays;
	std::vector<std::vector<float>> lightPoints;
};

namespace {

void put3(float* d, const Vec3f& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

void fillView(Scene& sc, rtx_view& v)
{
	sc.camera.ensureMatrix();
	v.width = (uint32_t)sc.options.width; v.height = (uint32_t)sc.options.height;
	v.bias = sc.options.bias; v.max_ray_depth = sc.options.maxRayDepth;
	put3(v.background, sc.options.backgroundColor);
	v.flags = (sc.cullingOn() ? RTX_FLAG_BACKFACE_CULL : 0u) | (sc.skyboxOn() ? RTX_FLAG_SKYBOX : 0u);
	put3(v.cam_pos, sc.camera.pos);
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) v.cam_matrix[i * 4 + j] = sc.camera.rMatrix[i][j];
	v.scale = tanf(sc.camera.fov * 0.5f / 180.0f * (float)(3.14159265358979323846));   // scene.cpp:447
	v.aspect = (sc.options.width) / (float)sc.options.height;                          // scene.cpp:448
}

} // namespace

FlatScene* flattenScene(Scene& sc)
{
	auto* fs = new FlatScene;
	fillView(sc, fs->desc.view);
	for (auto& op : sc.objects) {
		const Object& o = *op;
		rtx_object ro{};
		ro.type = o.objectType == ObjectType::Sphere ? RTX_OBJ_SPHERE : (o.objectType == ObjectType::Plane ? RTX_OBJ_PLANE : RTX_OBJ_MESH);
		ro.material = (int32_t)o.materialType;
		put3(ro.pos, o.pos); put3(ro.color, o.color);
		ro.ior = o.indexOfRefraction; ro.ambient = o.ambient; ro.diffuse = o.diffuse; ro.specular = o.specular; ro.n_specular = o.nSpecular;
		ro.mesh = -1;
		if (o.objectType == ObjectType::Sphere) ro.radius2 = static_cast<const Sphere&>(o).r2;
		else if (o.objectType == ObjectType::Plane) put3(ro.normal, static_cast<const Plane&>(o).normal);
		else {
			const Mesh& m = static_cast<const Mesh&>(o);
			if (!m.ac) { std::cout << "Mesh without acceleration structure (OBJ failed to load)\n"; LOG_ERROR(); }
			ro.mesh = (int32_t)fs->meshes.size();
			fs->meshArrays.push_back(std::make_unique<FlatScene::MeshArrays>());
			auto& A = *fs->meshArrays.back();
			const auto& nodes = m.ac->nodes;
			A.bounds.resize(nodes.size() * 6); A.skip.resize(nodes.size()); A.leafBegin.resize(nodes.size()); A.leafCount.resize(nodes.size());
			for (size_t i = 0; i < nodes.size(); ++i) {
				put3(&A.bounds[i * 6], nodes[i].bounds[0]); put3(&A.bounds[i * 6 + 3], nodes[i].bounds[1]);
				A.skip[i] = nodes[i].skip; A.leafBegin[i] = nodes[i].leafBegin; A.leafCount[i] = nodes[i].leafCount;
			}
			const size_t nt = m.allTris.size();
			A.pos.resize(nt * 9); A.nrm.resize(nt * 9); A.uv.resize(nt * 6); A.tb.resize(nt * 6);
			for (size_t i = 0; i < nt; ++i) {
				const Triangle& t = m.allTris[i];
				put3(&A.pos[i * 9], t.a); put3(&A.pos[i * 9 + 3], t.b); put3(&A.pos[i * 9 + 6], t.c);
				put3(&A.nrm[i * 9], t.n_a); put3(&A.nrm[i * 9 + 3], t.n_b); put3(&A.nrm[i * 9 + 6], t.n_c);
				A.uv[i * 6] = t.t_a.x; A.uv[i * 6 + 1] = t.t_a.y; A.uv[i * 6 + 2] = t.t_b.x; A.uv[i * 6 + 3] = t.t_b.y;
				A.uv[i * 6 + 4] = t.t_c.x; A.uv[i * 6 + 5] = t.t_c.y;
				put3(&A.tb[i * 6], t.tangent); put3(&A.tb[i * 6 + 3], t.bitangent);
			}
			rtx_mesh rm{};
			rm.n_nodes = (uint32_t)nodes.size(); rm.n_refs = (uint32_t)m.ac->refs.size(); rm.n_tris = (uint32_t)nt;
			rm.node_bounds = A.bounds.data(); rm.node_skip = A.skip.data(); rm.leaf_begin = A.leafBegin.data(); rm.leaf_count = A.leafCount.data();
			rm.refs = m.ac->refs.data();
			rm.tri_pos = A.pos.data(); rm.tri_nrm = A.nrm.data(); rm.tri_uv = A.uv.data(); rm.tri_tb = A.tb.data();
			auto flat3 = [](const std::vector<Vec3f>& src, std::vector<float>& dst) {
				dst.resize(src.size() * 3);
				for (size_t i = 0; i < src.size(); ++i) put3(&dst[i * 3], src[i]);
			};
			if (m.diffuseMapLoaded) { flat3(m.diffuseMap, A.diffuse); rm.diffuse_w = m.diffuseMapWidth; rm.diffuse_h = m.diffuseMapHeight; rm.diffuse_map = A.diffuse.data(); }
			if (m.normalMapLoaded) { flat3(m.normalMap, A.normal); rm.normal_w = m.normalMapWidth; rm.normal_h = m.normalMapHeight; rm.normal_map = A.normal.data(); }
			if (m.specularMapLoaded) { rm.specular_w = m.specularMapWidth; rm.specular_h = m.specularMapHeight; rm.specular_map = m.specularMap.data(); }
			fs->meshes.push_back(rm);
		}
		fs->objects.push_back(ro);
	}
	for (auto& lp : sc.lights) {
		rtx_light rl{};
		put3(rl.color, lp->color); rl.intensity = lp->intensity;
		if (lp->type == LightType::DistantLight) { rl.type = RTX_LIGHT_DISTANT; put3(rl.dir, static_cast<DistantLight&>(*lp).dir); }
		else if (lp->type == LightType::PointLight) { rl.type = RTX_LIGHT_POINT; put3(rl.pos, static_cast<PointLight&>(*lp).pos); }
		else if (lp->type == LightType::AreaLight) {
			auto& al = static_cast<AreaLight&>(*lp);
			al.setPoints();
			rl.type = RTX_LIGHT_AREA; put3(rl.pos, al.pos);
			fs->lightPoints.emplace_back(al.points.size() * 3);
			for (size_t i = 0; i < al.points.size(); ++i) put3(&fs->lightPoints.back()[i * 3], al.points[i]);
			rl.n_points = (uint32_t)al.points.size();
		}
		else LOG_ERROR();
		fs->lights.push_back(rl);
	}
	// area-light point arrays may have been reallocated while pushing: fix the pointers up now
	{
		size_t k = 0;
		for (auto& rl : fs->lights) if (rl.type == RTX_LIGHT_AREA) rl.points = fs->lightPoints[k++].data();
	}
	fs->desc.n_objects = (uint32_t)fs->objects.size(); fs->desc.objects = fs->objects.data();
	fs->desc.n_meshes = (uint32_t)fs->meshes.size(); fs->desc.meshes = fs->meshes.data();
	fs->desc.n_lights = (uint32_t)fs->lights.size(); fs->desc.lights = fs->lights.data();
	if (sc.skyboxOn() && sc.skyboxWidth > 0) {
		fs->desc.sky_w = (uint32_t)sc.skyboxWidth; fs->desc.sky_h = (uint32_t)sc.skyboxHeight;
		for (int k = 0; k < 6; ++k) fs->desc.sky[k] = &sc.skyboxes[k][0].x;   // Vec3f is 3 packed floats
	}
	return fs;
}

const rtx_scene_desc* flatDesc(const FlatScene* fs) { return &fs->desc; }
void freeFlatScene(FlatScene* fs) { delete fs; }

// ------------------------------------------------------------------------------------------------
// GPU side
// ------------------------------------------------------------------------------------------------
namespace {
void gpuCheck(int rc, const char* what)
{
	if (rc == RTX_OK) return;
	std::cout << what << " failed: " << rtx_last_error() << '\n';
	noteError(std::string(what) + " failed: " + rtx_last_error());
	LOG_ERROR();
}
}

void Scene::invalidateView() { viewDirty_ = true; }

namespace {
void probeOne(Scene& sc, const Ray& ray, float hit[8], float colour[3])
{
	const float r[6] = { ray.orig.x, ray.orig.y, ray.orig.z, ray.dir.x, ray.dir.y, ray.dir.z };
	gpuCheck(rtx_cast_rays(sc.gpu(), 1, r, hit, colour), "rtx_cast_rays");
}
}

bool Render::trace(const Ray& ray, Scene& scene, IntersectInfo& info)
{
	if (ray.rayType != RayType::PrimaryRay) { std::cout << "Render::trace: only primary rays can be probed\n"; LOG_ERROR(); }
	float h[8], c[3];
	probeOne(scene, ray, h, c);
	info = IntersectInfo();
	if (h[0] == 0.0f) return false;
	info.hitObject = scene.objects[(size_t)h[1]].get();
	info.tNear = h[3];
	info.uv = Vec2f(h[4], h[5]);
	if (info.hitObject->objectType == ObjectType::Mesh)
		info.triPtr = &static_cast<const Mesh*>(info.hitObject)->allTris[(size_t)h[2]];
	return true;
}

Vec3f Render::castRay(const Ray& ray, Scene& scene, const int depth)
{
	if (depth != 0) { std::cout << "Render::castRay: depth must be 0 at the API boundary\n"; LOG_ERROR(); }
	float h[8], c[3];
	probeOne(scene, ray, h, c);
	return Vec3f(c[0], c[1], c[2]);
}

rtx_scene* Scene::gpu()
{
	if (!gpu_) {
		FlatScene* fs = flattenScene(*this);
		const int rc = rtx_scene_create(flatDesc(fs), device, &gpu_);
		freeFlatScene(fs);
		gpuCheck(rc, "rtx_scene_create");
		viewDirty_ = false;
	}
	if (viewDirty_) {
		rtx_view v{};
		fillView(*this, v);
		gpuCheck(rtx_scene_set_view(gpu_, &v), "rtx_scene_set_view");
		viewDirty_ = false;
	}
	return gpu_;
}

namespace {
void hipCheck(hipError_t e, const char* what)
{
	if (e == hipSuccess) return;
	std::cout << what << ": " << hipGetErrorString(e) << '\n';
	noteError(std::string(what) + ": " + hipGetErrorString(e));
	LOG_ERROR();
}
}

// The frame lives in HBM across pass 1, the Sobel mask and the 4-ray pass (one allocation per Scene, re-used by every
// render): the host sees it once, at the end.
struct Scene::DeviceFrame {
	float* fb = nullptr; uint8_t* mask = nullptr; uint8_t* bgr = nullptr;
	size_t pixels = 0;
	~DeviceFrame() { if (fb) (void)hipFree(fb); if (mask) (void)hipFree(mask); if (bgr) (void)hipFree(bgr); }
};

Scene::~Scene()
{
	frame_.reset();
	if (gpu_) rtx_scene_destroy(gpu_);
}

Scene::DeviceFrame& Scene::deviceFrame()
{
	const size_t px = options.width * options.height;
	hipCheck(hipSetDevice(device), "hipSetDevice");
	if (!frame_ || frame_->pixels != px) {
		frame_.reset(new DeviceFrame);
		hipCheck(hipMalloc((void**)&frame_->fb, px * sizeof(Vec3f)), "hipMalloc");
		hipCheck(hipMalloc((void**)&frame_->mask, px), "hipMalloc");
		hipCheck(hipMalloc((void**)&frame_->bgr, px * 3), "hipMalloc");
		frame_->pixels = px;
	}
	return *frame_;
}

void Scene::pass1OnDevice()
{
	rtx_scene* g = gpu();
	DeviceFrame& d = deviceFrame();
	gpuCheck(rtx_counters_enable(g, statisticsOn()), "rtx_counters_enable");
	gpuCheck(rtx_render_pass1(g, 0, (uint32_t)options.height, d.fb, nullptr), "rtx_render_pass1");
}

void Scene::ssaaOnDevice()
{
	rtx_scene* g = gpu();
	DeviceFrame& d = deviceFrame();
	gpuCheck(rtx_counters_enable(g, statisticsOn()), "rtx_counters_enable");
	gpuCheck(rtx_sobel(g, d.fb, 0, (uint32_t)options.height, d.mask, nullptr), "rtx_sobel");
	gpuCheck(rtx_render_ssaa(g, d.mask, 0, (uint32_t)options.height, d.fb, nullptr), "rtx_render_ssaa");
}

void Scene::readTimes()
{
	float ms = 0;
	if (rtx_last_kernel_ms(gpu_, 0, &ms) == RTX_OK) lastPass1Ms = ms;
	if (rtx_last_kernel_ms(gpu_, 1, &ms) == RTX_OK) lastSobelMs = ms;
	if (rtx_last_kernel_ms(gpu_, 2, &ms) == RTX_OK) lastSsaaMs = ms;
	if (rtx_last_kernel_ms(gpu_, 3, &ms) == RTX_OK) lastFrameMs = ms;
}

// The reference's entry points keep their meaning for a caller that owns a host framebuffer (scene.h:68-100): the
// workers write their pixels into it and leave the others alone, so the buffer is uploaded, updated and copied back.
// Scene::render() does not go through them: it keeps the frame on the device.
void Scene::launchWorkers(Vec3f* frameBuffer)
{
	Timer t("Render scene");
	DeviceFrame& d = deviceFrame();
	const size_t bytes = options.width * options.height * sizeof(Vec3f);
	hipCheck(hipMemcpy(d.fb, frameBuffer, bytes, hipMemcpyHostToDevice), "hipMemcpy");
	pass1OnDevice();
	hipCheck(hipMemcpy(frameBuffer, d.fb, bytes, hipMemcpyDeviceToHost), "hipMemcpy");
	readTimes();
}

void Scene::launchSSAA(Vec3f* frameBuffer)
{
	Timer t("MSAA");
	DeviceFrame& d = deviceFrame();
	const size_t bytes = options.width * options.height * sizeof(Vec3f);
	hipCheck(hipMemcpy(d.fb, frameBuffer, bytes, hipMemcpyHostToDevice), "hipMemcpy");
	ssaaOnDevice();
	hipCheck(hipMemcpy(frameBuffer, d.fb, bytes, hipMemcpyDeviceToHost), "hipMemcpy");
	readTimes();
}

void Scene::attachComm(rtx_comm* comm, int nRanks, int rank)
{
	comm_ = comm; nRanks_ = nRanks < 1 ? 1 : nRanks; rank_ = rank;
}

// Scene::render (scene.cpp:595-606): pass 1, the adaptive 4-ray pass, saveImage.  The frame stays in HBM; what comes
// back to the host is the BGR8 image saveImage writes (3 bytes per pixel instead of 4 x 12).  With a communicator
// attached (one process per GPU) this rank renders its own bands of rows, and the bands are collected on rank 0 with
// rtx_gather (RCCL over xGMI) -- rank 0 writes the file.
// Rows per band: about eight bands per device, between 64 and 256 rows (every band costs two halo rows; the same rule as
// rendering_amd/parallel.py band_height, measured there).
static uint32_t bandHeight(uint32_t height, uint32_t nParts)
{
	const uint32_t b = height / (8u * (nParts ? nParts : 1u)) / 64u * 64u;
	return b < 64u ? 64u : (b > 256u ? 256u : b);
}

void Scene::render()
{
	if (!sceneLoadSuccess) return;
	Timer t("Total time");
	if (options::showAC || options::showNormals || !options::useAC) {
		std::cout << "showAC / showNormals / useAC=0 are debug modes outside the accelerated hot path\n";
		LOG_ERROR();
	}
	rtx_scene* g = gpu();
	DeviceFrame& d = deviceFrame();
	const bool sharded = comm_ && nRanks_ > 1;
	gpuCheck(rtx_set_row_ownership(g, sharded ? bandHeight((uint32_t)options.height, (uint32_t)nRanks_) : 0u, (uint32_t)nRanks_, (uint32_t)rank_, 1), "rtx_set_row_ownership");
	hipCheck(hipMemset(d.fb, 0, options.width * options.height * sizeof(Vec3f)), "hipMemset");    // new Vec3f[H*W]() (scene.cpp:599)
	// With other ranks waiting for this one's bands, a failure of ANY stage must not simply exit (the others would hang in rtx_gather):
	// the verdicts of the stages are collected in rc, and the ranks agree on them once, immediately before the gather.
	int rc = RTX_OK;
	auto stage = [&](int r, const char* what) { if (sharded) { if (rc == RTX_OK && r != RTX_OK) { rc = r; std::cout << "rank " << rank_ << ": " << what << ": " << rtx_last_error() << '\n'; } } else gpuCheck(r, what); };
	auto hipStage = [&](hipError_t e, const char* what) { if (sharded) { if (rc == RTX_OK && e != hipSuccess) { rc = RTX_ERR_DEVICE; std::cout << "rank " << rank_ << ": " << what << ": " << hipGetErrorString(e) << '\n'; } } else hipCheck(e, what); };
	uint32_t status = 0;
	if (options::enableSSAA && !statisticsOn()) {
		// launchWorkers + launchSSAA as one call: the stages overlap on the device (rtx_render_frame)
		Timer tp("Render scene + MSAA");
		stage(rtx_counters_enable(g, 0), "rtx_counters_enable");
		if (rc == RTX_OK) stage(rtx_render_frame(g, 0, (uint32_t)options.height, d.fb, d.mask, nullptr), "rtx_render_frame");
		// the host's synchronisation point: a single launch that gave up has been rendered again in three (include/rtx.h)
		if (rc == RTX_OK) stage(rtx_frame_status(g, &status), "rtx_frame_status");
		if (status && options::enableOutput) std::cout << "frame rendered again in three launches (single launch status " << (status & 0xffu) << ")\n";
	}
	else {
		{
			Timer tp("Render scene");
			stage(rtx_counters_enable(g, statisticsOn()), "rtx_counters_enable");
			if (rc == RTX_OK) stage(rtx_render_pass1(g, 0, (uint32_t)options.height, d.fb, nullptr), "rtx_render_pass1");
			hipStage(hipDeviceSynchronize(), "hipDeviceSynchronize");
		}
		if (options::enableSSAA) {
			Timer ts("MSAA");
			if (rc == RTX_OK) stage(rtx_sobel(g, d.fb, 0, (uint32_t)options.height, d.mask, nullptr), "rtx_sobel");
			if (rc == RTX_OK) stage(rtx_render_ssaa(g, d.mask, 0, (uint32_t)options.height, d.fb, nullptr), "rtx_render_ssaa");
			hipStage(hipDeviceSynchronize(), "hipDeviceSynchronize");
		}
	}
	readTimes();
	if (options::imageOutput || sharded) {
		if (options.width % 4 != 0) { std::cout << "saveImage is only defined for width % 4 == 0 (util.cpp:28-29)\n"; LOG_ERROR(); }
		if (rc == RTX_OK) stage(rtx_quantize_bgr8(g, d.fb, d.bgr, nullptr), "rtx_quantize_bgr8");
		if (sharded) {
			int allOk = 0;
			gpuCheck(rtx_comm_agree(comm_, rc == RTX_OK, &allOk, nullptr), "rtx_comm_agree");
			if (!allOk) { std::cout << "rank " << rank_ << ": " << (rc == RTX_OK ? "another rank failed to render its rows" : "this rank failed to render its rows") << '\n'; LOG_ERROR(); }
			gpuCheck(rtx_gather(g, comm_, d.bgr, options.width * 3, 1, 0, nullptr), "rtx_gather");
		}
		hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
		if (options::imageOutput && rank_ == 0) {
			std::vector<unsigned char> bgr(options.width * options.height * 3);
			hipCheck(hipMemcpy(bgr.data(), d.bgr, bgr.size(), hipMemcpyDeviceToHost), "hipMemcpy");
			saveImageBGR(bgr.data(), options);
		}
	}
	// launchWorkers / launchSSAA (the reference's entry points) render ALL rows: the sharding is this call's own business
	if (sharded) gpuCheck(rtx_set_row_ownership(g, 0u, 1u, 0u, 1), "rtx_set_row_ownership");
	if (statisticsOn()) {
		rtx_counters c{};
		if (rtx_counters_read(gpu(), &c) == RTX_OK) {
			stats::raysCasted = c.rays; stats::accelStructTests = c.box_tests; stats::rayTriTests = c.tri_tests;
		}
		stats::printStats();
	}
	if (options::enableOutput) std::cout << '\n';
}
