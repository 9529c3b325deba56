// Scene / Camera / Render (reference: include/scene.h).  Same public surface -- Scene(path), render(),
// launchWorkers(Vec3f*), launchSSAA(Vec3f*), getTiles(), objects / lights / options / camera -- but the
// two worker loops are launches of the gfx950 kernels through the C ABI in include/rtx.h.
#pragma once
#include <string>
#include <vector>

#include "geometry.h"
#include "lights.h"
#include "objects.h"
#include "options.h"

#include <memory>

struct rtx_scene;
struct rtx_scene_desc;
struct rtx_comm;

typedef struct { size_t x0, x1, y0, y1; } tileInfo;

class Camera {
public:
	Vec3f pos{ 0, 0, 0 };
	Vec3f rot{ 0, 0, 0 };
	Matrix44f rMatrix;
	float fov = 60.0f;
	float zNear = 0.1f, zFar = 100.0f;
	bool cameraRotated = false;
	void ensureMatrix();          // the lazy part of Camera::getRay (scene.cpp:22-49), done before upload
};

// Render::trace's result record (reference: scene.h:17-23)
struct IntersectInfo {
	const Object* hitObject = nullptr;
	float tNear = 3.402823466e+38f;
	const Triangle* triPtr = nullptr;
	Vec2f uv{ -1, -1 };
};

class Scene;

// Single-ray entry points of the reference (scene.h:31-48), evaluated on the GPU through rtx_cast_rays.
// They take the Scene (which owns the uploaded GPU twin) instead of the bare ObjectVector; castRay supports
// depth 0 only (deeper levels are internal to the kernel's recursion frames).  Meant for tools and tests --
// frames go through Scene::launchWorkers.
class Render {
public:
	static bool trace(const Ray& ray, Scene& scene, IntersectInfo& intrInfo);
	static Vec3f castRay(const Ray& ray, Scene& scene, const int depth);
};

class Scene {
public:
	bool sceneLoadSuccess = true;
	ObjectVector objects;
	LightsVector lights;
	Options options;
	Camera camera;
	int skyboxWidth = 0, skyboxHeight = 0;
	std::vector<Vec3f> skyboxes[6];

	explicit Scene(const std::string& sceneName);
	~Scene();
	Scene(const Scene&) = delete;
	Scene& operator=(const Scene&) = delete;

	bool loadScene(const std::string& sceneName);
	void loadSkybox();
	void render();
	void launchWorkers(Vec3f* frameBuffer);   // pass 1 on the GPU, result copied into the caller's buffer
	void launchSSAA(Vec3f* frameBuffer);      // Sobel + 4-ray re-render on the GPU
	std::vector<tileInfo> getTiles();

	// MI355X side
	int device = 0;
	rtx_scene* gpu();                          // flattens + uploads on first use; LOG_ERROR()s without a GPU
	void invalidateView();                     // call after changing options.width/height, camera, flags
	// Multi-GPU (one process per GPU): with a communicator of the C ABI attached (rtx_comm_create), render() renders this
	// rank's rows only and collects the image on rank 0 (rtx_gather), which writes the file.
	void attachComm(rtx_comm* comm, int nRanks, int rank);
	double lastPass1Ms = 0, lastSobelMs = 0, lastSsaaMs = 0, lastFrameMs = 0;      // (lastFrameMs: the whole of render(), rtx_render_frame)
	// The switches the render reads are process-global in the reference (options::useBackfaceCulling, useSkybox,
	// collectStatistics).  A host that keeps several scenes alive (the C API in capi.cpp) pins them per scene here:
	// -1 = follow the global (the reference's behaviour), 0 / 1 = this scene's own value.
	int useBackfaceCulling = -1, useSkybox = -1, collectStatistics = -1;
	bool cullingOn() const { return useBackfaceCulling < 0 ? options::useBackfaceCulling : useBackfaceCulling != 0; }
	bool skyboxOn() const { return useSkybox < 0 ? options::useSkybox : useSkybox != 0; }
	bool statisticsOn() const { return collectStatistics < 0 ? options::collectStatistics : collectStatistics != 0; }
	void pinFlags() { useBackfaceCulling = options::useBackfaceCulling; useSkybox = options::useSkybox; collectStatistics = options::collectStatistics; }

private:
	struct DeviceFrame;
	DeviceFrame& deviceFrame();                // the frame's device buffers (fp32 framebuffer, Sobel mask, BGR8 image)
	void pass1OnDevice();
	void ssaaOnDevice();
	void readTimes();
	rtx_scene* gpu_ = nullptr;
	bool viewDirty_ = true;
	std::unique_ptr<DeviceFrame> frame_;
	rtx_comm* comm_ = nullptr;
	int nRanks_ = 1, rank_ = 0;
};

// Host-side flattening used by Scene::gpu(); exposed for tests and for maintainers wiring the reference's own
// Scene to the C ABI.
struct FlatScene;
FlatScene* flattenScene(Scene& scene);
const rtx_scene_desc* flatDesc(const FlatScene*);
void freeFlatScene(FlatScene*);
