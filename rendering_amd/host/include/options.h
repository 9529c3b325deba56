// Per-scene Options and the process-global options:: switches -- same names, types and defaults as the
// reference's include/options.h:9-37, so scene files and host code written against it keep working.
#pragma once
#include <cstddef>
#include <string>

#include "geometry.h"

class Options {
public:
	size_t width = 800, height = 600;
	float bias = 0.0001f;
	int maxRayDepth = 10;
	int nWorkers = 32;              // kept for API compatibility; the GPU path has no worker threads
	Vec3f backgroundColor{ 0.0f, 0.0f, 0.0f };
	int acPenalty = 1;
	char skyboxNames[6][64] = { { 0 } };
	std::string imageName = "out";
};

namespace options {
inline bool outputProgress = true;
inline bool useBackfaceCulling = true;
inline bool collectStatistics = false;
inline bool enableOutput = true;
inline bool imageOutput = true;
inline bool useAC = true;       // the GPU path always walks the acceleration structure (useAC=0 is a debug quirk, out of scope)
inline bool showAC = false;     // debug heat-map, out of scope
inline bool useSkybox = false;
inline bool useTextures = true;
inline bool showNormals = false; // debug mode, out of scope
inline bool enableSSAA = true;
// Where Mesh::loadOBJ builds the acceleration structure: -1 = decide on first use (the device if one is visible,
// environment RENDERING_AMD_AC_BUILD=host|device overrides), 0 = host builder, 1 = rtx_bvh_build on the device.
// Both produce the same structure bit for bit (tests/test_gpu_bvh.py).
inline int acBuildOnDevice = -1;
inline int acBuildDevice = 0;     // HIP device index for the build
// restores the defaults above (the reference never resets them between scenes)
void reset();
}
