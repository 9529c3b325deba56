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
// restores the defaults above (the reference never resets them between scenes)
void reset();
}
