// Light sources (reference: include/lights.h).  illuminate() is evaluated on the device
// (rtx_kernels.hip: ST_NEXT_LIGHT); the host classes only carry the parameters.
#pragma once
#include <memory>
#include <vector>

#include "geometry.h"

enum class LightType { BaseLight, DistantLight, PointLight, AreaLight };

class Light {
public:
	virtual ~Light() = default;
	Vec3f color{ 1, 1, 1 };
	float intensity = 1;
	LightType type = LightType::BaseLight;
};
using LightsVector = std::vector<std::unique_ptr<Light>>;

class DistantLight : public Light {
public:
	DistantLight() { type = LightType::DistantLight; }
	Vec3f dir{ 0, 0, -1 };   // assigned raw by the loader, never re-normalised (scene.cpp:222)
};

class PointLight : public Light {
public:
	PointLight() { type = LightType::PointLight; }
	Vec3f pos{ 0, 0, 0 };
};

class AreaLight : public Light {
public:
	AreaLight() { type = LightType::AreaLight; }
	void setPoints();        // samples x samples grid over the parallelogram (lights.cpp:46-63)
	Vec3f pos, i, j;
	int samples = 1;
	bool pointsCreated = false;
	std::vector<Vec3f> points;
};
