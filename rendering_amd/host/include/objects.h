// Scene objects (reference: include/objects.h): Object / Sphere / Plane / Mesh / Triangle /
// AccelerationStructure with the reference's public members.  Intersection and shading live on the device;
// the host side loads, builds the acceleration structure bit-identically to objects.cpp:470-526,633-763
// and keeps it in the flat pre-order form the kernels consume.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "geometry.h"
#include "options.h"

enum class ObjectType { Object, Sphere, Plane, Mesh };
enum class MaterialType { Diffuse, Reflective, Transparent, Phong };

class Object {
public:
	virtual ~Object() = default;
	ObjectType objectType = ObjectType::Object;
	Vec3f pos{ 1, 1, 1 };
	Vec3f color{ 1, 1, 1 };
	MaterialType materialType = MaterialType::Diffuse;
	float indexOfRefraction = 1.4f;
	float ambient = 0.1f;
	float diffuse = 0.1f;
	float specular = 1.0f;
	float nSpecular = 5.0f;
};
using ObjectVector = std::vector<std::unique_ptr<Object>>;

class Sphere : public Object {
public:
	Sphere() { objectType = ObjectType::Sphere; pos = Vec3f(0, 0, 0); }
	float r = 1, r2 = 1;
};

class Plane : public Object {
public:
	Plane() { objectType = ObjectType::Plane; }
	Vec3f normal{ 0, 1, 0 };   // assigned raw by the loader, never re-normalised (scene.cpp:300)
};

class Triangle {
public:
	Vec3f a, b, c;
	Vec3f n_a, n_b, n_c;
	Vec2f t_a, t_b, t_c;
	Vec3f tangent, bitangent;
};

// Flat acceleration structure: nodes in pre-order (left subtree first), leaf references in visiting order.
class AccelerationStructure {
public:
	struct Node {
		Vec3f bounds[2];
		int32_t skip = 0;        // pre-order index of the first node after this subtree
		int32_t leafBegin = -1;  // -1 for inner nodes
		int32_t leafCount = -1;
	};
	void setBounds(const Vec3f& a, const Vec3f& b) { rootBounds[0] = a; rootBounds[1] = b; }
	bool setup(const std::vector<Triangle>& tris, const Options& options);   // false: the device build failed (message printed)
	float buildMs = 0;   // device build: time of the build on the GPU (HIP events)
	bool builtOnDevice = false;
	Vec3f rootBounds[2];
	std::vector<Node> nodes;
	std::vector<uint32_t> refs;
	int maxDepth = 0;
	size_t leafCount() const;
};

class Mesh : public Object {
public:
	Mesh() { objectType = ObjectType::Mesh; }
	bool loadOBJ(const std::string& filename, const Options& options);
	bool loadDiffuseMap(const std::string& filename);
	bool loadNormalMap(const std::string& filename);
	bool loadSpecularMap(const std::string& filename);

	Vec3f size, rot;
	std::vector<Triangle> allTris;
	std::unique_ptr<AccelerationStructure> ac;

	bool diffuseMapLoaded = false; int diffuseMapWidth = 0, diffuseMapHeight = 0; std::vector<Vec3f> diffuseMap;
	bool normalMapLoaded = false; int normalMapWidth = 0, normalMapHeight = 0; std::vector<Vec3f> normalMap;
	bool specularMapLoaded = false; int specularMapWidth = 0, specularMapHeight = 0; std::vector<float> specularMap;
};
