// Host-side math for the MI355X renderer: the subset of the reference's include/geometry.h that the scene
// API exposes (Vec2f / Vec3f / Matrix44f / Ray), written from scratch.  Only what the loaders, the BVH
// builder and the flattener need is provided; the reference's inverse/transpose/stream operators are dead
// code there and are not reproduced.
//
// Numerics contract: every expression keeps the reference's operand order and is compiled with
// -ffp-contract=off; normalize() goes through fp64 exactly like geometry.h:104-112.
#pragma once
#include <cmath>
#include <cstdint>

template <typename T> struct Vec2 {
	T x{}, y{};
	Vec2() = default;
	Vec2(T v) : x(v), y(v) {}
	Vec2(T a, T b) : x(a), y(b) {}
	Vec2 operator+(const Vec2& o) const { return { x + o.x, y + o.y }; }
	Vec2 operator-(const Vec2& o) const { return { x - o.x, y - o.y }; }
	Vec2 operator*(T s) const { return { x * s, y * s }; }
	T& operator[](int i) { return i == 0 ? x : y; }
	const T& operator[](int i) const { return i == 0 ? x : y; }
};
using Vec2f = Vec2<float>;

template <typename T> struct Vec3 {
	T x{}, y{}, z{};
	Vec3() = default;
	Vec3(T v) : x(v), y(v), z(v) {}
	Vec3(T a, T b, T c) : x(a), y(b), z(c) {}
	T dotProduct(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
	Vec3 crossProduct(const Vec3& o) const { return { y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x }; }
	T length2() const { return x * x + y * y + z * z; }
	T length() const { return (T)std::sqrt((double)length2()); }
	Vec3& normalize()
	{
		const T l2 = length2();
		if (l2 > 0) {
			const T f = (T)(1 / std::sqrt((double)l2));
			x *= f; y *= f; z *= f;
		}
		return *this;
	}
	Vec3 operator-() const { return { -x, -y, -z }; }
	Vec3 operator+(const Vec3& o) const { return { x + o.x, y + o.y, z + o.z }; }
	Vec3 operator-(const Vec3& o) const { return { x - o.x, y - o.y, z - o.z }; }
	Vec3 operator*(const Vec3& o) const { return { x * o.x, y * o.y, z * o.z }; }
	Vec3 operator/(const Vec3& o) const { return { x / o.x, y / o.y, z / o.z }; }
	Vec3 operator*(T s) const { return { x * s, y * s, z * s }; }
	Vec3 operator/(T s) const { return { x / s, y / s, z / s }; }
	Vec3& operator+=(const Vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
	T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
	const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
using Vec3f = Vec3<float>;

template <typename T> struct Matrix44 {
	T x[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
	T* operator[](int i) { return x[i]; }
	const T* operator[](int i) const { return x[i]; }
	Matrix44 operator*(const Matrix44& b) const
	{
		Matrix44 c;
		for (int i = 0; i < 4; ++i)
			for (int j = 0; j < 4; ++j)
				c.x[i][j] = x[i][0] * b.x[0][j] + x[i][1] * b.x[1][j] + x[i][2] * b.x[2][j] + x[i][3] * b.x[3][j];
		return c;
	}
	// row vector * matrix with the homogeneous divide of the reference (geometry.h:289-307)
	Vec3<T> multVecMatrix(const Vec3<T>& s) const
	{
		Vec3<T> d;
		d.x = s.x * x[0][0] + s.y * x[1][0] + s.z * x[2][0] + x[3][0];
		d.y = s.x * x[0][1] + s.y * x[1][1] + s.z * x[2][1] + x[3][1];
		d.z = s.x * x[0][2] + s.y * x[1][2] + s.z * x[2][2] + x[3][2];
		const T w = s.x * x[0][3] + s.y * x[1][3] + s.z * x[2][3] + x[3][3];
		if (w != 0 && w != 1) {
			const T wi = 1.0f / w;
			d.x *= wi; d.y *= wi; d.z *= wi;
		}
		return d;
	}
	// Euler rotation mz * my * mx from degrees (scene.cpp:24-48, objects.cpp:180-204)
	static Matrix44 fromEulerDegrees(const Vec3<T>& rot);
};
using Matrix44f = Matrix44<float>;

enum class RayType { PrimaryRay, ShadowRay };
struct Ray {
	RayType rayType = RayType::PrimaryRay;
	Vec3f orig{ 0, 0, 0 };
	Vec3f dir{ 0, 0, -1 };
};
