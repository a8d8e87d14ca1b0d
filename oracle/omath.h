// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Scalar math restated from the reference's src/core/math.h / math.cpp (file:line cited per function).
// Strict IEEE fp32, no FMA contraction (compile with -ffp-contract=off), sqrt/div correctly rounded.
// PARITY UNPINNED: the reference holds no tests / golden vectors for this path and
// cannot be compiled here (MSVC dialect + Windows PCH + empty EnTT submodule).
#pragma once
#include <cmath>
#include <cstdint>
#include <cfloat>
#include <cstring>
#include <algorithm>

namespace orc {

typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t u8;
typedef int32_t i32;
typedef uint64_t u64;

static const float M_PI_F = 3.14159265359f;      // core/math.h:13 (M_PI macro)
static const float EPSILON = 1e-6f;                // core/math.h:22

struct vec3
{
	float x, y, z;
	vec3() : x(0.f), y(0.f), z(0.f) {}
	vec3(float v) : x(v), y(v), z(v) {}
	vec3(float x, float y, float z) : x(x), y(y), z(z) {}
	float& operator[](u32 i) { return (&x)[i]; }
	float operator[](u32 i) const { return (&x)[i]; }
};

struct vec2 { float x, y; vec2() : x(0), y(0) {} vec2(float x, float y) : x(x), y(y) {} };

struct vec4
{
	float x, y, z, w;
	vec4() : x(0), y(0), z(0), w(0) {}
	vec4(vec3 v, float w) : x(v.x), y(v.y), z(v.z), w(w) {}
	vec4(float x, float y, float z, float w) : x(x), y(y), z(z), w(w) {}
	vec3 xyz() const { return vec3(x, y, z); }
};

// core/math.h:292-314: quat is {x,y,z,w}; v = xyz.
struct quat
{
	float x, y, z, w;
	quat() : x(0), y(0), z(0), w(1) {}
	quat(float x, float y, float z, float w) : x(x), y(y), z(z), w(w) {}
	// core/math.h:932-936
	quat(vec3 axis, float angle)
	{
		w = cosf(angle * 0.5f);
		float s = sinf(angle * 0.5f);
		x = axis.x * s; y = axis.y * s; z = axis.z * s;
	}
	vec3 v() const { return vec3(x, y, z); }
};

// core/math.h:385-396: column-major storage m00,m10,m20,m01,... ; mRC = row R col C.
struct mat3
{
	float m00, m10, m20, m01, m11, m21, m02, m12, m22;
	mat3() : m00(0), m10(0), m20(0), m01(0), m11(0), m21(0), m02(0), m12(0), m22(0) {}
	float* m() { return &m00; }
	const float* m() const { return &m00; }
	static mat3 identity() { mat3 r; r.m00 = r.m11 = r.m22 = 1.f; return r; }
	static mat3 zero() { return mat3(); }
};

struct mat2 { float m00, m10, m01, m11; mat2() : m00(0), m10(0), m01(0), m11(0) {} };

static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator*(vec3 a, float b) { return vec3(a.x * b, a.y * b, a.z * b); }
static inline vec3 operator*(float a, vec3 b) { return b * a; }
static inline vec3 operator/(vec3 a, float b) { return vec3(a.x / b, a.y / b, a.z / b); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
static inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
static inline vec3& operator*=(vec3& a, float b) { a = a * b; return a; }
static inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
static inline vec3& operator/=(vec3& a, float b) { a = a / b; return a; }
static inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
static inline vec2 operator*(vec2 a, float b) { return vec2(a.x * b, a.y * b); }
static inline vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }

// core/math.h:579-600
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float squaredLength(vec3 a) { return dot(a, a); }
static inline float length(vec3 a) { return sqrtf(squaredLength(a)); }
static inline vec3 noz(vec3 a) { float sl = squaredLength(a); return (sl < 1e-8f) ? vec3(0.f, 0.f, 0.f) : (a * (1.f / sqrtf(sl))); }
static inline vec3 normalize(vec3 a) { float l = length(a); return a * (1.f / l); }
static inline vec3 vabs(vec3 a) { return vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline vec3 vmin(vec3 a, vec3 b) { return vec3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
static inline vec3 vmax(vec3 a, vec3 b) { return vec3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float clamp01(float v) { return clampf(v, 0.f, 1.f); }
static inline float lerpf(float l, float u, float t) { return l + t * (u - l); }
static inline vec3 lerp(vec3 l, vec3 u, float t) { return l + t * (u - l); }          // core/math.h:672

// core/math.h:622-646
static inline quat conjugate(quat a) { return quat(-a.x, -a.y, -a.z, a.w); }
static inline quat operator+(quat a, quat b) { return quat(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline quat operator*(quat a, quat b)
{
	quat r;
	r.w = a.w * b.w - dot(a.v(), b.v());
	vec3 v = a.v() * b.w + b.v() * a.w + cross(a.v(), b.v());
	r.x = v.x; r.y = v.y; r.z = v.z;
	return r;
}
static inline quat operator*(quat q, float s) { return quat(q.x * s, q.y * s, q.z * s, q.w * s); }
static inline vec3 operator*(quat q, vec3 v)
{
	quat p(v.x, v.y, v.z, 0.f);
	return (q * p * conjugate(q)).v();
}
// normalize(quat) goes through normalize(vec4): length via dot(vec4) = addElements(a*b) (core/math.h:581,600,622).
static inline quat normalize(quat a)
{
	float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
	float inv = 1.f / l;
	return quat(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
}
static inline bool operator==(quat a, quat b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

static inline vec3 row(const mat3& a, u32 r) { const float* m = a.m(); return vec3(m[r], m[r + 3], m[r + 6]); } // core/math.h:486
static inline vec3 col(const mat3& a, u32 c) { const float* m = a.m(); return vec3(m[3 * c], m[3 * c + 1], m[3 * c + 2]); }
static inline vec3 operator*(const mat3& a, vec3 b) { return vec3(dot(row(a, 0), b), dot(row(a, 1), b), dot(row(a, 2), b)); } // core/math.h:660
// core/math.cpp:103-118
static inline mat3 operator*(const mat3& a, const mat3& b)
{
	vec3 r0 = row(a, 0), r1 = row(a, 1), r2 = row(a, 2);
	vec3 c0 = col(b, 0), c1 = col(b, 1), c2 = col(b, 2);
	mat3 r;
	r.m00 = dot(r0, c0); r.m01 = dot(r0, c1); r.m02 = dot(r0, c2);
	r.m10 = dot(r1, c0); r.m11 = dot(r1, c1); r.m12 = dot(r1, c2);
	r.m20 = dot(r2, c0); r.m21 = dot(r2, c1); r.m22 = dot(r2, c2);
	return r;
}
static inline mat3 operator+(const mat3& a, const mat3& b) { mat3 r; for (u32 i = 0; i < 9; ++i) r.m()[i] = a.m()[i] + b.m()[i]; return r; }
static inline mat3 operator-(const mat3& a, const mat3& b) { mat3 r; for (u32 i = 0; i < 9; ++i) r.m()[i] = a.m()[i] - b.m()[i]; return r; }
static inline mat3 operator*(const mat3& a, float b) { mat3 r; for (u32 i = 0; i < 9; ++i) r.m()[i] = a.m()[i] * b; return r; }
static inline mat3& operator+=(mat3& a, const mat3& b) { a = a + b; return a; }
// core/math.cpp:241-248
static inline mat3 transpose(const mat3& a)
{
	mat3 r;
	r.m00 = a.m00; r.m01 = a.m10; r.m02 = a.m20;
	r.m10 = a.m01; r.m11 = a.m11; r.m12 = a.m21;
	r.m20 = a.m02; r.m21 = a.m12; r.m22 = a.m22;
	return r;
}
// core/math.cpp:276-306
static inline mat3 invert(const mat3& m)
{
	mat3 inv;
	inv.m00 = m.m11 * m.m22 - m.m21 * m.m12;
	inv.m01 = m.m02 * m.m21 - m.m22 * m.m01;
	inv.m02 = m.m01 * m.m12 - m.m11 * m.m02;
	inv.m10 = m.m12 * m.m20 - m.m22 * m.m10;
	inv.m11 = m.m00 * m.m22 - m.m20 * m.m02;
	inv.m12 = m.m02 * m.m10 - m.m12 * m.m00;
	inv.m20 = m.m10 * m.m21 - m.m20 * m.m11;
	inv.m21 = m.m01 * m.m20 - m.m21 * m.m00;
	inv.m22 = m.m00 * m.m11 - m.m10 * m.m01;
	float det = m.m00 * (m.m11 * m.m22 - m.m21 * m.m12)
		- m.m01 * (m.m10 * m.m22 - m.m20 * m.m12)
		+ m.m02 * (m.m10 * m.m21 - m.m20 * m.m11);
	if (det == 0.f) { return mat3(); }
	det = 1.f / det;
	return inv * det;
}
// core/math.cpp:778-795
static inline mat3 outerProduct(vec3 a, vec3 b)
{
	vec3 c0 = a * b.x, c1 = a * b.y, c2 = a * b.z;
	mat3 r;
	r.m00 = c0.x; r.m10 = c0.y; r.m20 = c0.z;
	r.m01 = c1.x; r.m11 = c1.y; r.m21 = c1.z;
	r.m02 = c2.x; r.m12 = c2.y; r.m22 = c2.z;
	return r;
}
// core/math.cpp:797-810
static inline mat3 getSkewMatrix(vec3 r)
{
	mat3 s;
	s.m00 = 0.f; s.m01 = -r.z; s.m02 = r.y;
	s.m10 = r.z; s.m11 = 0.f; s.m12 = -r.x;
	s.m20 = -r.y; s.m21 = r.x; s.m22 = 0.f;
	return s;
}
// core/math.cpp:644-677
static inline mat3 quaternionToMat3(quat q)
{
	if (q.w == 1.f) { return mat3::identity(); }
	float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
	float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
	float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
	mat3 r;
	r.m00 = 1.f - 2.f * (qyy + qzz);
	r.m10 = 2.f * (qxy + qwz);
	r.m20 = 2.f * (qxz - qwy);
	r.m01 = 2.f * (qxy - qwz);
	r.m11 = 1.f - 2.f * (qxx + qzz);
	r.m21 = 2.f * (qyz + qwx);
	r.m02 = 2.f * (qxz + qwy);
	r.m12 = 2.f * (qyz - qwx);
	r.m22 = 1.f - 2.f * (qxx + qyy);
	return r;
}
// core/math.cpp:538-575
static inline quat rotateFromTo(vec3 _from, vec3 _to)
{
	vec3 from = normalize(_from);
	vec3 to = normalize(_to);
	float d = dot(from, to);
	if (d >= 1.f) { return quat(0.f, 0.f, 0.f, 1.f); }
	quat q;
	if (d < (1e-6f - 1.f))
	{
		vec3 axis = cross(vec3(1.f, 0.f, 0.f), from);
		if (squaredLength(axis) == 0.f) { axis = cross(vec3(0.f, 1.f, 0.f), from); }
		axis = normalize(axis);
		q = normalize(quat(axis, M_PI_F));
	}
	else
	{
		float s = sqrtf((1.f + d) * 2.f);
		float invs = 1.f / s;
		vec3 c = cross(from, to);
		q.x = c.x * invs; q.y = c.y * invs; q.z = c.z * invs; q.w = s * 0.5f;
		q = normalize(q);
	}
	return q;
}
// core/math.cpp:577-592
static inline void getAxisRotation(quat q, vec3& axis, float& angle)
{
	float sqLength = squaredLength(q.v());
	if (sqLength > 0.f)
	{
		angle = 2.f * acosf(q.w);
		float invLength = 1.f / sqrtf(sqLength);
		axis = q.v() * invLength;
	}
	else { angle = 0.f; axis = vec3(1.f, 0.f, 0.f); }
}
// core/math.cpp:1342-1371
static inline vec2 solveLinearSystem(const mat2& A, vec2 b)
{
	float a11 = A.m00, a12 = A.m01, a21 = A.m10, a22 = A.m11;
	float det = a11 * a22 - a12 * a21;
	if (det != 0.f) { det = 1.f / det; }
	vec2 x;
	x.x = det * (a22 * b.x - a12 * b.y);
	x.y = det * (a11 * b.y - a21 * b.x);
	return x;
}
static inline vec3 solveLinearSystem(const mat3& A, vec3 b)
{
	vec3 ex(A.m00, A.m10, A.m20), ey(A.m01, A.m11, A.m21), ez(A.m02, A.m12, A.m22);
	float det = dot(ex, cross(ey, ez));
	if (det != 0.f) { det = 1.f / det; }
	vec3 x;
	x.x = det * dot(b, cross(ey, ez));
	x.y = det * dot(ex, cross(b, ez));
	x.z = det * dot(ex, cross(ey, b));
	return x;
}
// core/math.cpp:1390-1408
static inline vec3 getBarycentricCoordinates(vec3 a, vec3 b, vec3 c, vec3 p)
{
	vec3 v0 = b - a, v1 = c - a, v2 = p - a;
	float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
	float denom = d00 * d11 - d01 * d01;
	denom = (fabsf(denom) < EPSILON) ? 1.f : denom;
	float v = (d11 * d20 - d01 * d21) / denom;
	float w = (d00 * d21 - d01 * d20) / denom;
	float u = 1.0f - v - w;
	return vec3(u, v, w);
}
// core/math.cpp:1416-1426
static inline vec3 getTangent(vec3 normal)
{
	vec3 tangent = (fabsf(normal.x) >= 0.57735f) ? vec3(normal.y, -normal.x, 0.f) : vec3(0.f, normal.z, -normal.y);
	return normalize(tangent);
}
static inline void getTangents(vec3 normal, vec3& outTangent, vec3& outBitangent)
{
	outTangent = getTangent(normal);
	outBitangent = cross(normal, outTangent);
}

// core/math.h: trs = {rotation, position, scale}; physics only uses rotation+position (scale 1).
struct trs { quat rotation; vec3 position; };
static inline vec3 transformPosition(const trs& m, vec3 pos) { return m.rotation * pos + m.position; }               // math.cpp:518 (scale=1)
static inline vec3 transformDirection(const trs& m, vec3 dir) { return m.rotation * dir; }                            // math.cpp:523
static inline vec3 inverseTransformPosition(const trs& m, vec3 pos) { return conjugate(m.rotation) * (pos - m.position); } // math.cpp:528
static inline vec3 inverseTransformDirection(const trs& m, vec3 dir) { return conjugate(m.rotation) * dir; }          // math.cpp:533

} // namespace orc
