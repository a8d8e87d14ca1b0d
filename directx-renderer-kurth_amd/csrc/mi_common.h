// Device-side math and record layouts shared by the HIP kernels of the MI355X rigid-body stepper.
// gfx950 only (wave64).  All device arithmetic is strict fp32: the translation units are compiled with
// -ffp-contract=off so that expression-for-expression the kernels round like the CPU restatement used as the
// parity checker (sqrt and divide are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring> // rocPRIM's texture_cache_iterator.hpp uses memset without including it

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef uint8_t u8;

#define MI_WAVE 64
#define MI_EPSILON 1e-6f
#define MI_PI 3.14159265359f
#define MI_FLT_MAX 3.402823466e+38f

// collider_type / enum order is load-bearing (reference physics.h:61)
enum { MI_SPHERE = 0, MI_CAPSULE = 1, MI_CYLINDER = 2, MI_AABB = 3, MI_OBB = 4, MI_HULL = 5, MI_TYPE_COUNT = 6 };

// ---------------------------------------------------------------------------------------------------
// V3 / Q4 / M3 — scalar math in the operation order of the reference's core/math.h (cited per function)
// ---------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };
struct M3 { float m00, m10, m20, m01, m11, m21, m02, m12, m22; }; // column-major like reference mat3

#define MI_DEV __host__ __device__ __forceinline__
#define mi_f2u(x) __builtin_bit_cast(u32, (float)(x))
#define mi_u2f(x) __builtin_bit_cast(float, (u32)(x))

MI_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MI_DEV V3 v3s(float s) { return v3(s, s, s); }
MI_DEV V3 v3f4(float4 f) { return v3(f.x, f.y, f.z); }
MI_DEV float vget(const V3& v, u32 i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
MI_DEV void vset(V3& v, u32 i, float f) { if (i == 0) v.x = f; else if (i == 1) v.y = f; else v.z = f; }
MI_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
MI_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MI_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
MI_DEV V3 operator*(V3 a, float b) { return v3(a.x * b, a.y * b, a.z * b); }
MI_DEV V3 operator*(float a, V3 b) { return b * a; }
MI_DEV V3 operator/(V3 a, float b) { return v3(a.x / b, a.y / b, a.z / b); }
MI_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
MI_DEV V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
MI_DEV V3& operator-=(V3& a, V3 b) { a = a - b; return a; }
MI_DEV V3& operator*=(V3& a, float b) { a = a * b; return a; }
MI_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       // math.h:582
MI_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); } // math.h:586
MI_DEV float sqlen(V3 a) { return dot(a, a); }
MI_DEV float length(V3 a) { return sqrtf(sqlen(a)); }
MI_DEV V3 noz(V3 a) { float sl = sqlen(a); return (sl < 1e-8f) ? v3(0.f, 0.f, 0.f) : (a * (1.f / sqrtf(sl))); } // math.h:595
MI_DEV V3 normalize(V3 a) { float l = length(a); return a * (1.f / l); }                         // math.h:599
MI_DEV V3 vabs(V3 a) { return v3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
MI_DEV V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
MI_DEV V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
MI_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
MI_DEV float clamp01(float v) { return clampf(v, 0.f, 1.f); }
MI_DEV V3 lerp(V3 l, V3 u, float t) { return l + t * (u - l); }

MI_DEV Q4 q4(float x, float y, float z, float w) { Q4 q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }
MI_DEV Q4 q4f4(float4 f) { return q4(f.x, f.y, f.z, f.w); }
MI_DEV V3 qv(Q4 q) { return v3(q.x, q.y, q.z); }
MI_DEV Q4 conjugate(Q4 a) { return q4(-a.x, -a.y, -a.z, a.w); }
MI_DEV Q4 operator*(Q4 a, Q4 b)                                                                  // math.h:627-633
{
	Q4 r;
	r.w = a.w * b.w - dot(qv(a), qv(b));
	V3 v = qv(a) * b.w + qv(b) * a.w + cross(qv(a), qv(b));
	r.x = v.x; r.y = v.y; r.z = v.z;
	return r;
}
MI_DEV V3 operator*(Q4 q, V3 v) { Q4 p = q4(v.x, v.y, v.z, 0.f); return qv(q * p * conjugate(q)); } // math.h:642-646
MI_DEV Q4 qnormalize(Q4 a)
{
	float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
	float inv = 1.f / l;
	return q4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
}

MI_DEV V3 mrow(const M3& a, u32 r) { return r == 0 ? v3(a.m00, a.m01, a.m02) : (r == 1 ? v3(a.m10, a.m11, a.m12) : v3(a.m20, a.m21, a.m22)); }
MI_DEV V3 mcol(const M3& a, u32 c) { return c == 0 ? v3(a.m00, a.m10, a.m20) : (c == 1 ? v3(a.m01, a.m11, a.m21) : v3(a.m02, a.m12, a.m22)); }
MI_DEV V3 operator*(const M3& a, V3 b) { return v3(dot(mrow(a, 0), b), dot(mrow(a, 1), b), dot(mrow(a, 2), b)); } // math.h:660
MI_DEV M3 operator*(const M3& a, const M3& b)                                                     // math.cpp:103-118
{
	V3 r0 = mrow(a, 0), r1 = mrow(a, 1), r2 = mrow(a, 2);
	V3 c0 = mcol(b, 0), c1 = mcol(b, 1), c2 = mcol(b, 2);
	M3 r;
	r.m00 = dot(r0, c0); r.m01 = dot(r0, c1); r.m02 = dot(r0, c2);
	r.m10 = dot(r1, c0); r.m11 = dot(r1, c1); r.m12 = dot(r1, c2);
	r.m20 = dot(r2, c0); r.m21 = dot(r2, c1); r.m22 = dot(r2, c2);
	return r;
}
MI_DEV M3 madd(const M3& a, const M3& b)
{
	M3 r;
	r.m00 = a.m00 + b.m00; r.m10 = a.m10 + b.m10; r.m20 = a.m20 + b.m20;
	r.m01 = a.m01 + b.m01; r.m11 = a.m11 + b.m11; r.m21 = a.m21 + b.m21;
	r.m02 = a.m02 + b.m02; r.m12 = a.m12 + b.m12; r.m22 = a.m22 + b.m22;
	return r;
}
MI_DEV M3 mtranspose(const M3& a)
{
	M3 r;
	r.m00 = a.m00; r.m01 = a.m10; r.m02 = a.m20;
	r.m10 = a.m01; r.m11 = a.m11; r.m12 = a.m21;
	r.m20 = a.m02; r.m21 = a.m12; r.m22 = a.m22;
	return r;
}
MI_DEV M3 midentity() { M3 r; r.m00 = r.m11 = r.m22 = 1.f; r.m10 = r.m20 = r.m01 = r.m21 = r.m02 = r.m12 = 0.f; return r; }
MI_DEV M3 mskew(V3 r)                                                                             // math.cpp:797-810
{
	M3 s;
	s.m00 = 0.f; s.m01 = -r.z; s.m02 = r.y;
	s.m10 = r.z; s.m11 = 0.f; s.m12 = -r.x;
	s.m20 = -r.y; s.m21 = r.x; s.m22 = 0.f;
	return s;
}
MI_DEV M3 quaternionToMat3(Q4 q)                                                                  // math.cpp:644-677
{
	if (q.w == 1.f) { return midentity(); }
	float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
	float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
	float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
	M3 r;
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
MI_DEV V3 solve3(const M3& A, V3 b)                                                               // math.cpp:1356-1371
{
	V3 ex = v3(A.m00, A.m10, A.m20), ey = v3(A.m01, A.m11, A.m21), ez = v3(A.m02, A.m12, A.m22);
	float det = dot(ex, cross(ey, ez));
	if (det != 0.f) { det = 1.f / det; }
	V3 x;
	x.x = det * dot(b, cross(ey, ez));
	x.y = det * dot(ex, cross(b, ez));
	x.z = det * dot(ex, cross(ey, b));
	return x;
}
MI_DEV Q4 rotateFromTo(V3 _from, V3 _to)                                                          // math.cpp:538-575
{
	V3 from = normalize(_from), to = normalize(_to);
	float d = dot(from, to);
	if (d >= 1.f) return q4(0.f, 0.f, 0.f, 1.f);
	Q4 q;
	if (d < (1e-6f - 1.f))
	{
		V3 axis = cross(v3(1.f, 0.f, 0.f), from);
		if (sqlen(axis) == 0.f) axis = cross(v3(0.f, 1.f, 0.f), from);
		axis = normalize(axis);
		float h = MI_PI * 0.5f, s = sinf(h);
		q = qnormalize(q4(axis.x * s, axis.y * s, axis.z * s, cosf(h)));
	}
	else
	{
		float s = sqrtf((1.f + d) * 2.f);
		float invs = 1.f / s;
		V3 c = cross(from, to);
		q = qnormalize(q4(c.x * invs, c.y * invs, c.z * invs, s * 0.5f));
	}
	return q;
}
MI_DEV V3 getTangent(V3 normal)                                                                   // math.cpp:1416-1420
{
	V3 t = (fabsf(normal.x) >= 0.57735f) ? v3(normal.y, -normal.x, 0.f) : v3(0.f, normal.z, -normal.y);
	return normalize(t);
}
MI_DEV float4 createPlane(V3 point, V3 normal) { float d = -dot(normal, point); return make_float4(normal.x, normal.y, normal.z, d); } // bounding_volumes.h:166
MI_DEV float signedDistanceToPlane(V3 p, float4 pl) { return p.x * pl.x + p.y * pl.y + p.z * pl.z + pl.w; }                           // bounding_volumes.h:296

// ---------------------------------------------------------------------------------------------------
// Records in HBM
// ---------------------------------------------------------------------------------------------------
// World-space / local collider: 64 B, four float4 — one gather = one 64-B segment (reference collider_union, physics.h:86-106).
//   f[0..9]  shape payload (sphere c,r | capsule A,B,r | aabb min,max | obb q,c,r)
//   f[10] restitution  f[11] friction  u[12] type  u[13] body index (numBodies = static)  f[14] density  u[15] flags
struct ColliderRec { float4 a, b, c, d; };
MI_DEV u32 colType(const ColliderRec& r) { return mi_f2u(r.d.x); }
MI_DEV u32 colBody(const ColliderRec& r) { return mi_f2u(r.d.y); }
MI_DEV float colRestitution(const ColliderRec& r) { return r.c.z; }
MI_DEV float colFriction(const ColliderRec& r) { return r.c.w; }

// Per candidate pair, written by the narrowphase, read once by contact init: 96 B.
//   p[k] = (point.xyz, penetrationDepth); nf = (normal.xyz, bits(friction_restitution)); ids = (bodyA, bodyB, count, colliderPairIndex)
struct ManifoldRec { float4 p[4]; float4 nf; uint4 ids; };

// Solver view of a body: 2 x float4, gathered/scattered as one 32-B segment.
//   vel[2i] = (v.xyz, invMass), vel[2i+1] = (w.xyz, 0).  Index numBodies is the zero-mass static dummy (never written).

#define MI_MAX_CONTACTS_PER_MANIFOLD 4
#define MI_ROW_PLANES 8          // read-only float4 planes per contact (30 floats; layout: solver_rows.h)
#define MI_MAX_COLORS 64         // colours 0..63 run in parallel; colour 64 is the serial overflow bucket
#define MI_SERIAL_COLOR 64

#define MI_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { mi_set_error(_e, __FILE__, __LINE__); } } while (0)
void mi_set_error(hipError_t e, const char* file, int line);
