// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// "AVX2 semantics" of the reference's wide math, one lane at a time: what the reference's 8-wide paths compute
// where they differ from its scalar code by more than the last bit —
//   * polynomial cos / sin / atan2 / acos (core/simd.h:28-49, 122-164; the w8_float overloads at :739-748 forward to them): absolute
//     errors of ~1e-3 (cos, sin), ~3e-3 rad (atan2) and ~7e-5 rad (acos) against libm;
//   * noz / normalize through the hardware reciprocal-square-root ESTIMATE (math_simd.h:283-289, simd.h:694: _mm256_rsqrt_ps, 12 bits,
//     no Newton step), switched by wideApproxRsqrt() because its bits differ between CPU vendors;
//   * dot / cross with fused multiply-adds in the reference's association (math_simd.h:240-246);
//   * rotateFromTo / getAxisRotation / quat(axis, angle) in their wide formulation (math_simd.h:512-573, 592-598).
// The joint initialisation uses them when wideJointMath() is on (orc_set_wide_joint_math): constraints.cpp:1309-1777 (hinge) and
// 2072-2634 (cone twist) evaluate their limit / motor angles with exactly these functions.  Lane-wise `ifThen` selections become
// plain branches (one lane).  Not restated: the fused multiply-adds inside the wide paths' mass / impulse algebra (last-bit differences).
#pragma once
#include "omath.h"

#if defined(__SSE__)
#include <xmmintrin.h>
#endif

namespace orc {

inline bool& wideApproxRsqrt() { static bool on = false; return on; }
inline bool& wideJointMath() { static bool on = false; return on; }
static inline float rsqrtEstimate(float x)
{
#if defined(__SSE__)
	return _mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(x)));
#else
	return 1.f / sqrtf(x);
#endif
}
static inline float wrsqrt(float x) { return wideApproxRsqrt() ? rsqrtEstimate(x) : (1.f / sqrtf(x)); }

// core/simd.h:28-44
static inline float polyCos(float x)
{
	const float tp = 1.f / (2.f * 3.14159265359f), q = 0.25f, h = 0.5f, o = 1.f, s = 16.f, v = 0.225f;
	x *= tp;
	x -= q + floorf(x + q);
	x *= s * (fabsf(x) - h);
	x += v * x * (fabsf(x) - o);
	return x;
}
// core/simd.h:46-49
static inline float polySin(float x) { return polyCos(x - (3.14159265359f * 0.5f)); }
// core/simd.h:122-148 (shifts are logical: simd.h:291)
static inline float polyAtan2(float y, float x)
{
	const u32 sign_mask = 0x80000000u;
	const float b = 0.596227f;
	u32 xb, yb; memcpy(&xb, &x, 4); memcpy(&yb, &y, 4);
	u32 ux_s = sign_mask & xb, uy_s = sign_mask & yb;
	float q = (float)(int)(((~ux_s & uy_s) >> 29) | (ux_s >> 30));
	float bxy_a = fabsf(b * x * y);
	float num = fmaf(y, y, bxy_a);
	float atan_1q = num / fmaf(x, x, bxy_a + num);
	u32 ab; memcpy(&ab, &atan_1q, 4);
	u32 uatan_2q = (ux_s ^ uy_s) | ab;
	float signedAtan; memcpy(&signedAtan, &uatan_2q, 4);
	float result04 = q + signedAtan;
	float result = (result04 >= 2.f) ? (result04 - 4.f) : result04;
	return result * (3.14159265359f * 0.5f);
}
// core/simd.h:150-164
static inline float polyAcos(float x)
{
	float negate = (x < 0.f) ? 1.f : 0.f;
	x = fabsf(x);
	float ret = -0.0187293f;
	ret = fmaf(ret, x, 0.0742610f);
	ret = fmaf(ret, x, -0.2121144f);
	ret = fmaf(ret, x, 1.5707288f);
	ret = ret * sqrtf(1.f - x);
	ret = ret - negate * ret * 2.f;
	return fmaf(negate, 3.14159265359f, ret);
}

// math_simd.h:240-246
static inline float wdot(vec3 a, vec3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline vec3 wcross(vec3 a, vec3 b) { return vec3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))); }
// math_simd.h:283-289, 299
static inline vec3 wnoz(vec3 a) { float sl = wdot(a, a); return (sl < 1e-8f) ? vec3(0.f) : a * wrsqrt(sl); }
static inline vec3 wnormalize(vec3 a) { return a * wrsqrt(wdot(a, a)); }
static inline quat wnormalize(quat a)
{
	float l2 = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, a.w * a.w)));
	float r = wrsqrt(l2);
	return quat(a.x * r, a.y * r, a.z * r, a.w * r);
}
// math_simd.h:592-598
static inline quat wquatAxisAngle(vec3 axis, float angle)
{
	float h = 0.5f;
	float w = polyCos(angle * h);
	vec3 v = axis * polySin(angle * h);
	return quat(v.x, v.y, v.z, w);
}
// math_simd.h:512-553
static inline quat wrotateFromTo(vec3 _from, vec3 _to)
{
	vec3 from = wnormalize(_from), to = wnormalize(_to);
	float d = wdot(from, to);
	bool same = d >= 1.f, largeRotation = d < (1e-6f - 1.f);
	quat q;
	if (largeRotation)
	{
		vec3 axis = wcross(vec3(1.f, 0.f, 0.f), from);
		if (wdot(axis, axis) == 0.f) { axis = wcross(vec3(0.f, 1.f, 0.f), from); }
		axis = wnormalize(axis);
		q = wnormalize(wquatAxisAngle(axis, M_PI_F));
	}
	else
	{
		float s = sqrtf((1.f + d) * 2.f);
		float invs = 1.f / s;
		vec3 c = wcross(from, to);
		q = wnormalize(quat(c.x * invs, c.y * invs, c.z * invs, s * 0.5f));
	}
	if (same) { q = quat(0.f, 0.f, 0.f, 1.f); }
	return q;
}
// math_simd.h:555-573
static inline void wgetAxisRotation(quat q, vec3& axis, float& angle)
{
	angle = 0.f; axis = vec3(1.f, 0.f, 0.f);
	float sqLength = wdot(q.v(), q.v());
	if (sqLength > 0.f)
	{
		angle = 2.f * polyAcos(q.w);
		float invLength = 1.f / sqrtf(sqLength);
		axis = q.v() * invLength;
	}
}

// What the joint initialisation calls: the scalar functions, or the wide ones above.
static inline float jointAtan2(float y, float x) { return wideJointMath() ? polyAtan2(y, x) : atan2f(y, x); }
static inline float jointAcos(float x) { return wideJointMath() ? polyAcos(x) : acosf(x); }
static inline float jointCos(float x) { return wideJointMath() ? polyCos(x) : cosf(x); }
static inline float jointSin(float x) { return wideJointMath() ? polySin(x) : sinf(x); }
static inline float jointDot(vec3 a, vec3 b) { return wideJointMath() ? wdot(a, b) : dot(a, b); }
static inline vec3 jointNoz(vec3 a) { return wideJointMath() ? wnoz(a) : noz(a); }
static inline quat jointRotateFromTo(vec3 a, vec3 b) { return wideJointMath() ? wrotateFromTo(a, b) : rotateFromTo(a, b); }
static inline void jointGetAxisRotation(quat q, vec3& axis, float& angle) { if (wideJointMath()) wgetAxisRotation(q, axis, angle); else getAxisRotation(q, axis, angle); }
static inline quat jointQuatAxisAngle(vec3 axis, float angle) { return wideJointMath() ? wquatAxisAngle(axis, angle) : quat(axis, angle); }

} // namespace orc
