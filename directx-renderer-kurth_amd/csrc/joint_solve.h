// Joint constraint solves (sequential impulses on the per-joint update record written by k_*_init), shared by the launch-per-colour
// joint kernels (k_joints.hip) and the LDS cluster sweep (k_cluster.hip), which runs them on body velocities held in LDS.
// Scalar formulations of the reference: constraints.cpp:189-264 (distance), 460-528 (ball), 736-823 (fixed), 1079-1307 (hinge),
// 1782-2070 (cone-twist), 2638-2846 (slider).  `o` is the joint's update record (layouts: k_joints.hip), v the two bodies'
// velocities, IA / IB their world-space inverse inertia.
#pragma once
#include "mi_common.h"

#define BETA_DISTANCE 0.1f       // constraints.cpp:9-17
#define BETA_BALL 0.1f
#define BETA_SLIDER 0.1f
#define BETA_HINGE_ROT 0.3f
#define BETA_HINGE_LIMIT 0.1f
#define BETA_TWIST_LIMIT 0.1f
#define BETA_SLIDER_LIMIT 0.1f
#define DT_THRESHOLD 1e-5f

struct BodyIn { Q4 rot; V3 localCOG; V3 pos; M3 invI; float invMass; };

MI_DEV M3 ldInvI(const float4* __restrict__ invIw, u32 i)
{
	float4 c0 = invIw[3 * i], c1 = invIw[3 * i + 1], c2 = invIw[3 * i + 2];
	M3 I; I.m00 = c0.x; I.m10 = c0.y; I.m20 = c0.z; I.m01 = c1.x; I.m11 = c1.y; I.m21 = c1.z; I.m02 = c2.x; I.m12 = c2.y; I.m22 = c2.z;
	return I;
}
MI_DEV BodyIn loadBody(u32 i, const float4* __restrict__ pose, const float4* __restrict__ bprops, const float4* __restrict__ cog, const float4* __restrict__ invIw)
{
	BodyIn b;
	b.rot = q4f4(pose[2 * i + 1]);
	b.localCOG = v3f4(bprops[5 * i]);
	float4 c = cog[i];
	b.pos = v3f4(c); b.invMass = c.w;
	b.invI = ldInvI(invIw, i);
	return b;
}
MI_DEV V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
MI_DEV void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
MI_DEV void stM3(float* p, const M3& m) { p[0] = m.m00; p[1] = m.m10; p[2] = m.m20; p[3] = m.m01; p[4] = m.m11; p[5] = m.m21; p[6] = m.m02; p[7] = m.m12; p[8] = m.m22; }
MI_DEV M3 ldM3(const float* p) { M3 m; m.m00 = p[0]; m.m10 = p[1]; m.m20 = p[2]; m.m01 = p[3]; m.m11 = p[4]; m.m21 = p[5]; m.m02 = p[6]; m.m12 = p[7]; m.m22 = p[8]; return m; }

MI_DEV M3 mscale(const M3& a, float s)
{
	M3 r; r.m00 = a.m00 * s; r.m10 = a.m10 * s; r.m20 = a.m20 * s; r.m01 = a.m01 * s; r.m11 = a.m11 * s; r.m21 = a.m21 * s; r.m02 = a.m02 * s; r.m12 = a.m12 * s; r.m22 = a.m22 * s;
	return r;
}
// skewA * IA * skewA^T + skewB * IB * skewB^T + I * (mA + mB)   (constraints.cpp:487-492 and the identical blocks of fixed/hinge/cone-twist)
MI_DEV M3 pointBlock(const BodyIn& A, const BodyIn& B, V3 rA, V3 rB)
{
	M3 sA = mskew(rA), sB = mskew(rB);
	return madd(madd(sA * A.invI * mtranspose(sA), sB * B.invI * mtranspose(sB)), mscale(midentity(), A.invMass + B.invMass));
}
MI_DEV void solve2(float m00, float m01, float m10, float m11, float bx, float by, float& x, float& y) // math.cpp:1342-1354
{
	float det = m00 * m11 - m01 * m10;
	if (det != 0.f) det = 1.f / det;
	x = det * (m11 * bx - m01 * by);
	y = det * (m00 * by - m10 * bx);
}

struct Vel { V3 vA, wA, vB, wB; float invMassA, invMassB; };
MI_DEV Vel loadVel(const float4* __restrict__ vel, u32 a, u32 b)
{
	Vel v; float4 la = vel[2 * a], lb = vel[2 * b];
	v.vA = v3f4(la); v.wA = v3f4(vel[2 * a + 1]); v.vB = v3f4(lb); v.wB = v3f4(vel[2 * b + 1]); v.invMassA = la.w; v.invMassB = lb.w;
	return v;
}
MI_DEV void storeVel(float4* __restrict__ vel, u32 nb, u32 a, u32 b, const Vel& v)
{
	if (a < nb) { vel[2 * a] = make_float4(v.vA.x, v.vA.y, v.vA.z, v.invMassA); vel[2 * a + 1] = make_float4(v.wA.x, v.wA.y, v.wA.z, 0.f); }
	if (b < nb) { vel[2 * b] = make_float4(v.vB.x, v.vB.y, v.vB.z, v.invMassB); vel[2 * b + 1] = make_float4(v.wB.x, v.wB.y, v.wB.z, 0.f); }
}
// "Position part" shared by ball/fixed/hinge/cone-twist (e.g. constraints.cpp:1288-1300)
MI_DEV void solvePointBlock(Vel& v, const M3& IA, const M3& IB, V3 rA, V3 rB, V3 bias, const M3& invEff)
{
	V3 Cdot = (v.vB + cross(v.wB, rB)) - (v.vA + cross(v.wA, rA)) + bias;
	V3 P = solve3(invEff, -Cdot);
	v.vA -= v.invMassA * P; v.wA -= IA * cross(rA, P);
	v.vB += v.invMassB * P; v.wB += IB * cross(rB, P);
}


MI_DEV void solveDistance(float* o, Vel& v, const M3& IA, const M3& IB)
{
	V3 rA = ld3(o), rB = ld3(o + 3), jA = ld3(o + 6), jB = ld3(o + 9), u = ld3(o + 12);
	float Cdot = dot(u, (v.vB + cross(v.wB, rB)) - (v.vA + cross(v.wA, rA))) + o[15];
	float lambda = -o[16] * Cdot;
	V3 P = lambda * u;
	v.vA -= v.invMassA * P; v.wA -= jA * lambda; v.vB += v.invMassB * P; v.wB += jB * lambda;
}

MI_DEV void solveBall(float* o, Vel& v, const M3& IA, const M3& IB)
{
	solvePointBlock(v, IA, IB, ld3(o), ld3(o + 3), ld3(o + 6), ldM3(o + 9));
}

MI_DEV void solveFixed(float* o, Vel& v, const M3& IA, const M3& IB)
{
	{
		V3 Cdot = v.wB - v.wA;
		V3 lam = solve3(ldM3(o + 21), -(Cdot + ld3(o + 18)));
		v.wA -= IA * lam; v.wB += IB * lam;
	}
	solvePointBlock(v, IA, IB, ld3(o), ld3(o + 3), ld3(o + 6), ldM3(o + 9));
}

MI_DEV void solveHinge(float* o, Vel& v, const M3& IA, const M3& IB)
{
	u32 flags = __float_as_uint(o[34]);
	V3 axis = ld3(o + 30), jA = ld3(o + 41), jB = ld3(o + 44);
	float effAxial = o[33];
	if (flags & 2u) // motor
	{
		float rel = dot(axis, v.wB) - dot(axis, v.wA);
		float lam = -effAxial * (rel - o[40]);
		float old = o[38];
		float imp = clampf(old + lam, -o[39], o[39]);
		o[38] = imp; lam = imp - old;
		v.wA -= jA * lam; v.wB += jB * lam;
	}
	if (flags & 1u) // limit
	{
		float sign = o[37];
		float rel = sign * (dot(axis, v.wB) - dot(axis, v.wA));
		float lam = -effAxial * (rel + o[36]);
		float imp = fmaxf(o[35] + lam, 0.f);
		lam = imp - o[35]; o[35] = imp;
		lam *= sign;
		v.wA -= jA * lam; v.wB += jB * lam;
	}
	{ // rotation
		V3 bxa = ld3(o + 24), cxa = ld3(o + 27);
		V3 dw = v.wB - v.wA;
		float lx, ly;
		solve2(o[20], o[21], o[22], o[23], -(dot(bxa, dw) + o[18]), -(dot(cxa, dw) + o[19]), lx, ly);
		V3 P = bxa * lx + cxa * ly;
		v.wA -= IA * P; v.wB += IB * P;
	}
	solvePointBlock(v, IA, IB, ld3(o), ld3(o + 3), ld3(o + 6), ldM3(o + 9));
}

MI_DEV void solveConeTwist(float* o, Vel& v, const M3& IA, const M3& IB)
{
	u32 flags = __float_as_uint(o[18]);
	V3 twistAxis = ld3(o + 31), tjA = ld3(o + 38), tjB = ld3(o + 41);
	if (flags & 8u) // twist motor
	{
		float rel = dot(twistAxis, v.wB) - dot(twistAxis, v.wA);
		float lam = -o[36] * (rel - o[59]);
		float old = o[57];
		float imp = clampf(old + lam, -o[58], o[58]);
		o[57] = imp; lam = imp - old;
		v.wA -= tjA * lam; v.wB += tjB * lam;
	}
	if (flags & 4u) // swing motor
	{
		V3 g = ld3(o + 47);
		float rel = dot(g, v.wB) - dot(g, v.wA);
		float lam = -o[50] * (rel - o[46]);
		float old = o[44];
		float imp = clampf(old + lam, -o[45], o[45]);
		o[44] = imp; lam = imp - old;
		v.wA -= ld3(o + 51) * lam; v.wB += ld3(o + 54) * lam;
	}
	if (flags & 2u) // twist limit
	{
		float sign = o[35];
		float rel = sign * (dot(twistAxis, v.wB) - dot(twistAxis, v.wA));
		float lam = -o[36] * (rel + o[37]);
		float imp = fmaxf(o[34] + lam, 0.f);
		lam = imp - o[34]; o[34] = imp;
		lam *= sign;
		v.wA -= tjA * lam; v.wB += tjB * lam;
	}
	if (flags & 1u) // swing (cone) limit
	{
		V3 g = ld3(o + 19);
		float cdot = dot(g, v.wA) - dot(g, v.wB) + o[24];
		float lam = -o[23] * cdot;
		float imp = fmaxf(o[22] + lam, 0.f);
		lam = imp - o[22]; o[22] = imp;
		v.wA += ld3(o + 25) * lam; v.wB -= ld3(o + 28) * lam;
	}
	solvePointBlock(v, IA, IB, ld3(o), ld3(o + 3), ld3(o + 6), ldM3(o + 9));
}

MI_DEV void solveSlider(float* o, Vel& v, const M3& IA, const M3& IB)
{
	u32 flags = __float_as_uint(o[36]);
	V3 axis = ld3(o + 37);
	if (flags & 2u)
	{
		float Cdot = dot(v.vB, axis) - dot(v.vA, axis) - o[56];
		float mass = 1.f / (v.invMassA + v.invMassB);
		float lam = -mass * Cdot;
		float old = o[57];
		float imp = clampf(old + lam, -o[58], o[58]);
		o[57] = imp; lam = imp - old;
		V3 P = lam * axis;
		v.vA -= v.invMassA * P; v.vB += v.invMassB * P;
	}
	if (flags & 1u)
	{
		float Cdot = dot(v.vB, axis) + dot(v.wB, ld3(o + 47)) - dot(v.vA, axis) - dot(v.wA, ld3(o + 44));
		float lam = -o[40] * (o[43] * Cdot + o[41]);
		float imp = fmaxf(o[42] + lam, 0.f);
		lam = imp - o[42]; o[42] = imp;
		lam *= o[43];
		V3 P = lam * axis;
		v.vA -= v.invMassA * P; v.wA -= ld3(o + 50) * lam;
		v.vB += v.invMassB * P; v.wB += ld3(o + 53) * lam;
	}
	{
		V3 Cdot = v.wB - v.wA;
		V3 lam = solve3(ldM3(o + 24), -(Cdot + ld3(o + 33)));
		v.wA -= IA * lam; v.wB += IB * lam;
	}
	{
		V3 rAuxt = ld3(o), rAuxb = ld3(o + 3), rBxt = ld3(o + 6), rBxb = ld3(o + 9), tangent = ld3(o + 12), bitangent = ld3(o + 15);
		float cx = dot(tangent, v.vB) + dot(rBxt, v.wB) - dot(tangent, v.vA) - dot(rAuxt, v.wA);
		float cy = dot(bitangent, v.vB) + dot(rBxb, v.wB) - dot(bitangent, v.vA) - dot(rAuxb, v.wA);
		float lx, ly;
		solve2(o[18], o[19], o[20], o[21], -(cx + o[22]), -(cy + o[23]), lx, ly);
		V3 tb = tangent * lx + bitangent * ly;
		v.vA -= v.invMassA * tb; v.wA -= IA * (rAuxt * lx + rAuxb * ly);
		v.vB += v.invMassB * tb; v.wB += IB * (rBxt * lx + rBxb * ly);
	}
}

// Update record size in floats, by constraint type (= MI_JOINT_UPDATE_FLOATS, world.h).
MI_DEV u32 jointUpdateFloats(u32 type) { return type == 0u ? 20u : (type == 1u ? 20u : (type == 2u ? 36u : (type == 3u ? 56u : (type == 4u ? 80u : 72u)))); }
MI_DEV void jointSolve(u32 type, float* o, Vel& v, const M3& IA, const M3& IB)
{
	switch (type)
	{
	case 0: solveDistance(o, v, IA, IB); break;
	case 1: solveBall(o, v, IA, IB); break;
	case 2: solveFixed(o, v, IA, IB); break;
	case 3: solveHinge(o, v, IA, IB); break;
	case 4: solveConeTwist(o, v, IA, IB); break;
	default: solveSlider(o, v, IA, IB); break;
	}
}
