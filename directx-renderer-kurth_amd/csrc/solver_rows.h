// Contact rows of the projected Gauss-Seidel sweep, shared by the launch-per-colour sweep (k_solver.hip) and the LDS cluster
// sweep (k_cluster.hip): the SoA plane layout written by k_contact_init and the two-row solve of one contact
// (friction, then normal: reference constraints.cpp:3381-3449 / 3618-3709).
//   plane p of contact k, slot s: rowPlanes[(k*6+p)*rowCap + s]
//   p0 = rA.xyz rB.x | p1 = rB.yz t.xy | p2 = t.z JnA.xyz | p3 = JtA.xyz JnB.x | p4 = JnB.yz JtB.xy | p5 = JtB.z mN mT bias
#pragma once
#include "mi_common.h"

struct ContactRow { float4 p0, p1, p2, p3, p4, p5; float2 lam; };

MI_DEV void loadRow(ContactRow& r, u32 k, u32 s, size_t rowCap, const float4* __restrict__ rowPlanes, const float2* __restrict__ rowLambda)
{
	const float4* P = rowPlanes + (size_t)(k * MI_ROW_PLANES) * rowCap + s;
	r.p0 = P[0]; r.p1 = P[rowCap]; r.p2 = P[2 * rowCap]; r.p3 = P[3 * rowCap]; r.p4 = P[4 * rowCap]; r.p5 = P[5 * rowCap];
	r.lam = rowLambda[(size_t)k * rowCap + s];
}

MI_DEV void solveRow(ContactRow& r, V3 n, float friction, float invMassA, float invMassB, V3& vA, V3& wA, V3& vB, V3& wB)
{
	V3 rA = v3(r.p0.x, r.p0.y, r.p0.z), rB = v3(r.p0.w, r.p1.x, r.p1.y), t = v3(r.p1.z, r.p1.w, r.p2.x);
	V3 JnA = v3(r.p2.y, r.p2.z, r.p2.w), JtA = v3(r.p3.x, r.p3.y, r.p3.z), JnB = v3(r.p3.w, r.p4.x, r.p4.y), JtB = v3(r.p4.z, r.p4.w, r.p5.x);
	float mN = r.p5.y, mT = r.p5.z, bias = r.p5.w;
	float impulseN = r.lam.x, impulseT = r.lam.y;
	{ // tangent (constraints.cpp:3404-3424)
		V3 rel = (vB + cross(wB, rB)) - (vA + cross(wA, rA));
		float vt = dot(rel, t);
		float lambda = -mT * vt;
		float maxFriction = friction * impulseN;
		float newImpulse = clampf(impulseT + lambda, -maxFriction, maxFriction);
		lambda = newImpulse - impulseT;
		impulseT = newImpulse;
		V3 Pv = lambda * t;
		vA -= invMassA * Pv; wA -= JtA * lambda;
		vB += invMassB * Pv; wB += JtB * lambda;
	}
	{ // normal (constraints.cpp:3426-3442)
		V3 rel = (vB + cross(wB, rB)) - (vA + cross(wA, rA));
		float vn = dot(rel, n);
		float lambda = -mN * (vn - bias);
		float impulse = fmaxf(impulseN + lambda, 0.f);
		lambda = impulse - impulseN;
		impulseN = impulse;
		V3 Pv = lambda * n;
		vA -= invMassA * Pv; wA -= JnA * lambda;
		vB += invMassB * Pv; wB += JnB * lambda;
	}
	r.lam = make_float2(impulseN, impulseT);
}
