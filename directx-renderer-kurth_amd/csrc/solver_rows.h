// Contact rows of the projected Gauss-Seidel sweep, shared by the launch-per-colour sweep (k_solver.hip) and the LDS cluster
// sweep (k_cluster.hip): the SoA plane layout written by k_contact_init and the two-row solve of one contact (friction, then
// normal: reference constraints.cpp:3381-3449 / 3618-3709).
//
// Row form.  The reference evaluates the relative anchor velocity as (vB + wB x rB) - (vA + wA x rA) and projects it on the row
// direction d; with the scalar triple product that is the Jacobian form
//     J v = d . (vB - vA) + (rB x d) . wB - (rA x d) . wA,
// whose angular parts rA x d, rB x d the row initialisation computes anyway (they feed the effective mass and the
// impulse-to-angular-velocity vectors).  On this chip the sweep is bound by the LATENCY of one row solve (a dependent chain issued
// by a single wave: ~7 cycles per vector instruction), so the row is kept in that form and evaluated with explicit fused
// multiply-adds (fma placement as in the reference's wide dot: fmadd(a.x, b.x, fmadd(a.y, b.y, a.z * b.z)), core/math_simd.h:241):
// 65 instructions per contact instead of 130.  Same mathematics, different rounding: the CPU oracle restates exactly this
// expression tree with fmaf (oracle/oconstraints.h: solveCollisionConstraintRowForm), so device and oracle stay bit-equal, and the
// oracle's reference-formula solver is compared with it within a stated tolerance (tests/test_oracle.py).
//   plane p of contact k, slot s: rowPlanes[(k * MI_ROW_PLANES + p) * rowCap + s]
//   p0 = t.xyz (rA x t).x | p1 = (rA x t).yz (rB x t).xy | p2 = (rB x t).z (rA x n).xyz | p3 = (rB x n).xyz JtA.x
//   p4 = JtA.yz JtB.xy | p5 = JtB.z JnA.xyz | p6 = JnB.xyz mN | p7 = mT bias - -
#pragma once
#include "mi_common.h"

struct ContactRow { float4 p0, p1, p2, p3, p4, p5, p6; float2 p7; float2 lam; };

MI_DEV void loadRow(ContactRow& r, u32 k, u32 s, size_t rowCap, const float4* __restrict__ rowPlanes, const float2* __restrict__ rowLambda)
{
	const float4* P = rowPlanes + (size_t)(k * MI_ROW_PLANES) * rowCap + s;
	r.p0 = P[0]; r.p1 = P[rowCap]; r.p2 = P[2 * rowCap]; r.p3 = P[3 * rowCap]; r.p4 = P[4 * rowCap]; r.p5 = P[5 * rowCap]; r.p6 = P[6 * rowCap];
	float4 q = P[7 * rowCap]; r.p7 = make_float2(q.x, q.y);
	r.lam = rowLambda[(size_t)k * rowCap + s];
}

// J v for direction d with angular parts cA = rA x d, cB = rB x d.
MI_DEV float rowVelocity(V3 d, V3 cA, V3 cB, V3 vA, V3 wA, V3 vB, V3 wB)
{
	V3 dv = vB - vA;
	float s = dv.z * d.z;
	s = __builtin_fmaf(dv.y, d.y, s); s = __builtin_fmaf(dv.x, d.x, s);
	s = __builtin_fmaf(wB.z, cB.z, s); s = __builtin_fmaf(wB.y, cB.y, s); s = __builtin_fmaf(wB.x, cB.x, s);
	s = __builtin_fmaf(-wA.z, cA.z, s); s = __builtin_fmaf(-wA.y, cA.y, s); s = __builtin_fmaf(-wA.x, cA.x, s);
	return s;
}
// v -+= invMass * lambda * d, w -+= J * lambda
MI_DEV void rowApply(float lambda, V3 d, V3 JA, V3 JB, float invMassA, float invMassB, V3& vA, V3& wA, V3& vB, V3& wB)
{
	float a = invMassA * lambda, b = invMassB * lambda;
	vA = v3(__builtin_fmaf(-a, d.x, vA.x), __builtin_fmaf(-a, d.y, vA.y), __builtin_fmaf(-a, d.z, vA.z));
	vB = v3(__builtin_fmaf(b, d.x, vB.x), __builtin_fmaf(b, d.y, vB.y), __builtin_fmaf(b, d.z, vB.z));
	wA = v3(__builtin_fmaf(-lambda, JA.x, wA.x), __builtin_fmaf(-lambda, JA.y, wA.y), __builtin_fmaf(-lambda, JA.z, wA.z));
	wB = v3(__builtin_fmaf(lambda, JB.x, wB.x), __builtin_fmaf(lambda, JB.y, wB.y), __builtin_fmaf(lambda, JB.z, wB.z));
}

MI_DEV void solveRow(ContactRow& r, V3 n, float friction, float invMassA, float invMassB, V3& vA, V3& wA, V3& vB, V3& wB)
{
	V3 t = v3(r.p0.x, r.p0.y, r.p0.z), cAt = v3(r.p0.w, r.p1.x, r.p1.y), cBt = v3(r.p1.z, r.p1.w, r.p2.x);
	V3 cAn = v3(r.p2.y, r.p2.z, r.p2.w), cBn = v3(r.p3.x, r.p3.y, r.p3.z);
	V3 JtA = v3(r.p3.w, r.p4.x, r.p4.y), JtB = v3(r.p4.z, r.p4.w, r.p5.x), JnA = v3(r.p5.y, r.p5.z, r.p5.w), JnB = v3(r.p6.x, r.p6.y, r.p6.z);
	float mN = r.p6.w, mT = r.p7.x, bias = r.p7.y;
	float impulseN = r.lam.x, impulseT = r.lam.y;
	{ // tangent (constraints.cpp:3404-3424)
		float vt = rowVelocity(t, cAt, cBt, vA, wA, vB, wB);
		float lambda = -mT * vt;
		float maxFriction = friction * impulseN;
		float newImpulse = clampf(impulseT + lambda, -maxFriction, maxFriction);
		lambda = newImpulse - impulseT;
		impulseT = newImpulse;
		rowApply(lambda, t, JtA, JtB, invMassA, invMassB, vA, wA, vB, wB);
	}
	{ // normal (constraints.cpp:3426-3442)
		float vn = rowVelocity(n, cAn, cBn, vA, wA, vB, wB);
		float lambda = -mN * (vn - bias);
		float impulse = fmaxf(impulseN + lambda, 0.f);
		lambda = impulse - impulseN;
		impulseN = impulse;
		rowApply(lambda, n, JnA, JnB, invMassA, invMassB, vA, wA, vB, wB);
	}
	r.lam = make_float2(impulseN, impulseT);
}
