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

// Quad form.  The cluster sweep solves a row with FOUR lanes, one per body vector (vA, wA, vB, wB), so the row velocity is the sum of
// four 3-term partial dots, each accumulated z, y, x with fused multiply-adds and combined as (pA_lin + pA_ang) + (pB_lin + pB_ang)
// with body A's terms negated in the operands; the impulse delta d is applied as x = fma(d, a, x) with the "apply vectors"
// a = -(invMassA * dir), -JA, invMassB * dir, JB.  Every solver path (this scalar-lane restatement, the quad lanes of k_cluster.hip)
// evaluates exactly this expression tree, and so does the oracle (oracle/oconstraints.h: solveCollisionConstraintRowForm).
MI_DEV float rowDot3(V3 x, V3 d) { float s = x.z * d.z; s = __builtin_fmaf(x.y, d.y, s); return __builtin_fmaf(x.x, d.x, s); }
MI_DEV float rowVelocity(V3 d, V3 cA, V3 cB, V3 vA, V3 wA, V3 vB, V3 wB)
{
	return (rowDot3(vA, -d) + rowDot3(wA, -cA)) + (rowDot3(vB, d) + rowDot3(wB, cB));
}
MI_DEV V3 rowFma(float d, V3 a, V3 x) { return v3(__builtin_fmaf(d, a.x, x.x), __builtin_fmaf(d, a.y, x.y), __builtin_fmaf(d, a.z, x.z)); }
// v -+= (invMass * dir) * lambda, w -+= J * lambda
MI_DEV void rowApply(float lambda, V3 d, V3 JA, V3 JB, float invMassA, float invMassB, V3& vA, V3& wA, V3& vB, V3& wB)
{
	vA = rowFma(lambda, -(invMassA * d), vA);
	wA = rowFma(lambda, -JA, wA);
	vB = rowFma(lambda, invMassB * d, vB);
	wB = rowFma(lambda, JB, wB);
}
// clamp(x, -m, m) for m >= 0 (median of three: one instruction on the device; the oracle's clampf gives the same value for every non-NaN input)
MI_DEV float rowClampSym(float x, float m)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_fmed3f(x, -m, m);
#else
	return x < -m ? -m : (x > m ? m : x);
#endif
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
		float maxFriction = friction * impulseN;
		float newImpulse = rowClampSym(__builtin_fmaf(-mT, vt, impulseT), maxFriction);
		float lambda = newImpulse - impulseT;
		impulseT = newImpulse;
		rowApply(lambda, t, JtA, JtB, invMassA, invMassB, vA, wA, vB, wB);
	}
	{ // normal (constraints.cpp:3426-3442)
		float vn = rowVelocity(n, cAn, cBn, vA, wA, vB, wB);
		float impulse = fmaxf(__builtin_fmaf(-mN, vn - bias, impulseN), 0.f);
		float lambda = impulse - impulseN;
		impulseN = impulse;
		rowApply(lambda, n, JnA, JnB, invMassA, invMassB, vA, wA, vB, wB);
	}
	r.lam = make_float2(impulseN, impulseT);
}
