// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// 8-lane ("AVX2-shaped") contact path restated from the reference's constraints.cpp:3451-3709 and
// constraints.h:641-662: SoA batches of 8 contacts scheduled by scheduleConstraintsSIMD, body state gathered
// per lane from the 104-byte AoS records, friction row then normal row, scattered back.  Written as plain
// 8-wide loops that gcc vectorises under -O3 -mavx2 -mfma; this is the `cpu_baseline` ("port") that bench.py
// times on the GPU box's host core.  Differences from the reference's wide math, all documented in DESIGN.md:
//   * noz() uses exact 1/sqrtf instead of _mm256_rsqrt_ps (math_simd.h:283-289) — vendor-independent bits — unless
//     wideApproxRsqrt() is switched on (orc_set_wide_rsqrt): then the host's own rsqrtss (12-bit estimate, no Newton step: what the
//     reference's noz executes; its bits differ between CPU vendors) is used, so that the distance between "exact" and "AVX2
//     semantics" can be reported (tests/test_oracle.py);
//   * contact indices are u32 (reference truncates to u16 at constraints.cpp:3473).
#pragma once
#include "oconstraints.h"

namespace orc {

static const u32 WIDE = 8;
// (wideApproxRsqrt() / rsqrtEstimate(): owidemath.h)

// constraints.h:641-662 (30 float rows x 8 + 2 x 8 ids)
struct alignas(32) simd_collision_constraint_batch
{
	float relGlobalAnchorA[3][WIDE];
	float relGlobalAnchorB[3][WIDE];
	float normal[3][WIDE];
	float tangent[3][WIDE];
	float normalImpulseToAngularVelocityA[3][WIDE];
	float tangentImpulseToAngularVelocityA[3][WIDE];
	float normalImpulseToAngularVelocityB[3][WIDE];
	float tangentImpulseToAngularVelocityB[3][WIDE];
	float effectiveMassInNormalDir[WIDE];
	float effectiveMassInTangentDir[WIDE];
	float friction[WIDE];
	float impulseInNormalDir[WIDE];
	float impulseInTangentDir[WIDE];
	float bias[WIDE];
	u32 rbAIndices[WIDE];
	u32 rbBIndices[WIDE];
};

struct wvec3 { float x[WIDE], y[WIDE], z[WIDE]; };

#define ORC_LANES for (u32 l = 0; l < WIDE; ++l)

static inline void wcross(const wvec3& a, const wvec3& b, wvec3& r)
{
	ORC_LANES { r.x[l] = a.y[l] * b.z[l] - a.z[l] * b.y[l]; r.y[l] = a.z[l] * b.x[l] - a.x[l] * b.z[l]; r.z[l] = a.x[l] * b.y[l] - a.y[l] * b.x[l]; }
}
static inline void wdot(const wvec3& a, const wvec3& b, float* r) { ORC_LANES { r[l] = a.x[l] * b.x[l] + a.y[l] * b.y[l] + a.z[l] * b.z[l]; } }

static inline void wload3(wvec3& v, const float src[3][WIDE]) { ORC_LANES { v.x[l] = src[0][l]; v.y[l] = src[1][l]; v.z[l] = src[2][l]; } }
static inline void wstore3(const wvec3& v, float dst[3][WIDE]) { ORC_LANES { dst[0][l] = v.x[l]; dst[1][l] = v.y[l]; dst[2][l] = v.z[l]; } }

// constraints.cpp:3451-3616
static inline void initializeCollisionBatchesWide(std::vector<simd_collision_constraint_batch>& batches, const std::vector<sched_slot>& slots,
	const rigid_body_global_state* rbs, const collision_contact* contacts, const constraint_body_pair* bodyPairs, float dt)
{
	batches.resize(slots.size());
	const float slop = -0.001f, scale = 0.1f, invDt = 1.f / dt;
	for (size_t i = 0; i < slots.size(); ++i)
	{
		const sched_slot& slot = slots[i];
		simd_collision_constraint_batch& batch = batches[i];

		wvec3 point, normal, vA, wA, vB, wB, posA, posB;
		float depth[WIDE], friction[WIDE], restitution[WIDE], invMassA[WIDE], invMassB[WIDE];
		float iA[9][WIDE], iB[9][WIDE];
		ORC_LANES // gather (load8 / 8x8 transposes in the reference, simd.h:852-867)
		{
			u32 ci = slot.indices[l];
			const collision_contact& c = contacts[ci];
			batch.rbAIndices[l] = bodyPairs[ci].rbA;
			batch.rbBIndices[l] = bodyPairs[ci].rbB;
			const rigid_body_global_state& a = rbs[bodyPairs[ci].rbA];
			const rigid_body_global_state& b = rbs[bodyPairs[ci].rbB];
			point.x[l] = c.point.x; point.y[l] = c.point.y; point.z[l] = c.point.z; depth[l] = c.penetrationDepth;
			normal.x[l] = c.normal.x; normal.y[l] = c.normal.y; normal.z[l] = c.normal.z;
			friction[l] = (float)(c.friction_restitution >> 16) / (float)0xFFFF;
			restitution[l] = (float)(c.friction_restitution & 0xFFFF) / (float)0xFFFF;
			for (u32 k = 0; k < 9; ++k) { iA[k][l] = a.invInertia.m()[k]; iB[k][l] = b.invInertia.m()[k]; }
			invMassA[l] = a.invMass; invMassB[l] = b.invMass;
			vA.x[l] = a.linearVelocity.x; vA.y[l] = a.linearVelocity.y; vA.z[l] = a.linearVelocity.z;
			wA.x[l] = a.angularVelocity.x; wA.y[l] = a.angularVelocity.y; wA.z[l] = a.angularVelocity.z;
			vB.x[l] = b.linearVelocity.x; vB.y[l] = b.linearVelocity.y; vB.z[l] = b.linearVelocity.z;
			wB.x[l] = b.angularVelocity.x; wB.y[l] = b.angularVelocity.y; wB.z[l] = b.angularVelocity.z;
			posA.x[l] = a.position.x; posA.y[l] = a.position.y; posA.z[l] = a.position.z;
			posB.x[l] = b.position.x; posB.y[l] = b.position.y; posB.z[l] = b.position.z;
		}

		wvec3 rA, rB, avA, avB, rel, tangent, tmp;
		ORC_LANES { rA.x[l] = point.x[l] - posA.x[l]; rA.y[l] = point.y[l] - posA.y[l]; rA.z[l] = point.z[l] - posA.z[l];
		            rB.x[l] = point.x[l] - posB.x[l]; rB.y[l] = point.y[l] - posB.y[l]; rB.z[l] = point.z[l] - posB.z[l]; }
		wcross(wA, rA, tmp); ORC_LANES { avA.x[l] = vA.x[l] + tmp.x[l]; avA.y[l] = vA.y[l] + tmp.y[l]; avA.z[l] = vA.z[l] + tmp.z[l]; }
		wcross(wB, rB, tmp); ORC_LANES { avB.x[l] = vB.x[l] + tmp.x[l]; avB.y[l] = vB.y[l] + tmp.y[l]; avB.z[l] = vB.z[l] + tmp.z[l]; }
		ORC_LANES { rel.x[l] = avB.x[l] - avA.x[l]; rel.y[l] = avB.y[l] - avA.y[l]; rel.z[l] = avB.z[l] - avA.z[l]; }
		float nDotRel[WIDE]; wdot(normal, rel, nDotRel);
		ORC_LANES
		{
			float tx = rel.x[l] - nDotRel[l] * normal.x[l], ty = rel.y[l] - nDotRel[l] * normal.y[l], tz = rel.z[l] - nDotRel[l] * normal.z[l];
			float sl = tx * tx + ty * ty + tz * tz;
			float inv = (sl < 1e-8f) ? 0.f : (wideApproxRsqrt() ? rsqrtEstimate(sl) : (1.f / sqrtf(sl))); // noz (math_simd.h:283-289): exact, or the hardware estimate on request
			tangent.x[l] = tx * inv; tangent.y[l] = ty * inv; tangent.z[l] = tz * inv;
		}
		wstore3(rA, batch.relGlobalAnchorA); wstore3(rB, batch.relGlobalAnchorB);
		wstore3(normal, batch.normal); wstore3(tangent, batch.tangent);
		ORC_LANES { batch.impulseInNormalDir[l] = 0.f; batch.impulseInTangentDir[l] = 0.f; batch.friction[l] = friction[l]; }

		auto direction = [&](const wvec3& d, float* effMass, float outA[3][WIDE], float outB[3][WIDE])
		{
			wvec3 crA, crB, jA, jB;
			wcross(rA, d, crA); wcross(rB, d, crB);
			ORC_LANES
			{
				jA.x[l] = iA[0][l] * crA.x[l] + iA[3][l] * crA.y[l] + iA[6][l] * crA.z[l];
				jA.y[l] = iA[1][l] * crA.x[l] + iA[4][l] * crA.y[l] + iA[7][l] * crA.z[l];
				jA.z[l] = iA[2][l] * crA.x[l] + iA[5][l] * crA.y[l] + iA[8][l] * crA.z[l];
				jB.x[l] = iB[0][l] * crB.x[l] + iB[3][l] * crB.y[l] + iB[6][l] * crB.z[l];
				jB.y[l] = iB[1][l] * crB.x[l] + iB[4][l] * crB.y[l] + iB[7][l] * crB.z[l];
				jB.z[l] = iB[2][l] * crB.x[l] + iB[5][l] * crB.y[l] + iB[8][l] * crB.z[l];
			}
			float dA[WIDE], dB[WIDE]; wdot(crA, jA, dA); wdot(crB, jB, dB);
			ORC_LANES { float inv = invMassA[l] + dA[l] + invMassB[l] + dB[l]; effMass[l] = (inv != 0.f) ? (1.f / inv) : 0.f; }
			wstore3(jA, outA); wstore3(jB, outB);
		};
		direction(tangent, batch.effectiveMassInTangentDir, batch.tangentImpulseToAngularVelocityA, batch.tangentImpulseToAngularVelocityB);
		direction(normal, batch.effectiveMassInNormalDir, batch.normalImpulseToAngularVelocityA, batch.normalImpulseToAngularVelocityB);

		ORC_LANES
		{
			float bias = 0.f;
			if (dt > ORC_DT_THRESHOLD)
			{
				float vRel = nDotRel[l];
				float bounceBias = -restitution[l] * vRel - scale * (-depth[l] - slop) * invDt;
				bias = ((-depth[l] < slop) & (vRel < 0.f)) ? bounceBias : bias;
			}
			batch.bias[l] = bias;
		}
	}
}

// constraints.cpp:3618-3709
static inline void solveCollisionBatchesWide(std::vector<simd_collision_constraint_batch>& batches, rigid_body_global_state* rbs)
{
	for (size_t i = 0; i < batches.size(); ++i)
	{
		simd_collision_constraint_batch& batch = batches[i];
		wvec3 vA, wA, vB, wB;
		float invMassA[WIDE], invMassB[WIDE];
		ORC_LANES // gather
		{
			const rigid_body_global_state& a = rbs[batch.rbAIndices[l]];
			const rigid_body_global_state& b = rbs[batch.rbBIndices[l]];
			invMassA[l] = a.invMass; invMassB[l] = b.invMass;
			vA.x[l] = a.linearVelocity.x; vA.y[l] = a.linearVelocity.y; vA.z[l] = a.linearVelocity.z;
			wA.x[l] = a.angularVelocity.x; wA.y[l] = a.angularVelocity.y; wA.z[l] = a.angularVelocity.z;
			vB.x[l] = b.linearVelocity.x; vB.y[l] = b.linearVelocity.y; vB.z[l] = b.linearVelocity.z;
			wB.x[l] = b.angularVelocity.x; wB.y[l] = b.angularVelocity.y; wB.z[l] = b.angularVelocity.z;
		}
		wvec3 rA, rB, n, t, jA, jB, tmpA, tmpB;
		wload3(rA, batch.relGlobalAnchorA); wload3(rB, batch.relGlobalAnchorB);
		wload3(n, batch.normal); wload3(t, batch.tangent);

		auto rowSolve = [&](const wvec3& d, const float* lambdaIn)
		{
			ORC_LANES
			{
				float lam = lambdaIn[l];
				float ia = invMassA[l] * lam, ib = invMassB[l] * lam;
				vA.x[l] -= ia * d.x[l]; vA.y[l] -= ia * d.y[l]; vA.z[l] -= ia * d.z[l];
				wA.x[l] -= jA.x[l] * lam; wA.y[l] -= jA.y[l] * lam; wA.z[l] -= jA.z[l] * lam;
				vB.x[l] += ib * d.x[l]; vB.y[l] += ib * d.y[l]; vB.z[l] += ib * d.z[l];
				wB.x[l] += jB.x[l] * lam; wB.y[l] += jB.y[l] * lam; wB.z[l] += jB.z[l] * lam;
			}
		};
		auto relDot = [&](const wvec3& d, float* out)
		{
			wcross(wA, rA, tmpA); wcross(wB, rB, tmpB);
			ORC_LANES
			{
				float rx = (vB.x[l] + tmpB.x[l]) - (vA.x[l] + tmpA.x[l]);
				float ry = (vB.y[l] + tmpB.y[l]) - (vA.y[l] + tmpA.y[l]);
				float rz = (vB.z[l] + tmpB.z[l]) - (vA.z[l] + tmpA.z[l]);
				out[l] = rx * d.x[l] + ry * d.y[l] + rz * d.z[l];
			}
		};

		float lambda[WIDE], vrel[WIDE];
		{ // tangent
			wload3(jA, batch.tangentImpulseToAngularVelocityA); wload3(jB, batch.tangentImpulseToAngularVelocityB);
			relDot(t, vrel);
			ORC_LANES
			{
				float lam = -batch.effectiveMassInTangentDir[l] * vrel[l];
				float maxFriction = batch.friction[l] * batch.impulseInNormalDir[l];
				float newImpulse = clampf(batch.impulseInTangentDir[l] + lam, -maxFriction, maxFriction);
				lambda[l] = newImpulse - batch.impulseInTangentDir[l];
				batch.impulseInTangentDir[l] = newImpulse;
			}
			rowSolve(t, lambda);
		}
		{ // normal
			wload3(jA, batch.normalImpulseToAngularVelocityA); wload3(jB, batch.normalImpulseToAngularVelocityB);
			relDot(n, vrel);
			ORC_LANES
			{
				float lam = -batch.effectiveMassInNormalDir[l] * (vrel[l] - batch.bias[l]);
				float impulse = std::max(batch.impulseInNormalDir[l] + lam, 0.f);
				lambda[l] = impulse - batch.impulseInNormalDir[l];
				batch.impulseInNormalDir[l] = impulse;
			}
			rowSolve(n, lambda);
		}
		ORC_LANES // scatter (store8, simd.h:895-910); padding lanes rewrite lane 0's identical values
		{
			rigid_body_global_state& a = rbs[batch.rbAIndices[l]];
			rigid_body_global_state& b = rbs[batch.rbBIndices[l]];
			a.linearVelocity = vec3(vA.x[l], vA.y[l], vA.z[l]); a.angularVelocity = vec3(wA.x[l], wA.y[l], wA.z[l]);
			b.linearVelocity = vec3(vB.x[l], vB.y[l], vB.z[l]); b.angularVelocity = vec3(wB.x[l], wB.y[l], wB.z[l]);
		}
	}
}

} // namespace orc
