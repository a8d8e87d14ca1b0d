// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// Constraint solver restated from the reference's src/physics/constraints.{h,cpp}: the greedy batch scheduler
// (:51-184), and the SCALAR init/solve of every constraint type (primary oracle).  An 8-lane
// batch variant of the contact path (the AVX2 shape, :3451-3709) lives in owide.h and is the cpu_baseline.
#pragma once
#include "oshapes.h"
#include "owidemath.h"
#include <vector>

namespace orc {

#define ORC_DISTANCE_CONSTRAINT_BETA 0.1f        // constraints.cpp:9-17
#define ORC_BALL_CONSTRAINT_BETA 0.1f
#define ORC_SLIDER_CONSTRAINT_BETA 0.1f
#define ORC_HINGE_ROTATION_CONSTRAINT_BETA 0.3f
#define ORC_HINGE_LIMIT_CONSTRAINT_BETA 0.1f
#define ORC_TWIST_LIMIT_CONSTRAINT_BETA 0.1f
#define ORC_SLIDER_LIMIT_CONSTRAINT_BETA 0.1f
#define ORC_DT_THRESHOLD 1e-5f

// rigid_body.h:6-16 (field order is what the reference's load8 offsets depend on; `position` is the world COG).
struct rigid_body_global_state
{
	quat rotation;
	vec3 localCOGPosition;
	vec3 position;
	mat3 invInertia;
	float invMass;
	vec3 linearVelocity;
	vec3 angularVelocity;
};
static_assert(sizeof(rigid_body_global_state) == 104, "104 B like the reference");

enum constraint_motor_type : u32 { constraint_velocity_motor, constraint_position_motor };

// constraints.h:73-80, 129-135, 175-183, 229-257, 346-380, 497-520 — persistent POD inputs (page_size members dropped).
struct distance_constraint { vec3 localAnchorA, localAnchorB; float globalLength; };
struct ball_constraint { vec3 localAnchorA, localAnchorB; };
struct fixed_constraint { quat initialInvRotationDifference; vec3 localAnchorA, localAnchorB; };
struct hinge_constraint
{
	vec3 localAnchorA, localAnchorB, localHingeAxisA, localHingeAxisB;
	float minRotationLimit, maxRotationLimit;
	float maxMotorTorque;
	constraint_motor_type motorType;
	float motorVelocity; // union with motorTargetAngle
	vec3 localHingeTangentA, localHingeBitangentA, localHingeTangentB;
};
struct cone_twist_constraint
{
	vec3 localAnchorA, localAnchorB;
	vec3 localLimitAxisA, localLimitAxisB;
	vec3 localLimitTangentA, localLimitBitangentA, localLimitTangentB;
	float swingLimit, twistLimit;
	constraint_motor_type swingMotorType;
	float swingMotorVelocity; // union with swingMotorTargetAngle
	float maxSwingMotorTorque;
	float swingMotorAxis;
	constraint_motor_type twistMotorType;
	float twistMotorVelocity; // union with twistMotorTargetAngle
	float maxTwistMotorTorque;
};
struct slider_constraint
{
	quat initialInvRotationDifference;
	vec3 localAnchorA, localAnchorB;
	vec3 localAxisA;
	float negDistanceLimit, posDistanceLimit;
	float maxMotorForce;
	constraint_motor_type motorType;
	float motorVelocity; // union with motorTargetDistance
};
static_assert(sizeof(distance_constraint) == 28 && sizeof(ball_constraint) == 24 && sizeof(fixed_constraint) == 40, "POD layout");
static_assert(sizeof(hinge_constraint) == 104 && sizeof(cone_twist_constraint) == 120 && sizeof(slider_constraint) == 72, "POD layout");

// ---------------------------------------------------------------------------------------------------
// scheduleConstraintsSIMD — constraints.cpp:51-184, W = 8 lanes, 4 buckets,
// indices widened to u32 (the packed 16-bit compare becomes two 32-bit compares per stored id).
// Output: slots of W constraint indices; padding lanes duplicate lane 0.
// ---------------------------------------------------------------------------------------------------
static const u32 SCHED_W = 8;
static const u32 SCHED_INVALID = 0xFFFFFFFFu;
struct sched_slot { u32 indices[SCHED_W]; };

static inline u32 scheduleConstraintsSIMD(const constraint_body_pair* bodyPairs, u32 numBodyPairs, u32 dummyRigidBodyIndex, std::vector<sched_slot>& out)
{
	struct entry { u32 a[SCHED_W], b[SCHED_W]; sched_slot slot; };
	const u32 numBuckets = 4;
	std::vector<entry> buckets[numBuckets];
	u32 count[numBuckets] = { 0, 0, 0, 0 };
	auto invalidEntry = []() { entry e; for (u32 l = 0; l < SCHED_W; ++l) { e.a[l] = e.b[l] = SCHED_INVALID; e.slot.indices[l] = 0; } return e; };
	for (u32 b = 0; b < numBuckets; ++b) { buckets[b].push_back(invalidEntry()); } // sentinel (:68-75)
	out.clear();

	for (u32 i = 0; i < numBodyPairs; ++i)
	{
		constraint_body_pair bp = bodyPairs[i];
		u32 rbA = (bp.rbA == dummyRigidBodyIndex) ? bp.rbB : bp.rbA; // :82-83
		u32 rbB = (bp.rbB == dummyRigidBodyIndex) ? bp.rbA : bp.rbB;
		u32 bucket = i % numBuckets;
		std::vector<entry>& es = buckets[bucket];

		u32 j = 0;
		for (;; ++j) // :111-121 — sentinel at es[count] always accepts
		{
			const entry& e = es[j];
			bool conflict = false;
			for (u32 l = 0; l < SCHED_W; ++l)
			{
				// an empty lane holds INVALID in both halves; a real id never equals INVALID
				if (e.a[l] == rbA || e.b[l] == rbA || e.a[l] == rbB || e.b[l] == rbB) { conflict = true; break; }
			}
			if (!conflict) { break; }
		}
		entry& e = es[j];
		u32 lane = 0;
		while (!(e.a[lane] == SCHED_INVALID && e.b[lane] == SCHED_INVALID)) { ++lane; } // lowest empty lane (:126)
		e.slot.indices[lane] = i;
		e.a[lane] = bp.rbA; e.b[lane] = bp.rbB; // original ids incl. dummy (:132)

		u32& c = count[bucket];
		if (j == c)
		{
			++c;
			if (es.size() <= c) { es.push_back(invalidEntry()); } else { es[c] = invalidEntry(); }
		}
		else if (lane == SCHED_W - 1)
		{
			sched_slot full = e.slot;
			--c;
			es[j] = es[c];        // swap and pop (:149-151)
			out.push_back(full);
			es[c] = invalidEntry();
		}
	}
	for (u32 b = 0; b < numBuckets; ++b) // :162-179
	{
		for (u32 i = 0; i < count[b]; ++i)
		{
			entry& e = buckets[b][i];
			sched_slot s = e.slot;
			for (u32 l = 0; l < SCHED_W; ++l) { if (e.a[l] == SCHED_INVALID && e.b[l] == SCHED_INVALID) { s.indices[l] = e.slot.indices[0]; } }
			out.push_back(s);
		}
	}
	return (u32)out.size();
}

// ---------------------------------------------------------------------------------------------------
// Collision constraint — constraints.h:615-633, constraints.cpp:3307-3449
// ---------------------------------------------------------------------------------------------------
struct collision_constraint
{
	vec3 relGlobalAnchorA, relGlobalAnchorB, tangent;
	vec3 tangentImpulseToAngularVelocityA, tangentImpulseToAngularVelocityB;
	vec3 normalImpulseToAngularVelocityA, normalImpulseToAngularVelocityB;
	float impulseInNormalDir, impulseInTangentDir, effectiveMassInNormalDir, effectiveMassInTangentDir, bias;
	vec3 crAt, crBt, crAn, crBn; // rA x tangent, rB x tangent, rA x normal, rB x normal: the angular parts of the row Jacobians (kept for the row-form solve)
};

static inline void initializeCollisionConstraint(collision_constraint& constraint, const rigid_body_global_state* rbs, const collision_contact& contact, constraint_body_pair pair, float dt)
{
	float invDt = 1.f / dt;
	const rigid_body_global_state& rbA = rbs[pair.rbA];
	const rigid_body_global_state& rbB = rbs[pair.rbB];
	constraint.impulseInNormalDir = 0.f;
	constraint.impulseInTangentDir = 0.f;
	constraint.relGlobalAnchorA = contact.point - rbA.position;
	constraint.relGlobalAnchorB = contact.point - rbB.position;
	vec3 anchorVelocityA = rbA.linearVelocity + cross(rbA.angularVelocity, constraint.relGlobalAnchorA);
	vec3 anchorVelocityB = rbB.linearVelocity + cross(rbB.angularVelocity, constraint.relGlobalAnchorB);
	vec3 relVelocity = anchorVelocityB - anchorVelocityA;
	constraint.tangent = relVelocity - dot(contact.normal, relVelocity) * contact.normal;
	constraint.tangent = noz(constraint.tangent);
	{
		vec3 crAt = cross(constraint.relGlobalAnchorA, constraint.tangent);
		vec3 crBt = cross(constraint.relGlobalAnchorB, constraint.tangent);
		float invMassInTangentDir = rbA.invMass + dot(crAt, rbA.invInertia * crAt) + rbB.invMass + dot(crBt, rbB.invInertia * crBt);
		constraint.effectiveMassInTangentDir = (invMassInTangentDir != 0.f) ? (1.f / invMassInTangentDir) : 0.f;
		constraint.tangentImpulseToAngularVelocityA = rbA.invInertia * crAt;
		constraint.tangentImpulseToAngularVelocityB = rbB.invInertia * crBt;
		constraint.crAt = crAt; constraint.crBt = crBt;
	}
	{
		vec3 crAn = cross(constraint.relGlobalAnchorA, contact.normal);
		vec3 crBn = cross(constraint.relGlobalAnchorB, contact.normal);
		float invMassInNormalDir = rbA.invMass + dot(crAn, rbA.invInertia * crAn) + rbB.invMass + dot(crBn, rbB.invInertia * crBn);
		constraint.effectiveMassInNormalDir = (invMassInNormalDir != 0.f) ? (1.f / invMassInNormalDir) : 0.f;
		constraint.bias = 0.f;
		if (dt > ORC_DT_THRESHOLD)
		{
			float vRel = dot(contact.normal, relVelocity);
			const float slop = -0.001f;
			if (-contact.penetrationDepth < slop && vRel < 0.f)
			{
				float restitution = (float)(contact.friction_restitution & 0xFFFF) / (float)0xFFFF;
				constraint.bias = -restitution * vRel - 0.1f * (-contact.penetrationDepth - slop) * invDt;
			}
		}
		constraint.normalImpulseToAngularVelocityA = rbA.invInertia * crAn;
		constraint.normalImpulseToAngularVelocityB = rbB.invInertia * crBn;
		constraint.crAn = crAn; constraint.crBn = crBn;
	}
}

// One contact, friction row then normal row (constraints.cpp:3385-3447).  skipStaticPair mirrors the scalar
// path's early-out (:3394-3397); the 8-wide path does not have it (effective masses are 0 anyway).
static inline void solveCollisionConstraint(collision_constraint& constraint, const collision_contact& contact, constraint_body_pair pair, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[pair.rbA];
	rigid_body_global_state& rbB = rbs[pair.rbB];
	if (rbA.invMass == 0.f && rbB.invMass == 0.f) { return; }
	vec3 vA = rbA.linearVelocity, wA = rbA.angularVelocity, vB = rbB.linearVelocity, wB = rbB.angularVelocity;
	{
		vec3 anchorVelocityA = vA + cross(wA, constraint.relGlobalAnchorA);
		vec3 anchorVelocityB = vB + cross(wB, constraint.relGlobalAnchorB);
		vec3 relVelocity = anchorVelocityB - anchorVelocityA;
		float vt = dot(relVelocity, constraint.tangent);
		float lambda = -constraint.effectiveMassInTangentDir * vt;
		float friction = (float)(contact.friction_restitution >> 16) / (float)0xFFFF;
		float maxFriction = friction * constraint.impulseInNormalDir;
		float newImpulse = clampf(constraint.impulseInTangentDir + lambda, -maxFriction, maxFriction);
		lambda = newImpulse - constraint.impulseInTangentDir;
		constraint.impulseInTangentDir = newImpulse;
		vec3 P = lambda * constraint.tangent;
		vA -= rbA.invMass * P;
		wA -= constraint.tangentImpulseToAngularVelocityA * lambda;
		vB += rbB.invMass * P;
		wB += constraint.tangentImpulseToAngularVelocityB * lambda;
	}
	{
		vec3 anchorVelocityA = vA + cross(wA, constraint.relGlobalAnchorA);
		vec3 anchorVelocityB = vB + cross(wB, constraint.relGlobalAnchorB);
		vec3 relVelocity = anchorVelocityB - anchorVelocityA;
		float vn = dot(relVelocity, contact.normal);
		float lambda = -constraint.effectiveMassInNormalDir * (vn - constraint.bias);
		float impulse = std::max(constraint.impulseInNormalDir + lambda, 0.f);
		lambda = impulse - constraint.impulseInNormalDir;
		constraint.impulseInNormalDir = impulse;
		vec3 P = lambda * contact.normal;
		vA -= rbA.invMass * P;
		wA -= constraint.normalImpulseToAngularVelocityA * lambda;
		vB += rbB.invMass * P;
		wB += constraint.normalImpulseToAngularVelocityB * lambda;
	}
	rbA.linearVelocity = vA; rbA.angularVelocity = wA;
	rbB.linearVelocity = vB; rbB.angularVelocity = wB;
}

// The same contact in ROW FORM, as the device evaluates it (directx-renderer-kurth_amd/csrc/solver_rows.h): the relative anchor velocity
// projected on the row direction d, dot((vB + wB x rB) - (vA + wA x rA), d), is written with the scalar triple product as
// d . (vB - vA) + (rB x d) . wB - (rA x d) . wA and evaluated with fused multiply-adds in the association of the reference's wide
// dot (fmadd(a.x, b.x, fmadd(a.y, b.y, a.z * b.z)), core/math_simd.h:241) per body vector; the impulse is applied as
// v -+= lambda * (invMass * d), w -+= lambda * J, one fma per component.  Same mathematics as solveCollisionConstraint above, different rounding: this is the
// function the device is compared with bit for bit; tests/test_oracle.py bounds its distance from the reference formula.
// (quad form: four 3-term partial dots, one per body vector, each accumulated z, y, x with fused multiply-adds, body A's negated in the
// operands, combined as (pA_lin + pA_ang) + (pB_lin + pB_ang) — the device's four lanes per row and their two in-quad adds)
static inline float rowDot3(vec3 x, vec3 d) { float s = x.z * d.z; s = __builtin_fmaf(x.y, d.y, s); return __builtin_fmaf(x.x, d.x, s); }
static inline float rowVelocity(vec3 d, vec3 cA, vec3 cB, vec3 vA, vec3 wA, vec3 vB, vec3 wB)
{
	return (rowDot3(vA, -d) + rowDot3(wA, -cA)) + (rowDot3(vB, d) + rowDot3(wB, cB));
}
static inline vec3 rowFma(float d, vec3 a, vec3 x) { return vec3(__builtin_fmaf(d, a.x, x.x), __builtin_fmaf(d, a.y, x.y), __builtin_fmaf(d, a.z, x.z)); }
static inline void rowApply(float lambda, vec3 d, vec3 JA, vec3 JB, float invMassA, float invMassB, vec3& vA, vec3& wA, vec3& vB, vec3& wB)
{
	vA = rowFma(lambda, -(invMassA * d), vA);
	wA = rowFma(lambda, -JA, wA);
	vB = rowFma(lambda, invMassB * d, vB);
	wB = rowFma(lambda, JB, wB);
}
static inline void solveCollisionConstraintRowForm(collision_constraint& constraint, const collision_contact& contact, constraint_body_pair pair, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[pair.rbA];
	rigid_body_global_state& rbB = rbs[pair.rbB];
	vec3 vA = rbA.linearVelocity, wA = rbA.angularVelocity, vB = rbB.linearVelocity, wB = rbB.angularVelocity;
	{
		float vt = rowVelocity(constraint.tangent, constraint.crAt, constraint.crBt, vA, wA, vB, wB);
		float friction = (float)(contact.friction_restitution >> 16) / (float)0xFFFF;
		float maxFriction = friction * constraint.impulseInNormalDir;
		float newImpulse = clampf(__builtin_fmaf(-constraint.effectiveMassInTangentDir, vt, constraint.impulseInTangentDir), -maxFriction, maxFriction);
		float lambda = newImpulse - constraint.impulseInTangentDir;
		constraint.impulseInTangentDir = newImpulse;
		rowApply(lambda, constraint.tangent, constraint.tangentImpulseToAngularVelocityA, constraint.tangentImpulseToAngularVelocityB, rbA.invMass, rbB.invMass, vA, wA, vB, wB);
	}
	{
		float vn = rowVelocity(contact.normal, constraint.crAn, constraint.crBn, vA, wA, vB, wB);
		float impulse = std::max(__builtin_fmaf(-constraint.effectiveMassInNormalDir, vn - constraint.bias, constraint.impulseInNormalDir), 0.f);
		float lambda = impulse - constraint.impulseInNormalDir;
		constraint.impulseInNormalDir = impulse;
		rowApply(lambda, contact.normal, constraint.normalImpulseToAngularVelocityA, constraint.normalImpulseToAngularVelocityB, rbA.invMass, rbB.invMass, vA, wA, vB, wB);
	}
	rbA.linearVelocity = vA; rbA.angularVelocity = wA;
	rbB.linearVelocity = vB; rbB.angularVelocity = wB;
}

// ---------------------------------------------------------------------------------------------------
// Distance — constraints.h:82-96, constraints.cpp:189-264
// ---------------------------------------------------------------------------------------------------
struct distance_constraint_update
{
	u32 rigidBodyIndexA, rigidBodyIndexB;
	vec3 relGlobalAnchorA, relGlobalAnchorB, impulseToAngularVelocityA, impulseToAngularVelocityB, u;
	float bias, effectiveMass;
};
static inline void initializeDistanceConstraint(distance_constraint_update& out, const rigid_body_global_state* rbs, const distance_constraint& in, constraint_body_pair bp, float dt)
{
	float invDt = 1.f / dt;
	out.rigidBodyIndexA = bp.rbA; out.rigidBodyIndexB = bp.rbB;
	const rigid_body_global_state& globalA = rbs[bp.rbA];
	const rigid_body_global_state& globalB = rbs[bp.rbB];
	out.relGlobalAnchorA = globalA.rotation * (in.localAnchorA - globalA.localCOGPosition);
	out.relGlobalAnchorB = globalB.rotation * (in.localAnchorB - globalB.localCOGPosition);
	vec3 globalAnchorA = globalA.position + out.relGlobalAnchorA;
	vec3 globalAnchorB = globalB.position + out.relGlobalAnchorB;
	out.u = globalAnchorB - globalAnchorA;
	float l = length(out.u);
	out.u = (l > 0.001f) ? (out.u * (1.f / l)) : vec3(0.f);
	vec3 crAu = cross(out.relGlobalAnchorA, out.u);
	vec3 crBu = cross(out.relGlobalAnchorB, out.u);
	float invMass = globalA.invMass + dot(crAu, globalA.invInertia * crAu) + globalB.invMass + dot(crBu, globalB.invInertia * crBu);
	out.effectiveMass = (invMass != 0.f) ? (1.f / invMass) : 0.f;
	out.bias = 0.f;
	if (dt > ORC_DT_THRESHOLD) { out.bias = (l - in.globalLength) * (ORC_DISTANCE_CONSTRAINT_BETA * invDt); }
	out.impulseToAngularVelocityA = globalA.invInertia * cross(out.relGlobalAnchorA, crAu);
	out.impulseToAngularVelocityB = globalB.invInertia * cross(out.relGlobalAnchorB, crBu);
}
static inline void solveDistanceConstraint(distance_constraint_update& con, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[con.rigidBodyIndexA];
	rigid_body_global_state& rbB = rbs[con.rigidBodyIndexB];
	vec3 anchorVelocityA = rbA.linearVelocity + cross(rbA.angularVelocity, con.relGlobalAnchorA);
	vec3 anchorVelocityB = rbB.linearVelocity + cross(rbB.angularVelocity, con.relGlobalAnchorB);
	float Cdot = dot(con.u, anchorVelocityB - anchorVelocityA) + con.bias;
	float lambda = -con.effectiveMass * Cdot;
	vec3 P = lambda * con.u;
	rbA.linearVelocity -= rbA.invMass * P;
	rbA.angularVelocity -= con.impulseToAngularVelocityA * lambda;
	rbB.linearVelocity += rbB.invMass * P;
	rbB.angularVelocity += con.impulseToAngularVelocityB * lambda;
}

// ---------------------------------------------------------------------------------------------------
// Ball — constraints.h:137-146, constraints.cpp:460-528
// ---------------------------------------------------------------------------------------------------
struct ball_constraint_update
{
	u32 rigidBodyIndexA, rigidBodyIndexB;
	vec3 relGlobalAnchorA, relGlobalAnchorB, bias;
	mat3 invEffectiveMass;
};
static inline mat3 pointBlockInvEffectiveMass(const rigid_body_global_state& globalA, const rigid_body_global_state& globalB, vec3 rA, vec3 rB)
{
	mat3 skewMatA = getSkewMatrix(rA);
	mat3 skewMatB = getSkewMatrix(rB);
	return skewMatA * globalA.invInertia * transpose(skewMatA)
		+ skewMatB * globalB.invInertia * transpose(skewMatB)
		+ mat3::identity() * (globalA.invMass + globalB.invMass);
}
static inline void initializeBallConstraint(ball_constraint_update& out, const rigid_body_global_state* rbs, const ball_constraint& in, constraint_body_pair bp, float dt)
{
	float invDt = 1.f / dt;
	out.rigidBodyIndexA = bp.rbA; out.rigidBodyIndexB = bp.rbB;
	const rigid_body_global_state& globalA = rbs[bp.rbA];
	const rigid_body_global_state& globalB = rbs[bp.rbB];
	out.relGlobalAnchorA = globalA.rotation * (in.localAnchorA - globalA.localCOGPosition);
	out.relGlobalAnchorB = globalB.rotation * (in.localAnchorB - globalB.localCOGPosition);
	vec3 globalAnchorA = globalA.position + out.relGlobalAnchorA;
	vec3 globalAnchorB = globalB.position + out.relGlobalAnchorB;
	out.invEffectiveMass = pointBlockInvEffectiveMass(globalA, globalB, out.relGlobalAnchorA, out.relGlobalAnchorB);
	out.bias = vec3(0.f);
	if (dt > ORC_DT_THRESHOLD) { out.bias = (globalAnchorB - globalAnchorA) * (ORC_BALL_CONSTRAINT_BETA * invDt); }
}
static inline void solvePointBlock(vec3& vA, vec3& wA, vec3& vB, vec3& wB, const rigid_body_global_state& rbA, const rigid_body_global_state& rbB,
	vec3 rA, vec3 rB, vec3 bias, const mat3& invEffectiveMass)
{
	vec3 anchorVelocityA = vA + cross(wA, rA);
	vec3 anchorVelocityB = vB + cross(wB, rB);
	vec3 Cdot = anchorVelocityB - anchorVelocityA + bias;
	vec3 P = solveLinearSystem(invEffectiveMass, -Cdot);
	vA -= rbA.invMass * P;
	wA -= rbA.invInertia * cross(rA, P);
	vB += rbB.invMass * P;
	wB += rbB.invInertia * cross(rB, P);
}
static inline void solveBallConstraint(ball_constraint_update& con, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[con.rigidBodyIndexA];
	rigid_body_global_state& rbB = rbs[con.rigidBodyIndexB];
	solvePointBlock(rbA.linearVelocity, rbA.angularVelocity, rbB.linearVelocity, rbB.angularVelocity, rbA, rbB,
		con.relGlobalAnchorA, con.relGlobalAnchorB, con.bias, con.invEffectiveMass);
}

// ---------------------------------------------------------------------------------------------------
// Fixed — constraints.h:185-197, constraints.cpp:736-823
// ---------------------------------------------------------------------------------------------------
struct fixed_constraint_update
{
	u32 rigidBodyIndexA, rigidBodyIndexB;
	vec3 relGlobalAnchorA, relGlobalAnchorB, translationBias;
	mat3 invEffectiveTranslationMass;
	vec3 rotationBias;
	mat3 invEffectiveRotationMass;
};
static inline void initializeFixedConstraint(fixed_constraint_update& out, const rigid_body_global_state* rbs, const fixed_constraint& in, constraint_body_pair bp, float dt)
{
	float invDt = 1.f / dt;
	out.rigidBodyIndexA = bp.rbA; out.rigidBodyIndexB = bp.rbB;
	const rigid_body_global_state& globalA = rbs[bp.rbA];
	const rigid_body_global_state& globalB = rbs[bp.rbB];
	out.relGlobalAnchorA = globalA.rotation * (in.localAnchorA - globalA.localCOGPosition);
	out.relGlobalAnchorB = globalB.rotation * (in.localAnchorB - globalB.localCOGPosition);
	vec3 globalAnchorA = globalA.position + out.relGlobalAnchorA;
	vec3 globalAnchorB = globalB.position + out.relGlobalAnchorB;
	out.invEffectiveTranslationMass = pointBlockInvEffectiveMass(globalA, globalB, out.relGlobalAnchorA, out.relGlobalAnchorB);
	out.invEffectiveRotationMass = globalA.invInertia + globalB.invInertia;
	out.translationBias = vec3(0.f);
	out.rotationBias = vec3(0.f);
	if (dt > ORC_DT_THRESHOLD)
	{
		out.translationBias = (globalAnchorB - globalAnchorA) * (ORC_BALL_CONSTRAINT_BETA * invDt);
		quat rotationError = globalB.rotation * in.initialInvRotationDifference * conjugate(globalA.rotation);
		out.rotationBias = rotationError.v() * (ORC_SLIDER_CONSTRAINT_BETA * invDt * 2.f);
	}
}
static inline void solveFixedConstraint(fixed_constraint_update& con, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[con.rigidBodyIndexA];
	rigid_body_global_state& rbB = rbs[con.rigidBodyIndexB];
	{
		vec3 Cdot = rbB.angularVelocity - rbA.angularVelocity;
		vec3 rotationLambda = solveLinearSystem(con.invEffectiveRotationMass, -(Cdot + con.rotationBias));
		rbA.angularVelocity -= rbA.invInertia * rotationLambda;
		rbB.angularVelocity += rbB.invInertia * rotationLambda;
	}
	solvePointBlock(rbA.linearVelocity, rbA.angularVelocity, rbB.linearVelocity, rbB.angularVelocity, rbA, rbB,
		con.relGlobalAnchorA, con.relGlobalAnchorB, con.translationBias, con.invEffectiveTranslationMass);
}

// ---------------------------------------------------------------------------------------------------
// Hinge — constraints.h:259-297, constraints.cpp:1079-1307
// ---------------------------------------------------------------------------------------------------
struct hinge_constraint_update
{
	u32 rigidBodyIndexA, rigidBodyIndexB;
	vec3 relGlobalAnchorA, relGlobalAnchorB, translationBias;
	mat3 invEffectiveTranslationMass;
	vec2 rotationBias;
	mat2 invEffectiveRotationMass;
	vec3 bxa, cxa;
	vec3 globalRotationAxis;
	float effectiveAxialMass;
	bool solveLimit, solveMotor;
	float limitImpulse, limitBias, limitSign;
	float motorImpulse, maxMotorImpulse, motorVelocity;
	vec3 motorAndLimitImpulseToAngularVelocityA, motorAndLimitImpulseToAngularVelocityB;
};
static inline void initializeHingeConstraint(hinge_constraint_update& out, const rigid_body_global_state* rbs, const hinge_constraint& in, constraint_body_pair bp, float dt)
{
	float invDt = 1.f / dt;
	out.rigidBodyIndexA = bp.rbA; out.rigidBodyIndexB = bp.rbB;
	const rigid_body_global_state& globalA = rbs[bp.rbA];
	const rigid_body_global_state& globalB = rbs[bp.rbB];
	out.relGlobalAnchorA = globalA.rotation * (in.localAnchorA - globalA.localCOGPosition);
	out.relGlobalAnchorB = globalB.rotation * (in.localAnchorB - globalB.localCOGPosition);
	vec3 globalAnchorA = globalA.position + out.relGlobalAnchorA;
	vec3 globalAnchorB = globalB.position + out.relGlobalAnchorB;

	out.invEffectiveTranslationMass = pointBlockInvEffectiveMass(globalA, globalB, out.relGlobalAnchorA, out.relGlobalAnchorB);
	out.translationBias = vec3(0.f);
	if (dt > ORC_DT_THRESHOLD) { out.translationBias = (globalAnchorB - globalAnchorA) * (ORC_BALL_CONSTRAINT_BETA * invDt); }

	vec3 globalHingeAxisA = globalA.rotation * in.localHingeAxisA;
	vec3 globalHingeAxisB = globalB.rotation * in.localHingeAxisB;
	vec3 globalTangentB, globalBitangentB;
	getTangents(globalHingeAxisB, globalTangentB, globalBitangentB);
	vec3 bxa = cross(globalTangentB, globalHingeAxisA);
	vec3 cxa = cross(globalBitangentB, globalHingeAxisA);
	vec3 iAbxa = globalA.invInertia * bxa, iBbxa = globalB.invInertia * bxa;
	vec3 iAcxa = globalA.invInertia * cxa, iBcxa = globalB.invInertia * cxa;
	out.invEffectiveRotationMass.m00 = dot(bxa, iAbxa) + dot(bxa, iBbxa);
	out.invEffectiveRotationMass.m01 = dot(bxa, iAcxa) + dot(bxa, iBcxa);
	out.invEffectiveRotationMass.m10 = dot(cxa, iAbxa) + dot(cxa, iBbxa);
	out.invEffectiveRotationMass.m11 = dot(cxa, iAcxa) + dot(cxa, iBcxa);
	out.bxa = bxa; out.cxa = cxa;
	out.rotationBias = vec2(0.f, 0.f);
	if (dt > ORC_DT_THRESHOLD)
	{
		out.rotationBias = vec2(dot(globalHingeAxisA, globalTangentB), dot(globalHingeAxisA, globalBitangentB)) * (ORC_HINGE_ROTATION_CONSTRAINT_BETA * invDt);
	}

	out.solveLimit = false; out.solveMotor = false;
	out.globalRotationAxis = vec3(0.f); out.effectiveAxialMass = 0.f;
	out.limitImpulse = out.limitBias = out.limitSign = 0.f;
	out.motorImpulse = out.maxMotorImpulse = out.motorVelocity = 0.f;
	out.motorAndLimitImpulseToAngularVelocityA = out.motorAndLimitImpulseToAngularVelocityB = vec3(0.f);

	if (in.minRotationLimit <= 0.f || in.maxRotationLimit >= 0.f || in.maxMotorTorque > 0.f)
	{
		vec3 localHingeCompareA = conjugate(globalA.rotation) * (globalB.rotation * in.localHingeTangentB);
		float angle = jointAtan2(jointDot(localHingeCompareA, in.localHingeBitangentA), jointDot(localHingeCompareA, in.localHingeTangentA)); // (wide path: constraints.cpp:1545, polynomial atan2)
		bool minLimitViolated = in.minRotationLimit <= 0.f && angle <= in.minRotationLimit;
		bool maxLimitViolated = in.maxRotationLimit >= 0.f && angle >= in.maxRotationLimit;
		out.solveLimit = minLimitViolated || maxLimitViolated;
		out.solveMotor = in.maxMotorTorque > 0.f;
		if (out.solveLimit || out.solveMotor)
		{
			out.globalRotationAxis = globalHingeAxisA;
			out.limitImpulse = 0.f;
			float invEffectiveAxialMass = dot(globalHingeAxisA, globalA.invInertia * globalHingeAxisA) + dot(globalHingeAxisA, globalB.invInertia * globalHingeAxisA);
			out.effectiveAxialMass = (invEffectiveAxialMass != 0.f) ? (1.f / invEffectiveAxialMass) : 0.f;
			out.limitSign = minLimitViolated ? 1.f : -1.f;
			out.maxMotorImpulse = in.maxMotorTorque * dt;
			out.motorImpulse = 0.f;
			out.motorAndLimitImpulseToAngularVelocityA = globalA.invInertia * out.globalRotationAxis;
			out.motorAndLimitImpulseToAngularVelocityB = globalB.invInertia * out.globalRotationAxis;
			out.motorVelocity = in.motorVelocity;
			if (in.motorType == constraint_position_motor)
			{
				float minLimit = (in.minRotationLimit <= 0.f) ? in.minRotationLimit : -M_PI_F;
				float maxLimit = (in.maxRotationLimit >= 0.f) ? in.maxRotationLimit : M_PI_F;
				float targetAngle = clampf(in.motorVelocity, minLimit, maxLimit);
				out.motorVelocity = (dt > ORC_DT_THRESHOLD) ? ((targetAngle - angle) * invDt) : 0.f;
			}
			out.limitBias = 0.f;
			if (dt > ORC_DT_THRESHOLD)
			{
				float d = minLimitViolated ? (angle - in.minRotationLimit) : (in.maxRotationLimit - angle);
				out.limitBias = d * ORC_HINGE_LIMIT_CONSTRAINT_BETA * invDt;
			}
		}
	}
}
static inline void solveHingeConstraint(hinge_constraint_update& con, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[con.rigidBodyIndexA];
	rigid_body_global_state& rbB = rbs[con.rigidBodyIndexB];
	vec3 vA = rbA.linearVelocity, wA = rbA.angularVelocity, vB = rbB.linearVelocity, wB = rbB.angularVelocity;
	vec3 globalRotationAxis = con.globalRotationAxis;
	if (con.solveMotor)
	{
		float aDotWA = dot(globalRotationAxis, wA);
		float aDotWB = dot(globalRotationAxis, wB);
		float relAngularVelocity = (aDotWB - aDotWA);
		float motorCdot = relAngularVelocity - con.motorVelocity;
		float motorLambda = -con.effectiveAxialMass * motorCdot;
		float oldImpulse = con.motorImpulse;
		con.motorImpulse = clampf(con.motorImpulse + motorLambda, -con.maxMotorImpulse, con.maxMotorImpulse);
		motorLambda = con.motorImpulse - oldImpulse;
		wA -= con.motorAndLimitImpulseToAngularVelocityA * motorLambda;
		wB += con.motorAndLimitImpulseToAngularVelocityB * motorLambda;
	}
	if (con.solveLimit)
	{
		float limitSign = con.limitSign;
		float aDotWA = dot(globalRotationAxis, wA);
		float aDotWB = dot(globalRotationAxis, wB);
		float relAngularVelocity = limitSign * (aDotWB - aDotWA);
		float limitCdot = relAngularVelocity + con.limitBias;
		float limitLambda = -con.effectiveAxialMass * limitCdot;
		float impulse = std::max(con.limitImpulse + limitLambda, 0.f);
		limitLambda = impulse - con.limitImpulse;
		con.limitImpulse = impulse;
		limitLambda *= limitSign;
		wA -= con.motorAndLimitImpulseToAngularVelocityA * limitLambda;
		wB += con.motorAndLimitImpulseToAngularVelocityB * limitLambda;
	}
	{
		vec3 deltaAngularVelocity = wB - wA;
		vec2 rotationCdot(dot(con.bxa, deltaAngularVelocity), dot(con.cxa, deltaAngularVelocity));
		vec2 rotLambda = solveLinearSystem(con.invEffectiveRotationMass, -(rotationCdot + con.rotationBias));
		vec3 rotationP = con.bxa * rotLambda.x + con.cxa * rotLambda.y;
		wA -= rbA.invInertia * rotationP;
		wB += rbB.invInertia * rotationP;
	}
	solvePointBlock(vA, wA, vB, wB, rbA, rbB, con.relGlobalAnchorA, con.relGlobalAnchorB, con.translationBias, con.invEffectiveTranslationMass);
	rbA.linearVelocity = vA; rbA.angularVelocity = wA;
	rbB.linearVelocity = vB; rbB.angularVelocity = wB;
}

// ---------------------------------------------------------------------------------------------------
// Cone-twist — constraints.h:382-431, constraints.cpp:1782-2070
// ---------------------------------------------------------------------------------------------------
struct cone_twist_constraint_update
{
	u32 rigidBodyIndexA, rigidBodyIndexB;
	vec3 relGlobalAnchorA, relGlobalAnchorB, bias;
	mat3 invEffectiveMass;
	bool solveSwingLimit, solveTwistLimit, solveSwingMotor, solveTwistMotor;
	vec3 globalSwingAxis; float swingImpulse, effectiveSwingLimitMass, swingLimitBias;
	vec3 globalTwistAxis; float twistImpulse, twistLimitSign, effectiveTwistMass, twistLimitBias;
	float swingMotorImpulse, maxSwingMotorImpulse, swingMotorVelocity; vec3 globalSwingMotorAxis; float effectiveSwingMotorMass;
	float twistMotorImpulse, maxTwistMotorImpulse, twistMotorVelocity;
	vec3 twistMotorAndLimitImpulseToAngularVelocityA, twistMotorAndLimitImpulseToAngularVelocityB;
	vec3 swingMotorImpulseToAngularVelocityA, swingMotorImpulseToAngularVelocityB;
	vec3 swingLimitImpulseToAngularVelocityA, swingLimitImpulseToAngularVelocityB;
};
static inline void initializeConeTwistConstraint(cone_twist_constraint_update& out, const rigid_body_global_state* rbs, const cone_twist_constraint& in, constraint_body_pair bp, float dt)
{
	float invDt = 1.f / dt;
	memset(&out, 0, sizeof(out));
	out.rigidBodyIndexA = bp.rbA; out.rigidBodyIndexB = bp.rbB;
	const rigid_body_global_state& globalA = rbs[bp.rbA];
	const rigid_body_global_state& globalB = rbs[bp.rbB];
	out.relGlobalAnchorA = globalA.rotation * (in.localAnchorA - globalA.localCOGPosition);
	out.relGlobalAnchorB = globalB.rotation * (in.localAnchorB - globalB.localCOGPosition);
	vec3 globalAnchorA = globalA.position + out.relGlobalAnchorA;
	vec3 globalAnchorB = globalB.position + out.relGlobalAnchorB;
	out.invEffectiveMass = pointBlockInvEffectiveMass(globalA, globalB, out.relGlobalAnchorA, out.relGlobalAnchorB);
	out.bias = vec3(0.f);
	if (dt > ORC_DT_THRESHOLD) { out.bias = (globalAnchorB - globalAnchorA) * (ORC_BALL_CONSTRAINT_BETA * invDt); }

	quat btoa = conjugate(globalA.rotation) * globalB.rotation;
	vec3 localLimitAxisA = in.localLimitAxisA;
	vec3 localLimitAxisCompareA = btoa * in.localLimitAxisB;
	quat swingRotation = jointRotateFromTo(localLimitAxisA, localLimitAxisCompareA); // (wide path: constraints.cpp:2258-2272)
	vec3 twistTangentA = swingRotation * in.localLimitTangentA;
	vec3 twistBitangentA = swingRotation * in.localLimitBitangentA;
	vec3 localLimitTangentCompareA = btoa * in.localLimitTangentB;
	float twistAngle = jointAtan2(jointDot(localLimitTangentCompareA, twistBitangentA), jointDot(localLimitTangentCompareA, twistTangentA));

	vec3 swingAxis; float swingAngle;
	jointGetAxisRotation(swingRotation, swingAxis, swingAngle);
	if (swingAngle < 0.f) { swingAngle *= -1.f; swingAxis *= -1.f; }

	out.solveSwingLimit = in.swingLimit >= 0.f && swingAngle >= in.swingLimit;
	if (out.solveSwingLimit)
	{
		out.swingImpulse = 0.f;
		out.globalSwingAxis = globalA.rotation * swingAxis;
		float invEffectiveLimitMass = dot(out.globalSwingAxis, globalA.invInertia * out.globalSwingAxis) + dot(out.globalSwingAxis, globalB.invInertia * out.globalSwingAxis);
		out.effectiveSwingLimitMass = (invEffectiveLimitMass != 0.f) ? (1.f / invEffectiveLimitMass) : 0.f;
		out.swingLimitBias = 0.f;
		if (dt > ORC_DT_THRESHOLD) { out.swingLimitBias = (in.swingLimit - swingAngle) * (ORC_HINGE_LIMIT_CONSTRAINT_BETA * invDt); }
		out.swingLimitImpulseToAngularVelocityA = globalA.invInertia * out.globalSwingAxis;
		out.swingLimitImpulseToAngularVelocityB = globalB.invInertia * out.globalSwingAxis;
	}

	out.solveSwingMotor = in.maxSwingMotorTorque > 0.f;
	if (out.solveSwingMotor)
	{
		out.maxSwingMotorImpulse = in.maxSwingMotorTorque * dt;
		out.swingMotorImpulse = 0.f;
		float axisX = jointCos(in.swingMotorAxis), axisY = jointSin(in.swingMotorAxis);
		vec3 localSwingMotorAxis = axisX * in.localLimitTangentA + axisY * in.localLimitBitangentA;
		if (in.swingMotorType == constraint_velocity_motor)
		{
			out.globalSwingMotorAxis = globalA.rotation * localSwingMotorAxis;
			out.swingMotorVelocity = in.swingMotorVelocity;
		}
		else
		{
			float targetAngle = in.swingMotorVelocity;
			if (in.swingLimit >= 0.f) { targetAngle = clampf(targetAngle, -in.swingLimit, in.swingLimit); }
			vec3 localTargetDirection = jointQuatAxisAngle(localSwingMotorAxis, targetAngle) * localLimitAxisA;
			vec3 localSwingMotorAxis2 = jointNoz(wideJointMath() ? wcross(localLimitAxisCompareA, localTargetDirection) : cross(localLimitAxisCompareA, localTargetDirection));
			out.globalSwingMotorAxis = globalA.rotation * localSwingMotorAxis2;
			float cosAngle = jointDot(localTargetDirection, localLimitAxisCompareA);
			float deltaAngle = jointAcos(clamp01(cosAngle));
			out.swingMotorVelocity = (dt > ORC_DT_THRESHOLD) ? (deltaAngle * invDt * 0.2f) : 0.f;
		}
		out.swingMotorImpulseToAngularVelocityA = globalA.invInertia * out.globalSwingMotorAxis;
		out.swingMotorImpulseToAngularVelocityB = globalB.invInertia * out.globalSwingMotorAxis;
		float invEffectiveMotorMass = dot(out.globalSwingMotorAxis, globalA.invInertia * out.globalSwingMotorAxis) + dot(out.globalSwingMotorAxis, globalB.invInertia * out.globalSwingMotorAxis);
		out.effectiveSwingMotorMass = (invEffectiveMotorMass != 0.f) ? (1.f / invEffectiveMotorMass) : 0.f;
	}

	bool minTwistLimitViolated = in.twistLimit >= 0.f && twistAngle <= -in.twistLimit;
	bool maxTwistLimitViolated = in.twistLimit >= 0.f && twistAngle >= in.twistLimit;
	out.solveTwistLimit = minTwistLimitViolated || maxTwistLimitViolated;
	out.solveTwistMotor = in.maxTwistMotorTorque > 0.f;
	if (out.solveTwistLimit || out.solveTwistMotor)
	{
		out.twistImpulse = 0.f;
		out.globalTwistAxis = globalA.rotation * localLimitAxisA;
		float invEffectiveMass = dot(out.globalTwistAxis, globalA.invInertia * out.globalTwistAxis) + dot(out.globalTwistAxis, globalB.invInertia * out.globalTwistAxis);
		out.effectiveTwistMass = (invEffectiveMass != 0.f) ? (1.f / invEffectiveMass) : 0.f;
		out.twistLimitSign = minTwistLimitViolated ? 1.f : -1.f;
		out.maxTwistMotorImpulse = in.maxTwistMotorTorque * dt;
		out.twistMotorImpulse = 0.f;
		out.twistMotorAndLimitImpulseToAngularVelocityA = globalA.invInertia * out.globalTwistAxis;
		out.twistMotorAndLimitImpulseToAngularVelocityB = globalB.invInertia * out.globalTwistAxis;
		out.twistMotorVelocity = in.twistMotorVelocity;
		if (in.twistMotorType == constraint_position_motor)
		{
			float limit = (in.twistLimit >= 0.f) ? in.twistLimit : M_PI_F;
			float targetAngle = clampf(in.twistMotorVelocity, -limit, limit);
			out.twistMotorVelocity = (dt > ORC_DT_THRESHOLD) ? ((targetAngle - twistAngle) * invDt) : 0.f;
		}
		out.twistLimitBias = 0.f;
		if (dt > ORC_DT_THRESHOLD)
		{
			float d = minTwistLimitViolated ? (in.twistLimit + twistAngle) : (in.twistLimit - twistAngle);
			out.twistLimitBias = d * ORC_TWIST_LIMIT_CONSTRAINT_BETA * invDt;
		}
	}
}
static inline void solveConeTwistConstraint(cone_twist_constraint_update& con, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[con.rigidBodyIndexA];
	rigid_body_global_state& rbB = rbs[con.rigidBodyIndexB];
	vec3 vA = rbA.linearVelocity, wA = rbA.angularVelocity, vB = rbB.linearVelocity, wB = rbB.angularVelocity;
	vec3 globalTwistAxis = con.globalTwistAxis;
	if (con.solveTwistMotor)
	{
		float aDotWA = dot(globalTwistAxis, wA), aDotWB = dot(globalTwistAxis, wB);
		float relAngularVelocity = (aDotWB - aDotWA);
		float motorCdot = relAngularVelocity - con.twistMotorVelocity;
		float motorLambda = -con.effectiveTwistMass * motorCdot;
		float oldImpulse = con.twistMotorImpulse;
		con.twistMotorImpulse = clampf(con.twistMotorImpulse + motorLambda, -con.maxTwistMotorImpulse, con.maxTwistMotorImpulse);
		motorLambda = con.twistMotorImpulse - oldImpulse;
		wA -= con.twistMotorAndLimitImpulseToAngularVelocityA * motorLambda;
		wB += con.twistMotorAndLimitImpulseToAngularVelocityB * motorLambda;
	}
	if (con.solveSwingMotor)
	{
		vec3 axis = con.globalSwingMotorAxis;
		float aDotWA = dot(axis, wA), aDotWB = dot(axis, wB);
		float relAngularVelocity = (aDotWB - aDotWA);
		float motorCdot = relAngularVelocity - con.swingMotorVelocity;
		float motorLambda = -con.effectiveSwingMotorMass * motorCdot;
		float oldImpulse = con.swingMotorImpulse;
		con.swingMotorImpulse = clampf(con.swingMotorImpulse + motorLambda, -con.maxSwingMotorImpulse, con.maxSwingMotorImpulse);
		motorLambda = con.swingMotorImpulse - oldImpulse;
		wA -= con.swingMotorImpulseToAngularVelocityA * motorLambda;
		wB += con.swingMotorImpulseToAngularVelocityB * motorLambda;
	}
	if (con.solveTwistLimit)
	{
		float limitSign = con.twistLimitSign;
		float aDotWA = dot(globalTwistAxis, wA), aDotWB = dot(globalTwistAxis, wB);
		float relAngularVelocity = limitSign * (aDotWB - aDotWA);
		float limitCdot = relAngularVelocity + con.twistLimitBias;
		float limitLambda = -con.effectiveTwistMass * limitCdot;
		float impulse = std::max(con.twistImpulse + limitLambda, 0.f);
		limitLambda = impulse - con.twistImpulse;
		con.twistImpulse = impulse;
		limitLambda *= limitSign;
		wA -= con.twistMotorAndLimitImpulseToAngularVelocityA * limitLambda;
		wB += con.twistMotorAndLimitImpulseToAngularVelocityB * limitLambda;
	}
	if (con.solveSwingLimit)
	{
		float aDotWA = dot(con.globalSwingAxis, wA), aDotWB = dot(con.globalSwingAxis, wB);
		float swingLimitCdot = aDotWA - aDotWB + con.swingLimitBias;
		float limitLambda = -con.effectiveSwingLimitMass * swingLimitCdot;
		float impulse = std::max(con.swingImpulse + limitLambda, 0.f);
		limitLambda = impulse - con.swingImpulse;
		con.swingImpulse = impulse;
		wA += con.swingLimitImpulseToAngularVelocityA * limitLambda;
		wB -= con.swingLimitImpulseToAngularVelocityB * limitLambda;
	}
	solvePointBlock(vA, wA, vB, wB, rbA, rbB, con.relGlobalAnchorA, con.relGlobalAnchorB, con.bias, con.invEffectiveMass);
	rbA.linearVelocity = vA; rbA.angularVelocity = wA;
	rbB.linearVelocity = vB; rbB.angularVelocity = wB;
}

// ---------------------------------------------------------------------------------------------------
// Slider — constraints.h:522-560, constraints.cpp:2638-2846
// ---------------------------------------------------------------------------------------------------
struct slider_constraint_update
{
	u32 rigidBodyIndexA, rigidBodyIndexB;
	vec3 rAuxt, rAuxb, rBxt, rBxb, tangent, bitangent;
	mat2 invEffectiveTranslationMass; vec2 translationBias;
	mat3 invEffectiveRotationMass; vec3 rotationBias;
	bool solveLimit, solveMotor;
	vec3 globalSliderAxis;
	float effectiveAxialMass, limitBias, limitImpulse, limitSign;
	vec3 rAuxs, rBxs, limitImpulseToAngularVelocityA, limitImpulseToAngularVelocityB;
	float motorVelocity, motorImpulse, maxMotorImpulse;
};
static inline void initializeSliderConstraint(slider_constraint_update& out, const rigid_body_global_state* rbs, const slider_constraint& in, constraint_body_pair bp, float dt)
{
	float invDt = 1.f / dt;
	memset(&out, 0, sizeof(out));
	out.rigidBodyIndexA = bp.rbA; out.rigidBodyIndexB = bp.rbB;
	const rigid_body_global_state& globalA = rbs[bp.rbA];
	const rigid_body_global_state& globalB = rbs[bp.rbB];
	vec3 relGlobalAnchorA = globalA.rotation * (in.localAnchorA - globalA.localCOGPosition);
	vec3 relGlobalAnchorB = globalB.rotation * (in.localAnchorB - globalB.localCOGPosition);
	vec3 globalAnchorA = globalA.position + relGlobalAnchorA;
	vec3 globalAnchorB = globalB.position + relGlobalAnchorB;
	vec3 globalSliderAxis = globalA.rotation * in.localAxisA;
	getTangents(globalSliderAxis, out.tangent, out.bitangent);
	vec3 u = globalAnchorB - globalAnchorA;
	vec3 rAu = relGlobalAnchorA + u;
	out.rBxt = cross(relGlobalAnchorB, out.tangent);
	out.rBxb = cross(relGlobalAnchorB, out.bitangent);
	out.rAuxt = cross(rAu, out.tangent);
	out.rAuxb = cross(rAu, out.bitangent);
	vec3 iArAuxt = globalA.invInertia * out.rAuxt, iArAuxb = globalA.invInertia * out.rAuxb;
	vec3 iBrBxt = globalB.invInertia * out.rBxt, iBrBxb = globalB.invInertia * out.rBxb;
	float invMassSum = globalA.invMass + globalB.invMass;
	out.invEffectiveTranslationMass.m00 = dot(out.rAuxt, iArAuxt) + dot(out.rBxt, iBrBxt) + invMassSum;
	out.invEffectiveTranslationMass.m01 = dot(out.rAuxt, iArAuxb) + dot(out.rBxt, iBrBxb);
	out.invEffectiveTranslationMass.m10 = dot(out.rAuxb, iArAuxt) + dot(out.rBxb, iBrBxt);
	out.invEffectiveTranslationMass.m11 = dot(out.rAuxb, iArAuxb) + dot(out.rBxb, iBrBxb) + invMassSum;
	out.invEffectiveRotationMass = globalA.invInertia + globalB.invInertia;
	out.translationBias = vec2(0.f, 0.f);
	out.rotationBias = vec3(0.f);
	if (dt > ORC_DT_THRESHOLD)
	{
		float a = dot(u, out.tangent), b = dot(u, out.bitangent);
		out.translationBias = vec2(a, b) * (ORC_SLIDER_CONSTRAINT_BETA * invDt);
		quat rotationError = globalB.rotation * in.initialInvRotationDifference * conjugate(globalA.rotation);
		out.rotationBias = rotationError.v() * (ORC_SLIDER_CONSTRAINT_BETA * invDt * 2.f);
	}
	out.globalSliderAxis = globalSliderAxis;
	float distanceAlongSlider = dot(u, globalSliderAxis);
	out.solveLimit = false;
	if (in.negDistanceLimit <= 0.f || in.posDistanceLimit >= 0.f)
	{
		bool minLimitViolated = (in.negDistanceLimit <= 0.f) && (distanceAlongSlider < in.negDistanceLimit);
		bool maxLimitViolated = (in.posDistanceLimit >= 0.f) && (distanceAlongSlider > in.posDistanceLimit);
		if (minLimitViolated || maxLimitViolated)
		{
			out.solveLimit = true;
			out.limitImpulse = 0.f;
			out.rAuxs = cross(rAu, globalSliderAxis);
			out.rBxs = cross(relGlobalAnchorB, globalSliderAxis);
			float invEffectiveAxialMass = invMassSum + dot(out.rAuxs, globalA.invInertia * out.rAuxs) + dot(out.rBxs, globalB.invInertia * out.rBxs);
			out.effectiveAxialMass = (invEffectiveAxialMass != 0.f) ? (1.f / invEffectiveAxialMass) : 0.f;
			out.limitSign = minLimitViolated ? 1.f : -1.f;
			out.limitBias = 0.f;
			if (dt > ORC_DT_THRESHOLD)
			{
				float error = minLimitViolated ? (distanceAlongSlider - in.negDistanceLimit) : (in.posDistanceLimit - distanceAlongSlider);
				out.limitBias = error * (ORC_SLIDER_LIMIT_CONSTRAINT_BETA * invDt);
			}
			out.limitImpulseToAngularVelocityA = globalA.invInertia * out.rAuxs;
			out.limitImpulseToAngularVelocityB = globalB.invInertia * out.rBxs;
		}
	}
	out.solveMotor = false;
	if (in.maxMotorForce > 0.f)
	{
		out.solveMotor = true;
		out.maxMotorImpulse = in.maxMotorForce * dt;
		out.motorImpulse = 0.f;
		out.motorVelocity = in.motorVelocity;
		if (in.motorType == constraint_position_motor)
		{
			float minLimit = (in.negDistanceLimit <= 0.f) ? in.negDistanceLimit : -INFINITY;
			float maxLimit = (in.posDistanceLimit >= 0.f) ? in.posDistanceLimit : INFINITY;
			float targetDistance = clampf(in.motorVelocity, minLimit, maxLimit);
			out.motorVelocity = (dt > ORC_DT_THRESHOLD) ? ((targetDistance - distanceAlongSlider) * invDt) : 0.f;
		}
	}
}
static inline void solveSliderConstraint(slider_constraint_update& con, rigid_body_global_state* rbs)
{
	rigid_body_global_state& rbA = rbs[con.rigidBodyIndexA];
	rigid_body_global_state& rbB = rbs[con.rigidBodyIndexB];
	vec3 vA = rbA.linearVelocity, wA = rbA.angularVelocity, vB = rbB.linearVelocity, wB = rbB.angularVelocity;
	if (con.solveMotor)
	{
		float Cdot = dot(vB, con.globalSliderAxis) - dot(vA, con.globalSliderAxis) - con.motorVelocity;
		float mass = 1.f / (rbA.invMass + rbB.invMass);
		float motorLambda = -mass * Cdot;
		float oldImpulse = con.motorImpulse;
		con.motorImpulse = clampf(con.motorImpulse + motorLambda, -con.maxMotorImpulse, con.maxMotorImpulse);
		motorLambda = con.motorImpulse - oldImpulse;
		vec3 P = motorLambda * con.globalSliderAxis;
		vA -= rbA.invMass * P;
		vB += rbB.invMass * P;
	}
	if (con.solveLimit)
	{
		float Cdot = dot(vB, con.globalSliderAxis) + dot(wB, con.rBxs) - dot(vA, con.globalSliderAxis) - dot(wA, con.rAuxs);
		float limitLambda = -con.effectiveAxialMass * (con.limitSign * Cdot + con.limitBias);
		float impulse = std::max(con.limitImpulse + limitLambda, 0.f);
		limitLambda = impulse - con.limitImpulse;
		con.limitImpulse = impulse;
		limitLambda *= con.limitSign;
		vec3 P = limitLambda * con.globalSliderAxis;
		vA -= rbA.invMass * P;
		wA -= con.limitImpulseToAngularVelocityA * limitLambda;
		vB += rbB.invMass * P;
		wB += con.limitImpulseToAngularVelocityB * limitLambda;
	}
	{
		vec3 Cdot = wB - wA;
		vec3 rotationLambda = solveLinearSystem(con.invEffectiveRotationMass, -(Cdot + con.rotationBias));
		wA -= rbA.invInertia * rotationLambda;
		wB += rbB.invInertia * rotationLambda;
	}
	{
		vec2 Cdot;
		Cdot.x = dot(con.tangent, vB) + dot(con.rBxt, wB) - dot(con.tangent, vA) - dot(con.rAuxt, wA);
		Cdot.y = dot(con.bitangent, vB) + dot(con.rBxb, wB) - dot(con.bitangent, vA) - dot(con.rAuxb, wA);
		vec2 translationLambda = solveLinearSystem(con.invEffectiveTranslationMass, -(Cdot + con.translationBias));
		vec3 tb = con.tangent * translationLambda.x + con.bitangent * translationLambda.y;
		vA -= rbA.invMass * tb;
		wA -= rbA.invInertia * (con.rAuxt * translationLambda.x + con.rAuxb * translationLambda.y);
		vB += rbB.invMass * tb;
		wB += rbB.invInertia * (con.rBxt * translationLambda.x + con.rBxb * translationLambda.y);
	}
	rbA.linearVelocity = vA; rbA.angularVelocity = wA;
	rbB.linearVelocity = vB; rbB.angularVelocity = wB;
}

} // namespace orc
