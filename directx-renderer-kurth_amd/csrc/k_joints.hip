// Joint constraints on the GPU: distance, ball, fixed, hinge, cone-twist, slider — init and sequential-impulse solve, following the
// SCALAR formulations of the reference (constraints.cpp:189-264, 460-528, 736-823, 1079-1307, 1782-2070, 2638-2846), whose results
// the 8-wide variants reproduce up to their polynomial trig / rsqrt approximations (SURVEY finding 3).
// Joints change rarely, so they are greedily coloured on the host when the joint set changes (world.hip) and stored colour-sorted;
// each colour of each type is one launch, one lane per joint.  Per-joint scratch is an AoS record in HBM (joint counts are small:
// config 4 has 3 328 joints); body state is the same 32-B velocity record + 48-B world inverse inertia the contact path uses.
#include "world.h"

#include "joint_solve.h"

#define JOINT_KERNEL_ARGS u32 start, u32 end, u32 nb, float dt, const uint8_t* __restrict__ pods, const uint2* __restrict__ pairs, float* __restrict__ upd, \
	const float4* __restrict__ pose, const float4* __restrict__ bprops, const float4* __restrict__ cog, const float4* __restrict__ invIw, float4* __restrict__ vel
#define JOINT_INDEX u32 j = start + blockIdx.x * blockDim.x + threadIdx.x; if (j >= end) return; uint2 ab = pairs[j]

// ---- distance (update: rA3 rB3 jA3 jB3 u3 bias effMass = 17 of 20 floats) ------------------------------------------------
__global__ void k_distance_init(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	const mi_distance_constraint& in = *(const mi_distance_constraint*)(pods + (size_t)j * sizeof(mi_distance_constraint));
	float* o = upd + (size_t)j * 20;
	BodyIn A = loadBody(ab.x, pose, bprops, cog, invIw), B = loadBody(ab.y, pose, bprops, cog, invIw);
	float invDt = 1.f / dt;
	V3 rA = A.rot * (ld3(in.localAnchorA) - A.localCOG), rB = B.rot * (ld3(in.localAnchorB) - B.localCOG);
	V3 u = (B.pos + rB) - (A.pos + rA);
	float l = length(u);
	u = (l > 0.001f) ? (u * (1.f / l)) : v3s(0.f);
	V3 crAu = cross(rA, u), crBu = cross(rB, u);
	float invMass = A.invMass + dot(crAu, A.invI * crAu) + B.invMass + dot(crBu, B.invI * crBu);
	float eff = (invMass != 0.f) ? (1.f / invMass) : 0.f;
	float bias = 0.f;
	if (dt > DT_THRESHOLD) bias = (l - in.globalLength) * (BETA_DISTANCE * invDt);
	st3(o, rA); st3(o + 3, rB); st3(o + 6, A.invI * cross(rA, crAu)); st3(o + 9, B.invI * cross(rB, crBu)); st3(o + 12, u); o[15] = bias; o[16] = eff;
}
__global__ void k_distance_solve(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	float* o = upd + (size_t)j * 20;
	Vel v = loadVel(vel, ab.x, ab.y);
	solveDistance(o, v, ldInvI(invIw, ab.x), ldInvI(invIw, ab.y));
	storeVel(vel, nb, ab.x, ab.y, v);
}

// ---- ball (update: rA3 rB3 bias3 invEff9 = 18 of 20) -------------------------------------------------------------------
__global__ void k_ball_init(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	const mi_ball_constraint& in = *(const mi_ball_constraint*)(pods + (size_t)j * sizeof(mi_ball_constraint));
	float* o = upd + (size_t)j * 20;
	BodyIn A = loadBody(ab.x, pose, bprops, cog, invIw), B = loadBody(ab.y, pose, bprops, cog, invIw);
	float invDt = 1.f / dt;
	V3 rA = A.rot * (ld3(in.localAnchorA) - A.localCOG), rB = B.rot * (ld3(in.localAnchorB) - B.localCOG);
	V3 bias = v3s(0.f);
	if (dt > DT_THRESHOLD) bias = ((B.pos + rB) - (A.pos + rA)) * (BETA_BALL * invDt);
	st3(o, rA); st3(o + 3, rB); st3(o + 6, bias); stM3(o + 9, pointBlock(A, B, rA, rB));
}
__global__ void k_ball_solve(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	float* o = upd + (size_t)j * 20;
	Vel v = loadVel(vel, ab.x, ab.y);
	solveBall(o, v, ldInvI(invIw, ab.x), ldInvI(invIw, ab.y));
	storeVel(vel, nb, ab.x, ab.y, v);
}

// ---- fixed (update: rA3 rB3 tBias3 invEffT9 rBias3 invEffR9 = 30 of 36) ------------------------------------------------
__global__ void k_fixed_init(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	const mi_fixed_constraint& in = *(const mi_fixed_constraint*)(pods + (size_t)j * sizeof(mi_fixed_constraint));
	float* o = upd + (size_t)j * 36;
	BodyIn A = loadBody(ab.x, pose, bprops, cog, invIw), B = loadBody(ab.y, pose, bprops, cog, invIw);
	float invDt = 1.f / dt;
	V3 rA = A.rot * (ld3(in.localAnchorA) - A.localCOG), rB = B.rot * (ld3(in.localAnchorB) - B.localCOG);
	V3 tBias = v3s(0.f), rBias = v3s(0.f);
	if (dt > DT_THRESHOLD)
	{
		tBias = ((B.pos + rB) - (A.pos + rA)) * (BETA_BALL * invDt);
		const float* q = in.initialInvRotationDifference;
		Q4 err = B.rot * q4(q[0], q[1], q[2], q[3]) * conjugate(A.rot);
		rBias = qv(err) * (BETA_SLIDER * invDt * 2.f);
	}
	st3(o, rA); st3(o + 3, rB); st3(o + 6, tBias); stM3(o + 9, pointBlock(A, B, rA, rB)); st3(o + 18, rBias); stM3(o + 21, madd(A.invI, B.invI));
}
__global__ void k_fixed_solve(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	float* o = upd + (size_t)j * 36;
	Vel v = loadVel(vel, ab.x, ab.y);
	solveFixed(o, v, ldInvI(invIw, ab.x), ldInvI(invIw, ab.y));
	storeVel(vel, nb, ab.x, ab.y, v);
}

// ---- hinge (update, 56 floats) -------------------------------------------------------------------------------------------
//  0 rA3 | 3 rB3 | 6 tBias3 | 9 invEffT9 | 18 rotBias2 | 20 invEffR(m00 m01 m10 m11) | 24 bxa3 | 27 cxa3 | 30 axis3 | 33 effAxial | 34 flags(bit0 limit, bit1 motor)
//  35 limitImpulse | 36 limitBias | 37 limitSign | 38 motorImpulse | 39 maxMotorImpulse | 40 motorVelocity | 41 jA3 | 44 jB3
__global__ void k_hinge_init(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	const mi_hinge_constraint& in = *(const mi_hinge_constraint*)(pods + (size_t)j * sizeof(mi_hinge_constraint));
	float* o = upd + (size_t)j * 56;
	BodyIn A = loadBody(ab.x, pose, bprops, cog, invIw), B = loadBody(ab.y, pose, bprops, cog, invIw);
	float invDt = 1.f / dt;
	V3 rA = A.rot * (ld3(in.localAnchorA) - A.localCOG), rB = B.rot * (ld3(in.localAnchorB) - B.localCOG);
	V3 tBias = v3s(0.f);
	if (dt > DT_THRESHOLD) tBias = ((B.pos + rB) - (A.pos + rA)) * (BETA_BALL * invDt);
	st3(o, rA); st3(o + 3, rB); st3(o + 6, tBias); stM3(o + 9, pointBlock(A, B, rA, rB));

	V3 axisA = A.rot * ld3(in.localHingeAxisA), axisB = B.rot * ld3(in.localHingeAxisB);
	V3 tanB = getTangent(axisB), bitB = cross(axisB, tanB);
	V3 bxa = cross(tanB, axisA), cxa = cross(bitB, axisA);
	V3 iAbxa = A.invI * bxa, iBbxa = B.invI * bxa, iAcxa = A.invI * cxa, iBcxa = B.invI * cxa;
	o[20] = dot(bxa, iAbxa) + dot(bxa, iBbxa);
	o[21] = dot(bxa, iAcxa) + dot(bxa, iBcxa);
	o[22] = dot(cxa, iAbxa) + dot(cxa, iBbxa);
	o[23] = dot(cxa, iAcxa) + dot(cxa, iBcxa);
	st3(o + 24, bxa); st3(o + 27, cxa);
	o[18] = 0.f; o[19] = 0.f;
	if (dt > DT_THRESHOLD) { o[18] = dot(axisA, tanB) * (BETA_HINGE_ROT * invDt); o[19] = dot(axisA, bitB) * (BETA_HINGE_ROT * invDt); }

	u32 flags = 0;
	for (u32 k = 30; k < 47; ++k) o[k] = 0.f;
	if (in.minRotationLimit <= 0.f || in.maxRotationLimit >= 0.f || in.maxMotorTorque > 0.f)
	{
		V3 cmp = conjugate(A.rot) * (B.rot * ld3(in.localHingeTangentB));
		float angle = atan2f(dot(cmp, ld3(in.localHingeBitangentA)), dot(cmp, ld3(in.localHingeTangentA)));
		bool minV = in.minRotationLimit <= 0.f && angle <= in.minRotationLimit;
		bool maxV = in.maxRotationLimit >= 0.f && angle >= in.maxRotationLimit;
		bool solveLimit = minV || maxV, solveMotor = in.maxMotorTorque > 0.f;
		if (solveLimit || solveMotor)
		{
			flags = (solveLimit ? 1u : 0u) | (solveMotor ? 2u : 0u);
			st3(o + 30, axisA);
			float invEff = dot(axisA, A.invI * axisA) + dot(axisA, B.invI * axisA);
			o[33] = (invEff != 0.f) ? (1.f / invEff) : 0.f;
			o[37] = minV ? 1.f : -1.f;
			o[39] = in.maxMotorTorque * dt;
			st3(o + 41, A.invI * axisA); st3(o + 44, B.invI * axisA);
			float motorVelocity = in.motorVelocity;
			if (in.motorType == MI_MOTOR_POSITION)
			{
				float minLimit = (in.minRotationLimit <= 0.f) ? in.minRotationLimit : -MI_PI;
				float maxLimit = (in.maxRotationLimit >= 0.f) ? in.maxRotationLimit : MI_PI;
				float target = clampf(in.motorVelocity, minLimit, maxLimit);
				motorVelocity = (dt > DT_THRESHOLD) ? ((target - angle) * invDt) : 0.f;
			}
			o[40] = motorVelocity;
			if (dt > DT_THRESHOLD)
			{
				float d = minV ? (angle - in.minRotationLimit) : (in.maxRotationLimit - angle);
				o[36] = d * BETA_HINGE_LIMIT * invDt;
			}
		}
	}
	o[34] = __uint_as_float(flags);
}
__global__ void k_hinge_solve(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	float* o = upd + (size_t)j * 56;
	Vel v = loadVel(vel, ab.x, ab.y);
	solveHinge(o, v, ldInvI(invIw, ab.x), ldInvI(invIw, ab.y));
	storeVel(vel, nb, ab.x, ab.y, v);
}

// ---- cone-twist (update, 80 floats) ----------------------------------------------------------------------------------------
//  0 rA3 | 3 rB3 | 6 bias3 | 9 invEff9 | 18 flags (1 swingLimit, 2 twistLimit, 4 swingMotor, 8 twistMotor)
//  19 swingAxis3 | 22 swingImpulse | 23 effSwingLimit | 24 swingLimitBias | 25 swingLimJA3 | 28 swingLimJB3
//  31 twistAxis3 | 34 twistImpulse | 35 twistSign | 36 effTwist | 37 twistLimitBias | 38 twistJA3 | 41 twistJB3
//  44 swingMotorImpulse | 45 maxSwingMotorImpulse | 46 swingMotorVelocity | 47 swingMotorAxis3 | 50 effSwingMotor | 51 swingMotJA3 | 54 swingMotJB3
//  57 twistMotorImpulse | 58 maxTwistMotorImpulse | 59 twistMotorVelocity
__global__ void k_cone_twist_init(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	const mi_cone_twist_constraint& in = *(const mi_cone_twist_constraint*)(pods + (size_t)j * sizeof(mi_cone_twist_constraint));
	float* o = upd + (size_t)j * 80;
	BodyIn A = loadBody(ab.x, pose, bprops, cog, invIw), B = loadBody(ab.y, pose, bprops, cog, invIw);
	float invDt = 1.f / dt;
	for (u32 k = 18; k < 60; ++k) o[k] = 0.f;
	V3 rA = A.rot * (ld3(in.localAnchorA) - A.localCOG), rB = B.rot * (ld3(in.localAnchorB) - B.localCOG);
	V3 bias = v3s(0.f);
	if (dt > DT_THRESHOLD) bias = ((B.pos + rB) - (A.pos + rA)) * (BETA_BALL * invDt);
	st3(o, rA); st3(o + 3, rB); st3(o + 6, bias); stM3(o + 9, pointBlock(A, B, rA, rB));

	Q4 btoa = conjugate(A.rot) * B.rot;
	V3 limitAxisA = ld3(in.localLimitAxisA);
	V3 limitAxisCompareA = btoa * ld3(in.localLimitAxisB);
	Q4 swingRotation = rotateFromTo(limitAxisA, limitAxisCompareA);
	V3 twistTangentA = swingRotation * ld3(in.localLimitTangentA);
	V3 twistBitangentA = swingRotation * ld3(in.localLimitBitangentA);
	V3 tangentCompareA = btoa * ld3(in.localLimitTangentB);
	float twistAngle = atan2f(dot(tangentCompareA, twistBitangentA), dot(tangentCompareA, twistTangentA));

	V3 swingAxis; float swingAngle; // getAxisRotation (math.cpp:577-592)
	{
		float sq = sqlen(qv(swingRotation));
		if (sq > 0.f) { swingAngle = 2.f * acosf(swingRotation.w); swingAxis = qv(swingRotation) * (1.f / sqrtf(sq)); }
		else { swingAngle = 0.f; swingAxis = v3(1.f, 0.f, 0.f); }
	}
	if (swingAngle < 0.f) { swingAngle *= -1.f; swingAxis *= -1.f; }

	u32 flags = 0;
	if (in.swingLimit >= 0.f && swingAngle >= in.swingLimit)
	{
		flags |= 1u;
		V3 g = A.rot * swingAxis;
		float invEff = dot(g, A.invI * g) + dot(g, B.invI * g);
		st3(o + 19, g);
		o[23] = (invEff != 0.f) ? (1.f / invEff) : 0.f;
		if (dt > DT_THRESHOLD) o[24] = (in.swingLimit - swingAngle) * (BETA_HINGE_LIMIT * invDt);
		st3(o + 25, A.invI * g); st3(o + 28, B.invI * g);
	}
	if (in.maxSwingMotorTorque > 0.f)
	{
		flags |= 4u;
		o[45] = in.maxSwingMotorTorque * dt;
		float axisX = cosf(in.swingMotorAxis), axisY = sinf(in.swingMotorAxis);
		V3 localMotorAxis = axisX * ld3(in.localLimitTangentA) + axisY * ld3(in.localLimitBitangentA);
		V3 g;
		if (in.swingMotorType == MI_MOTOR_VELOCITY) { g = A.rot * localMotorAxis; o[46] = in.swingMotorVelocity; }
		else
		{
			float target = in.swingMotorVelocity;
			if (in.swingLimit >= 0.f) target = clampf(target, -in.swingLimit, in.swingLimit);
			float h = target * 0.5f, sh = sinf(h);
			Q4 tq = q4(localMotorAxis.x * sh, localMotorAxis.y * sh, localMotorAxis.z * sh, cosf(h)); // quat(axis, angle) (math.h:932-936)
			V3 targetDir = tq * limitAxisA;
			V3 axis2 = noz(cross(limitAxisCompareA, targetDir));
			g = A.rot * axis2;
			float deltaAngle = acosf(clamp01(dot(targetDir, limitAxisCompareA)));
			o[46] = (dt > DT_THRESHOLD) ? (deltaAngle * invDt * 0.2f) : 0.f;
		}
		st3(o + 47, g);
		st3(o + 51, A.invI * g); st3(o + 54, B.invI * g);
		float invEff = dot(g, A.invI * g) + dot(g, B.invI * g);
		o[50] = (invEff != 0.f) ? (1.f / invEff) : 0.f;
	}
	bool minTwist = in.twistLimit >= 0.f && twistAngle <= -in.twistLimit;
	bool maxTwist = in.twistLimit >= 0.f && twistAngle >= in.twistLimit;
	bool twistLimit = minTwist || maxTwist, twistMotor = in.maxTwistMotorTorque > 0.f;
	if (twistLimit || twistMotor)
	{
		flags |= (twistLimit ? 2u : 0u) | (twistMotor ? 8u : 0u);
		V3 g = A.rot * limitAxisA;
		st3(o + 31, g);
		float invEff = dot(g, A.invI * g) + dot(g, B.invI * g);
		o[36] = (invEff != 0.f) ? (1.f / invEff) : 0.f;
		o[35] = minTwist ? 1.f : -1.f;
		o[58] = in.maxTwistMotorTorque * dt;
		st3(o + 38, A.invI * g); st3(o + 41, B.invI * g);
		float mv = in.twistMotorVelocity;
		if (in.twistMotorType == MI_MOTOR_POSITION)
		{
			float limit = (in.twistLimit >= 0.f) ? in.twistLimit : MI_PI;
			float target = clampf(in.twistMotorVelocity, -limit, limit);
			mv = (dt > DT_THRESHOLD) ? ((target - twistAngle) * invDt) : 0.f;
		}
		o[59] = mv;
		if (dt > DT_THRESHOLD)
		{
			float d = minTwist ? (in.twistLimit + twistAngle) : (in.twistLimit - twistAngle);
			o[37] = d * BETA_TWIST_LIMIT * invDt;
		}
	}
	o[18] = __uint_as_float(flags);
}
__global__ void k_cone_twist_solve(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	float* o = upd + (size_t)j * 80;
	Vel v = loadVel(vel, ab.x, ab.y);
	solveConeTwist(o, v, ldInvI(invIw, ab.x), ldInvI(invIw, ab.y));
	storeVel(vel, nb, ab.x, ab.y, v);
}

// ---- slider (update, 72 floats) -------------------------------------------------------------------------------------------
//  0 rAuxt3 | 3 rAuxb3 | 6 rBxt3 | 9 rBxb3 | 12 tangent3 | 15 bitangent3 | 18 invEffT(m00 m01 m10 m11) | 22 tBias2 | 24 invEffR9 | 33 rBias3
//  36 flags (1 limit, 2 motor) | 37 axis3 | 40 effAxial | 41 limitBias | 42 limitImpulse | 43 limitSign | 44 rAuxs3 | 47 rBxs3 | 50 limJA3 | 53 limJB3
//  56 motorVelocity | 57 motorImpulse | 58 maxMotorImpulse
__global__ void k_slider_init(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	const mi_slider_constraint& in = *(const mi_slider_constraint*)(pods + (size_t)j * sizeof(mi_slider_constraint));
	float* o = upd + (size_t)j * 72;
	BodyIn A = loadBody(ab.x, pose, bprops, cog, invIw), B = loadBody(ab.y, pose, bprops, cog, invIw);
	float invDt = 1.f / dt;
	for (u32 k = 36; k < 59; ++k) o[k] = 0.f;
	V3 rA = A.rot * (ld3(in.localAnchorA) - A.localCOG), rB = B.rot * (ld3(in.localAnchorB) - B.localCOG);
	V3 gA = A.pos + rA, gB = B.pos + rB;
	V3 axis = A.rot * ld3(in.localAxisA);
	V3 tangent = getTangent(axis), bitangent = cross(axis, tangent);
	V3 u = gB - gA;
	V3 rAu = rA + u;
	V3 rBxt = cross(rB, tangent), rBxb = cross(rB, bitangent), rAuxt = cross(rAu, tangent), rAuxb = cross(rAu, bitangent);
	V3 iArAuxt = A.invI * rAuxt, iArAuxb = A.invI * rAuxb, iBrBxt = B.invI * rBxt, iBrBxb = B.invI * rBxb;
	float invMassSum = A.invMass + B.invMass;
	st3(o, rAuxt); st3(o + 3, rAuxb); st3(o + 6, rBxt); st3(o + 9, rBxb); st3(o + 12, tangent); st3(o + 15, bitangent);
	o[18] = dot(rAuxt, iArAuxt) + dot(rBxt, iBrBxt) + invMassSum;
	o[19] = dot(rAuxt, iArAuxb) + dot(rBxt, iBrBxb);
	o[20] = dot(rAuxb, iArAuxt) + dot(rBxb, iBrBxt);
	o[21] = dot(rAuxb, iArAuxb) + dot(rBxb, iBrBxb) + invMassSum;
	stM3(o + 24, madd(A.invI, B.invI));
	o[22] = 0.f; o[23] = 0.f; st3(o + 33, v3s(0.f));
	if (dt > DT_THRESHOLD)
	{
		o[22] = dot(u, tangent) * (BETA_SLIDER * invDt); o[23] = dot(u, bitangent) * (BETA_SLIDER * invDt);
		const float* q = in.initialInvRotationDifference;
		Q4 err = B.rot * q4(q[0], q[1], q[2], q[3]) * conjugate(A.rot);
		st3(o + 33, qv(err) * (BETA_SLIDER * invDt * 2.f));
	}
	st3(o + 37, axis);
	float dist = dot(u, axis);
	u32 flags = 0;
	if (in.negDistanceLimit <= 0.f || in.posDistanceLimit >= 0.f)
	{
		bool minV = (in.negDistanceLimit <= 0.f) && (dist < in.negDistanceLimit);
		bool maxV = (in.posDistanceLimit >= 0.f) && (dist > in.posDistanceLimit);
		if (minV || maxV)
		{
			flags |= 1u;
			V3 rAuxs = cross(rAu, axis), rBxs = cross(rB, axis);
			float invEff = invMassSum + dot(rAuxs, A.invI * rAuxs) + dot(rBxs, B.invI * rBxs);
			o[40] = (invEff != 0.f) ? (1.f / invEff) : 0.f;
			o[43] = minV ? 1.f : -1.f;
			if (dt > DT_THRESHOLD) { float err = minV ? (dist - in.negDistanceLimit) : (in.posDistanceLimit - dist); o[41] = err * (BETA_SLIDER_LIMIT * invDt); }
			st3(o + 44, rAuxs); st3(o + 47, rBxs); st3(o + 50, A.invI * rAuxs); st3(o + 53, B.invI * rBxs);
		}
	}
	if (in.maxMotorForce > 0.f)
	{
		flags |= 2u;
		o[58] = in.maxMotorForce * dt;
		float mv = in.motorVelocity;
		if (in.motorType == MI_MOTOR_POSITION)
		{
			float minLimit = (in.negDistanceLimit <= 0.f) ? in.negDistanceLimit : -INFINITY;
			float maxLimit = (in.posDistanceLimit >= 0.f) ? in.posDistanceLimit : INFINITY;
			float target = clampf(in.motorVelocity, minLimit, maxLimit);
			mv = (dt > DT_THRESHOLD) ? ((target - dist) * invDt) : 0.f;
		}
		o[56] = mv;
	}
	o[36] = __uint_as_float(flags);
}
__global__ void k_slider_solve(JOINT_KERNEL_ARGS)
{
	JOINT_INDEX;
	float* o = upd + (size_t)j * 72;
	Vel v = loadVel(vel, ab.x, ab.y);
	solveSlider(o, v, ldInvI(invIw, ab.x), ldInvI(invIw, ab.y));
	storeVel(vel, nb, ab.x, ab.y, v);
}

typedef void (*joint_kernel_t)(u32, u32, u32, float, const uint8_t*, const uint2*, float*, const float4*, const float4*, const float4*, const float4*, float4*);
static const joint_kernel_t INIT_KERNELS[MI_JOINT_TYPES] = { k_distance_init, k_ball_init, k_fixed_init, k_hinge_init, k_cone_twist_init, k_slider_init };
static const joint_kernel_t SOLVE_KERNELS[MI_JOINT_TYPES] = { k_distance_solve, k_ball_solve, k_fixed_solve, k_hinge_solve, k_cone_twist_solve, k_slider_solve };

static void launchJoint(World& w, joint_kernel_t k, JointSet& js, u32 start, u32 end, float dt)
{
	if (end <= start) return;
	hipLaunchKernelGGL(k, dim3((end - start + 63) / 64), dim3(64), 0, w.stream, start, end, w.nb, dt, js.dPods.p, js.dPairs.p, js.dUpdate.p,
		w.pose.p, w.bprops.p, w.cog.p, w.invIw.p, w.vel.p);
}

void launch_joint_init(World& w, float dt)
{
	for (u32 t = 0; t < MI_JOINT_TYPES; ++t)
	{
		JointSet& js = w.joints[t];
		if (js.order.empty()) continue;
		launchJoint(w, INIT_KERNELS[t], js, 0, (u32)js.order.size(), dt);
	}
}

// Reference order inside one iteration: distance, ball, fixed, hinge, cone-twist, slider (constraints.cpp:3748-3772); contacts follow.
void launch_joint_solve_iteration(World& w)
{
	for (u32 t = 0; t < MI_JOINT_TYPES; ++t)
	{
		JointSet& js = w.joints[t];
		for (size_t c = 0; c + 1 < js.colorStart.size(); ++c) launchJoint(w, SOLVE_KERNELS[t], js, js.colorStart[c], js.colorStart[c + 1], 0.f);
	}
}
