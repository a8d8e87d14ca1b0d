// locomotion_env.cpp — the reference's one application of the physics path, rebuilt over the C-ABI:
// the humanoid ragdoll (src/physics/ragdoll.cpp:10-158) and the reinforcement-learning environment that the reference exports from
// its Physics-Lib DLL (src/learning/learned_locomotion.cpp:395-489; state / action / reward: :73-357).  Built as libmi_locomotion.so
// with the SAME five exports, so learning/loco_env.py binds it by changing the library path:
//     int  getPhysicsStateSize();  int getPhysicsActionSize();
//     void getPhysicsRanges(float* stateMin, float* stateMax, float* actionMin, float* actionMax);
//     void resetPhysics(float* outState);
//     int  updatePhysics(float* action, float* outState, float* outReward);   // returns 1 when the ragdoll has fallen
// One addition: setPhysicsSeed(uint64) — the reference seeds its random pushes with time(0); here the seed is explicit (default
// fixed) so that runs are reproducible.  resetPhysics also fills outState (the reference leaves it untouched).
// Everything physical happens in libmi_physics.so on the GPU; this file is host logic only, like the reference's.
//
// Build: g++ -std=c++17 -O2 -fPIC -shared -I../../include locomotion_env.cpp -L.. -lmi_physics -Wl,-rpath,'$ORIGIN' -o ../libmi_locomotion.so
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mi_physics.h"

namespace
{
	struct vec3 { float x, y, z; };
	struct quat { float x, y, z, w; };
	inline vec3 v3(float x, float y, float z) { return { x, y, z }; }
	inline vec3 operator+(vec3 a, vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
	inline vec3 operator-(vec3 a, vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
	inline vec3 operator*(vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
	inline vec3 operator*(float s, vec3 a) { return a * s; }
	inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
	inline vec3 cross(vec3 a, vec3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
	inline float length(vec3 a) { return sqrtf(dot(a, a)); }
	inline vec3 normalize(vec3 a) { return a * (1.f / length(a)); }
	inline quat conjugate(quat q) { return { -q.x, -q.y, -q.z, q.w }; }
	inline quat operator*(quat a, quat b) // core/math.cpp quat product
	{
		return { a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
			a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
	}
	inline vec3 operator*(quat q, vec3 v) { quat p = { v.x, v.y, v.z, 0.f }; quat r = q * p * conjugate(q); return { r.x, r.y, r.z }; }
	inline quat axisAngle(vec3 axis, float angle) { float h = angle * 0.5f, s = sinf(h); return { axis.x * s, axis.y * s, axis.z * s, cosf(h) }; }
	inline float deg2rad(float d) { return d * (3.14159265358979323846f / 180.f); }
	inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
	inline float clampf(float v, float l, float u) { return fminf(u, fmaxf(l, v)); }
	const float PI = 3.14159265358979323846f;

	struct trs { quat rotation; vec3 position; };
	inline vec3 transformPosition(const trs& m, vec3 p) { return m.rotation * p + m.position; }

	enum { NUM_BODY_PARTS = 14, NUM_CONE_TWIST = 7, NUM_HINGE = 6, ACTION_SIZE = NUM_CONE_TWIST * 3 + NUM_HINGE, STATE_SIZE = 13 * 3 + ACTION_SIZE };
	enum part { torso, head, leftUpperArm, leftLowerArm, rightUpperArm, rightLowerArm, leftUpperLeg, leftLowerLeg, leftFoot, leftToes, rightUpperLeg, rightLowerLeg, rightFoot, rightToes };
	const int NO_PARENT = -1;
	const int parentOf[NUM_BODY_PARTS] = { NO_PARENT, torso, torso, leftUpperArm, torso, rightUpperArm, torso, leftUpperLeg, leftLowerLeg, leftFoot, torso, rightUpperLeg, rightLowerLeg, rightFoot }; // ragdoll.cpp:155-168

	// byte-identical PODs (constraints.h:229-257, 346-380)
	struct hinge_pod { float a[12]; float minRotationLimit, maxRotationLimit, maxMotorTorque; uint32_t motorType; float motorTargetAngle; float t[9]; };
	struct cone_twist_pod { float a[21]; float swingLimit, twistLimit; uint32_t swingMotorType; float swingMotorTargetAngle, maxSwingMotorTorque, swingMotorAxis; uint32_t twistMotorType; float twistMotorTargetAngle, maxTwistMotorTorque; };
	static_assert(sizeof(hinge_pod) == 104 && sizeof(cone_twist_pod) == 120, "POD layout");

	struct rng64 // core/random.h:5-49
	{
		uint64_t state = 0x9E3779B97F4A7C15ull;
		uint64_t u64() { uint64_t x = state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; state = x; return x; }
		uint32_t u32() { return (uint32_t)u64(); }
		float f01() { return u32() / (float)UINT32_MAX; }
		float between(float lo, float hi) { return lo + f01() * (hi - lo); }
		uint32_t u32Between(uint32_t lo, uint32_t hi) { return u32() % (hi - lo) + lo; }
	};

	struct ragdoll
	{
		uint32_t body[NUM_BODY_PARTS];
		uint32_t coneTwist[NUM_CONE_TWIST], hinge[NUM_HINGE];
		vec3 boxMin[NUM_BODY_PARTS], boxMax[NUM_BODY_PARTS]; // local AABB of each part's colliders (learned_locomotion.cpp:199-246)
	};

	void grow(vec3& mn, vec3& mx, vec3 p) { mn = { fminf(mn.x, p.x), fminf(mn.y, p.y), fminf(mn.z, p.z) }; mx = { fmaxf(mx.x, p.x), fmaxf(mx.y, p.y), fmaxf(mx.z, p.z) }; }

	// humanoid_ragdoll::initialize — ragdoll.cpp:10-133.  Bodies are created in the un-placed pose, the joints from global points
	// in that pose (local anchors do not care about the later placement), then every part is rotated about the hip and moved.
	ragdoll createRagdoll(mi_world* w, vec3 hip, float initialRotation)
	{
		const float scale = 0.42f;
		mi_material material = { 0.2f, 1.f, 985.f };
		const vec3 Z = v3(0.f, 0.f, 1.f);
		const quat I = { 0.f, 0.f, 0.f, 1.f };
		trs t[NUM_BODY_PARTS] = {
			{ I, scale * v3(0.f, 0.f, 0.f) }, { I, scale * v3(0.f, 1.45f, 0.f) },
			{ axisAngle(Z, deg2rad(-30.f)), scale * v3(-0.6f, 0.75f, 0.f) }, { axisAngle(Z, deg2rad(-20.f)), scale * v3(-0.884f, 0.044f, -0.043f) },
			{ axisAngle(Z, deg2rad(30.f)), scale * v3(0.6f, 0.75f, 0.f) }, { axisAngle(Z, deg2rad(20.f)), scale * v3(0.884f, 0.044f, -0.043f) },
			{ axisAngle(Z, deg2rad(-10.f)), scale * v3(-0.371f, -0.812f, 0.f) }, { axisAngle(Z, deg2rad(-3.5f)), scale * v3(-0.452f, -1.955f, 0.f) },
			{ I, scale * v3(-0.498f, -2.585f, -0.18f) }, { I, scale * v3(-0.498f, -2.585f, -0.637f) },
			{ axisAngle(Z, deg2rad(10.f)), scale * v3(0.371f, -0.812f, 0.f) }, { axisAngle(Z, deg2rad(3.5f)), scale * v3(0.452f, -1.955f, 0.f) },
			{ I, scale * v3(0.498f, -2.585f, -0.18f) }, { I, scale * v3(0.498f, -2.585f, -0.637f) } };
		ragdoll r;
		for (int i = 0; i < NUM_BODY_PARTS; ++i)
		{
			r.body[i] = mi_add_body(w, 0, 1.f, 0.4f, 0.4f, &t[i].position.x, &t[i].rotation.x);
			r.boxMin[i] = v3(FLT_MAX, FLT_MAX, FLT_MAX); r.boxMax[i] = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
		}
		auto capsule = [&](int p, vec3 a, vec3 b, float radius)
		{
			float s[7] = { scale * a.x, scale * a.y, scale * a.z, scale * b.x, scale * b.y, scale * b.z, scale * radius };
			mi_add_collider(w, r.body[p], MI_COLLIDER_CAPSULE, s, &material);
			vec3 r3 = v3(s[6], s[6], s[6]), A = v3(s[0], s[1], s[2]), B = v3(s[3], s[4], s[5]);
			grow(r.boxMin[p], r.boxMax[p], A + r3); grow(r.boxMin[p], r.boxMax[p], A - r3); grow(r.boxMin[p], r.boxMax[p], B + r3); grow(r.boxMin[p], r.boxMax[p], B - r3);
		};
		auto box = [&](int p, vec3 radius)
		{
			float s[6] = { -scale * radius.x, -scale * radius.y, -scale * radius.z, scale * radius.x, scale * radius.y, scale * radius.z };
			mi_add_collider(w, r.body[p], MI_COLLIDER_AABB, s, &material);
			grow(r.boxMin[p], r.boxMax[p], v3(s[0], s[1], s[2])); grow(r.boxMin[p], r.boxMax[p], v3(s[3], s[4], s[5]));
		};
		capsule(torso, v3(-0.2f, 0.f, 0.f), v3(0.2f, 0.f, 0.f), 0.25f); capsule(torso, v3(-0.16f, 0.32f, 0.f), v3(0.16f, 0.32f, 0.f), 0.2f);
		capsule(torso, v3(-0.14f, 0.62f, 0.f), v3(0.14f, 0.62f, 0.f), 0.22f); capsule(torso, v3(-0.14f, 0.92f, 0.f), v3(0.14f, 0.92f, 0.f), 0.2f);
		capsule(head, v3(0.f, -0.075f, 0.f), v3(0.f, 0.075f, 0.f), 0.25f);
		for (int p : { leftUpperArm, leftLowerArm, rightUpperArm, rightLowerArm }) capsule(p, v3(0.f, -0.2f, 0.f), v3(0.f, 0.2f, 0.f), 0.15f);
		capsule(leftUpperLeg, v3(0.f, -0.3f, 0.f), v3(0.f, 0.3f, 0.f), 0.25f); capsule(leftLowerLeg, v3(0.f, -0.3f, 0.f), v3(0.f, 0.3f, 0.f), 0.18f);
		box(leftFoot, v3(0.1587f, 0.1f, 0.3424f)); capsule(leftToes, v3(-0.0587f, 0.f, 0.f), v3(0.0587f, 0.f, 0.f), 0.1f);
		capsule(rightUpperLeg, v3(0.f, -0.3f, 0.f), v3(0.f, 0.3f, 0.f), 0.25f); capsule(rightLowerLeg, v3(0.f, -0.3f, 0.f), v3(0.f, 0.3f, 0.f), 0.18f);
		box(rightFoot, v3(0.1587f, 0.1f, 0.3424f)); capsule(rightToes, v3(-0.0587f, 0.f, 0.f), v3(0.0587f, 0.f, 0.f), 0.1f);

		auto tp = [&](int p, vec3 local) { return transformPosition(t[p], scale * local); };
		auto td = [&](int p, vec3 d) { return t[p].rotation * d; };
		auto coneTwistJ = [&](int a, int b, vec3 anchor, vec3 axis, float swing, float twist) { return mi_add_cone_twist_constraint_global(w, r.body[a], r.body[b], &anchor.x, &axis.x, swing, twist); };
		auto hingeJ = [&](int a, int b, vec3 anchor, vec3 axis, float mn, float mx) { return mi_add_hinge_constraint_global(w, r.body[a], r.body[b], &anchor.x, &axis.x, mn, mx); };
		// per-type ids follow the add order; the reference's handle arrays (ragdoll.h:60-83) are in exactly this order
		uint32_t neck = coneTwistJ(torso, head, tp(torso, v3(0.f, 1.2f, 0.f)), v3(0.f, 1.f, 0.f), deg2rad(50.f), deg2rad(90.f));
		uint32_t lShoulder = coneTwistJ(torso, leftUpperArm, tp(torso, v3(-0.4f, 1.f, 0.f)), v3(-1.f, 0.f, 0.f), deg2rad(130.f), deg2rad(90.f));
		uint32_t lElbow = hingeJ(leftUpperArm, leftLowerArm, tp(leftUpperArm, v3(0.f, -0.42f, 0.f)), normalize(v3(1.f, 0.f, 1.f)), deg2rad(-5.f), deg2rad(85.f));
		uint32_t rShoulder = coneTwistJ(torso, rightUpperArm, tp(torso, v3(0.4f, 1.f, 0.f)), v3(1.f, 0.f, 0.f), deg2rad(130.f), deg2rad(90.f));
		uint32_t rElbow = hingeJ(rightUpperArm, rightLowerArm, tp(rightUpperArm, v3(0.f, -0.42f, 0.f)), normalize(v3(1.f, 0.f, -1.f)), deg2rad(-5.f), deg2rad(85.f));
		uint32_t lHip = coneTwistJ(torso, leftUpperLeg, tp(torso, v3(-0.3f, -0.25f, 0.f)), td(leftUpperLeg, v3(0.f, -1.f, 0.f)), -1.f, deg2rad(30.f));
		uint32_t lKnee = hingeJ(leftUpperLeg, leftLowerLeg, tp(leftUpperLeg, v3(0.f, -0.6f, 0.f)), v3(1.f, 0.f, 0.f), deg2rad(-90.f), deg2rad(5.f));
		uint32_t lAnkle = coneTwistJ(leftLowerLeg, leftFoot, tp(leftLowerLeg, v3(0.f, -0.52f, 0.f)), td(leftLowerLeg, v3(0.f, -1.f, 0.f)), deg2rad(75.f), deg2rad(20.f));
		uint32_t lToes = hingeJ(leftFoot, leftToes, tp(leftFoot, v3(0.f, 0.f, -0.36f)), v3(1.f, 0.f, 0.f), deg2rad(-45.f), deg2rad(45.f));
		uint32_t rHip = coneTwistJ(torso, rightUpperLeg, tp(torso, v3(0.3f, -0.25f, 0.f)), td(rightUpperLeg, v3(0.f, -1.f, 0.f)), -1.f, deg2rad(30.f));
		uint32_t rKnee = hingeJ(rightUpperLeg, rightLowerLeg, tp(rightUpperLeg, v3(0.f, -0.6f, 0.f)), v3(1.f, 0.f, 0.f), deg2rad(-90.f), deg2rad(5.f));
		uint32_t rAnkle = coneTwistJ(rightLowerLeg, rightFoot, tp(rightLowerLeg, v3(0.f, -0.52f, 0.f)), td(rightLowerLeg, v3(0.f, -1.f, 0.f)), deg2rad(75.f), deg2rad(20.f));
		uint32_t rToes = hingeJ(rightFoot, rightToes, tp(rightFoot, v3(0.f, 0.f, -0.36f)), v3(1.f, 0.f, 0.f), deg2rad(-45.f), deg2rad(45.f));
		const uint32_t ct[NUM_CONE_TWIST] = { neck, lShoulder, rShoulder, lHip, lAnkle, rHip, rAnkle };
		const uint32_t hg[NUM_HINGE] = { lElbow, rElbow, lKnee, lToes, rKnee, rToes };
		memcpy(r.coneTwist, ct, sizeof(ct)); memcpy(r.hinge, hg, sizeof(hg));

		quat rotation = axisAngle(v3(0.f, 1.f, 0.f), initialRotation);
		for (int i = 0; i < NUM_BODY_PARTS; ++i) // ragdoll.cpp:125-133
		{
			quat q = rotation * t[i].rotation;
			vec3 p = rotation * t[i].position + hip;
			mi_set_transform(w, r.body[i], &p.x, &q.x);
		}
		return r;
	}

	struct target { vec3 positions[6], velocities[6]; quat localRotation; };

	struct environment
	{
		mi_world* world = nullptr;
		ragdoll doll;
		float lastSmoothedAction[ACTION_SIZE] = {};
		float headTargetHeight = 0.f;
		vec3 torsoVelocityTarget = { 0.f, 0.f, 0.f };
		vec3 localPositions[NUM_BODY_PARTS][6];
		target targets[NUM_BODY_PARTS];
		vec3 localCOG[NUM_BODY_PARTS];
		float totalReward = 0.f;
		rng64 rng;
		// snapshot of the device state, refreshed after every step: transform_component (interpolated), velocities
		trs transform[NUM_BODY_PARTS]; vec3 linearVelocity[NUM_BODY_PARTS], angularVelocity[NUM_BODY_PARTS];

		void snapshot()
		{
			uint32_t n = mi_num_bodies(world);
			std::vector<float> t(7 * (size_t)n), v(6 * (size_t)n);
			mi_read_transforms(world, 0 /* transform_component: what the reference's getState reads */, t.data(), n);
			mi_read_velocities(world, v.data(), n);
			for (int i = 0; i < NUM_BODY_PARTS; ++i)
			{
				const float* p = &t[7 * (size_t)doll.body[i]]; const float* q = &v[6 * (size_t)doll.body[i]];
				transform[i] = { { p[3], p[4], p[5], p[6] }, { p[0], p[1], p[2] } };
				linearVelocity[i] = v3(q[0], q[1], q[2]); angularVelocity[i] = v3(q[3], q[4], q[5]);
			}
		}
		vec3 globalCOG(int i) const { return transform[i].position + transform[i].rotation * localCOG[i]; } // rigid_body.cpp:83-86
		vec3 pointVelocity(int i, vec3 localP) const { return linearVelocity[i] + cross(angularVelocity[i], transformPosition(transform[i], localP) - globalCOG(i)); } // :88-93

		void applyAction(const float* action) // learned_locomotion.cpp:73-109
		{
			for (int i = 0; i < ACTION_SIZE; ++i) lastSmoothedAction[i] = lerpf(lastSmoothedAction[i], action[i], 0.1f);
			for (int i = 0; i < NUM_CONE_TWIST; ++i)
			{
				cone_twist_pod c; mi_constraint_get(world, MI_CONSTRAINT_CONE_TWIST, doll.coneTwist[i], &c);
				c.maxSwingMotorTorque = 200.f; c.maxTwistMotorTorque = 200.f; c.swingMotorType = MI_MOTOR_POSITION; c.twistMotorType = MI_MOTOR_POSITION;
				c.twistMotorTargetAngle = lastSmoothedAction[3 * i]; c.swingMotorTargetAngle = lastSmoothedAction[3 * i + 1]; c.swingMotorAxis = lastSmoothedAction[3 * i + 2];
				mi_constraint_set(world, MI_CONSTRAINT_CONE_TWIST, doll.coneTwist[i], &c);
			}
			for (int i = 0; i < NUM_HINGE; ++i)
			{
				hinge_pod c; mi_constraint_get(world, MI_CONSTRAINT_HINGE, doll.hinge[i], &c);
				c.maxMotorTorque = 200.f; c.motorType = MI_MOTOR_POSITION; c.motorTargetAngle = lastSmoothedAction[3 * NUM_CONE_TWIST + i];
				mi_constraint_set(world, MI_CONSTRAINT_HINGE, doll.hinge[i], &c);
			}
		}
		vec3 frameOrigin() const { vec3 c = globalCOG(torso); c.y = 0.f; return c; } // getCoordinateSystem (:111-122): torso COG on the ground, identity rotation
		void getState(float* out) const // :135-152; field order of learning_state (learned_locomotion.h:42-68)
		{
			vec3 o = frameOrigin();
			auto put = [&](int slot, vec3 v) { out[3 * slot] = v.x; out[3 * slot + 1] = v.y; out[3 * slot + 2] = v.z; };
			put(0, linearVelocity[torso]);
			put(1, globalCOG(leftToes) - o); put(2, linearVelocity[leftToes]);
			put(3, globalCOG(rightToes) - o); put(4, linearVelocity[rightToes]);
			put(5, globalCOG(torso) - o); put(6, linearVelocity[torso]);
			put(7, globalCOG(head) - o); put(8, linearVelocity[head]);
			put(9, globalCOG(leftLowerArm) - o); put(10, linearVelocity[leftLowerArm]);
			put(11, globalCOG(rightLowerArm) - o); put(12, linearVelocity[rightLowerArm]);
			memcpy(out + 39, lastSmoothedAction, sizeof(lastSmoothedAction));
		}
		quat localRotation(int i) const { quat parentRotation = parentOf[i] == NO_PARENT ? quat{ 0.f, 0.f, 0.f, 1.f } : transform[parentOf[i]].rotation; return transform[i].rotation * conjugate(parentRotation); }
		void resetTraining() // training_locomotion::reset (:300-311) + learned_locomotion::reset (:36-44)
		{
			for (int i = 0; i < NUM_BODY_PARTS; ++i)
			{
				vec3 c = (doll.boxMin[i] + doll.boxMax[i]) * 0.5f, r = (doll.boxMax[i] - doll.boxMin[i]) * 0.5f;
				vec3* p = localPositions[i];
				p[0] = c - v3(r.x, 0.f, 0.f); p[1] = c - v3(0.f, r.y, 0.f); p[2] = c - v3(0.f, 0.f, r.z); p[3] = c + v3(r.x, 0.f, 0.f); p[4] = c + v3(0.f, r.y, 0.f); p[5] = c + v3(0.f, 0.f, r.z);
				for (int k = 0; k < 6; ++k) { targets[i].positions[k] = transformPosition(transform[i], p[k]); targets[i].velocities[k] = pointVelocity(i, p[k]); }
				targets[i].localRotation = localRotation(i);
			}
			memset(lastSmoothedAction, 0, sizeof(lastSmoothedAction));
			float zero[ACTION_SIZE] = {};
			applyAction(zero);
			headTargetHeight = transform[head].position.y;
			torsoVelocityTarget = v3(0.f, 0.f, 0.f);
		}
		float getReward() const // :325-357
		{
			float positionError = 0.f, velocityError = 0.f, rotationError = 0.f;
			for (int i = 0; i < NUM_BODY_PARTS; ++i)
			{
				for (int k = 0; k < 6; ++k)
				{
					positionError += length(transformPosition(transform[i], localPositions[i][k]) - targets[i].positions[k]);
					velocityError += length(pointVelocity(i, localPositions[i][k]) - targets[i].velocities[k]);
				}
				quat d = targets[i].localRotation * conjugate(localRotation(i));
				rotationError += 2.f * acosf(clampf(d.w, -1.f, 1.f));
			}
			float vcmError = length(linearVelocity[torso] - torsoVelocityTarget);
			float rp = expf(-10.f / NUM_BODY_PARTS * positionError), rv = expf(-1.f / NUM_BODY_PARTS * velocityError);
			float rlocal = expf(-10.f / NUM_BODY_PARTS * rotationError), rvcm = expf(-vcmError);
			float fall = clampf(1.3f - 1.4f * (headTargetHeight - transform[head].position.y), 0.f, 1.f);
			return fall * (rp + rv + rlocal + rvcm);
		}
	};

	environment* env = nullptr;
	uint64_t seed = 0x9E3779B97F4A7C15ull;

	void fillRanges(mi_world* w, const ragdoll& r, float* actionMin, float* actionMax) // getLimits (:365-385)
	{
		int k = 0;
		for (int i = 0; i < NUM_CONE_TWIST; ++i)
		{
			cone_twist_pod c; mi_constraint_get(w, MI_CONSTRAINT_CONE_TWIST, r.coneTwist[i], &c);
			actionMin[k] = c.twistLimit >= 0.f ? -c.twistLimit : -PI; actionMax[k++] = c.twistLimit >= 0.f ? c.twistLimit : PI;
			actionMin[k] = c.swingLimit >= 0.f ? -c.swingLimit : -PI; actionMax[k++] = c.swingLimit >= 0.f ? c.swingLimit : PI;
			actionMin[k] = -PI; actionMax[k++] = PI;
		}
		for (int i = 0; i < NUM_HINGE; ++i)
		{
			hinge_pod c; mi_constraint_get(w, MI_CONSTRAINT_HINGE, r.hinge[i], &c);
			actionMin[k] = c.minRotationLimit <= 0.f ? c.minRotationLimit : -PI; actionMax[k++] = c.maxRotationLimit >= 0.f ? c.maxRotationLimit : PI;
		}
	}
}

extern "C"
{
	int getPhysicsStateSize() { return STATE_SIZE; }   // learned_locomotion.cpp:401
	int getPhysicsActionSize() { return ACTION_SIZE; } // :402

	void setPhysicsSeed(unsigned long long s) { seed = s ? s : 0x9E3779B97F4A7C15ull; if (env) env->rng.state = seed; }

	void getPhysicsRanges(float* stateMin, float* stateMax, float* actionMin, float* actionMax) // :404-433
	{
		for (int i = 0; i < STATE_SIZE; ++i) { stateMin[i] = -FLT_MAX; stateMax[i] = FLT_MAX; }
		mi_world_desc d = { -1, 0, 0, 0 };
		mi_world* w = mi_world_create(&d);
		if (!w) { fprintf(stderr, "getPhysicsRanges: %s\n", mi_last_error(nullptr)); return; }
		ragdoll r = createRagdoll(w, v3(0.f, 0.f, 0.f), 0.f);
		fillRanges(w, r, actionMin, actionMax);
		mi_world_destroy(w);
	}

	void resetPhysics(float* outState) // :435-461
	{
		if (!env) { env = new environment; env->rng.state = seed; }
		if (env->world) mi_world_destroy(env->world);
		mi_world_desc d = { -1, 0, 0, 0 };
		env->world = mi_world_create(&d);
		if (!env->world) { fprintf(stderr, "resetPhysics: %s\n", mi_last_error(nullptr)); return; }
		env->totalReward = 0.f;
		mi_material ground = { 0.1f, 1.f, 4.f };
		float box[6] = { -20.f, -4.f, -20.f, 20.f, 4.f, 20.f }, pos[3] = { 0.f, -4.f, 0.f }, rot[4] = { 0.f, 0.f, 0.f, 1.f };
		mi_add_static_collider(env->world, MI_COLLIDER_AABB, box, &ground, pos, rot);
		env->doll = createRagdoll(env->world, v3(0.f, 1.25f, 0.f), 0.f);
		std::vector<float> mp(13 * (size_t)mi_num_bodies(env->world));
		mi_read_mass_properties(env->world, mp.data(), mi_num_bodies(env->world));
		for (int i = 0; i < NUM_BODY_PARTS; ++i) { const float* m = &mp[13 * (size_t)env->doll.body[i]]; env->localCOG[i] = v3(m[0], m[1], m[2]); }
		env->snapshot();
		env->resetTraining();
		if (outState) env->getState(outState);
	}

	int updatePhysics(float* action, float* outState, float* outReward) // :463-489
	{
		if (!env || !env->world) { if (outReward) *outReward = 0.f; return 1; }
		env->applyAction(action);
		if (env->rng.f01() < 0.02f) // a random push every ~50 steps
		{
			uint32_t bodyPartIndex = env->rng.u32Between(0, NUM_BODY_PARTS - 1);
			vec3 part = env->transform[bodyPartIndex].position + v3(0.f, 0.2f, 0.f);
			float dx = env->rng.between(-1.f, 1.f), dz = env->rng.between(-1.f, 1.f);
			vec3 direction = normalize(v3(dx, 0.f, dz));
			vec3 origin = part - direction * 5.f;
			mi_test_physics_interaction(env->world, &origin.x, &direction.x, 1000.f);
		}
		mi_physics_settings s = { 1, 60, 4, 30, 0, 1, 0, 1, 1, 1 };
		float timer = 0.f;
		mi_step(env->world, &timer, &s, 1.f / 60.f);
		env->snapshot();
		env->getState(outState);
		bool failure = outState[3 * 7 + 1] < 1.f; // hasFallen: headPosition.y < 1 (:154-157)
		*outReward = 0.f;
		if (!failure) { *outReward = env->getReward(); env->totalReward += *outReward; }
		return failure ? 1 : 0;
	}
}
