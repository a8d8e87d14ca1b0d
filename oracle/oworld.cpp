// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).  PARITY UNPINNED by the reference (it has no tests).
// World, mass properties, broadphase, narrowphase driver, step and C API of the CPU restatement.
// Follows src/physics/physics.cpp:631-756 (world-space colliders), :1180-1413 (physicsStepInternal/physicsStep),
// :1416-1519 (mass properties), rigid_body.cpp:6-142, collision_broad.cpp:87-166 + 297-447,
// collision_narrow.cpp:2221-2253 + 2328-2603.  Body index = add order; collider index = add order
// (the reference's EnTT storage order is not observable here, SURVEY §8c "Third-party arithmetic").
#include "onarrow.h"
#include "oconstraints.h"
#include "owide.h"
#include "ocloth.h"
#include "oheightmap.h"
#include <vector>
#include <chrono>
#include <cstdio>
#include <map>

namespace orc {

u32 g_gjkMaxItersSeen = 0;
u32 g_epaMaxTriangles = 0, g_epaMaxEdges = 0, g_epaMaxBorder = 0;

static const float GRAVITY = -9.81f; // physics.h:11
static const u32 STATIC_BODY = 0xFFFFFFFFu;

// rigid_body.h:18-46 + physics_transform0/1 (rigid_body.h:48-58) + the entity's transform_component
struct body
{
	vec3 localCOGPosition; float invMass; mat3 invInertia;
	float gravityFactor, linearDamping, angularDamping;
	vec3 linearVelocity, angularVelocity, forceAccumulator, torqueAccumulator;
	trs transform, transform0, transform1;
	std::vector<u32> colliders; // add order; the reference's intrusive list walks it newest-first (scene.h:56-58)
	bool removed = false;       // entity deleted: the body and its colliders no longer take part (indices stay valid)
};

struct collider
{
	collider_union local;  // shape in the parent's local space
	u32 parent;            // body id or STATIC_BODY
	trs staticTransform;   // transform of a static collider's entity (or of the force field / trigger entity it belongs to)
	u32 zoneType = physics_object_type_static_collider; // physics_object_type_force_field / _trigger: the collider of such an entity (physics.cpp:657-666)
	u32 zoneIndex = 0;     // force field / trigger id
};

// force_field_component (physics.h:182-185) + its entity's optional transform_component
struct force_field { vec3 force; bool hasTransform; trs transform; u32 numColliders; };
// trigger_component (physics.h:200-203) without the callback: events are drained by the caller
struct trigger { trs transform; u32 numColliders; };
// trigger_event / collision_begin_event / collision_end_event (physics.h:187-198, 356-376) as one plain record; same layout as mi_event
enum event_kind : u32 { event_trigger_enter = 0, event_trigger_leave = 1, event_collision_begin = 2, event_collision_end = 3 };
struct event_record { u32 kind, step, a, b, bodyA, bodyB; float position[3], normal[3], relativeVelocity[3]; };
struct entity_pair { u32 a, b; bool operator<(const entity_pair& o) const { return a < o.a || (a == o.a && b < o.b); } bool operator==(const entity_pair& o) const { return a == o.a && b == o.b; } };
struct collision_entity_pair { u32 a, b, contactOffset, numContacts; bool operator<(const collision_entity_pair& o) const { return a < o.a || (a == o.a && b < o.b); } bool operator==(const collision_entity_pair& o) const { return a == o.a && b == o.b; } };

struct sap_endpoint { float value; u32 collider; bool start; };

// physics.h:382-397 minus the std::function callbacks
struct physics_settings
{
	u32 fixedFrameRate, frameRate, maxPhysicsIterationsPerFrame, numRigidSolverIterations;
	u32 numClothVelocityIterations, numClothPositionIterations, numClothDriftIterations;
	u32 simdBroadPhase, simdNarrowPhase, simdConstraintSolver;
};

enum solver_mode : u32 { solver_scalar = 0, solver_wide8 = 1, solver_custom_order = 2, solver_replay = 3 };

struct world
{
	std::vector<body> bodies;
	std::vector<collider> colliders;
	std::vector<bounding_hull_geometry> hullGeometries; // boundingHullGeometries (physics.cpp:47), per world instead of global

	std::vector<distance_constraint> distanceConstraints; std::vector<constraint_body_pair> distancePairs;
	std::vector<ball_constraint> ballConstraints; std::vector<constraint_body_pair> ballPairs;
	std::vector<fixed_constraint> fixedConstraints; std::vector<constraint_body_pair> fixedPairs;
	std::vector<hinge_constraint> hingeConstraints; std::vector<constraint_body_pair> hingePairs;
	std::vector<cone_twist_constraint> coneTwistConstraints; std::vector<constraint_body_pair> coneTwistPairs;
	std::vector<slider_constraint> sliderConstraints; std::vector<constraint_body_pair> sliderPairs;

	// force fields, triggers, events (physics.cpp:759-787, 952-1178)
	std::vector<force_field> forceFields;
	std::vector<trigger> triggers;
	std::vector<entity_pair> prevFrameTriggerOverlaps;              // event_context
	std::vector<collision_entity_pair> prevFrameCollisions;
	bool collisionBeginEvents = false, collisionEndEvents = false;  // physics_settings::collisionBeginCallback / collisionEndCallback set
	std::vector<event_record> events;                               // callbacks in call order, drained by orc_drain_events
	u32 stepIndex = 0;
	u32 terrainSlotMismatch = 0;                                    // follow mode: colliders whose terrain contact count differs from the slots handed in
	heightmap terrain; bool hasTerrain = false;                     // heightmap_collider_component (physics.cpp:1236-1249)
	std::vector<cloth> cloths;                                      // cloth_component, stepped after the rigid bodies (physics.cpp:1354-1358)
	u32 clothIterations[3] = { 0, 1, 0 };                           // physics_settings::numCloth{Velocity,Position,Drift}Iterations (physics.h:387-389)
	bool clothColourOrder = false;                                  // Gauss-Seidel order of the cloth constraints: storage (reference) or colour (device)
	u32 zoneTested[36] = {}, zoneHit[36] = {};                      // overlap checks per type pair (typeA * 6 + typeB), for test coverage reports

	// sap_context (collision_broad.cpp:20-24)
	std::vector<sap_endpoint> endpoints;
	u32 sortingAxis = 0;
	bool wideBroadphase = false; // the sweep of determineOverlapsSIMD (8 active boxes per compare) instead of determineOverlapsScalar: same pair list, what the reference's AVX2 build runs
	std::vector<u8> simOff; // per body: 1 = not simulated in this world (a spatial slab of a multi-GPU run simulates what it owns plus ghosts): its colliders take no part, its state is frozen
	float lastVariance[3] = { 0.f, 0.f, 0.f }; // of the last broadphase's AABB centres (the next axis is their argmax): tests look at near ties

	// Per-step arrays kept for inspection by tests.
	std::vector<bounding_box> worldSpaceAABBs;
	std::vector<collider_union> worldSpaceColliders;
	std::vector<collider_pair> broadphasePairs;
	std::vector<collision_contact> contacts;
	std::vector<constraint_body_pair> contactBodyPairs;
	std::vector<collider_pair> collidingPairs;       // ordered (A,B) as the contact normal sees them
	std::vector<u8> contactCountPerCollision;
	std::vector<u32> contactCollisionIndex;          // contact -> index into collidingPairs
	std::vector<rigid_body_global_state> rbGlobal;   // after applyGravityAndIntegrateForces (+ solve)
	std::vector<rigid_body_global_state> rbGlobalPreSolve;
	std::vector<collision_constraint> contactConstraints;
	std::vector<sched_slot> contactSlots;
	u32 usedSortingAxis = 0;

	// Optional external GS order for contacts: a permutation (or subset order) of contact indices.
	std::vector<u32> customOrder;
	double stageSeconds[5] = { 0, 0, 0, 0, 0 }; // cumulative wall time per stage (colliders + broadphase, narrowphase, forces + constraint setup, solve, velocity integration): cpu_baseline's breakdown
	bool rowForm = true; // custom-order solves use the device's row form (solveCollisionConstraintRowForm); false = the reference formula
	bool scalarRowForm = false; // the scalar solver (emission order) in row form: lets tests bound row form vs reference formula on the same order
	// "Follow" mode for whole-step parity with a device run: the narrowphase consumes an externally ordered candidate-pair list
	// (slots) instead of prune/classify/bucket, contacts are solved manifold by manifold in `slotOrder`, joints in `jointOrder[t]`.
	bool usePairOverride = false;
	std::vector<collider_pair> pairOverride;
	std::vector<u32> slotOrder;
	std::vector<u8> slotCounts;          // contacts per candidate slot of the last narrowphase (override mode)
	std::vector<u32> jointOrder[6];
};

// ---------------------------------------------------------------------------------------------------
// Mass properties — physics.cpp:1416-1519 (per collider) + rigid_body.cpp:29-81 (combine)
// ---------------------------------------------------------------------------------------------------
struct physics_properties { mat3 inertia; vec3 cog; float mass; };

// core/math.cpp:443-448
static float determinant3(const mat3& m)
{
	return m.m00 * (m.m11 * m.m22 - m.m21 * m.m12)
		- m.m01 * (m.m10 * m.m22 - m.m20 * m.m12)
		+ m.m02 * (m.m10 * m.m21 - m.m20 * m.m11);
}

static physics_properties calculatePhysicsProperties(const collider_union& c, const std::vector<bounding_hull_geometry>& hullGeometries)
{
	physics_properties result;
	switch (c.type)
	{
		case collider_type_hull: // physics.cpp:1520-1580 (http://number-none.com/blow/inertia/)
		{
			bounding_hull hull = c.hull();
			const bounding_hull_geometry& geom = hullGeometries[hull.geometryIndex];
			const float s60 = 1.f / 60.f, s120 = 1.f / 120.f;
			mat3 Ccanonical; // covariance of the unit tetrahedron
			Ccanonical.m00 = s60; Ccanonical.m01 = s120; Ccanonical.m02 = s120;
			Ccanonical.m10 = s120; Ccanonical.m11 = s60; Ccanonical.m12 = s120;
			Ccanonical.m20 = s120; Ccanonical.m21 = s120; Ccanonical.m22 = s60;
			float totalMass = 0.f;
			mat3 totalCovariance = mat3::zero();
			vec3 totalCOG(0.f);
			for (const bounding_hull_face& face : geom.faces)
			{
				vec3 w1 = hull.position + hull.rotation * geom.vertices[face.a];
				vec3 w2 = hull.position + hull.rotation * geom.vertices[face.b];
				vec3 w3 = hull.position + hull.rotation * geom.vertices[face.c];
				mat3 A; // columns w1, w2, w3
				A.m00 = w1.x; A.m01 = w2.x; A.m02 = w3.x;
				A.m10 = w1.y; A.m11 = w2.y; A.m12 = w3.y;
				A.m20 = w1.z; A.m21 = w2.z; A.m22 = w3.z;
				float detA = determinant3(A);
				mat3 covariance = ((A * detA) * Ccanonical) * transpose(A);
				float volume = 1.f / 6.f * detA;
				float mass = volume;
				vec3 cog = (w1 + w2 + w3) * 0.25f;
				totalMass += mass;
				totalCovariance = totalCovariance + covariance;
				totalCOG += cog * mass;
			}
			totalCOG /= totalMass;
			mat3 CprimeTotal = totalCovariance - outerProduct(totalCOG, totalCOG) * totalMass;
			result.cog = totalCOG;
			result.mass = totalMass * c.material.density;
			result.inertia = mat3::identity() * (CprimeTotal.m00 + CprimeTotal.m11 + CprimeTotal.m22) - CprimeTotal;
			result.inertia = result.inertia * c.material.density;
		} break;
		case collider_type_sphere:
		{
			bounding_sphere s = c.sphere();
			result.mass = sphereVolume(s.radius) * c.material.density;
			result.cog = s.center;
			result.inertia = mat3::identity() * (2.f / 5.f * result.mass * s.radius * s.radius);
		} break;
		case collider_type_capsule:
		{
			bounding_capsule cap = c.capsule();
			vec3 axis = cap.positionA - cap.positionB;
			if (axis.y < 0.f) { axis *= -1.f; }
			float height = length(axis);
			axis *= (1.f / height);
			quat rotation = rotateFromTo(vec3(0.f, 1.f, 0.f), axis);
			mat3 rot = quaternionToMat3(rotation);
			result.mass = capsuleVolume(cap) * c.material.density;
			result.cog = (cap.positionA + cap.positionB) * 0.5f;
			float sqRadius = cap.radius * cap.radius;
			float sqRadiusPI = M_PI_F * sqRadius;
			float cylinderMass = c.material.density * sqRadiusPI * height;
			float hemiSphereMass = c.material.density * 2.f / 3.f * sqRadiusPI * cap.radius;
			float sqCapsuleHeight = height * height;
			mat3 I;
			I.m11 = sqRadius * cylinderMass * 0.5f;
			I.m00 = I.m22 = I.m11 * 0.5f + cylinderMass * sqCapsuleHeight / 12.f;
			float temp0 = hemiSphereMass * 2.f * sqRadius / 5.f;
			I.m11 += temp0 * 2.f;
			float temp1 = height * 0.5f;
			float temp2 = temp0 + hemiSphereMass * (temp1 * temp1 + 3.f / 8.f * sqCapsuleHeight);
			I.m00 += temp2 * 2.f;
			I.m22 += temp2 * 2.f;
			result.inertia = transpose(rot) * I * rot;
		} break;
		case collider_type_cylinder:
		{
			bounding_cylinder cyl = c.cylinder();
			vec3 axis = cyl.positionA - cyl.positionB;
			if (axis.y < 0.f) { axis *= -1.f; }
			float height = length(axis);
			axis *= (1.f / height);
			quat rotation = rotateFromTo(vec3(0.f, 1.f, 0.f), axis);
			mat3 rot = quaternionToMat3(rotation);
			result.mass = cylinderVolume(cyl) * c.material.density;
			result.cog = (cyl.positionA + cyl.positionB) * 0.5f;
			float sqRadius = cyl.radius * cyl.radius;
			float sqHeight = height * height;
			mat3 I;
			I.m11 = sqRadius * result.mass * 0.5f;
			I.m00 = I.m22 = 1.f / 12.f * result.mass * (3.f * sqRadius + sqHeight);
			result.inertia = transpose(rot) * I * rot;
		} break;
		case collider_type_aabb:
		{
			bounding_box b = c.aabb();
			result.mass = b.volume() * c.material.density;
			result.cog = b.getCenter();
			vec3 diameter = b.getRadius() * 2.f;
			result.inertia = mat3::zero();
			result.inertia.m00 = 1.f / 12.f * result.mass * (diameter.y * diameter.y + diameter.z * diameter.z);
			result.inertia.m11 = 1.f / 12.f * result.mass * (diameter.x * diameter.x + diameter.z * diameter.z);
			result.inertia.m22 = 1.f / 12.f * result.mass * (diameter.x * diameter.x + diameter.y * diameter.y);
		} break;
		case collider_type_obb:
		{
			bounding_oriented_box o = c.obb();
			result.mass = o.volume() * c.material.density;
			result.cog = o.center;
			vec3 diameter = o.radius * 2.f;
			mat3 I = mat3::zero();
			I.m00 = 1.f / 12.f * result.mass * (diameter.y * diameter.y + diameter.z * diameter.z);
			I.m11 = 1.f / 12.f * result.mass * (diameter.x * diameter.x + diameter.z * diameter.z);
			I.m22 = 1.f / 12.f * result.mass * (diameter.x * diameter.x + diameter.y * diameter.y);
			mat3 rot = quaternionToMat3(o.rotation);
			result.inertia = transpose(rot) * I * rot;
		} break;
		default: result.mass = 0.f; result.inertia = mat3::zero(); break;
	}
	return result;
}

// rigid_body.cpp:29-81
static void recalculateProperties(world& w, body& rb)
{
	if (rb.invMass == 0.f) { return; } // kinematic
	u32 numColliders = (u32)rb.colliders.size();
	if (!numColliders) { return; }
	std::vector<physics_properties> properties(numColliders);
	for (u32 i = 0; i < numColliders; ++i) // newest first, like the reference's intrusive list
	{
		properties[i] = calculatePhysicsProperties(w.colliders[rb.colliders[numColliders - 1 - i]].local, w.hullGeometries);
	}
	mat3 inertia = mat3::zero();
	vec3 cog(0.f);
	float mass = 0.f;
	for (u32 i = 0; i < numColliders; ++i) { mass += properties[i].mass; cog += properties[i].cog * properties[i].mass; }
	rb.invMass = 1.f / mass;
	rb.localCOGPosition = cog = cog * rb.invMass;
	for (u32 i = 0; i < numColliders; ++i)
	{
		vec3 r = properties[i].cog - cog;
		inertia += properties[i].inertia + (mat3::identity() * dot(r, r) - outerProduct(r, r)) * properties[i].mass;
	}
	rb.invInertia = invert(inertia);
}

// ---------------------------------------------------------------------------------------------------
// getWorldSpaceColliders — physics.cpp:631-756
// ---------------------------------------------------------------------------------------------------
static void getWorldSpaceColliders(world& w)
{
	u32 n = (u32)w.colliders.size();
	u32 dummyRigidBodyIndex = (u32)w.bodies.size();
	w.worldSpaceAABBs.resize(n);
	w.worldSpaceColliders.resize(n);
	for (u32 i = 0; i < n; ++i)
	{
		const collider& c = w.colliders[i];
		bounding_box& bb = w.worldSpaceAABBs[i];
		collider_union& col = w.worldSpaceColliders[i];
		const trs& transform = (c.parent != STATIC_BODY) ? w.bodies[c.parent].transform1 : c.staticTransform;
		col = c.local;
		if (c.parent != STATIC_BODY) { col.objectIndex = c.parent; col.objectType = physics_object_type_rigid_body; }
		else if (c.zoneType != physics_object_type_static_collider) { col.objectIndex = c.zoneIndex; col.objectType = c.zoneType; } // physics.cpp:657-666
		else { col.objectIndex = dummyRigidBodyIndex; col.objectType = physics_object_type_static_collider; }
		if (c.parent != STATIC_BODY && (w.bodies[c.parent].removed || (c.parent < w.simOff.size() && w.simOff[c.parent])))
		{
			// deleted entity: the reference takes the collider out of the sweep (collision_broad.cpp:42-75); keeping collider indices
			// stable, it is parked where it can overlap nothing instead
			bb = bounding_box::fromCenterRadius(vec3(1.0e7f + 16.f * (float)i, -1.0e7f, 0.f), 0.25f);
			continue;
		}
		switch (c.local.type)
		{
			case collider_type_sphere:
			{
				bounding_sphere s = c.local.sphere();
				vec3 center = transform.position + transform.rotation * s.center;
				bb = bounding_box::fromCenterRadius(center, s.radius);
				col.set(bounding_sphere{ center, s.radius });
			} break;
			case collider_type_capsule:
			{
				bounding_capsule cap = c.local.capsule();
				vec3 posA = transform.rotation * cap.positionA + transform.position;
				vec3 posB = transform.rotation * cap.positionB + transform.position;
				vec3 radius3(cap.radius);
				bb = bounding_box::negativeInfinity();
				bb.grow(posA + radius3); bb.grow(posA - radius3); bb.grow(posB + radius3); bb.grow(posB - radius3);
				col.set(bounding_capsule{ posA, posB, cap.radius });
			} break;
			case collider_type_cylinder:
			{
				bounding_cylinder cyl = c.local.cylinder();
				vec3 posA = transform.rotation * cyl.positionA + transform.position;
				vec3 posB = transform.rotation * cyl.positionB + transform.position;
				vec3 a = posB - posA;
				float aa = dot(a, a);
				float x = 1.f - a.x * a.x / aa, y = 1.f - a.y * a.y / aa, z = 1.f - a.z * a.z / aa;
				x = sqrtf(std::max(0.f, x)); y = sqrtf(std::max(0.f, y)); z = sqrtf(std::max(0.f, z));
				vec3 e = cyl.radius * vec3(x, y, z);
				bb = bounding_box::fromMinMax(vmin(posA - e, posB - e), vmax(posA + e, posB + e));
				col.set(bounding_capsule{ posA, posB, cyl.radius });
			} break;
			case collider_type_aabb:
			{
				bounding_box b = c.local.aabb();
				bb = b.transformToAABB(transform.rotation, transform.position);
				if (transform.rotation == quat(0.f, 0.f, 0.f, 1.f)) { col.set(bb); }
				else { col.type = collider_type_obb; col.set(b.transformToOBB(transform.rotation, transform.position)); }
			} break;
			case collider_type_obb:
			{
				bounding_oriented_box o = c.local.obb();
				bb = o.transformToAABB(transform.rotation, transform.position);
				col.set(o.transformToOBB(transform.rotation, transform.position));
			} break;
			case collider_type_hull: // physics.cpp:742-753
			{
				bounding_hull h = c.local.hull();
				const bounding_hull_geometry& geometry = w.hullGeometries[h.geometryIndex];
				quat rotation = transform.rotation * h.rotation;
				vec3 position = transform.rotation * h.position + transform.position;
				bb = geometry.aabb.transformToAABB(rotation, position);
				col.set(bounding_hull{ rotation, position, h.geometryIndex });
			} break;
			default: break;
		}
	}
}

// ---------------------------------------------------------------------------------------------------
// broadphase — collision_broad.cpp:297-447 with determineOverlapsScalar (:87-166)
// ---------------------------------------------------------------------------------------------------
static void broadphase(world& w)
{
	u32 numColliders = (u32)w.colliders.size();
	w.broadphasePairs.clear();
	if (numColliders == 0) { return; }
	std::vector<sap_endpoint>& endpoints = w.endpoints;
	u32 numEndpoints = numColliders * 2;

	vec3 s(0.f), s2(0.f);
	u32 numCounted = 0;
	u32 sortingAxis = w.sortingAxis;
	w.usedSortingAxis = sortingAxis;
	{
		// the reference walks indirections (collider -> endpoint slots); equivalent here: scan endpoints.
		for (u32 e = 0; e < numEndpoints; ++e)
		{
			sap_endpoint& ep = endpoints[e];
			const bounding_box& aabb = w.worldSpaceAABBs[ep.collider];
			ep.value = ep.start ? aabb.minCorner[sortingAxis] : aabb.maxCorner[sortingAxis];
		}
		for (u32 i = 0; i < numColliders; ++i)
		{
			const u32 parent = w.colliders[i].parent;
			if (parent != STATIC_BODY && (w.bodies[parent].removed || (parent < w.simOff.size() && w.simOff[parent]))) { continue; } // (parked: not a collider of this world's sweep; the reference removes a deleted entity's endpoints, collision_broad.cpp:42-75)
			vec3 center = w.worldSpaceAABBs[i].getCenter();
			s += center;
			s2 += center * center;
			++numCounted;
		}
	}
	for (u32 i = 1; i < numEndpoints; ++i) // insertion sort (:387-398), stable, strict >
	{
		sap_endpoint key = endpoints[i];
		u32 j = i - 1;
		while (j != UINT32_MAX && endpoints[j].value > key.value) { endpoints[j + 1] = endpoints[j]; j = j - 1; }
		endpoints[j + 1] = key;
	}

	if (w.wideBroadphase) // determineOverlapsSIMD (:168-295): the active boxes in SoA blocks of 8, one start endpoint against 8 of them at a time (physics_settings::simdBroadPhase)
	{
		const u32 W = 8;
		struct soa_bounding_box { float minX[8], minY[8], minZ[8], maxX[8], maxY[8], maxZ[8]; };
		std::vector<u32> activeList((numColliders + W - 1) / W * W), positionInActiveList(numColliders);
		std::vector<soa_bounding_box> activeBBs((numColliders + W - 1) / W);
		u32 numActive = 0;
		for (u32 i = 0; i < numEndpoints; ++i)
		{
			sap_endpoint ep = endpoints[i];
			if (ep.start)
			{
				const bounding_box& a = w.worldSpaceAABBs[ep.collider];
				const u32 count = (numActive + W - 1) / W;
				for (u32 active = 0; active < count; ++active)
				{
					const soa_bounding_box& b = activeBBs[active];
					const u32 numValidLanes = std::min(numActive - active * W, W);
					u32 mask = 0;
					for (u32 k = 0; k < W; ++k) // aabbVsAABB on 8 lanes (bounding_volumes_simd.h:59-66): inclusive compares, combined with &
						mask |= (u32)((a.maxCorner.x >= b.minX[k]) & (a.minCorner.x <= b.maxX[k]) & (a.maxCorner.y >= b.minY[k]) & (a.minCorner.y <= b.maxY[k]) & (a.maxCorner.z >= b.minZ[k]) & (a.minCorner.z <= b.maxZ[k])) << k;
					mask &= (1u << numValidLanes) - 1u;
					for (u32 k = 0; k < W; ++k) if (mask & (1u << k)) w.broadphasePairs.push_back({ ep.collider, activeList[active * W + k] });
				}
				positionInActiveList[ep.collider] = numActive;
				soa_bounding_box& out = activeBBs[numActive / W]; const u32 slot = numActive % W;
				out.minX[slot] = a.minCorner.x; out.minY[slot] = a.minCorner.y; out.minZ[slot] = a.minCorner.z; out.maxX[slot] = a.maxCorner.x; out.maxY[slot] = a.maxCorner.y; out.maxZ[slot] = a.maxCorner.z;
				activeList[numActive++] = ep.collider;
			}
			else
			{
				const u32 pos = positionInActiveList[ep.collider];
				--numActive;
				const u32 last = activeList[numActive];
				positionInActiveList[last] = pos;
				activeList[pos] = activeList[numActive];
				soa_bounding_box& out = activeBBs[pos / W]; const u32 slot = pos % W;
				const soa_bounding_box& from = activeBBs[numActive / W]; const u32 fromSlot = numActive % W;
				out.minX[slot] = from.minX[fromSlot]; out.minY[slot] = from.minY[fromSlot]; out.minZ[slot] = from.minZ[fromSlot]; out.maxX[slot] = from.maxX[fromSlot]; out.maxY[slot] = from.maxY[fromSlot]; out.maxZ[slot] = from.maxZ[fromSlot];
			}
		}
	}
	else
	{
	// determineOverlapsScalar (:87-166)
	std::vector<u32> activeList(numColliders), positionInActiveList(numColliders);
	std::vector<bounding_box> activeBBs(numColliders);
	u32 numActive = 0;
	for (u32 i = 0; i < numEndpoints; ++i)
	{
		sap_endpoint ep = endpoints[i];
		if (ep.start)
		{
			const bounding_box& a = w.worldSpaceAABBs[ep.collider];
			for (u32 active = 0; active < numActive; ++active)
			{
				if (aabbVsAABB(a, activeBBs[active])) { w.broadphasePairs.push_back({ ep.collider, activeList[active] }); }
			}
			positionInActiveList[ep.collider] = numActive;
			activeBBs[numActive] = a;
			activeList[numActive++] = ep.collider;
		}
		else
		{
			u32 pos = positionInActiveList[ep.collider];
			--numActive;
			u32 last = activeList[numActive];
			positionInActiveList[last] = pos;
			activeList[pos] = activeList[numActive];
			activeBBs[pos] = activeBBs[numActive];
		}
	}
	}
	vec3 variance = s2 - s * s / (float)(numCounted ? numCounted : 1u);
	w.lastVariance[0] = variance.x; w.lastVariance[1] = variance.y; w.lastVariance[2] = variance.z;
	w.sortingAxis = (variance.x > variance.y) ? ((variance.x > variance.z) ? 0 : 2) : ((variance.y > variance.z) ? 1 : 2);
}

// ---------------------------------------------------------------------------------------------------
// narrowphase — collision_narrow.cpp:2328-2603 (collision pairs only; triggers/force fields are row N2)
// ---------------------------------------------------------------------------------------------------
static u32 packFrictionRestitution(const collider_union& A, const collider_union& B) // :2231-2237
{
	float friction = clamp01(sqrtf(A.material.friction * B.material.friction));
	float restitution = clamp01(std::max(A.material.restitution, B.material.restitution));
	return ((u32)(friction * 0xFFFF) << 16) | (u32)(restitution * 0xFFFF);
}

static u32 packTerrainFrictionRestitution(const collider_union& A, const heightmap& hm) // heightmap_collision.cpp:588-591
{
	float friction = clamp01(sqrtf(A.material.friction * hm.material.friction));
	float restitution = clamp01(std::max(A.material.restitution, hm.material.restitution));
	return ((u32)(friction * 0xFFFF) << 16) | (u32)(restitution * 0xFFFF);
}
static const u32 TERRAIN_SLOT = 0x80000000u; // follow mode: slot (collider, TERRAIN_SLOT | k) = the collider's k-th terrain contact in the device's emission order

// heightmapCollision (heightmap_collision.cpp:522-618): every rigid-body collider against the terrain, contacts appended after the narrowphase's
static void heightmapCollision(world& w)
{
	for (u32 i = 0; i < (u32)w.worldSpaceColliders.size(); ++i)
	{
		const collider_union& collider = w.worldSpaceColliders[i];
		if (collider.objectType != physics_object_type_rigid_body) { continue; }
		if (w.bodies[collider.objectIndex].removed) { continue; }
		std::vector<terrain_contact> found;
		heightmapContacts(w.terrain, collider, w.worldSpaceAABBs[i], found);
		if (found.empty()) { continue; }
		u32 fr = packTerrainFrictionRestitution(collider, w.terrain);
		u32 collisionIndex = (u32)w.collidingPairs.size();
		w.collidingPairs.push_back(collider_pair{ i, 0xFFFFFFFFu });
		w.contactCountPerCollision.push_back((u8)std::min<size_t>(found.size(), 255));
		for (const terrain_contact& t : found)
		{
			collision_contact c = t.contact; c.friction_restitution = fr;
			w.contacts.push_back(c);
			w.contactBodyPairs.push_back({ collider.objectIndex, (u32)w.bodies.size() });
			w.contactCollisionIndex.push_back(collisionIndex);
		}
	}
}
// the device's emission order: chunk z, chunk x, cell z, cell x, triangle; the lowest-point contact last
static void sortTerrainContactsDeviceOrder(std::vector<terrain_contact>& v)
{
	std::stable_sort(v.begin(), v.end(), [](const terrain_contact& l, const terrain_contact& r) { for (int i = 0; i < 5; ++i) if (l.key[i] != r.key[i]) return l.key[i] < r.key[i]; return false; });
}

static void narrowphaseOverride(world& w)
{
	const collider_union* cols = w.worldSpaceColliders.data();
	w.contacts.clear(); w.contactBodyPairs.clear(); w.collidingPairs.clear(); w.contactCountPerCollision.clear(); w.contactCollisionIndex.clear();
	w.slotCounts.assign(w.pairOverride.size(), 0);
	std::vector<u32> slotStart(w.pairOverride.size(), 0);
	std::map<u32, std::vector<terrain_contact>> terrainCache; std::map<u32, u32> terrainSeen;
	for (size_t s = 0; s < w.pairOverride.size(); ++s)
	{
		collider_pair pair = w.pairOverride[s];
		slotStart[s] = (u32)w.contacts.size();
		if (pair.colliderB & TERRAIN_SLOT) // one terrain contact of collider A
		{
			if (!w.hasTerrain) { continue; }
			auto it = terrainCache.find(pair.colliderA);
			if (it == terrainCache.end())
			{
				std::vector<terrain_contact> found;
				if (cols[pair.colliderA].objectType == physics_object_type_rigid_body) { heightmapContacts(w.terrain, cols[pair.colliderA], w.worldSpaceAABBs[pair.colliderA], found); }
				sortTerrainContactsDeviceOrder(found);
				it = terrainCache.emplace(pair.colliderA, std::move(found)).first;
			}
			u32 k = pair.colliderB & ~TERRAIN_SLOT;
			terrainSeen[pair.colliderA]++;
			if (k >= it->second.size()) { continue; }
			const collider_union& A = cols[pair.colliderA];
			collision_contact c = it->second[k].contact; c.friction_restitution = packTerrainFrictionRestitution(A, w.terrain);
			u32 collisionIndex = (u32)w.collidingPairs.size();
			w.collidingPairs.push_back(pair);
			w.contactCountPerCollision.push_back(1);
			w.slotCounts[s] = 1;
			w.contacts.push_back(c);
			w.contactBodyPairs.push_back({ A.objectIndex, (u32)w.bodies.size() });
			w.contactCollisionIndex.push_back(collisionIndex);
			continue;
		}
		const collider_union& A = cols[pair.colliderA];
		const collider_union& B = cols[pair.colliderB];
		contact_manifold contact; contact.numContacts = 0;
		if (intersectColliders(A, B, contact))
		{
			u32 fr = packFrictionRestitution(A, B);
			u32 collisionIndex = (u32)w.collidingPairs.size();
			w.collidingPairs.push_back(pair);
			w.contactCountPerCollision.push_back((u8)contact.numContacts);
			w.slotCounts[s] = (u8)contact.numContacts;
			for (u32 k = 0; k < contact.numContacts; ++k)
			{
				collision_contact c;
				c.normal = contact.collisionNormal; c.penetrationDepth = contact.contacts[k].penetrationDepth; c.point = contact.contacts[k].point; c.friction_restitution = fr;
				w.contacts.push_back(c);
				w.contactBodyPairs.push_back({ A.objectIndex, B.objectIndex });
				w.contactCollisionIndex.push_back(collisionIndex);
			}
		}
	}
	// terrain contacts the device did not report: the oracle's own sweep over all colliders must find exactly the slots it was given
	w.terrainSlotMismatch = 0;
	if (w.hasTerrain)
	{
		for (u32 i = 0; i < (u32)w.worldSpaceColliders.size(); ++i)
		{
			if (cols[i].objectType != physics_object_type_rigid_body || w.bodies[cols[i].objectIndex].removed) { continue; }
			std::vector<terrain_contact> found;
			auto it = terrainCache.find(i);
			if (it == terrainCache.end()) { heightmapContacts(w.terrain, cols[i], w.worldSpaceAABBs[i], found); }
			size_t expected = (it == terrainCache.end()) ? found.size() : it->second.size();
			if (expected != terrainSeen[i]) { w.terrainSlotMismatch++; }
		}
	}
	w.customOrder.clear();
	for (u32 s : w.slotOrder) { if (s < slotStart.size()) for (u32 k = 0; k < w.slotCounts[s]; ++k) w.customOrder.push_back(slotStart[s] + k); }
}

static void narrowphasePairs(world& w, const collider_pair* inPairs, u32 numPairs)
{
	const collider_union* cols = w.worldSpaceColliders.data();
	std::vector<collider_pair> kept;
	kept.reserve(numPairs);
	u32 countMatrix[collider_type_count][collider_type_count] = {};
	for (u32 i = 0; i < numPairs; ++i) // prune, classify, count (:2346-2397)
	{
		collider_pair pair = inPairs[i];
		const collider_union* a = cols + pair.colliderA;
		const collider_union* b = cols + pair.colliderB;
		if (a->objectType != physics_object_type_rigid_body && b->objectType != physics_object_type_rigid_body) { continue; }
		if (a->objectType == physics_object_type_rigid_body && b->objectType == physics_object_type_rigid_body && a->objectIndex == b->objectIndex) { continue; }
		pair = (a->type < b->type) ? pair : collider_pair{ pair.colliderB, pair.colliderA }; // NB swaps on equal types (:2374)
		a = cols + pair.colliderA; b = cols + pair.colliderB;
		bool collides = (a->objectType == physics_object_type_rigid_body && b->objectType == physics_object_type_rigid_body)
			|| a->objectType == physics_object_type_static_collider || b->objectType == physics_object_type_static_collider; // :2378-2379
		if (!collides) { continue; } // force field / trigger vs rigid body: overlap check only (nonCollisionInteractions)
		++countMatrix[a->type][b->type];
		kept.push_back(pair);
	}
	u32 offsetMatrix[collider_type_count][collider_type_count] = {};
	u32 total = 0;
	for (u32 i = 0; i < collider_type_count; ++i) for (u32 j = i; j < collider_type_count; ++j) { offsetMatrix[i][j] = total; total += countMatrix[i][j]; }
	std::vector<collider_pair> sorted(kept.size());
	u32 writeMatrix[collider_type_count][collider_type_count] = {};
	for (const collider_pair& pair : kept) // stable bucket sort (:2431-2440)
	{
		u32 ta = cols[pair.colliderA].type, tb = cols[pair.colliderB].type;
		sorted[offsetMatrix[ta][tb] + writeMatrix[ta][tb]++] = pair;
	}

	w.contacts.clear(); w.contactBodyPairs.clear(); w.collidingPairs.clear(); w.contactCountPerCollision.clear(); w.contactCollisionIndex.clear();
	for (const collider_pair& pair : sorted) // dispatch in bucket order (:2473-2570), collisionScalar (:2285-2302)
	{
		const collider_union& A = cols[pair.colliderA];
		const collider_union& B = cols[pair.colliderB];
		contact_manifold contact;
		contact.numContacts = 0;
		if (intersectColliders(A, B, contact))
		{
			u32 fr = packFrictionRestitution(A, B); // writeScalarContact (:2221-2253)
			u32 collisionIndex = (u32)w.collidingPairs.size();
			w.collidingPairs.push_back(pair);
			w.contactCountPerCollision.push_back((u8)contact.numContacts);
			for (u32 k = 0; k < contact.numContacts; ++k)
			{
				collision_contact c;
				c.normal = contact.collisionNormal;
				c.penetrationDepth = contact.contacts[k].penetrationDepth;
				c.point = contact.contacts[k].point;
				c.friction_restitution = fr;
				w.contacts.push_back(c);
				w.contactBodyPairs.push_back({ A.objectIndex, B.objectIndex });
				w.contactCollisionIndex.push_back(collisionIndex);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------
// Non-collision interactions (force fields, triggers) and events — collision_narrow.cpp:2573-2593, physics.cpp:759-787, 952-1178.
// Two deliberate re-statements, shared with the device path: (1) a body inside several localized force fields receives their
// forces in ascending field id (the reference adds them in the bucket order of its pair list; the sums differ in the last ulp
// only when three or more terms meet); (2) entity handles are collider / trigger / body ids here, so "sorted by entity pair"
// means sorted by those ids.
// ---------------------------------------------------------------------------------------------------
static vec3 fieldForceWorld(const force_field& f) { return f.hasTransform ? (f.transform.rotation * f.force) : f.force; } // physics.cpp:767-771

static void handleNonCollisionInteractions(world& w)
{
	const collider_union* cols = w.worldSpaceColliders.data();
	u32 numFields = (u32)w.forceFields.size();
	std::vector<u8> inField((size_t)numFields * w.bodies.size(), 0);
	std::vector<entity_pair> triggerOverlaps;
	for (const collider_pair& in : w.broadphasePairs)
	{
		collider_pair pair = in;
		const collider_union* a = cols + pair.colliderA;
		const collider_union* b = cols + pair.colliderB;
		bool rbA = a->objectType == physics_object_type_rigid_body, rbB = b->objectType == physics_object_type_rigid_body;
		if (rbA == rbB) { continue; }                                   // one rigid body ...
		const collider_union* other = rbA ? b : a;
		if (other->objectType != physics_object_type_force_field && other->objectType != physics_object_type_trigger) { continue; } // ... and one zone
		pair = (a->type < b->type) ? pair : collider_pair{ pair.colliderB, pair.colliderA }; // :2374
		u32 typeKey = cols[pair.colliderA].type * 6 + cols[pair.colliderB].type;
		w.zoneTested[typeKey]++;
		if (!overlapColliders(cols[pair.colliderA], cols[pair.colliderB])) { continue; }
		w.zoneHit[typeKey]++;
		u32 rigidBodyIndex = (rbA ? a : b)->objectIndex;                 // overlapCheck, :1601-1612
		if (other->objectType == physics_object_type_force_field) { inField[(size_t)rigidBodyIndex * numFields + other->objectIndex] = 1; }
		else { triggerOverlaps.push_back(entity_pair{ other->objectIndex, rigidBodyIndex }); }
	}
	for (size_t b = 0; b < w.bodies.size(); ++b)
		for (u32 f = 0; f < numFields; ++f)
			if (inField[b * numFields + f]) { w.bodies[b].forceAccumulator += fieldForceWorld(w.forceFields[f]); } // physics.cpp:963-967

	std::sort(triggerOverlaps.begin(), triggerOverlaps.end());
	triggerOverlaps.erase(std::unique(triggerOverlaps.begin(), triggerOverlaps.end()), triggerOverlaps.end()); // several colliders may report the same overlap
	auto triggerEvent = [&w](entity_pair pair, u32 kind)
	{
		event_record e; memset(&e, 0, sizeof(e));
		e.kind = kind; e.step = w.stepIndex; e.a = pair.a; e.b = pair.b; e.bodyA = STATIC_BODY; e.bodyB = pair.b;
		w.events.push_back(e);
	};
	auto prevIt = w.prevFrameTriggerOverlaps.begin(), prevEnd = w.prevFrameTriggerOverlaps.end();
	auto thisIt = triggerOverlaps.begin(), thisEnd = triggerOverlaps.end();
	while (prevIt != prevEnd && thisIt != thisEnd) // physics.cpp:1000-1022
	{
		if (*prevIt == *thisIt) { ++prevIt; ++thisIt; continue; }
		if (*prevIt < *thisIt) { triggerEvent(*prevIt++, event_trigger_leave); }
		else { triggerEvent(*thisIt++, event_trigger_enter); }
	}
	while (prevIt != prevEnd) { triggerEvent(*prevIt++, event_trigger_leave); }
	while (thisIt != thisEnd) { triggerEvent(*thisIt++, event_trigger_enter); }
	w.prevFrameTriggerOverlaps = std::move(triggerOverlaps);
}

static bool globalForceField(const world& w, vec3& sum) // getForceFieldStates, physics.cpp:759-787: fields without a collider act everywhere; EnTT walks newest first
{
	bool any = false;
	sum = vec3(0.f);
	for (size_t i = w.forceFields.size(); i-- > 0;) { if (!w.forceFields[i].numColliders) { sum += fieldForceWorld(w.forceFields[i]); any = true; } }
	return any;
}

static void handleCollisionCallbacks(world& w) // physics.cpp:1037-1178; runs after the force integration: velocities are the pre-solve ones
{
	std::vector<collision_entity_pair> collisions;
	u32 contactOffset = 0;
	for (size_t i = 0; i < w.collidingPairs.size(); ++i)
	{
		u32 numContacts = w.contactCountPerCollision[i];
		collisions.push_back(collision_entity_pair{ w.collidingPairs[i].colliderA, w.collidingPairs[i].colliderB, contactOffset, numContacts });
		contactOffset += numContacts;
	}
	std::sort(collisions.begin(), collisions.end());
	if (w.collisionBeginEvents || w.collisionEndEvents)
	{
		auto bodyOf = [&w](u32 colliderIndex) { return w.colliders[colliderIndex].parent; };
		auto beginEvent = [&](const collision_entity_pair& pair)
		{
			if (!w.collisionBeginEvents) { return; }
			const collision_contact* c = w.contacts.data() + pair.contactOffset;
			float norm = 1.f / pair.numContacts;
			vec3 point(0.f), normal(0.f);
			for (u32 i = 0; i < pair.numContacts; ++i) { point += c[i].point; normal += c[i].normal; }
			point *= norm; normal *= norm;
			u32 dummy = (u32)w.bodies.size();
			u32 bodyA = bodyOf(pair.a), bodyB = bodyOf(pair.b);
			const rigid_body_global_state& rbA = w.rbGlobal[bodyA == STATIC_BODY ? dummy : bodyA];
			const rigid_body_global_state& rbB = w.rbGlobal[bodyB == STATIC_BODY ? dummy : bodyB];
			vec3 velA = rbA.linearVelocity + cross(rbA.angularVelocity, point - rbA.position);
			vec3 velB = rbB.linearVelocity + cross(rbB.angularVelocity, point - rbB.position);
			vec3 rel = velB - velA;
			event_record e; memset(&e, 0, sizeof(e));
			e.kind = event_collision_begin; e.step = w.stepIndex; e.a = pair.a; e.b = pair.b; e.bodyA = bodyA; e.bodyB = bodyB;
			e.position[0] = point.x; e.position[1] = point.y; e.position[2] = point.z;
			e.normal[0] = normal.x; e.normal[1] = normal.y; e.normal[2] = normal.z;
			e.relativeVelocity[0] = rel.x; e.relativeVelocity[1] = rel.y; e.relativeVelocity[2] = rel.z;
			w.events.push_back(e);
		};
		auto endEvent = [&](const collision_entity_pair& pair)
		{
			if (!w.collisionEndEvents) { return; }
			event_record e; memset(&e, 0, sizeof(e));
			e.kind = event_collision_end; e.step = w.stepIndex; e.a = pair.a; e.b = pair.b; e.bodyA = bodyOf(pair.a); e.bodyB = bodyOf(pair.b);
			w.events.push_back(e);
		};
		auto prevIt = w.prevFrameCollisions.begin(), prevEnd = w.prevFrameCollisions.end();
		auto thisIt = collisions.begin(), thisEnd = collisions.end();
		while (prevIt != prevEnd && thisIt != thisEnd)
		{
			if (*prevIt == *thisIt) { ++prevIt; ++thisIt; continue; }
			if (*prevIt < *thisIt) { endEvent(*prevIt++); }
			else { beginEvent(*thisIt++); }
		}
		while (prevIt != prevEnd) { endEvent(*prevIt++); }
		while (thisIt != thisEnd) { beginEvent(*thisIt++); }
	}
	w.prevFrameCollisions = std::move(collisions);
}

// ---------------------------------------------------------------------------------------------------
// Integrators — rigid_body.cpp:95-142
// ---------------------------------------------------------------------------------------------------
static void applyGravityAndIntegrateForces(body& rb, rigid_body_global_state& global, const trs& transform, float dt)
{
	global.rotation = transform.rotation;
	global.position = transform.position + transform.rotation * rb.localCOGPosition;
	mat3 rot = quaternionToMat3(global.rotation);
	global.invInertia = rot * rb.invInertia * transpose(rot);
	global.invMass = rb.invMass;
	if (rb.invMass > 0.f) { rb.forceAccumulator.y += (GRAVITY / rb.invMass * rb.gravityFactor); }
	vec3 linearAcceleration = rb.forceAccumulator * rb.invMass;
	vec3 angularAcceleration = global.invInertia * rb.torqueAccumulator;
	rb.linearVelocity += linearAcceleration * dt;
	rb.angularVelocity += angularAcceleration * dt;
	rb.linearVelocity *= 1.f / (1.f + dt * rb.linearDamping);
	rb.angularVelocity *= 1.f / (1.f + dt * rb.angularDamping);
	global.linearVelocity = rb.linearVelocity;
	global.angularVelocity = rb.angularVelocity;
	global.localCOGPosition = rb.localCOGPosition;
}
static void integrateVelocity(body& rb, const rigid_body_global_state& global, trs& transform, float dt)
{
	rb.linearVelocity = global.linearVelocity;
	rb.angularVelocity = global.angularVelocity;
	quat deltaRot(0.5f * rb.angularVelocity.x, 0.5f * rb.angularVelocity.y, 0.5f * rb.angularVelocity.z, 0.f);
	deltaRot = deltaRot * global.rotation;
	quat rotation = normalize(global.rotation + (deltaRot * dt));
	vec3 position = global.position + rb.linearVelocity * dt;
	rb.forceAccumulator = vec3(0.f);
	rb.torqueAccumulator = vec3(0.f);
	transform.rotation = rotation;
	transform.position = position - rotation * rb.localCOGPosition;
}

// ---------------------------------------------------------------------------------------------------
// physicsStepInternal — physics.cpp:1180-1362
// ---------------------------------------------------------------------------------------------------
static void physicsStepInternal(world& w, u32 iterations, u32 mode, float dt)
{
	u32 numRigidBodies = (u32)w.bodies.size();
	if (numRigidBodies == 0) { return; }
	u32 dummyRigidBodyIndex = numRigidBodies;

	auto stageClock = std::chrono::steady_clock::now();
	auto stageEnd = [&](int stage) { auto now = std::chrono::steady_clock::now(); w.stageSeconds[stage] += std::chrono::duration<double>(now - stageClock).count(); stageClock = now; };
	getWorldSpaceColliders(w);
	broadphase(w);
	stageEnd(0);
	hullGeometryTable() = &w.hullGeometries;
	if (w.usePairOverride) { narrowphaseOverride(w); }
	else
	{
		narrowphasePairs(w, w.broadphasePairs.data(), (u32)w.broadphasePairs.size());
		if (w.hasTerrain) { heightmapCollision(w); }                               // :1236-1249
	}

	stageEnd(1);
	vec3 globalForce;
	bool anyGlobalForce = globalForceField(w, globalForce);                        // :1253 (a world without global fields skips the += 0 of :1273)
	if (!w.forceFields.empty() || !w.triggers.empty() || !w.prevFrameTriggerOverlaps.empty()) { handleNonCollisionInteractions(w); } // :1255

	w.rbGlobal.resize(numRigidBodies + 1);
	for (u32 i = 0; i < numRigidBodies; ++i)
	{
		if (i < w.simOff.size() && w.simOff[i]) { memset(&w.rbGlobal[i], 0, sizeof(rigid_body_global_state)); continue; } // simulated elsewhere: frozen here
		if (anyGlobalForce) { w.bodies[i].forceAccumulator += globalForce; }        // :1273
		applyGravityAndIntegrateForces(w.bodies[i], w.rbGlobal[i], w.bodies[i].transform1, dt);
	}
	memset(&w.rbGlobal[dummyRigidBodyIndex], 0, sizeof(rigid_body_global_state)); // :1279
	w.rbGlobalPreSolve = w.rbGlobal;
	handleCollisionCallbacks(w);                                                   // :1284
	++w.stepIndex;
	rigid_body_global_state* rbs = w.rbGlobal.data();

	// constraint_solver::initialize (constraints.cpp:3711-3746): joints scalar (joint SIMD variants schedule with
	// dummy = UINT16_MAX, see solveJointsInSlotOrder), contacts per `mode`.
	std::vector<distance_constraint_update> dU(w.distanceConstraints.size());
	std::vector<ball_constraint_update> bU(w.ballConstraints.size());
	std::vector<fixed_constraint_update> fU(w.fixedConstraints.size());
	std::vector<hinge_constraint_update> hU(w.hingeConstraints.size());
	std::vector<cone_twist_constraint_update> cU(w.coneTwistConstraints.size());
	std::vector<slider_constraint_update> sU(w.sliderConstraints.size());
	for (size_t i = 0; i < dU.size(); ++i) initializeDistanceConstraint(dU[i], rbs, w.distanceConstraints[i], w.distancePairs[i], dt);
	for (size_t i = 0; i < bU.size(); ++i) initializeBallConstraint(bU[i], rbs, w.ballConstraints[i], w.ballPairs[i], dt);
	for (size_t i = 0; i < fU.size(); ++i) initializeFixedConstraint(fU[i], rbs, w.fixedConstraints[i], w.fixedPairs[i], dt);
	for (size_t i = 0; i < hU.size(); ++i) initializeHingeConstraint(hU[i], rbs, w.hingeConstraints[i], w.hingePairs[i], dt);
	for (size_t i = 0; i < cU.size(); ++i) initializeConeTwistConstraint(cU[i], rbs, w.coneTwistConstraints[i], w.coneTwistPairs[i], dt);
	for (size_t i = 0; i < sU.size(); ++i) initializeSliderConstraint(sU[i], rbs, w.sliderConstraints[i], w.sliderPairs[i], dt);

	// Joint GS order: scalar = storage order; wide8 = scheduler output order with dummy UINT16_MAX (constraints.cpp:271,
	// 535, 830, 1314, 2077, 2853); duplicated padding lanes are idempotent re-solves of lane 0 and are skipped.
	auto jointOrder = [&](const std::vector<constraint_body_pair>& pairs)
	{
		std::vector<u32> order;
		if (mode == solver_wide8)
		{
			std::vector<sched_slot> slots;
			scheduleConstraintsSIMD(pairs.data(), (u32)pairs.size(), 0xFFFFu, slots);
			for (const sched_slot& s : slots)
			{
				for (u32 l = 0; l < SCHED_W; ++l) { if (l == 0 || s.indices[l] != s.indices[0]) order.push_back(s.indices[l]); }
			}
		}
		else { for (u32 i = 0; i < pairs.size(); ++i) order.push_back(i); }
		return order;
	};
	std::vector<u32> dO = jointOrder(w.distancePairs), bO = jointOrder(w.ballPairs), fO = jointOrder(w.fixedPairs);
	std::vector<u32> hO = jointOrder(w.hingePairs), cO = jointOrder(w.coneTwistPairs), sO = jointOrder(w.sliderPairs);
	if (w.usePairOverride) // follow mode: external joint orders (same size as the joint arrays)
	{
		std::vector<u32>* os[6] = { &dO, &bO, &fO, &hO, &cO, &sO };
		for (int t = 0; t < 6; ++t) if (w.jointOrder[t].size() == os[t]->size()) *os[t] = w.jointOrder[t];
	}

	u32 numContacts = (u32)w.contacts.size();
	std::vector<simd_collision_constraint_batch> batches;
	w.contactConstraints.clear(); w.contactSlots.clear();
	if (mode == solver_wide8)
	{
		scheduleConstraintsSIMD(w.contactBodyPairs.data(), numContacts, dummyRigidBodyIndex, w.contactSlots);
		initializeCollisionBatchesWide(batches, w.contactSlots, rbs, w.contacts.data(), w.contactBodyPairs.data(), dt);
	}
	else
	{
		w.contactConstraints.resize(numContacts);
		for (u32 i = 0; i < numContacts; ++i) initializeCollisionConstraint(w.contactConstraints[i], rbs, w.contacts[i], w.contactBodyPairs[i], dt);
		if (mode == solver_replay) scheduleConstraintsSIMD(w.contactBodyPairs.data(), numContacts, dummyRigidBodyIndex, w.contactSlots);
	}

	stageEnd(2);
	for (u32 it = 0; it < iterations; ++it) // solveOneIteration (constraints.cpp:3748-3772)
	{
		for (u32 i : dO) solveDistanceConstraint(dU[i], rbs);
		for (u32 i : bO) solveBallConstraint(bU[i], rbs);
		for (u32 i : fO) solveFixedConstraint(fU[i], rbs);
		for (u32 i : hO) solveHingeConstraint(hU[i], rbs);
		for (u32 i : cO) solveConeTwistConstraint(cU[i], rbs);
		for (u32 i : sO) solveSliderConstraint(sU[i], rbs);
		if (mode == solver_wide8) { solveCollisionBatchesWide(batches, rbs); }
		else if (mode == solver_replay) // the reference's batch order (scheduleConstraintsSIMD), each batch's constraints in the device's row-form arithmetic
		{
			for (const sched_slot& slot : w.contactSlots)
				for (u32 l = 0; l < SCHED_W; ++l)
				{
					u32 ci = slot.indices[l];
					bool padding = false; // (empty lanes duplicate lane 0: the SIMD code solves it again with identical operands)
					for (u32 m = 0; m < l; ++m) if (slot.indices[m] == ci) padding = true;
					if (!padding && ci < numContacts) solveCollisionConstraintRowForm(w.contactConstraints[ci], w.contacts[ci], w.contactBodyPairs[ci], rbs);
				}
		}
		else if (mode == solver_custom_order)
		{
			if (w.rowForm) { for (u32 ci : w.customOrder) { if (ci < numContacts) solveCollisionConstraintRowForm(w.contactConstraints[ci], w.contacts[ci], w.contactBodyPairs[ci], rbs); } }
			else for (u32 ci : w.customOrder) { if (ci < numContacts) solveCollisionConstraint(w.contactConstraints[ci], w.contacts[ci], w.contactBodyPairs[ci], rbs); }
		}
		else
		{
			if (w.scalarRowForm) { for (u32 i = 0; i < numContacts; ++i) solveCollisionConstraintRowForm(w.contactConstraints[i], w.contacts[i], w.contactBodyPairs[i], rbs); }
			else for (u32 i = 0; i < numContacts; ++i) solveCollisionConstraint(w.contactConstraints[i], w.contacts[i], w.contactBodyPairs[i], rbs);
		}
	}

	stageEnd(3);
	for (u32 i = 0; i < numRigidBodies; ++i) { if (i < w.simOff.size() && w.simOff[i]) continue; integrateVelocity(w.bodies[i], w.rbGlobal[i], w.bodies[i].transform1, dt); }
	stageEnd(4);

	for (cloth& c : w.cloths) // physics.cpp:1354-1358
	{
		c.applyWindForce(globalForce);
		c.simulate(w.clothIterations[0], w.clothIterations[1], w.clothIterations[2], dt, w.clothColourOrder);
	}
}

// physics.cpp:1364-1413
static void physicsStep(world& w, float& timer, const physics_settings& settings, u32 mode, float dt)
{
	if (settings.fixedFrameRate)
	{
		const float physicsFixedTimeStep = 1.f / (float)settings.frameRate;
		timer += dt;
		u32 physicsIterations = 0;
		if (timer >= physicsFixedTimeStep)
		{
			for (body& b : w.bodies) { b.transform0 = b.transform1; }
			w.clothIterations[0] = settings.numClothVelocityIterations; w.clothIterations[1] = settings.numClothPositionIterations; w.clothIterations[2] = settings.numClothDriftIterations;
			while (timer >= physicsFixedTimeStep && physicsIterations++ < settings.maxPhysicsIterationsPerFrame)
			{
				physicsStepInternal(w, settings.numRigidSolverIterations, mode, physicsFixedTimeStep);
				timer -= physicsFixedTimeStep;
			}
		}
		if (timer >= physicsFixedTimeStep) { timer = fmodf(timer, physicsFixedTimeStep); }
		float t = timer / physicsFixedTimeStep;
		for (body& b : w.bodies) // lerp(trs) (core/math.h:676-683): position lerp, rotation nlerp
		{
			b.transform.position = lerp(b.transform0.position, b.transform1.position, t);
			quat l = b.transform0.rotation, u = b.transform1.rotation;
			quat q(l.x + t * (u.x - l.x), l.y + t * (u.y - l.y), l.z + t * (u.z - l.z), l.w + t * (u.w - l.w));
			b.transform.rotation = normalize(q);
		}
	}
	else
	{
		w.clothIterations[0] = settings.numClothVelocityIterations; w.clothIterations[1] = settings.numClothPositionIterations; w.clothIterations[2] = settings.numClothDriftIterations;
		physicsStepInternal(w, settings.numRigidSolverIterations, mode, dt);
		for (body& b : w.bodies) { b.transform = b.transform1; }
	}
}

} // namespace orc

// =====================================================================================================
// C API (ctypes-friendly).  Mirrors include/mi_physics.h so the parity tests drive both the same way.
// =====================================================================================================
using namespace orc;
extern "C" {

world* orc_world_create() { return new world(); }
void orc_world_destroy(world* w) { delete w; }

// rigid_body.cpp:6-27 + scene.h:69-84
u32 orc_add_body(world* w, int kinematic, float gravityFactor, float linearDamping, float angularDamping, const float* pos, const float* rot)
{
	body b;
	if (kinematic) { b.invMass = 0.f; b.invInertia = mat3::zero(); }
	else { b.invMass = 1.f; b.invInertia = mat3::identity(); }
	b.gravityFactor = gravityFactor; b.linearDamping = linearDamping; b.angularDamping = angularDamping;
	b.transform.position = vec3(pos[0], pos[1], pos[2]);
	b.transform.rotation = quat(rot[0], rot[1], rot[2], rot[3]);
	b.transform0 = b.transform1 = b.transform;
	w->bodies.push_back(b);
	return (u32)w->bodies.size() - 1;
}

static u32 addColliderCommon(world* w, u32 parent, u32 type, const float* shape, const float* material, const float* pos, const float* rot)
{
	collider c;
	memset(&c.local, 0, sizeof(c.local));
	for (u32 i = 0; i < 10; ++i) c.local.shape[i] = shape[i];
	c.local.material = physics_material{ material[0], material[1], material[2] };
	c.local.type = type;
	c.parent = parent;
	c.staticTransform.position = pos ? vec3(pos[0], pos[1], pos[2]) : vec3(0.f);
	c.staticTransform.rotation = rot ? quat(rot[0], rot[1], rot[2], rot[3]) : quat(0.f, 0.f, 0.f, 1.f);
	u32 id = (u32)w->colliders.size();
	w->colliders.push_back(c);
	w->endpoints.push_back(sap_endpoint{ 0.f, id, true });   // addColliderToBroadphase (collision_broad.cpp:27-40)
	w->endpoints.push_back(sap_endpoint{ 0.f, id, false });
	if (parent != STATIC_BODY)
	{
		w->bodies[parent].colliders.push_back(id);
		recalculateProperties(*w, w->bodies[parent]); // scene.h:60-63
	}
	return id;
}
// bounding_hull_geometry::fromMesh (bounding_volumes.cpp:1394-1442) without the edge table (unused on this path);
// replaces allocateBoundingHullGeometry(meshFilepath) (physics.cpp:58-84), whose mesh loading belongs to the asset pipeline.
u32 orc_add_hull_geometry(world* w, const float* vertices3, u32 numVertices, const u32* triangles3, u32 numTriangles)
{
	bounding_hull_geometry g;
	g.aabb = bounding_box::negativeInfinity();
	for (u32 i = 0; i < numVertices; ++i) { vec3 v(vertices3[3 * i], vertices3[3 * i + 1], vertices3[3 * i + 2]); g.vertices.push_back(v); g.aabb.grow(v); }
	for (u32 i = 0; i < numTriangles; ++i) g.faces.push_back(bounding_hull_face{ triangles3[3 * i], triangles3[3 * i + 1], triangles3[3 * i + 2] });
	w->hullGeometries.push_back(g);
	return (u32)w->hullGeometries.size() - 1;
}
// makes this world's hull geometries the table the stage-level entry points (orc_narrowphase_ordered) read
void orc_use_hull_geometries(world* w) { hullGeometryTable() = &w->hullGeometries; }

u32 orc_add_collider(world* w, u32 bodyId, u32 type, const float* shape, const float* material) { return addColliderCommon(w, bodyId, type, shape, material, 0, 0); }
u32 orc_add_static_collider(world* w, u32 type, const float* shape, const float* material, const float* pos, const float* rot) { return addColliderCommon(w, STATIC_BODY, type, shape, material, pos, rot); }

// ---- force fields, triggers, events ----
static trs makeTrs(const float* pos, const float* rot)
{
	trs t; t.position = pos ? vec3(pos[0], pos[1], pos[2]) : vec3(0.f); t.rotation = rot ? quat(rot[0], rot[1], rot[2], rot[3]) : quat(0.f, 0.f, 0.f, 1.f);
	return t;
}
u32 orc_add_force_field(world* w, const float* force, const float* pos, const float* rot)
{
	force_field f; f.force = vec3(force[0], force[1], force[2]); f.hasTransform = (pos || rot); f.transform = makeTrs(pos, rot); f.numColliders = 0;
	w->forceFields.push_back(f);
	return (u32)w->forceFields.size() - 1;
}
int orc_set_force_field(world* w, u32 field, const float* force) { if (field >= w->forceFields.size()) return 1; w->forceFields[field].force = vec3(force[0], force[1], force[2]); return 0; }
u32 orc_add_trigger(world* w, const float* pos, const float* rot) { w->triggers.push_back(trigger{ makeTrs(pos, rot), 0 }); return (u32)w->triggers.size() - 1; }
static u32 addZoneCollider(world* w, u32 zoneType, u32 zoneIndex, const trs& transform, u32 type, const float* shape)
{
	const float material[3] = { 0.f, 0.f, 0.f };
	float pos[3] = { transform.position.x, transform.position.y, transform.position.z }, rot[4] = { transform.rotation.x, transform.rotation.y, transform.rotation.z, transform.rotation.w };
	u32 id = addColliderCommon(w, STATIC_BODY, type, shape, material, pos, rot);
	w->colliders[id].zoneType = zoneType; w->colliders[id].zoneIndex = zoneIndex;
	return id;
}
u32 orc_add_force_field_collider(world* w, u32 field, u32 type, const float* shape)
{
	if (field >= w->forceFields.size()) return 0xFFFFFFFFu;
	w->forceFields[field].numColliders++;
	return addZoneCollider(w, physics_object_type_force_field, field, w->forceFields[field].transform, type, shape);
}
u32 orc_add_trigger_collider(world* w, u32 trig, u32 type, const float* shape)
{
	if (trig >= w->triggers.size()) return 0xFFFFFFFFu;
	w->triggers[trig].numColliders++;
	return addZoneCollider(w, physics_object_type_trigger, trig, w->triggers[trig].transform, type, shape);
}
void orc_zone_pair_stats(world* w, u32* outTested36, u32* outHit36) { memcpy(outTested36, w->zoneTested, sizeof(w->zoneTested)); memcpy(outHit36, w->zoneHit, sizeof(w->zoneHit)); }
// overlapCheck on an ordered pair list (typeA <= typeB per pair), one flag per pair; the boolean twin of orc_narrowphase_ordered
void orc_overlap_ordered(const void* colliders64, const u32* pairs2, u32 numPairs, u8* outOverlaps)
{
	const collider_union* cols = (const collider_union*)colliders64;
	for (u32 i = 0; i < numPairs; ++i) outOverlaps[i] = overlapColliders(cols[pairs2[2 * i]], cols[pairs2[2 * i + 1]]) ? 1 : 0;
}
// ---- heightmap terrain (heightmap_collider.h:127-152) ----
void orc_set_heightmap(world* w, u32 chunksPerDim, float chunkSize, const float* material, const float* minCorner, float amplitudeScale)
{
	w->terrain.create(chunksPerDim, chunkSize, physics_material{ material[0], material[1], material[2] });
	w->terrain.update(vec3(minCorner[0], minCorner[1], minCorner[2]), amplitudeScale);
	w->hasTerrain = true;
}
int orc_heightmap_set_chunk(world* w, u32 x, u32 z, const u16* heights)
{
	if (!w->hasTerrain || x >= w->terrain.chunksPerDim || z >= w->terrain.chunksPerDim) return 1;
	w->terrain.chunks[z * w->terrain.chunksPerDim + x].setHeights(heights);
	return 0;
}
void orc_heightmap_update(world* w, const float* minCorner, float amplitudeScale) { w->terrain.update(vec3(minCorner[0], minCorner[1], minCorner[2]), amplitudeScale); }
float orc_heightmap_height_at(world* w, float x, float z) { return w->hasTerrain ? w->terrain.getHeightAt(x, z) : -FLT_MAX; }
u32 orc_terrain_slot_mismatch(world* w) { return w->terrainSlotMismatch; }
// terrain contacts of one collider at the poses of the last step, in the device's emission order: {point3, depth, normal3, key0} per contact
u32 orc_terrain_contacts(world* w, u32 colliderIndex, float* out8, u32 capacity)
{
	if (!w->hasTerrain || colliderIndex >= w->worldSpaceColliders.size()) return 0;
	std::vector<terrain_contact> found;
	heightmapContacts(w->terrain, w->worldSpaceColliders[colliderIndex], w->worldSpaceAABBs[colliderIndex], found);
	sortTerrainContactsDeviceOrder(found);
	u32 n = std::min<u32>((u32)found.size(), capacity);
	for (u32 i = 0; i < n; ++i)
	{
		const collision_contact& c = found[i].contact;
		float* o = out8 + 8 * i;
		o[0] = c.point.x; o[1] = c.point.y; o[2] = c.point.z; o[3] = c.penetrationDepth; o[4] = c.normal.x; o[5] = c.normal.y; o[6] = c.normal.z; o[7] = (found[i].key[0] == ~0u) ? 1.f : 0.f;
	}
	return (u32)found.size();
}

// ---- cloth (cloth.h:5-60) ----
u32 orc_add_cloth(world* w, float width, float height, u32 gridSizeX, u32 gridSizeY, float totalMass, float stiffness, float damping, float gravityFactor)
{
	w->cloths.emplace_back(width, height, gridSizeX, gridSizeY, totalMass, stiffness, damping, gravityFactor);
	return (u32)w->cloths.size() - 1;
}
int orc_cloth_set_fixed_vertices(world* w, u32 c, const float* pos, const float* rot, int moveRigid)
{
	if (c >= w->cloths.size()) return 1;
	w->cloths[c].setWorldPositionOfFixedVertices(makeTrs(pos, rot), moveRigid != 0);
	return 0;
}
int orc_cloth_set_properties(world* w, u32 c, float totalMass, float stiffness, float damping, float gravityFactor)
{
	if (c >= w->cloths.size()) return 1;
	cloth& cl = w->cloths[c]; cl.totalMass = totalMass; cl.stiffness = stiffness; cl.damping = damping; cl.gravityFactor = gravityFactor;
	return 0;
}
void orc_set_cloth_iterations(world* w, u32 velocityIterations, u32 positionIterations, u32 driftIterations) { w->clothIterations[0] = velocityIterations; w->clothIterations[1] = positionIterations; w->clothIterations[2] = driftIterations; }
void orc_set_cloth_colour_order(world* w, int on) { w->clothColourOrder = on != 0; }
u32 orc_cloth_num_particles(world* w, u32 c) { return c < w->cloths.size() ? w->cloths[c].gridSizeX * w->cloths[c].gridSizeY : 0; }
u32 orc_cloth_num_constraints(world* w, u32 c) { return c < w->cloths.size() ? (u32)w->cloths[c].constraints.size() : 0; }
int orc_cloth_read(world* w, u32 c, float* positions3, float* velocities3)
{
	if (c >= w->cloths.size()) return 1;
	const cloth& cl = w->cloths[c];
	for (size_t i = 0; i < cl.positions.size(); ++i)
	{
		if (positions3) { positions3[3 * i] = cl.positions[i].x; positions3[3 * i + 1] = cl.positions[i].y; positions3[3 * i + 2] = cl.positions[i].z; }
		if (velocities3) { velocities3[3 * i] = cl.velocities[i].x; velocities3[3 * i + 1] = cl.velocities[i].y; velocities3[3 * i + 2] = cl.velocities[i].z; }
	}
	return 0;
}
void orc_cloth_read_constraints(world* w, u32 c, void* out16, u32* outColour) // {a, b, restDistance, inverseMassSum} + colour per constraint, storage order
{
	const cloth& cl = w->cloths[c];
	memcpy(out16, cl.constraints.data(), sizeof(cloth_constraint) * cl.constraints.size());
	memcpy(outColour, cl.colour.data(), sizeof(u32) * cl.colour.size());
}
static void moveZoneColliders(world* w, u32 zoneType, u32 zoneIndex, const trs& t) { for (collider& c : w->colliders) if (c.zoneType == zoneType && c.zoneIndex == zoneIndex && c.parent == STATIC_BODY) c.staticTransform = t; }
int orc_set_force_field_transform(world* w, u32 field, const float* pos, const float* rot)
{
	if (field >= w->forceFields.size()) return 1;
	w->forceFields[field].hasTransform = true; w->forceFields[field].transform = makeTrs(pos, rot);
	moveZoneColliders(w, physics_object_type_force_field, field, w->forceFields[field].transform);
	return 0;
}
int orc_set_trigger_transform(world* w, u32 trig, const float* pos, const float* rot)
{
	if (trig >= w->triggers.size()) return 1;
	w->triggers[trig].transform = makeTrs(pos, rot);
	moveZoneColliders(w, physics_object_type_trigger, trig, w->triggers[trig].transform);
	return 0;
}
void orc_enable_collision_events(world* w, int begin, int end) { w->collisionBeginEvents = begin != 0; w->collisionEndEvents = end != 0; }
u32 orc_drain_events(world* w, void* out60, u32 capacity)
{
	u32 n = std::min<u32>((u32)w->events.size(), capacity);
	if (n) memcpy(out60, w->events.data(), sizeof(event_record) * n);
	w->events.erase(w->events.begin(), w->events.begin() + n);
	return n;
}

static vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }

// physics.cpp:128-333
u32 orc_add_distance_constraint_local(world* w, u32 a, u32 b, const float* la, const float* lb, float distance)
{
	w->distanceConstraints.push_back(distance_constraint{ v3(la), v3(lb), distance });
	w->distancePairs.push_back({ a, b });
	return (u32)w->distanceConstraints.size() - 1;
}
u32 orc_add_distance_constraint_global(world* w, u32 a, u32 b, const float* ga, const float* gb)
{
	vec3 la = inverseTransformPosition(w->bodies[a].transform, v3(ga));
	vec3 lb = inverseTransformPosition(w->bodies[b].transform, v3(gb));
	float distance = length(v3(ga) - v3(gb));
	return orc_add_distance_constraint_local(w, a, b, &la.x, &lb.x, distance);
}
u32 orc_add_ball_constraint_local(world* w, u32 a, u32 b, const float* la, const float* lb)
{
	w->ballConstraints.push_back(ball_constraint{ v3(la), v3(lb) });
	w->ballPairs.push_back({ a, b });
	return (u32)w->ballConstraints.size() - 1;
}
u32 orc_add_ball_constraint_global(world* w, u32 a, u32 b, const float* g)
{
	vec3 la = inverseTransformPosition(w->bodies[a].transform, v3(g));
	vec3 lb = inverseTransformPosition(w->bodies[b].transform, v3(g));
	return orc_add_ball_constraint_local(w, a, b, &la.x, &lb.x);
}
u32 orc_add_fixed_constraint_global(world* w, u32 a, u32 b, const float* g)
{
	const trs& tA = w->bodies[a].transform; const trs& tB = w->bodies[b].transform;
	fixed_constraint c;
	c.localAnchorA = inverseTransformPosition(tA, v3(g));
	c.localAnchorB = inverseTransformPosition(tB, v3(g));
	c.initialInvRotationDifference = conjugate(tB.rotation) * tA.rotation;
	w->fixedConstraints.push_back(c);
	w->fixedPairs.push_back({ a, b });
	return (u32)w->fixedConstraints.size() - 1;
}
u32 orc_add_hinge_constraint_global(world* w, u32 a, u32 b, const float* anchor, const float* axis, float minLimit, float maxLimit)
{
	const trs& tA = w->bodies[a].transform; const trs& tB = w->bodies[b].transform;
	hinge_constraint c;
	c.localAnchorA = inverseTransformPosition(tA, v3(anchor));
	c.localAnchorB = inverseTransformPosition(tB, v3(anchor));
	c.localHingeAxisA = inverseTransformDirection(tA, v3(axis));
	c.localHingeAxisB = inverseTransformDirection(tB, v3(axis));
	getTangents(c.localHingeAxisA, c.localHingeTangentA, c.localHingeBitangentA);
	c.localHingeTangentB = conjugate(tB.rotation) * (tA.rotation * c.localHingeTangentA);
	c.minRotationLimit = minLimit; c.maxRotationLimit = maxLimit;
	c.motorType = constraint_velocity_motor; c.motorVelocity = 0.f; c.maxMotorTorque = -1.f;
	w->hingeConstraints.push_back(c);
	w->hingePairs.push_back({ a, b });
	return (u32)w->hingeConstraints.size() - 1;
}
u32 orc_add_cone_twist_constraint_global(world* w, u32 a, u32 b, const float* anchor, const float* axis, float swingLimit, float twistLimit)
{
	const trs& tA = w->bodies[a].transform; const trs& tB = w->bodies[b].transform;
	cone_twist_constraint c;
	c.localAnchorA = inverseTransformPosition(tA, v3(anchor));
	c.localAnchorB = inverseTransformPosition(tB, v3(anchor));
	c.swingLimit = swingLimit; c.twistLimit = twistLimit;
	c.localLimitAxisA = inverseTransformDirection(tA, v3(axis));
	c.localLimitAxisB = inverseTransformDirection(tB, v3(axis));
	getTangents(c.localLimitAxisA, c.localLimitTangentA, c.localLimitBitangentA);
	c.localLimitTangentB = conjugate(tB.rotation) * (tA.rotation * c.localLimitTangentA);
	c.swingMotorType = constraint_velocity_motor; c.swingMotorVelocity = 0.f; c.maxSwingMotorTorque = -1.f; c.swingMotorAxis = 0.f;
	c.twistMotorType = constraint_velocity_motor; c.twistMotorVelocity = 0.f; c.maxTwistMotorTorque = -1.f;
	w->coneTwistConstraints.push_back(c);
	w->coneTwistPairs.push_back({ a, b });
	return (u32)w->coneTwistConstraints.size() - 1;
}
u32 orc_add_slider_constraint_global(world* w, u32 a, u32 b, const float* anchor, const float* axis, float minLimit, float maxLimit)
{
	const trs& tA = w->bodies[a].transform; const trs& tB = w->bodies[b].transform;
	slider_constraint c;
	c.localAnchorA = inverseTransformPosition(tA, v3(anchor));
	c.localAnchorB = inverseTransformPosition(tB, v3(anchor));
	c.localAxisA = inverseTransformDirection(tA, v3(axis));
	c.initialInvRotationDifference = conjugate(tB.rotation) * tA.rotation;
	c.negDistanceLimit = minLimit; c.posDistanceLimit = maxLimit;
	c.motorType = constraint_velocity_motor; c.motorVelocity = 0.f; c.maxMotorForce = -1.f;
	w->sliderConstraints.push_back(c);
	w->sliderPairs.push_back({ a, b });
	return (u32)w->sliderConstraints.size() - 1;
}

// getConstraint() mutable access (physics.h:248-253): type 0..5 = distance..slider (constraints.h:14-28)
static void* constraintPtr(world* w, u32 type, u32 id, u32* size)
{
	switch (type)
	{
		case 0: *size = sizeof(distance_constraint); return id < w->distanceConstraints.size() ? &w->distanceConstraints[id] : 0;
		case 1: *size = sizeof(ball_constraint); return id < w->ballConstraints.size() ? &w->ballConstraints[id] : 0;
		case 2: *size = sizeof(fixed_constraint); return id < w->fixedConstraints.size() ? &w->fixedConstraints[id] : 0;
		case 3: *size = sizeof(hinge_constraint); return id < w->hingeConstraints.size() ? &w->hingeConstraints[id] : 0;
		case 4: *size = sizeof(cone_twist_constraint); return id < w->coneTwistConstraints.size() ? &w->coneTwistConstraints[id] : 0;
		case 5: *size = sizeof(slider_constraint); return id < w->sliderConstraints.size() ? &w->sliderConstraints[id] : 0;
	}
	return 0;
}
int orc_constraint_get(world* w, u32 type, u32 id, void* pod) { u32 s; void* p = constraintPtr(w, type, id, &s); if (!p) return 1; memcpy(pod, p, s); return 0; }
int orc_constraint_set(world* w, u32 type, u32 id, const void* pod) { u32 s; void* p = constraintPtr(w, type, id, &s); if (!p) return 1; memcpy(p, pod, s); return 0; }

int orc_delete_body(world* w, u32 b)
{
	if (b >= w->bodies.size()) return 1;
	body& rb = w->bodies[b];
	rb.removed = true; rb.invMass = 0.f; rb.invInertia = mat3::zero(); rb.gravityFactor = 0.f;
	rb.linearVelocity = rb.angularVelocity = rb.forceAccumulator = rb.torqueAccumulator = vec3(0.f);
	return 0;
}

// ---- testPhysicsInteraction — physics.cpp:556-628; ray tests bounding_volumes.cpp:197-394, 677-705; pointInTriangle math.cpp:1273-1290
struct ray { vec3 origin, direction; };
static bool intersectPlane(const ray& r, vec3 normal, float d, float& outT)
{
	float ndotd = dot(r.direction, normal);
	if (fabsf(ndotd) < 1e-6f) { return false; }
	outT = -(dot(r.origin, normal) + d) / ndotd;
	return true;
}
static bool intersectAABB(const ray& r, const bounding_box& a, float& outT)
{
	vec3 invDir = vec3(1.f / r.direction.x, 1.f / r.direction.y, 1.f / r.direction.z);
	float tx1 = (a.minCorner.x - r.origin.x) * invDir.x, tx2 = (a.maxCorner.x - r.origin.x) * invDir.x;
	outT = std::min(tx1, tx2);
	float tmax = std::max(tx1, tx2);
	float ty1 = (a.minCorner.y - r.origin.y) * invDir.y, ty2 = (a.maxCorner.y - r.origin.y) * invDir.y;
	outT = std::max(outT, std::min(ty1, ty2)); tmax = std::min(tmax, std::max(ty1, ty2));
	float tz1 = (a.minCorner.z - r.origin.z) * invDir.z, tz2 = (a.maxCorner.z - r.origin.z) * invDir.z;
	outT = std::max(outT, std::min(tz1, tz2)); tmax = std::min(tmax, std::max(tz1, tz2));
	return tmax >= outT && outT > 0.f;
}
static bool intersectSphere(const ray& r, vec3 center, float radius, float& outT)
{
	vec3 m = r.origin - center;
	float b = dot(m, r.direction), c = dot(m, m) - radius * radius;
	if (c > 0.f && b > 0.f) { return false; }
	float discr = b * b - c;
	if (discr < 0.f) { return false; }
	outT = -b - sqrtf(discr);
	if (outT < 0.f) { outT = 0.f; }
	return true;
}
static bool intersectDisk(const ray& r, vec3 pos, vec3 normal, float radius, float& outT)
{
	if (intersectPlane(r, normal, -dot(normal, pos), outT)) { return length(r.origin + outT * r.direction - pos) <= radius; }
	return false;
}
static bool intersectCylinder(const ray& r, const bounding_cylinder& cylinder, float& outT)
{
	vec3 d = r.direction, o = r.origin;
	vec3 axis = cylinder.positionB - cylinder.positionA;
	float height = length(axis);
	quat q = rotateFromTo(axis, vec3(0.f, 1.f, 0.f));
	o = q * (o - cylinder.positionA);
	d = q * d;
	float epsilon = 1e-6f;
	float y = -1.f;
	if (o.x * o.x + o.z * o.z > cylinder.radius * cylinder.radius)
	{
		float a = d.x * d.x + d.z * d.z, b = d.x * o.x + d.z * o.z, c = o.x * o.x + o.z * o.z - cylinder.radius * cylinder.radius;
		float delta = b * b - a * c;
		if (delta < epsilon) { return false; }
		outT = (-b - sqrtf(delta)) / a;
		if (outT <= epsilon) { return false; }
		y = o.y + outT * d.y;
	}
	if (y > height + epsilon || y < -epsilon)
	{
		ray localRay = { o, d };
		float dist;
		bool b1 = d.y < 0.f && intersectDisk(localRay, vec3(0.f, height, 0.f), vec3(0.f, 1.f, 0.f), cylinder.radius, dist);
		if (b1) { outT = dist; }
		bool b2 = d.y > 0.f && intersectDisk(localRay, vec3(0.f, 0.f, 0.f), vec3(0.f, -1.f, 0.f), cylinder.radius, dist);
		if (b2) { outT = dist; }
		y = o.y + outT * d.y;
	}
	return y > -epsilon && y < height + epsilon;
}
static bool intersectCapsule(const ray& r, const bounding_capsule& capsule, float& outT)
{
	outT = FLT_MAX;
	float t; bool result = false;
	if (intersectCylinder(r, bounding_cylinder{ capsule.positionA, capsule.positionB, capsule.radius }, t)) { outT = t; result = true; }
	if (intersectSphere(r, capsule.positionA, capsule.radius, t)) { outT = std::min(outT, t); result = true; }
	if (intersectSphere(r, capsule.positionB, capsule.radius, t)) { outT = std::min(outT, t); result = true; }
	return result;
}
static bool pointInTriangle(vec3 point, vec3 triA, vec3 triB, vec3 triC)
{
	vec3 e10 = triB - triA, e20 = triC - triA;
	float a = dot(e10, e10), b = dot(e10, e20), c = dot(e20, e20);
	float ac_bb = (a * c) - (b * b);
	vec3 vp = point - triA;
	float d = dot(vp, e10), e = dot(vp, e20);
	float x = (d * c) - (e * b), y = (e * a) - (d * b), z = x + y - ac_bb;
	u32 ux, uy, uz; memcpy(&ux, &x, 4); memcpy(&uy, &y, 4); memcpy(&uz, &z, 4);
	return ((uz & ~(ux | uy)) & 0x80000000u) != 0;
}
static bool intersectTriangle(const ray& r, vec3 a, vec3 b, vec3 c, float& outT)
{
	vec3 normal = noz(cross(b - a, c - a));
	float d = -dot(normal, a);
	float nDotR = dot(r.direction, normal);
	if (fabsf(nDotR) <= 1e-6f) { return false; }
	outT = -(dot(r.origin, normal) + d) / nDotR;
	vec3 q = r.origin + outT * r.direction;
	return outT >= 0.f && pointInTriangle(q, a, b, c);
}
// returns 1 + the index of the pushed body, 0 if nothing was hit
u32 orc_test_physics_interaction(world* w, const float* origin, const float* direction, float strength)
{
	ray r = { v3(origin), v3(direction) };
	float minT = FLT_MAX; int minRB = -1;
	vec3 force(0.f), torque(0.f);
	for (const collider& col : w->colliders)
	{
		if (col.parent == STATIC_BODY || w->bodies[col.parent].removed) { continue; }
		const body& rb = w->bodies[col.parent];
		const trs& transform = rb.transform1;
		ray localR = { conjugate(transform.rotation) * (r.origin - transform.position), conjugate(transform.rotation) * r.direction };
		float t = 0.f; bool hit = false;
		const collider_union& c = col.local;
		switch (c.type)
		{
			case collider_type_sphere: { bounding_sphere s = c.sphere(); hit = intersectSphere(localR, s.center, s.radius, t); } break;
			case collider_type_capsule: hit = intersectCapsule(localR, c.capsule(), t); break;
			case collider_type_cylinder: hit = intersectCylinder(localR, c.cylinder(), t); break;
			case collider_type_aabb: hit = intersectAABB(localR, c.aabb(), t); break;
			case collider_type_obb:
			{
				bounding_oriented_box a = c.obb();
				ray lr = { conjugate(a.rotation) * (localR.origin - a.center), conjugate(a.rotation) * localR.direction };
				hit = intersectAABB(lr, bounding_box::fromCenterRadius(vec3(0.f), a.radius), t);
			} break;
			case collider_type_hull:
			{
				bounding_hull h = c.hull();
				const bounding_hull_geometry& g = w->hullGeometries[h.geometryIndex];
				ray lr = { conjugate(h.rotation) * (localR.origin - h.position), conjugate(h.rotation) * localR.direction };
				float best = FLT_MAX;
				for (const bounding_hull_face& f : g.faces)
				{
					float tt;
					if (intersectTriangle(lr, g.vertices[f.a], g.vertices[f.b], g.vertices[f.c], tt) && tt < best) { best = tt; hit = true; }
				}
				t = best;
			} break;
			default: break;
		}
		if (hit && t < minT)
		{
			minT = t; minRB = (int)col.parent;
			vec3 localHit = localR.origin + t * localR.direction;
			vec3 globalHit = transformPosition(transform, localHit);
			vec3 cogPosition = transform.position + transform.rotation * rb.localCOGPosition; // getGlobalCOGPosition (rigid_body.cpp:83-87)
			force = r.direction * strength;
			torque = cross(globalHit - cogPosition, force);
		}
	}
	if (minRB < 0) { return 0; }
	w->bodies[minRB].torqueAccumulator += torque;
	w->bodies[minRB].forceAccumulator += force;
	return 1u + (u32)minRB;
}

int orc_apply_force_torque(world* w, u32 b, const float* f, const float* t)
{
	if (b >= w->bodies.size()) return 1;
	w->bodies[b].forceAccumulator += v3(f); w->bodies[b].torqueAccumulator += v3(t);
	return 0;
}
// Bulk state injection (n x {pos3, quat4}, n x {lin3, ang3}) — lets bench.py time the CPU path on the device's settled scene.
void orc_write_state(world* w, const float* in7, const float* in6, u32 n)
{
	for (u32 i = 0; i < n && i < w->bodies.size(); ++i)
	{
		body& b = w->bodies[i];
		const float* t = in7 + 7 * (size_t)i; const float* v = in6 + 6 * (size_t)i;
		b.transform.position = vec3(t[0], t[1], t[2]); b.transform.rotation = quat(t[3], t[4], t[5], t[6]);
		b.transform0 = b.transform1 = b.transform;
		b.linearVelocity = vec3(v[0], v[1], v[2]); b.angularVelocity = vec3(v[3], v[4], v[5]);
	}
}
// Sorts the SAP endpoints of the CURRENT configuration with std::stable_sort so that the next broadphase's insertion sort starts
// from temporal coherence (the reference pays its O(N^2) first-frame insertion sort once at scene load; a baseline sample that
// starts from an injected state would otherwise time that one-off cost).
void orc_presort_endpoints(world* w)
{
	getWorldSpaceColliders(*w);
	u32 axis = w->sortingAxis;
	for (sap_endpoint& ep : w->endpoints)
	{
		const bounding_box& aabb = w->worldSpaceAABBs[ep.collider];
		ep.value = ep.start ? aabb.minCorner[axis] : aabb.maxCorner[axis];
	}
	std::stable_sort(w->endpoints.begin(), w->endpoints.end(), [](const sap_endpoint& a, const sap_endpoint& b) { return a.value < b.value; });
}

int orc_set_velocity(world* w, u32 b, const float* lin, const float* ang)
{
	if (b >= w->bodies.size()) return 1;
	w->bodies[b].linearVelocity = v3(lin); w->bodies[b].angularVelocity = v3(ang);
	return 0;
}

int orc_step(world* w, float* timer, const physics_settings* settings, u32 mode, float dt) { physicsStep(*w, *timer, *settings, mode, dt); return 0; }
int orc_step_internal(world* w, u32 iterations, u32 mode, float dt) { physicsStepInternal(*w, iterations, mode, dt); return 0; }
void orc_set_custom_order(world* w, const u32* order, u32 n) { w->customOrder.assign(order, order + n); }
void orc_set_row_form(world* w, int on) { w->rowForm = on != 0; }                  // custom-order (follow) solves: the device's row form (default) or the reference formula
void orc_set_scalar_row_form(world* w, int on) { w->scalarRowForm = on != 0; }     // the scalar solver in emission order, in row form (default: the reference formula)
void orc_set_wide_rsqrt(int on) { wideApproxRsqrt() = on != 0; }
// Joint initialisation with the reference's wide math (polynomial trig, rsqrt-based normalisation: owidemath.h) — row a33.
void orc_set_wide_joint_math(int on) { wideJointMath() = on != 0; }
// The polynomial functions themselves, for the tests (which: 0 cos, 1 sin, 2 atan2(a, b), 3 acos).
float orc_poly_trig(int which, float a, float b) { return which == 0 ? polyCos(a) : which == 1 ? polySin(a) : which == 2 ? polyAtan2(a, b) : polyAcos(a); }
void orc_stage_seconds(world* w, double* out5, int reset) { for (int i = 0; i < 5; ++i) { out5[i] = w->stageSeconds[i]; if (reset) w->stageSeconds[i] = 0; } }
// Follow mode (see struct world): ordered candidate pairs + manifold execution order; n = 0 switches it off.
void orc_set_follow(world* w, const u32* pairs2, u32 numPairs, const u32* slotOrder, u32 numOrder)
{
	w->usePairOverride = numPairs > 0 || numOrder > 0 || pairs2 != 0;
	w->pairOverride.assign((const collider_pair*)pairs2, (const collider_pair*)pairs2 + numPairs);
	w->slotOrder.assign(slotOrder, slotOrder + numOrder);
}
void orc_clear_follow(world* w) { w->usePairOverride = false; }
void orc_set_joint_order(world* w, u32 type, const u32* order, u32 n) { if (type < 6) w->jointOrder[type].assign(order, order + n); }
u32 orc_read_slot_counts(world* w, u8* out) { memcpy(out, w->slotCounts.data(), w->slotCounts.size()); return (u32)w->slotCounts.size(); }

u32 orc_num_bodies(world* w) { return (u32)w->bodies.size(); }
u32 orc_num_colliders(world* w) { return (u32)w->colliders.size(); }
u32 orc_num_pairs(world* w) { return (u32)w->broadphasePairs.size(); }
u32 orc_num_contacts(world* w) { return (u32)w->contacts.size(); }
u32 orc_num_collisions(world* w) { return (u32)w->collidingPairs.size(); }
u32 orc_sorting_axis_used(world* w) { return w->usedSortingAxis; }
u32 orc_sorting_axis_next(world* w) { return w->sortingAxis; }
void orc_set_wide_broadphase(world* w, int on) { w->wideBroadphase = on != 0; }
void orc_set_sim_mask(world* w, const u8* simulate, u32 n) { w->simOff.assign(w->bodies.size(), 0); for (u32 i = 0; i < n && i < w->simOff.size(); ++i) w->simOff[i] = simulate[i] ? 0 : 1; }
void orc_set_sorting_axis(world* w, u32 axis) { if (axis < 3u) w->sortingAxis = axis; } // a world that takes over another's state mid-run takes its sap_context::sortingAxis too
void orc_sorting_variance(world* w, float* out3) { for (int k = 0; k < 3; ++k) out3[k] = w->lastVariance[k]; }

// which: 0 = interpolated transform, 1 = physics_transform1, 2 = physics_transform0
void orc_read_transforms(world* w, u32 which, float* out7)
{
	for (size_t i = 0; i < w->bodies.size(); ++i)
	{
		const trs& t = which == 0 ? w->bodies[i].transform : (which == 1 ? w->bodies[i].transform1 : w->bodies[i].transform0);
		float* o = out7 + 7 * i;
		o[0] = t.position.x; o[1] = t.position.y; o[2] = t.position.z;
		o[3] = t.rotation.x; o[4] = t.rotation.y; o[5] = t.rotation.z; o[6] = t.rotation.w;
	}
}
void orc_read_velocities(world* w, float* out6)
{
	for (size_t i = 0; i < w->bodies.size(); ++i)
	{
		float* o = out6 + 6 * i; const body& b = w->bodies[i];
		o[0] = b.linearVelocity.x; o[1] = b.linearVelocity.y; o[2] = b.linearVelocity.z;
		o[3] = b.angularVelocity.x; o[4] = b.angularVelocity.y; o[5] = b.angularVelocity.z;
	}
}
// localCOG(3) invMass(1) invInertia(9, column-major like the reference's mat3)
void orc_read_mass_properties(world* w, float* out13)
{
	for (size_t i = 0; i < w->bodies.size(); ++i)
	{
		float* o = out13 + 13 * i; const body& b = w->bodies[i];
		o[0] = b.localCOGPosition.x; o[1] = b.localCOGPosition.y; o[2] = b.localCOGPosition.z; o[3] = b.invMass;
		memcpy(o + 4, b.invInertia.m(), 36);
	}
}
void orc_read_world_colliders(world* w, void* outColliders64, float* outAABBs6)
{
	memcpy(outColliders64, w->worldSpaceColliders.data(), w->worldSpaceColliders.size() * sizeof(collider_union));
	memcpy(outAABBs6, w->worldSpaceAABBs.data(), w->worldSpaceAABBs.size() * sizeof(bounding_box));
}
void orc_read_pairs(world* w, u32* out2) { memcpy(out2, w->broadphasePairs.data(), w->broadphasePairs.size() * sizeof(collider_pair)); }
void orc_read_contacts(world* w, void* outContacts32, u32* outBodyPairs2, u32* outCollisionIndex)
{
	memcpy(outContacts32, w->contacts.data(), w->contacts.size() * sizeof(collision_contact));
	memcpy(outBodyPairs2, w->contactBodyPairs.data(), w->contactBodyPairs.size() * sizeof(constraint_body_pair));
	memcpy(outCollisionIndex, w->contactCollisionIndex.data(), w->contactCollisionIndex.size() * sizeof(u32));
}
void orc_read_collisions(world* w, u32* outPairs2, u8* outCounts)
{
	memcpy(outPairs2, w->collidingPairs.data(), w->collidingPairs.size() * sizeof(collider_pair));
	memcpy(outCounts, w->contactCountPerCollision.data(), w->contactCountPerCollision.size());
}
// which: 0 = after solve, 1 = before solve (after gravity).  104 B records incl. the dummy at index N.
void orc_read_rb_global(world* w, u32 which, void* out104)
{
	const std::vector<rigid_body_global_state>& v = which ? w->rbGlobalPreSolve : w->rbGlobal;
	memcpy(out104, v.data(), v.size() * sizeof(rigid_body_global_state));
}
u32 orc_num_contact_slots(world* w) { return (u32)w->contactSlots.size(); }
void orc_read_contact_slots(world* w, u32* out8) { memcpy(out8, w->contactSlots.data(), w->contactSlots.size() * sizeof(sched_slot)); }

// ---- stage-level entry points for per-kernel parity tests -------------------------------------------
// Narrowphase on caller-provided world-space colliders and ORDERED pairs (A,B as given, no reordering):
// returns number of contacts; outCounts[numPairs] contact count per pair (0 = no collision).
u32 orc_narrowphase_ordered(const void* colliders64, const u32* pairs2, u32 numPairs, void* outContacts32, u8* outCounts)
{
	const collider_union* cols = (const collider_union*)colliders64;
	collision_contact* out = (collision_contact*)outContacts32;
	u32 n = 0;
	for (u32 i = 0; i < numPairs; ++i)
	{
		const collider_union& A = cols[pairs2[2 * i]];
		const collider_union& B = cols[pairs2[2 * i + 1]];
		contact_manifold m; m.numContacts = 0;
		outCounts[i] = 0;
		if (intersectColliders(A, B, m))
		{
			u32 fr = packFrictionRestitution(A, B);
			outCounts[i] = (u8)m.numContacts;
			for (u32 k = 0; k < m.numContacts; ++k)
			{
				out[n].point = m.contacts[k].point; out[n].penetrationDepth = m.contacts[k].penetrationDepth;
				out[n].normal = m.collisionNormal; out[n].friction_restitution = fr;
				++n;
			}
		}
	}
	return n;
}
// Scheduler on caller-provided body pairs; out8 must hold ceil(n/1)*8 u32 in the worst case.
u32 orc_schedule(const u32* bodyPairs2, u32 n, u32 dummy, u32* out8)
{
	std::vector<sched_slot> slots;
	scheduleConstraintsSIMD((const constraint_body_pair*)bodyPairs2, n, dummy, slots);
	memcpy(out8, slots.data(), slots.size() * sizeof(sched_slot));
	return (u32)slots.size();
}
// Contact init + `iterations` GS sweeps in the given order over caller-provided 104-byte body records (in/out).
// outRows: 17 floats per contact {rA3 rB3 t3 mn mt bias lambdaN lambdaT ... } see oracle.py for the layout.
void orc_solve_contacts(void* rb104, u32 numBodiesPlusDummy, const void* contacts32, const u32* bodyPairs2, u32 numContacts,
	const u32* order, u32 orderLen, u32 iterations, float dt, float* outImpulses2)
{
	(void)numBodiesPlusDummy;
	rigid_body_global_state* rbs = (rigid_body_global_state*)rb104;
	const collision_contact* contacts = (const collision_contact*)contacts32;
	const constraint_body_pair* pairs = (const constraint_body_pair*)bodyPairs2;
	std::vector<collision_constraint> cons(numContacts);
	for (u32 i = 0; i < numContacts; ++i) initializeCollisionConstraint(cons[i], rbs, contacts[i], pairs[i], dt);
	for (u32 it = 0; it < iterations; ++it)
	{
		for (u32 k = 0; k < orderLen; ++k) { u32 i = order[k]; solveCollisionConstraint(cons[i], contacts[i], pairs[i], rbs); }
	}
	for (u32 i = 0; i < numContacts; ++i) { outImpulses2[2 * i] = cons[i].impulseInNormalDir; outImpulses2[2 * i + 1] = cons[i].impulseInTangentDir; }
}

void orc_stats(u32* out4) { out4[0] = g_gjkMaxItersSeen; out4[1] = g_epaMaxTriangles; out4[2] = g_epaMaxEdges; out4[3] = g_epaMaxBorder; }

} // extern "C"
