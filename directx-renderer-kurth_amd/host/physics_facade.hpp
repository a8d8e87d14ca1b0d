// physics_facade.hpp — header-only C++ host side over the C-ABI (include/mi_physics.h) that keeps the reference's call shapes for the
// rigid-body path, so engine code that today says
//
//     auto e = scene.createEntity("box").addComponent<transform_component>(pos, rot)
//                   .addComponent<collider_component>(collider_component::asOBB(box, material))
//                   .addComponent<rigid_body_component>(false, 1.f);
//     auto h = addHingeConstraintFromGlobalPoints(a, b, anchor, axis, -0.5f, 0.5f);
//     getConstraint(scene, h).maxMotorTorque = 200.f;
//     physicsStep(scene, arena, timer, settings, dt);
//
// compiles against this header with the same statements (reference: physics.h:108-157, 209-264, 382-405; rigid_body.h:18-46;
// scene.h:38-84).  What differs, and why:
//   * game_scene here owns an mi_world (device memory + one HIP stream) instead of an EnTT registry; entities are light handles.
//   * getConstraint() returns a write-back proxy instead of T&: the POD lives in device-side tables, so the proxy reads it on
//     construction and stores it when it goes out of scope (same statement syntax for field writes).
//   * memory_arena& is accepted and ignored (per-step arrays live in the world's device arena).
//   * Errors surface as mi::physics_error instead of ASSERT/__debugbreak (pch.h:33-34).
// Nothing here computes physics; there is no CPU path.
#pragma once

#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "mi_physics.h"

namespace mi
{
	struct physics_error : std::runtime_error { using std::runtime_error::runtime_error; };

	struct vec3 { float x = 0, y = 0, z = 0; vec3() = default; vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
	struct quat { float x = 0, y = 0, z = 0, w = 1; quat() = default; quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {} };

	// bounding volumes as the collider factories take them (bounding_volumes.h:25-148)
	struct bounding_sphere { vec3 center; float radius; };
	struct bounding_capsule { vec3 positionA, positionB; float radius; };
	struct bounding_cylinder { vec3 positionA, positionB; float radius; };
	struct bounding_box { vec3 minCorner, maxCorner; static bounding_box fromCenterRadius(vec3 c, vec3 r) { return { { c.x - r.x, c.y - r.y, c.z - r.z }, { c.x + r.x, c.y + r.y, c.z + r.z } }; } };
	struct bounding_oriented_box { quat rotation; vec3 center, radius; };
	struct bounding_hull { quat rotation; vec3 position; uint32_t geometryIndex; }; // geometryIndex from game_scene::allocateBoundingHullGeometry

	struct physics_material { float restitution, friction, density; }; // physics.h:40-47 without the sound tag

	struct transform_component { vec3 position; quat rotation; transform_component(vec3 p = {}, quat r = {}) : position(p), rotation(r) {} };

	// collider_component::as*(shape, material), physics.h:108-157
	struct collider_component
	{
		uint32_t type = MI_COLLIDER_SPHERE; float shape[10] = {}; physics_material material{};
		static collider_component asSphere(bounding_sphere s, physics_material m) { collider_component c; c.type = MI_COLLIDER_SPHERE; c.set({ s.center.x, s.center.y, s.center.z, s.radius }); c.material = m; return c; }
		static collider_component asCapsule(bounding_capsule s, physics_material m) { collider_component c; c.type = MI_COLLIDER_CAPSULE; c.set({ s.positionA.x, s.positionA.y, s.positionA.z, s.positionB.x, s.positionB.y, s.positionB.z, s.radius }); c.material = m; return c; }
		static collider_component asCylinder(bounding_cylinder s, physics_material m) { collider_component c; c.type = MI_COLLIDER_CYLINDER; c.set({ s.positionA.x, s.positionA.y, s.positionA.z, s.positionB.x, s.positionB.y, s.positionB.z, s.radius }); c.material = m; return c; }
		static collider_component asAABB(bounding_box b, physics_material m) { collider_component c; c.type = MI_COLLIDER_AABB; c.set({ b.minCorner.x, b.minCorner.y, b.minCorner.z, b.maxCorner.x, b.maxCorner.y, b.maxCorner.z }); c.material = m; return c; }
		static collider_component asOBB(bounding_oriented_box b, physics_material m) { collider_component c; c.type = MI_COLLIDER_OBB; c.set({ b.rotation.x, b.rotation.y, b.rotation.z, b.rotation.w, b.center.x, b.center.y, b.center.z, b.radius.x, b.radius.y, b.radius.z }); c.material = m; return c; }
		static collider_component asHull(bounding_hull h, physics_material m) { collider_component c; c.type = MI_COLLIDER_HULL; c.set({ h.rotation.x, h.rotation.y, h.rotation.z, h.rotation.w, h.position.x, h.position.y, h.position.z, (float)h.geometryIndex }); c.material = m; return c; }
	private:
		void set(std::initializer_list<float> v) { int i = 0; for (float f : v) shape[i++] = f; }
	};

	// rigid_body_component(bool kinematic, float gravityFactor = 1, float linearDamping = .4, float angularDamping = .4), rigid_body.h:21
	struct rigid_body_component
	{
		bool kinematic = false; float gravityFactor = 1.f, linearDamping = 0.4f, angularDamping = 0.4f;
		rigid_body_component(bool k = false, float g = 1.f, float l = 0.4f, float a = 0.4f) : kinematic(k), gravityFactor(g), linearDamping(l), angularDamping(a) {}
	};

	struct scene_entity;
	// force_field_component (physics.h:182-185): on an entity without colliders the force acts on every rigid body, with colliders on
	// the bodies overlapping them; the entity's transform_component (if it has one) rotates the force
	struct force_field_component { vec3 force; force_field_component(vec3 f = {}) : force(f) {} };
	// cloth_component(width, height, gridSizeX, gridSizeY, totalMass, stiffness, damping, gravityFactor), cloth.h:8-9; hung at the entity's transform
	struct cloth_component
	{
		float width, height; uint32_t gridSizeX, gridSizeY; float totalMass, stiffness, damping, gravityFactor;
		cloth_component(float w, float h, uint32_t gx, uint32_t gy, float mass, float stiff = 0.5f, float damp = 0.3f, float gravity = 1.f)
			: width(w), height(h), gridSizeX(gx), gridSizeY(gy), totalMass(mass), stiffness(stiff), damping(damp), gravityFactor(gravity) {}
	};
	// heightmap_collider_component(chunksPerDim, chunkSize, material), heightmap_collider.h:127-152: one per scene; update() and the chunks'
	// setHeights() go through game_scene::heightmapUpdate / heightmapSetHeights once the component is added
	struct heightmap_collider_component
	{
		uint32_t chunksPerDim; float chunkSize; physics_material material;
		heightmap_collider_component(uint32_t c, float s, physics_material m) : chunksPerDim(c), chunkSize(s), material(m) {}
	};
	// trigger_event / trigger_component (physics.h:187-203)
	enum trigger_event_type { trigger_event_enter, trigger_event_leave };
	struct trigger_event;
	struct trigger_component { std::function<void(trigger_event)> callback; trigger_component(std::function<void(trigger_event)> cb = {}) : callback(std::move(cb)) {} };
	struct collision_begin_event;
	struct collision_end_event;
	typedef std::function<void(const collision_begin_event&)> collision_begin_event_func; // physics.h:379-380
	typedef std::function<void(const collision_end_event&)> collision_end_event_func;

	// physics_settings, physics.h:382-397
	struct physics_settings
	{
		collision_begin_event_func collisionBeginCallback; collision_end_event_func collisionEndCallback;
		bool fixedFrameRate = true; uint32_t frameRate = 120; uint32_t maxPhysicsIterationsPerFrame = 4; uint32_t numRigidSolverIterations = 30;
		uint32_t numClothVelocityIterations = 0, numClothPositionIterations = 1, numClothDriftIterations = 0;
		bool simdBroadPhase = true, simdNarrowPhase = true, simdConstraintSolver = true;
	};

	struct memory_arena {}; // accepted and ignored by physicsStep

	// Joint PODs: byte-identical to the reference structs (constraints.h:73-80,129-135,175-183,229-257,346-380,497-520).
	struct distance_constraint { vec3 localAnchorA, localAnchorB; float globalLength; };
	struct ball_constraint { vec3 localAnchorA, localAnchorB; };
	struct fixed_constraint { quat initialInvRotationDifference; vec3 localAnchorA, localAnchorB; };
	enum constraint_motor_type : uint32_t { constraint_velocity_motor = MI_MOTOR_VELOCITY, constraint_position_motor = MI_MOTOR_POSITION };
	struct hinge_constraint
	{
		vec3 localAnchorA, localAnchorB, localHingeAxisA, localHingeAxisB;
		float minRotationLimit, maxRotationLimit, maxMotorTorque;
		constraint_motor_type motorType; union { float motorVelocity; float motorTargetAngle; };
		vec3 localHingeTangentA, localHingeBitangentA, localHingeTangentB;
	};
	struct cone_twist_constraint
	{
		vec3 localAnchorA, localAnchorB, localLimitAxisA, localLimitAxisB, localLimitTangentA, localLimitBitangentA, localLimitTangentB;
		float swingLimit, twistLimit;
		constraint_motor_type swingMotorType; union { float swingMotorVelocity; float swingMotorTargetAngle; }; float maxSwingMotorTorque, swingMotorAxis;
		constraint_motor_type twistMotorType; union { float twistMotorVelocity; float twistMotorTargetAngle; }; float maxTwistMotorTorque;
	};
	struct slider_constraint
	{
		quat initialInvRotationDifference; vec3 localAnchorA, localAnchorB, localAxisA;
		float negDistanceLimit, posDistanceLimit, maxMotorForce;
		constraint_motor_type motorType; union { float motorVelocity; float motorTargetDistance; };
	};
	static_assert(sizeof(distance_constraint) == 28 && sizeof(ball_constraint) == 24 && sizeof(fixed_constraint) == 40, "POD layout");
	static_assert(sizeof(hinge_constraint) == 104 && sizeof(cone_twist_constraint) == 120 && sizeof(slider_constraint) == 72, "POD layout");

	template <typename T> struct constraint_type_of;
	template <> struct constraint_type_of<distance_constraint> { static constexpr uint32_t value = MI_CONSTRAINT_DISTANCE; };
	template <> struct constraint_type_of<ball_constraint> { static constexpr uint32_t value = MI_CONSTRAINT_BALL; };
	template <> struct constraint_type_of<fixed_constraint> { static constexpr uint32_t value = MI_CONSTRAINT_FIXED; };
	template <> struct constraint_type_of<hinge_constraint> { static constexpr uint32_t value = MI_CONSTRAINT_HINGE; };
	template <> struct constraint_type_of<cone_twist_constraint> { static constexpr uint32_t value = MI_CONSTRAINT_CONE_TWIST; };
	template <> struct constraint_type_of<slider_constraint> { static constexpr uint32_t value = MI_CONSTRAINT_SLIDER; };

	template <typename T> struct constraint_handle { uint32_t id = 0xFFFFFFFFu; };
	using distance_constraint_handle = constraint_handle<distance_constraint>;
	using ball_constraint_handle = constraint_handle<ball_constraint>;
	using fixed_constraint_handle = constraint_handle<fixed_constraint>;
	using hinge_constraint_handle = constraint_handle<hinge_constraint>;
	using cone_twist_constraint_handle = constraint_handle<cone_twist_constraint>;
	using slider_constraint_handle = constraint_handle<slider_constraint>;

	struct game_scene;

	// scene_entity: a light handle (scene.h:38-84).  Colliders added before the rigid body are held back until either the body is
	// added (they become its colliders and its mass properties are recomputed, rigid_body.cpp:29-81) or the first step (they become
	// static colliders, physics.cpp:667-671).
	struct scene_entity
	{
		game_scene* scene = nullptr; uint32_t index = 0xFFFFFFFFu;
		template <typename T, typename... Args> scene_entity& addComponent(Args&&... args);
		uint32_t body() const;
		transform_component transform() const; // transform_component after the last physicsStep (interpolated)
		vec3 linearVelocity() const;
		std::vector<vec3> clothPositions() const; // cloth_component::positions (what cloth_render_component::getRenderData uploads, cloth.cpp:357-363)
		bool operator==(const scene_entity& o) const { return scene == o.scene && index == o.index; }
	};
	struct trigger_event { scene_entity trigger, other; trigger_event_type type; };                                       // physics.h:193-198
	struct collision_begin_event { scene_entity entityA, entityB; const collider_component& colliderA; const collider_component& colliderB; vec3 position, normal, relativeVelocity; }; // physics.h:356-367
	struct collision_end_event { scene_entity entityA, entityB; const collider_component& colliderA; const collider_component& colliderB; };                                          // physics.h:369-376

	struct game_scene
	{
		struct entity_record
		{
			transform_component transform; bool hasTransform = false; uint32_t body = MI_STATIC_BODY, field = 0xFFFFFFFFu, trigger = 0xFFFFFFFFu, cloth = 0xFFFFFFFFu;
			std::vector<collider_component> pending; std::vector<uint32_t> colliders; trigger_component triggerComponent;
		};

		explicit game_scene(int device = -1)
		{
			mi_world_desc d{}; d.device = device;
			world = mi_world_create(&d);
			if (!world) throw physics_error(std::string("mi_world_create: ") + mi_last_error(nullptr));
		}
		~game_scene() { if (world) mi_world_destroy(world); }
		game_scene(const game_scene&) = delete; game_scene& operator=(const game_scene&) = delete;

		// allocateBoundingHullGeometry (physics.h:207) from the mesh on: vertices + outward-facing triangles instead of a model file
		uint32_t allocateBoundingHullGeometry(const std::vector<vec3>& vertices, const std::vector<uint32_t>& triangles)
		{
			static_assert(sizeof(vec3) == 12, "vec3 is three floats");
			return checkId(mi_add_hull_geometry(world, &vertices[0].x, (uint32_t)vertices.size(), triangles.data(), (uint32_t)(triangles.size() / 3)), "allocateBoundingHullGeometry");
		}

		// scene.deleteEntity for an entity with a rigid body: body, colliders and constraints leave the simulation
		void deleteEntity(uint32_t entityIndex) { if (entities[entityIndex].body != MI_STATIC_BODY) check(mi_delete_body(world, entities[entityIndex].body), "deleteEntity"); }

		scene_entity createEntity(const char* /*name*/ = nullptr) { entities.emplace_back(); return scene_entity{ this, (uint32_t)entities.size() - 1 }; }

		void check(int status, const char* what) const { if (status != MI_OK) throw physics_error(std::string(what) + ": " + mi_last_error(world)); }
		uint32_t checkId(uint32_t id, const char* what) const { if (id == 0xFFFFFFFFu) throw physics_error(std::string(what) + ": " + mi_last_error(world)); return id; }

		void flushStaticColliders()
		{
			for (uint32_t i = 0; i < (uint32_t)entities.size(); ++i)
			{
				auto& e = entities[i];
				if (e.body != MI_STATIC_BODY) continue;
				for (auto& c : e.pending)
				{
					mi_material m{ c.material.restitution, c.material.friction, c.material.density };
					uint32_t id = checkId(mi_add_static_collider(world, c.type, c.shape, &m, &e.transform.position.x, &e.transform.rotation.x), "mi_add_static_collider");
					registerCollider(i, id, c);
				}
				e.pending.clear();
			}
		}
		// collider id -> owning entity + component (the events name colliders by id); body / trigger id -> entity
		void registerCollider(uint32_t entityIndex, uint32_t id, const collider_component& c)
		{
			entities[entityIndex].colliders.push_back(id);
			if (colliderEntity.size() <= id) { colliderEntity.resize(id + 1, 0xFFFFFFFFu); colliderComponents.resize(id + 1); }
			colliderEntity[id] = entityIndex; colliderComponents[id] = c;
		}
		void addZoneCollider(uint32_t entityIndex, const collider_component& c) // collider of a force-field / trigger entity (physics.cpp:657-666)
		{
			auto& e = entities[entityIndex];
			uint32_t id = (e.field != 0xFFFFFFFFu) ? mi_add_force_field_collider(world, e.field, c.type, c.shape) : mi_add_trigger_collider(world, e.trigger, c.type, c.shape);
			registerCollider(entityIndex, checkId(id, "zone collider"), c);
		}
		// heightmap_collider_component::update(minCorner, amplitudeScale) / collider(x, z).setHeights(heights) (heightmap_collider.h:131, 17)
		void heightmapUpdate(vec3 minCorner, float amplitudeScale) { check(mi_heightmap_update(world, &minCorner.x, amplitudeScale), "heightmap update"); }
		void heightmapSetHeights(uint32_t x, uint32_t z, const uint16_t* heights129x129) { check(mi_heightmap_set_chunk(world, x, z, heights129x129), "heightmap setHeights"); }
		float heightmapHeightAt(float x, float z) const { return mi_heightmap_height_at(world, x, z); }
		// drains the device's events and calls back in the reference's order (physics.cpp:1000-1032, 1128-1174)
		void dispatchEvents(const physics_settings& settings)
		{
			mi_event buffer[256];
			for (;;)
			{
				uint32_t n = mi_drain_events(world, buffer, 256);
				for (uint32_t i = 0; i < n; ++i)
				{
					const mi_event& e = buffer[i];
					if (e.kind == MI_EVENT_TRIGGER_ENTER || e.kind == MI_EVENT_TRIGGER_LEAVE)
					{
						uint32_t t = triggerEntity[e.a];
						if (entities[t].triggerComponent.callback)
							entities[t].triggerComponent.callback(trigger_event{ scene_entity{ this, t }, scene_entity{ this, bodyEntity[e.b] }, e.kind == MI_EVENT_TRIGGER_ENTER ? trigger_event_enter : trigger_event_leave });
						continue;
					}
					scene_entity a{ this, colliderEntity[e.a] }, b{ this, colliderEntity[e.b] };
					if (e.kind == MI_EVENT_COLLISION_BEGIN && settings.collisionBeginCallback)
						settings.collisionBeginCallback(collision_begin_event{ a, b, colliderComponents[e.a], colliderComponents[e.b],
							vec3(e.position[0], e.position[1], e.position[2]), vec3(e.normal[0], e.normal[1], e.normal[2]), vec3(e.relativeVelocity[0], e.relativeVelocity[1], e.relativeVelocity[2]) });
					else if (e.kind == MI_EVENT_COLLISION_END && settings.collisionEndCallback)
						settings.collisionEndCallback(collision_end_event{ a, b, colliderComponents[e.a], colliderComponents[e.b] });
				}
				if (n < 256) break;
			}
		}

		mi_world* world = nullptr;
		std::vector<entity_record> entities;
		std::vector<uint32_t> colliderEntity, bodyEntity, triggerEntity; std::vector<collider_component> colliderComponents;
		int collisionEventMask = -1;
	};

	template <typename T, typename... Args> inline scene_entity& scene_entity::addComponent(Args&&... args)
	{
		auto& e = scene->entities[index];
		if constexpr (std::is_same_v<T, transform_component>)
		{
			e.transform = transform_component(std::forward<Args>(args)...); e.hasTransform = true;
			if (e.cloth != 0xFFFFFFFFu) scene->check(mi_cloth_set_fixed_vertices(scene->world, e.cloth, &e.transform.position.x, &e.transform.rotation.x, 1), "setWorldPositionOfFixedVertices"); // scene.h:96-101
		}
		else if constexpr (std::is_same_v<T, collider_component>)
		{
			collider_component c(std::forward<Args>(args)...);
			if (e.field != 0xFFFFFFFFu || e.trigger != 0xFFFFFFFFu) scene->addZoneCollider(index, c);
			else if (e.body == MI_STATIC_BODY) e.pending.push_back(c);
			else
			{
				mi_material m{ c.material.restitution, c.material.friction, c.material.density };
				scene->registerCollider(index, scene->checkId(mi_add_collider(scene->world, e.body, c.type, c.shape, &m), "mi_add_collider"), c);
			}
		}
		else if constexpr (std::is_same_v<T, heightmap_collider_component>)
		{
			heightmap_collider_component h(std::forward<Args>(args)...);
			mi_material m{ h.material.restitution, h.material.friction, h.material.density };
			const float origin[3] = { 0.f, 0.f, 0.f };
			scene->check(mi_set_heightmap(scene->world, h.chunksPerDim, h.chunkSize, &m, origin, 1.f), "heightmap_collider_component");
		}
		else if constexpr (std::is_same_v<T, cloth_component>)
		{
			cloth_component c(std::forward<Args>(args)...);
			e.cloth = scene->checkId(mi_add_cloth(scene->world, c.width, c.height, c.gridSizeX, c.gridSizeY, c.totalMass, c.stiffness, c.damping, c.gravityFactor), "mi_add_cloth");
			if (e.hasTransform) scene->check(mi_cloth_set_fixed_vertices(scene->world, e.cloth, &e.transform.position.x, &e.transform.rotation.x, 1), "setWorldPositionOfFixedVertices"); // scene.h:86-94
		}
		else if constexpr (std::is_same_v<T, force_field_component> || std::is_same_v<T, trigger_component>)
		{
			const float* pos = e.hasTransform ? &e.transform.position.x : nullptr; const float* rot = e.hasTransform ? &e.transform.rotation.x : nullptr;
			if constexpr (std::is_same_v<T, force_field_component>)
			{
				force_field_component f(std::forward<Args>(args)...);
				e.field = scene->checkId(mi_add_force_field(scene->world, &f.force.x, pos, rot), "mi_add_force_field");
			}
			else
			{
				e.triggerComponent = trigger_component(std::forward<Args>(args)...);
				e.trigger = scene->checkId(mi_add_trigger(scene->world, pos, rot), "mi_add_trigger");
				if (scene->triggerEntity.size() <= e.trigger) scene->triggerEntity.resize(e.trigger + 1, 0xFFFFFFFFu);
				scene->triggerEntity[e.trigger] = index;
			}
			std::vector<collider_component> pending; pending.swap(e.pending);
			for (auto& c : pending) scene->addZoneCollider(index, c);
		}
		else if constexpr (std::is_same_v<T, rigid_body_component>)
		{
			rigid_body_component rb(std::forward<Args>(args)...);
			e.body = scene->checkId(mi_add_body(scene->world, rb.kinematic, rb.gravityFactor, rb.linearDamping, rb.angularDamping, &e.transform.position.x, &e.transform.rotation.x), "mi_add_body");
			if (scene->bodyEntity.size() <= e.body) scene->bodyEntity.resize(e.body + 1, 0xFFFFFFFFu);
			scene->bodyEntity[e.body] = index;
			std::vector<collider_component> pending; pending.swap(e.pending);
			for (auto& c : pending)
			{
				mi_material m{ c.material.restitution, c.material.friction, c.material.density };
				scene->registerCollider(index, scene->checkId(mi_add_collider(scene->world, e.body, c.type, c.shape, &m), "mi_add_collider"), c);
			}
		}
		else static_assert(sizeof(T) == 0, "component type not on the rigid-body path");
		return *this;
	}

	inline uint32_t scene_entity::body() const
	{
		uint32_t b = scene->entities[index].body;
		if (b == MI_STATIC_BODY) throw physics_error("entity has no rigid_body_component");
		return b;
	}

	inline transform_component scene_entity::transform() const
	{
		auto& e = scene->entities[index];
		if (e.body == MI_STATIC_BODY) return e.transform;
		uint32_t n = e.body + 1; std::vector<float> t(7 * (size_t)n);
		scene->check(mi_read_transforms(scene->world, 0, t.data(), n), "mi_read_transforms");
		const float* p = &t[7 * (size_t)e.body];
		return transform_component({ p[0], p[1], p[2] }, { p[3], p[4], p[5], p[6] });
	}

	inline std::vector<vec3> scene_entity::clothPositions() const
	{
		uint32_t c = scene->entities[index].cloth;
		if (c == 0xFFFFFFFFu) throw physics_error("entity has no cloth_component");
		std::vector<vec3> p(mi_cloth_num_particles(scene->world, c));
		scene->check(mi_cloth_read(scene->world, c, &p[0].x, nullptr), "mi_cloth_read");
		return p;
	}

	inline vec3 scene_entity::linearVelocity() const
	{
		uint32_t n = body() + 1; std::vector<float> v(6 * (size_t)n);
		scene->check(mi_read_velocities(scene->world, v.data(), n), "mi_read_velocities");
		return { v[6 * (size_t)(n - 1)], v[6 * (size_t)(n - 1) + 1], v[6 * (size_t)(n - 1) + 2] };
	}

	// ---- add*ConstraintFrom{Local,Global}Points, physics.h:209-235
	inline distance_constraint_handle addDistanceConstraintFromLocalPoints(scene_entity& a, scene_entity& b, vec3 localAnchorA, vec3 localAnchorB, float distance)
	{ return { a.scene->checkId(mi_add_distance_constraint_local(a.scene->world, a.body(), b.body(), &localAnchorA.x, &localAnchorB.x, distance), "addDistanceConstraint") }; }
	inline distance_constraint_handle addDistanceConstraintFromGlobalPoints(scene_entity& a, scene_entity& b, vec3 globalAnchorA, vec3 globalAnchorB)
	{ return { a.scene->checkId(mi_add_distance_constraint_global(a.scene->world, a.body(), b.body(), &globalAnchorA.x, &globalAnchorB.x), "addDistanceConstraint") }; }
	inline ball_constraint_handle addBallConstraintFromLocalPoints(scene_entity& a, scene_entity& b, vec3 localAnchorA, vec3 localAnchorB)
	{ return { a.scene->checkId(mi_add_ball_constraint_local(a.scene->world, a.body(), b.body(), &localAnchorA.x, &localAnchorB.x), "addBallConstraint") }; }
	inline ball_constraint_handle addBallConstraintFromGlobalPoints(scene_entity& a, scene_entity& b, vec3 globalAnchor)
	{ return { a.scene->checkId(mi_add_ball_constraint_global(a.scene->world, a.body(), b.body(), &globalAnchor.x), "addBallConstraint") }; }
	inline fixed_constraint_handle addFixedConstraintFromGlobalPoints(scene_entity& a, scene_entity& b, vec3 globalAnchor)
	{ return { a.scene->checkId(mi_add_fixed_constraint_global(a.scene->world, a.body(), b.body(), &globalAnchor.x), "addFixedConstraint") }; }
	inline hinge_constraint_handle addHingeConstraintFromGlobalPoints(scene_entity& a, scene_entity& b, vec3 globalAnchor, vec3 globalHingeAxis, float minLimit = 1.f, float maxLimit = -1.f)
	{ return { a.scene->checkId(mi_add_hinge_constraint_global(a.scene->world, a.body(), b.body(), &globalAnchor.x, &globalHingeAxis.x, minLimit, maxLimit), "addHingeConstraint") }; }
	inline cone_twist_constraint_handle addConeTwistConstraintFromGlobalPoints(scene_entity& a, scene_entity& b, vec3 globalAnchor, vec3 globalAxis, float swingLimit, float twistLimit)
	{ return { a.scene->checkId(mi_add_cone_twist_constraint_global(a.scene->world, a.body(), b.body(), &globalAnchor.x, &globalAxis.x, swingLimit, twistLimit), "addConeTwistConstraint") }; }
	// addConstraint(a, b, const T&), physics.h:239-244: one overload per constraint POD (the reference's template, spelt out)
	inline distance_constraint_handle addConstraint(scene_entity& a, scene_entity& b, const distance_constraint& c)
	{ return { a.scene->checkId(mi_add_constraint(a.scene->world, MI_CONSTRAINT_DISTANCE, a.body(), b.body(), &c), "addConstraint") }; }
	inline ball_constraint_handle addConstraint(scene_entity& a, scene_entity& b, const ball_constraint& c)
	{ return { a.scene->checkId(mi_add_constraint(a.scene->world, MI_CONSTRAINT_BALL, a.body(), b.body(), &c), "addConstraint") }; }
	inline fixed_constraint_handle addConstraint(scene_entity& a, scene_entity& b, const fixed_constraint& c)
	{ return { a.scene->checkId(mi_add_constraint(a.scene->world, MI_CONSTRAINT_FIXED, a.body(), b.body(), &c), "addConstraint") }; }
	inline hinge_constraint_handle addConstraint(scene_entity& a, scene_entity& b, const hinge_constraint& c)
	{ return { a.scene->checkId(mi_add_constraint(a.scene->world, MI_CONSTRAINT_HINGE, a.body(), b.body(), &c), "addConstraint") }; }
	inline cone_twist_constraint_handle addConstraint(scene_entity& a, scene_entity& b, const cone_twist_constraint& c)
	{ return { a.scene->checkId(mi_add_constraint(a.scene->world, MI_CONSTRAINT_CONE_TWIST, a.body(), b.body(), &c), "addConstraint") }; }
	inline slider_constraint_handle addConstraint(scene_entity& a, scene_entity& b, const slider_constraint& c)
	{ return { a.scene->checkId(mi_add_constraint(a.scene->world, MI_CONSTRAINT_SLIDER, a.body(), b.body(), &c), "addConstraint") }; }
	inline slider_constraint_handle addSliderConstraintFromGlobalPoints(scene_entity& a, scene_entity& b, vec3 globalAnchor, vec3 globalAxis, float minLimit = 1.f, float maxLimit = -1.f)
	{ return { a.scene->checkId(mi_add_slider_constraint_global(a.scene->world, a.body(), b.body(), &globalAnchor.x, &globalAxis.x, minLimit, maxLimit), "addSliderConstraint") }; }

	// ---- getConstraint(scene, handle), physics.h:248-253: write-back proxy (see the header comment)
	template <typename T> struct constraint_ref
	{
		constraint_ref(game_scene& s, uint32_t id_) : scene(s), id(id_) { scene.check(mi_constraint_get(scene.world, constraint_type_of<T>::value, id, &value), "getConstraint"); std::memcpy(&loaded, &value, sizeof(T)); }
		~constraint_ref() { if (std::memcmp(&loaded, &value, sizeof(T)) != 0) mi_constraint_set(scene.world, constraint_type_of<T>::value, id, &value); }
		constraint_ref(const constraint_ref&) = delete;
		T* operator->() { return &value; }
		T& operator*() { return value; }
		game_scene& scene; uint32_t id; T value, loaded;
	};
	template <typename T> inline constraint_ref<T> getConstraint(game_scene& scene, constraint_handle<T> handle) { return constraint_ref<T>(scene, handle.id); }
	template <typename T> inline void deleteConstraint(game_scene& scene, constraint_handle<T> handle) { scene.check(mi_delete_constraint(scene.world, constraint_type_of<T>::value, handle.id), "deleteConstraint"); }
	inline void deleteAllConstraints(game_scene& scene) { scene.check(mi_delete_all_constraints(scene.world), "deleteAllConstraints"); }

	inline void deleteAllConstraintsFromEntity(scene_entity& entity) { entity.scene->check(mi_delete_all_constraints_from_body(entity.scene->world, entity.body()), "deleteAllConstraintsFromEntity"); } // physics.h:264
	struct ray { vec3 origin, direction; };
	// void testPhysicsInteraction(game_scene&, ray, float strength = 1000.f), physics.h:404
	inline void testPhysicsInteraction(game_scene& scene, ray r, float strength = 1000.f) { scene.flushStaticColliders(); mi_test_physics_interaction(scene.world, &r.origin.x, &r.direction.x, strength); }

	// ---- void physicsStep(game_scene&, memory_arena&, float& timer, const physics_settings&, float dt), physics.h:405
	inline void physicsStep(game_scene& scene, memory_arena& /*arena*/, float& timer, const physics_settings& settings, float dt)
	{
		scene.flushStaticColliders();
		mi_physics_settings s{};
		s.fixedFrameRate = settings.fixedFrameRate; s.frameRate = settings.frameRate; s.maxPhysicsIterationsPerFrame = settings.maxPhysicsIterationsPerFrame;
		s.numRigidSolverIterations = settings.numRigidSolverIterations;
		s.numClothVelocityIterations = settings.numClothVelocityIterations; s.numClothPositionIterations = settings.numClothPositionIterations; s.numClothDriftIterations = settings.numClothDriftIterations;
		s.simdBroadPhase = settings.simdBroadPhase; s.simdNarrowPhase = settings.simdNarrowPhase; s.simdConstraintSolver = settings.simdConstraintSolver;
		int mask = (settings.collisionBeginCallback ? 1 : 0) | (settings.collisionEndCallback ? 2 : 0);
		if (mask != scene.collisionEventMask) { scene.check(mi_enable_collision_events(scene.world, mask & 1, mask & 2), "mi_enable_collision_events"); scene.collisionEventMask = mask; }
		scene.check(mi_step(scene.world, &timer, &s, dt), "physicsStep");
		scene.dispatchEvents(settings);
	}
}
