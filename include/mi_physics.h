/* mi_physics.h — C-ABI of the MI355X-native rigid-body stepper (libmi_physics.so).
 *
 * Drop-in boundary for the `src/physics` hot path of study-game-engines/directx-renderer-kurth: every entry point
 * below names the reference interface it replaces (file:line under /root/reference/src).  Plain pointers and sizes
 * only; all host pointers are caller-owned and only touched during the call; a world owns its device memory and one
 * HIP stream; a world is not re-entrant (the reference's physics step is single-threaded per scene).
 * Every function returning `int` returns MI_OK (0) or an MI_ERR_* code instead of the reference's ASSERT/__debugbreak
 * (pch.h:33-34); mi_last_error() gives the text.  There is NO CPU fallback: without a usable HIP device
 * mi_world_create() fails with MI_ERR_NO_DEVICE.
 *
 * The existing C-ABI precedent in the reference is the Physics-Lib DLL (learning/learned_locomotion.cpp:395-489,
 * premake5.lua:388-462): global singletons + caller-owned float buffers; we keep caller-owned buffers and replace
 * the singletons by an opaque handle.
 */
#ifndef MI_PHYSICS_H
#define MI_PHYSICS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_world mi_world;

enum { MI_OK = 0, MI_ERR_NO_DEVICE = 1, MI_ERR_INVALID_ARGUMENT = 2, MI_ERR_HIP = 3, MI_ERR_CAPACITY = 4, MI_ERR_UNSUPPORTED = 5, MI_ERR_INVALID_STATE = 6 };

/* collider_type, physics.h:58-70 — the enum order is load-bearing (pair kernels are bucketed by it). */
enum { MI_COLLIDER_SPHERE = 0, MI_COLLIDER_CAPSULE = 1, MI_COLLIDER_CYLINDER = 2, MI_COLLIDER_AABB = 3, MI_COLLIDER_OBB = 4, MI_COLLIDER_HULL = 5 };
/* constraint_type, constraints.h:14-28 */
enum { MI_CONSTRAINT_DISTANCE = 0, MI_CONSTRAINT_BALL = 1, MI_CONSTRAINT_FIXED = 2, MI_CONSTRAINT_HINGE = 3, MI_CONSTRAINT_CONE_TWIST = 4, MI_CONSTRAINT_SLIDER = 5 };
/* constraint_motor_type, constraints.h:39-43 */
enum { MI_MOTOR_VELOCITY = 0, MI_MOTOR_POSITION = 1 };

#define MI_STATIC_BODY 0xFFFFFFFFu

/* physics_material (physics.h:40-47) without the sound-selection tag. */
typedef struct mi_material { float restitution, friction, density; } mi_material;

/* physics_settings (physics.h:382-397), field for field, minus the two std::function callbacks. */
typedef struct mi_physics_settings
{
	uint32_t fixedFrameRate;               /* bool */
	uint32_t frameRate;                    /* 120 */
	uint32_t maxPhysicsIterationsPerFrame; /* 4 */
	uint32_t numRigidSolverIterations;     /* 30 */
	uint32_t numClothVelocityIterations, numClothPositionIterations, numClothDriftIterations; /* iteration counts of the cloth solver (cloth.cpp:331-347) */
	uint32_t simdBroadPhase, simdNarrowPhase, simdConstraintSolver;                           /* accepted, unused: there is one (HIP) path */
} mi_physics_settings;

typedef struct mi_world_desc
{
	int32_t device;            /* HIP device ordinal; -1 = current device */
	uint32_t reserveBodies;    /* capacity hints, 0 = grow on demand */
	uint32_t reserveColliders;
	uint32_t reservePairs;
} mi_world_desc;

/* Counters of the last internal step — the reference's CPU_PROFILE_STATs (physics.cpp:1258-1262) plus per-stage times. */
typedef struct mi_stats
{
	uint32_t numRigidBodies, numColliders, numBroadphaseOverlaps, numCollisions, numContacts;
	uint32_t numColors, numJoints, numInternalSteps;
	uint32_t numGraphBuilds, coloringRounds; /* solver-sweep hipGraph (re)builds so far; colouring round budget of the last step */
	uint32_t flowProbes;                     /* unused (kept for layout compatibility) */
	uint32_t numFlowRecoveries;              /* steps whose cluster contact sweep gave up and was redone with the launch sweep (should stay 0) */
	float msCollidersBroad, msNarrow, msSolverSetup, msSolve, msIntegrate, msTotal; /* HIP-event times, mean over the timed steps since the previous mi_get_stats (timing enabled only) */
	float avgContacts, avgCollisions, avgColors, avgBroadphaseOverlaps, avgFlowProbes; /* means over the internal steps since the previous mi_get_stats */
	uint32_t avgSteps;                                                                 /* ... and how many steps that was */
	/* cluster contact sweep of the last step (all 0 after a launch-sweep step): tasks and manifolds per phase ([4] = the rest task),
	 * bodies handed between tasks (summed over the tasks that touch them), partition phases prepared */
	uint32_t clusterTasks[5], clusterManifolds[5], clusterSharedBodies, clusterParts;
	uint32_t numNarrowphaseRedone;           /* steps whose pair list + narrowphase, launched before the host knew the pair count, had to be launched again (the count jumped by more than 12 %) */
} mi_stats;

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
mi_world* mi_world_create(const mi_world_desc* desc);                 /* replaces game_scene + memory_arena ownership (physics.cpp:1205,1361) */
void mi_world_destroy(mi_world* w);
/* Checkpoint / resume (the engine's scene files — serialization_yaml.cpp:72-230, serialization_binary.cpp:225-260 —
 * are asset formats and out of scope): a self-contained binary image of the world (bodies with current state and mass properties,
 * colliders, hull geometries, joints).  A world restored from it continues bit-identically. */
uint64_t mi_snapshot_size(mi_world* w);
int mi_snapshot_save(mi_world* w, void* buffer, uint64_t capacity);
mi_world* mi_world_restore(const mi_world_desc* desc, const void* buffer, uint64_t size); /* NULL on error: mi_last_error(NULL) */
const char* mi_last_error(mi_world* w);                               /* w may be NULL for create-time errors */

/* ---- add API --------------------------------------------------------------------------------------------------- */
/* entity.addComponent<rigid_body_component>(kinematic, gravityFactor, linearDamping=.4, angularDamping=.4) on an entity with
 * transform {pos, rot}: rigid_body.h:21, rigid_body.cpp:6-27, scene.h:69-84.  Returns the body index (= add order). */
uint32_t mi_add_body(mi_world* w, int kinematic, float gravityFactor, float linearDamping, float angularDamping, const float pos[3], const float rot[4]);
/* Convex hull geometry shared by hull colliders: replaces allocateBoundingHullGeometry(meshFilepath) (physics.h:207, physics.cpp:58-84)
 * from the mesh on — loading a model file belongs to the asset pipeline — i.e. bounding_hull_geometry::fromMesh(vertices, triangles)
 * (bounding_volumes.cpp:1394-1452).  Triangles face outwards.  Returns the geometry index (INVALID = 0xFFFFFFFF). */
uint32_t mi_add_hull_geometry(mi_world* w, const float* vertices3, uint32_t numVertices, const uint32_t* triangles3, uint32_t numTriangles);
/* entity.addComponent<collider_component>(collider_component::as{Sphere,Capsule,Cylinder,AABB,OBB,Hull}(shape, material)): physics.h:110-157,
 * scene.h:38-63 (recomputes the body's mass properties, rigid_body.cpp:29-81).  shape = up to 10 floats in the parent's local space:
 * sphere c3,r | capsule/cylinder A3,B3,r | aabb min3,max3 | obb quat4,center3,radius3 | hull quat4,position3,geometry index (as a float
 * value).  Returns the collider index. */
uint32_t mi_add_collider(mi_world* w, uint32_t body, uint32_t type, const float* shape, const mi_material* material);
/* Collider on an entity without a rigid body (static_collider, physics.cpp:667-671); {pos, rot} is that entity's transform. */
uint32_t mi_add_static_collider(mi_world* w, uint32_t type, const float* shape, const mi_material* material, const float pos[3], const float rot[4]);

/* add{Distance,Ball,Fixed,Hinge,ConeTwist,Slider}ConstraintFrom{Local,Global}Points: physics.h:209-235, physics.cpp:128-333.
 * Global variants derive the local anchors/axes from the bodies' CURRENT transforms.  Return the per-type constraint id. */
uint32_t mi_add_distance_constraint_local(mi_world* w, uint32_t a, uint32_t b, const float localAnchorA[3], const float localAnchorB[3], float distance);
uint32_t mi_add_distance_constraint_global(mi_world* w, uint32_t a, uint32_t b, const float globalAnchorA[3], const float globalAnchorB[3]);
uint32_t mi_add_ball_constraint_local(mi_world* w, uint32_t a, uint32_t b, const float localAnchorA[3], const float localAnchorB[3]);
uint32_t mi_add_ball_constraint_global(mi_world* w, uint32_t a, uint32_t b, const float globalAnchor[3]);
uint32_t mi_add_fixed_constraint_global(mi_world* w, uint32_t a, uint32_t b, const float globalAnchor[3]);
uint32_t mi_add_hinge_constraint_global(mi_world* w, uint32_t a, uint32_t b, const float globalAnchor[3], const float globalHingeAxis[3], float minLimit, float maxLimit);
uint32_t mi_add_cone_twist_constraint_global(mi_world* w, uint32_t a, uint32_t b, const float globalAnchor[3], const float globalAxis[3], float swingLimit, float twistLimit);
uint32_t mi_add_slider_constraint_global(mi_world* w, uint32_t a, uint32_t b, const float globalAnchor[3], const float globalAxis[3], float minLimit, float maxLimit);
/* addConstraint(scene_entity& a, scene_entity& b, const T& c) for the six constraint types (physics.h:239-244): the constraint is given
 * as the reference's POD (the layouts mi_constraint_get returns: 28/24/40/104/120/72 bytes), as the deserialisers hold it
 * (serialization_binary.cpp:237-260).  type = MI_CONSTRAINT_*.  Returns the constraint id within its type, 0xFFFFFFFF on error. */
uint32_t mi_add_constraint(mi_world* w, uint32_t type, uint32_t a, uint32_t b, const void* pod);

/* T& getConstraint(scene, handle) (physics.h:248-253) as get/set of the POD whose layout is byte-identical to the reference's
 * distance/ball/fixed/hinge/cone_twist/slider_constraint structs (constraints.h:73-80,129-135,175-183,229-257,346-380,497-520;
 * 28/24/40/104/120/72 bytes).  Motors are driven by writing fields, as learned_locomotion.cpp:73-91 does. */
int mi_constraint_get(mi_world* w, uint32_t type, uint32_t id, void* pod);
int mi_constraint_set(mi_world* w, uint32_t type, uint32_t id, const void* pod);
int mi_delete_constraint(mi_world* w, uint32_t type, uint32_t id);    /* deleteConstraint, physics.h:257-262 */
int mi_delete_all_constraints(mi_world* w);                           /* deleteAllConstraints, physics.h:255 */
int mi_delete_all_constraints_from_body(mi_world* w, uint32_t body);  /* deleteAllConstraintsFromEntity, physics.h:264 */
/* Deleting the entity of a rigid body (scene.deleteEntity: body, colliders and constraints go away; the colliders leave the sweep,
 * collision_broad.cpp:42-75).  Indices are add-order positions and stay valid; the body is switched off for good. */
int mi_delete_body(mi_world* w, uint32_t body);

/* ---- force fields, triggers, collision events (row N2 of SURVEY §8f) --------------------------------------------------------------
 * force_field_component (physics.h:182-185): `force` is rotated by the entity's transform when it has one (pass pos/rot, or NULL for
 * none; physics.cpp:767-771).  A field without colliders acts on every body (global, :782, :1273); a field with colliders acts on
 * the rigid bodies whose colliders overlap them (:963-967).  A body inside several localized fields receives their forces in
 * ascending field id.  Returns the field id. */
uint32_t mi_add_force_field(mi_world* w, const float force[3], const float pos[3], const float rot[4]);
int mi_set_force_field(mi_world* w, uint32_t field, const float force[3]);
/* trigger_component (physics.h:200-203): the std::function callback becomes mi_drain_events.  Returns the trigger id. */
uint32_t mi_add_trigger(mi_world* w, const float pos[3], const float rot[4]);
/* Colliders of a force-field / trigger entity (collider_component on that entity, physics.cpp:657-666): shape payload as in
 * mi_add_collider, in the entity's local space; no material.  Return the collider id. */
uint32_t mi_add_force_field_collider(mi_world* w, uint32_t field, uint32_t type, const float* shape);
uint32_t mi_add_trigger_collider(mi_world* w, uint32_t trigger, uint32_t type, const float* shape);
/* Moving a force-field / trigger entity (writing its transform_component): its colliders follow, a field's force is rotated by the new
 * rotation.  Takes effect at the next step. */
int mi_set_force_field_transform(mi_world* w, uint32_t field, const float pos[3], const float rot[4]);
int mi_set_trigger_transform(mi_world* w, uint32_t trigger, const float pos[3], const float rot[4]);
/* physics_settings::collisionBeginCallback / collisionEndCallback (physics.h:394-395) set or not.  Enable before the first step:
 * the previous step's collision set is only kept while one of the two is on. */
int mi_enable_collision_events(mi_world* w, int begin, int end);
/* trigger_event (physics.h:193-198), collision_begin_event / collision_end_event (physics.h:356-376) as one record.
 * kind: MI_EVENT_*; step: internal step that raised it; trigger events: a = trigger id, b = bodyB = body id;
 * collision events: a, b = collider ids in contact-normal order, bodyA / bodyB = their bodies (MI_STATIC_BODY for a static
 * collider); begin events carry the mean contact point, mean normal and the relative velocity (B - A) at that point. */
enum { MI_EVENT_TRIGGER_ENTER = 0, MI_EVENT_TRIGGER_LEAVE = 1, MI_EVENT_COLLISION_BEGIN = 2, MI_EVENT_COLLISION_END = 3 };
typedef struct mi_event { uint32_t kind, step, a, b, bodyA, bodyB; float position[3], normal[3], relativeVelocity[3]; } mi_event;
/* Copies out up to `capacity` pending events in the order the reference calls back in (per step: trigger events, then collision
 * events, each sorted by pair) and returns how many; call again until it returns 0.  Synchronises with the device.
 * If the device-side event ring overflowed since the last drain, mi_last_error() says so and the call after this one fails with
 * MI_ERR_CAPACITY (events were lost). */
uint32_t mi_drain_events(mi_world* w, mi_event* out, uint32_t capacity);

/* ---- heightmap terrain (row N4 of SURVEY §8f): heightmap_collider_component(chunksPerDim, chunkSize, material) + update(minCorner,
 * amplitudeScale), heightmap_collider.h:127-152.  chunksPerDim x chunksPerDim chunks, each 129 x 129 uint16 heights (row-major, z rows)
 * spanning chunkSize metres; height = minCorner.y + h / 65535 * amplitudeScale.  Every internal step, after the narrowphase, each
 * sphere / capsule / AABB / OBB collider of a rigid body is collided with the triangles under it, plus one contact if its lowest point
 * is below the surface (heightmap_collision.cpp:522-618; cylinders and hulls have no terrain case in the reference either).
 * Terrain contacts raise no collision events (physics.cpp:1049). */
int mi_set_heightmap(mi_world* w, uint32_t chunksPerDim, float chunkSize, const mi_material* material, const float minCorner[3], float amplitudeScale);
int mi_heightmap_set_chunk(mi_world* w, uint32_t x, uint32_t z, const uint16_t* heights129x129); /* heightmap_collider_chunk::setHeights; chunks never set collide with nothing */
int mi_heightmap_update(mi_world* w, const float minCorner[3], float amplitudeScale);
float mi_heightmap_height_at(mi_world* w, float x, float z);   /* getHeightAt: -FLT_MAX outside the terrain */

/* ---- cloth (row N4 of SURVEY §8f): cloth_component(width, height, gridSizeX, gridSizeY, totalMass, stiffness = .5, damping = .3,
 * gravityFactor = 1), cloth.h:8-9.  A grid of particles in the cloth's local frame (x across, -z down, upper row locked) with stretch /
 * shear / bend distance constraints; every internal step, after the rigid bodies, it receives the global force field as wind and runs
 * numCloth{Velocity,Position,Drift}Iterations of mi_physics_settings (physics.cpp:1354-1358).  The constraints are solved in a fixed
 * colour order (12 colours), not the reference's storage order.  Returns the cloth id. */
uint32_t mi_add_cloth(mi_world* w, float width, float height, uint32_t gridSizeX, uint32_t gridSizeY, float totalMass, float stiffness, float damping, float gravityFactor);
/* cloth_component::setWorldPositionOfFixedVertices(transform, moveRigid), cloth.h:17: puts the locked upper row at `transform`;
 * moveRigid also carries the rest of the cloth along rigidly. */
int mi_cloth_set_fixed_vertices(mi_world* w, uint32_t cloth, const float pos[3], const float rot[4], int moveRigid);
/* The public fields totalMass / stiffness / damping / gravityFactor (cloth.h:21-24); mass and stiffness changes re-derive the
 * particle and constraint masses at the next step (cloth.cpp:198-204, 331-347). */
int mi_cloth_set_properties(mi_world* w, uint32_t cloth, float totalMass, float stiffness, float damping, float gravityFactor);
/* Cloth iteration counts for mi_step_internal (mi_step takes them from its settings).  Default 0, 1, 0 (physics.h:387-389). */
int mi_set_cloth_iterations(mi_world* w, uint32_t velocityIterations, uint32_t positionIterations, uint32_t driftIterations);
uint32_t mi_num_cloths(mi_world* w);
uint32_t mi_cloth_num_particles(mi_world* w, uint32_t cloth);
/* Particle positions / velocities, row-major over the grid, n x 3 floats each (either pointer may be NULL). */
int mi_cloth_read(mi_world* w, uint32_t cloth, float* positions3, float* velocities3);

/* void testPhysicsInteraction(game_scene&, ray, float strength = 1000.f): physics.h:404, physics.cpp:556-628 — the closest rigid-body
 * collider hit by the ray gets force = direction * strength at the hit point.  Returns 1 + the index of the body that was
 * pushed, 0 when the ray hits nothing (this one function does not return a status code). */
int mi_test_physics_interaction(mi_world* w, const float origin[3], const float direction[3], float strength);
/* rigid_body_component::{forceAccumulator,torqueAccumulator} += (testPhysicsInteraction applies them the same way, physics.cpp:624-628). */
int mi_apply_force_torque(mi_world* w, uint32_t body, const float force[3], const float torque[3]);
int mi_set_velocity(mi_world* w, uint32_t body, const float linear[3], const float angular[3]);
int mi_set_transform(mi_world* w, uint32_t body, const float pos[3], const float rot[4]);
/* Bulk forms of the two setters above for the first n bodies: n x {pos3, quat4} / n x {linear3, angular3} (scene deserialisation,
 * serialization_binary.cpp:237-260, and state hand-over between worlds). */
int mi_write_transforms(mi_world* w, const float* in7, uint32_t n);
int mi_write_velocities(mi_world* w, const float* in6, uint32_t n);

/* ---- per-frame -------------------------------------------------------------------------------------------------- */
/* void physicsStep(game_scene&, memory_arena&, float& timer, const physics_settings&, float dt): physics.h:405, physics.cpp:1364-1413.
 * The arena argument has no counterpart (per-step arrays live in device memory owned by the world). */
int mi_step(mi_world* w, float* timer, const mi_physics_settings* settings, float dt);
/* One physicsStepInternal (physics.cpp:1180-1362) at exactly dt — what bench.py and the parity tests time and compare. */
int mi_step_internal(mi_world* w, float dt, uint32_t numRigidSolverIterations);
int mi_synchronize(mi_world* w);                                       /* waits for the world's stream */

/* ---- results (transform_component / physics_transform1 / rigid_body_component.{v,w} write-back, physics.cpp:1342-1347,1399-1411) */
/* which: 0 = interpolated transform_component, 1 = physics_transform1, 2 = physics_transform0.  out = n x {pos3, quat4}. */
int mi_read_transforms(mi_world* w, uint32_t which, float* out7, uint32_t n);
int mi_read_velocities(mi_world* w, float* out6, uint32_t n);          /* n x {linear3, angular3} */
int mi_read_mass_properties(mi_world* w, float* out13, uint32_t n);    /* n x {localCOG3, invMass, invInertia9 (column-major)} */
int mi_get_stats(mi_world* w, mi_stats* out);
/* Debug guard: the reference's VALIDATE macros (physics.cpp:807-926, compiled out there): with the guard on, every stage's output is
 * scanned for NaN / Inf on the device and the step after the one that produced one fails with MI_ERR_INVALID_STATE (the message names
 * the stage and the first offending element).  Also switched on by MI_PHYSICS_VALIDATE=1.  Costs a few small launches per step. */
int mi_enable_validation(mi_world* w, int enable);
int mi_enable_stage_timing(mi_world* w, int enable);
uint32_t mi_num_bodies(mi_world* w);
uint32_t mi_num_colliders(mi_world* w);

/* ---- device-resident access for multi-GPU halo exchange and zero-copy callers ------------------------------------- */
/* Raw device pointers (valid until the next add call): pose = 2 x float4 per body {pos.xyz,0},{quat}; vel = 2 x float4 per body
 * {v.xyz, invMass},{w.xyz,0}.  The caller may read/write them on `stream` between steps (ghost-body refresh). */
int mi_device_pointers(mi_world* w, void** pose, void** vel, void** stream);

/* Spatial-slab runs (one world per GPU holding ALL bodies, each simulating its slab + ghosts): copy the whole pose / velocity arrays
 * (layout as mi_device_pointers) to / from caller-owned DEVICE buffers, and set the per-body simulate mask (1 byte per body, device
 * memory; 0 = body lives on another GPU: no AABB, no integration).  The halo exchange itself (RCCL send/recv of boundary bodies) is
 * host-side logic above this ABI; the reference has no counterpart (single process). */
int mi_state_to_device_buffers(mi_world* w, void* dPose, void* dVel);
int mi_state_from_device_buffers(mi_world* w, const void* dPose, const void* dVel, const uint8_t* dMask);

/* Ghost-body halo of a spatial slab, on the device (the reference has no counterpart).  mi_slab_configure makes this
 * world rank `rank` of `size` slabs along `axis` owning [lo, hi) (use -/+INFINITY at the ends) with ghosts within `margin` of a cut,
 * and classifies every body from its current pose (owned / ghost of the left or right neighbour / inactive).  Once per step, before
 * mi_step_internal:
 *   mi_slab_pack    one kernel: every owned body within `margin` of a cut goes into that neighbour's message; one that has crossed
 *                   the cut goes with MI_SLAB_MIGRATE and becomes a ghost here.  A message is a caller-owned DEVICE buffer of
 *                   mi_slab_message_bytes(capacity) bytes: a 16-byte header {count, dropped, 0, 0} + `capacity` records of 72 bytes
 *                   {body index, flag, pose 2 x float4, velocity 2 x float4}.  Fixed capacity: no size round trip; `dropped` != 0
 *                   means the band held more bodies than records (raise the capacity).  Pass NULL for a missing neighbour.
 *   (the caller exchanges the messages: one send + one receive per neighbour, e.g. RCCL over xGMI, on the stream of mi_device_pointers)
 *   mi_slab_unpack  two kernels: the received bodies are written (owner if MI_SLAB_MIGRATE, else ghost), ghosts the neighbour no longer
 *                   sends go inactive, the simulate mask follows.
 * Nothing here synchronises with the host.  mi_slab_read_codes copies the per-body codes out (tests / assembling results). */
enum { MI_SLAB_INACTIVE = 0, MI_SLAB_OWNED = 1, MI_SLAB_GHOST_LEFT = 2, MI_SLAB_GHOST_RIGHT = 3, MI_SLAB_GHOST = 0, MI_SLAB_MIGRATE = 1 };
int mi_slab_configure(mi_world* w, uint32_t rank, uint32_t size, uint32_t axis, float lo, float hi, float margin);
uint64_t mi_slab_message_bytes(uint32_t capacity);
int mi_slab_pack(mi_world* w, void* dMessageLeft, void* dMessageRight, uint32_t capacity);
int mi_slab_unpack(mi_world* w, const void* dMessageLeft, const void* dMessageRight, uint32_t capacity);
int mi_slab_read_codes(mi_world* w, uint8_t* outCodes, uint32_t n);

/* ---- inspection of the last internal step (parity tests; mirrors the arrays of physics.cpp:1207-1228) ---------------- */
uint32_t mi_debug_num_pairs(mi_world* w);
int mi_debug_read_pairs(mi_world* w, uint32_t* outPairs2);                              /* broadphase overlaps, (A,B) collider indices */
/* sap_context::sortingAxis (collision_broad.cpp:20-24, 443-444): out[0] = the axis the last step's sweep order refers to (it orients
 * equal-type pairs: A = the collider whose box starts first on it, collision_broad.cpp:127 + collision_narrow.cpp:2374), out[1] = the
 * axis of largest AABB-centre variance of the last step = the next step's axis. */
int mi_debug_sorting_axis(mi_world* w, uint32_t out[2]);
int mi_debug_read_world_colliders(mi_world* w, void* outColliders64, float* outAabbs6); /* worldSpaceColliders / worldSpaceAABBs */
uint32_t mi_debug_num_manifold_slots(mi_world* w);
/* Per candidate pair after prune/classify/bucket (collision_narrow.cpp:2346-2453): ordered collider pair, contact count, and up to 4
 * contacts in the reference's 32-byte collision_contact layout (physics.h:347-354). */
int mi_debug_read_manifolds(mi_world* w, uint32_t* outPairs2, uint32_t* outCounts, void* outContacts4x32, uint32_t* outBodyPairs2);
/* Gauss-Seidel schedule of the contact solve: manifold slots in execution order + colour boundaries. */
uint32_t mi_debug_num_colors(mi_world* w);
int mi_debug_read_schedule(mi_world* w, uint32_t* outManifoldSlots, uint32_t* outColorStart /* numColors+1 */);
int mi_debug_read_joint_order(mi_world* w, uint32_t type, uint32_t* outJointIds);
int mi_debug_read_body_state(mi_world* w, float* outCog4, float* outInvInertia12, uint32_t nPlusOne); /* rbGlobal: {cog.xyz, invMass}, 3 x float4 columns */
/* Replay of the reference's own Gauss-Seidel order (SURVEY section 7 / 8c "replay mode"): on != 0 makes the following steps run the reference's
 * greedy 8-wide batch scheduler (scheduleConstraintsSIMD, constraints.cpp:51-184) over the step's contacts in emission order and sweep
 * the batches one after the other (constraints.cpp:3618-3709) instead of the device's own schedule.  A parity facility (one workgroup
 * sweeps all contacts); mi_debug_read_replay_batches returns the last step's batches, 8 entries each: schedule position | contact << 28,
 * 0xFFFFFFFF = empty lane. */
int mi_debug_set_replay(mi_world* w, int on);
uint32_t mi_debug_num_replay_batches(mi_world* w);
int mi_debug_read_replay_batches(mi_world* w, uint32_t* outEntries8PerBatch);
/* Developer timeline of the cluster contact sweep: enable != 0 allocates it (16 rows of 32 u64 per workgroup of the solve launch), out (may be
 * NULL) receives numSlots rows: per task and iteration the wall-clock stamps (10 ns ticks) "shared bodies acquired" / "colours done", the cost of
 * every colour step of iteration 10, and the stages of the task's colouring (csrc/k_cluster.hip documents the rows; tests/cluster_timeline.py prints them). */
int mi_debug_flow_trace(mi_world* w, int enable, unsigned long long* out, uint32_t numSlots);

#ifdef __cplusplus
}
#endif
#endif
