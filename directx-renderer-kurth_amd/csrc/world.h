// Host-side World of the MI355X rigid-body stepper: owns the SoA device buffers (HBM layout in DESIGN.md), one HIP
// stream, and the per-step launch sequence.  The C-ABI in include/mi_physics.h is a thin shell over this class.
#pragma once
#include "mi_common.h"
#include "../../include/mi_physics.h"
#include <vector>
#include <string>
#include <algorithm>

// ---- persistent joint PODs: byte-for-byte the reference's structs (constraints.h:73-80,129-135,175-183,229-257,346-380,497-520)
struct mi_distance_constraint { float localAnchorA[3], localAnchorB[3], globalLength; };
struct mi_ball_constraint { float localAnchorA[3], localAnchorB[3]; };
struct mi_fixed_constraint { float initialInvRotationDifference[4], localAnchorA[3], localAnchorB[3]; };
struct mi_hinge_constraint
{
	float localAnchorA[3], localAnchorB[3], localHingeAxisA[3], localHingeAxisB[3];
	float minRotationLimit, maxRotationLimit, maxMotorTorque; u32 motorType; float motorVelocity;
	float localHingeTangentA[3], localHingeBitangentA[3], localHingeTangentB[3];
};
struct mi_cone_twist_constraint
{
	float localAnchorA[3], localAnchorB[3], localLimitAxisA[3], localLimitAxisB[3];
	float localLimitTangentA[3], localLimitBitangentA[3], localLimitTangentB[3];
	float swingLimit, twistLimit; u32 swingMotorType; float swingMotorVelocity, maxSwingMotorTorque, swingMotorAxis;
	u32 twistMotorType; float twistMotorVelocity, maxTwistMotorTorque;
};
struct mi_slider_constraint
{
	float initialInvRotationDifference[4], localAnchorA[3], localAnchorB[3], localAxisA[3];
	float negDistanceLimit, posDistanceLimit, maxMotorForce; u32 motorType; float motorVelocity;
};
static_assert(sizeof(mi_distance_constraint) == 28 && sizeof(mi_ball_constraint) == 24 && sizeof(mi_fixed_constraint) == 40, "POD layout");
static_assert(sizeof(mi_hinge_constraint) == 104 && sizeof(mi_cone_twist_constraint) == 120 && sizeof(mi_slider_constraint) == 72, "POD layout");

static const u32 MI_JOINT_TYPES = 6;
static const u32 MI_JOINT_POD_SIZE[MI_JOINT_TYPES] = { 28, 24, 40, 104, 120, 72 };
// Per-joint solver scratch ("update" record) in floats, by type; laid out by the kernels in k_joints.hip.
static const u32 MI_JOINT_UPDATE_FLOATS[MI_JOINT_TYPES] = { 20, 20, 36, 56, 80, 72 };

template <typename T> struct DevBuf
{
	T* p = nullptr; size_t cap = 0;
	void ensure(size_t n, hipStream_t s, bool keep = false);
	void release();
};

struct JointSet
{
	std::vector<uint8_t> pods;            // host, add order
	std::vector<u32> a, b;                // body ids
	std::vector<uint8_t> alive;
	// colour-sorted device view
	DevBuf<uint8_t> dPods; DevBuf<uint2> dPairs; DevBuf<float> dUpdate;
	std::vector<u32> order;               // sorted slot -> joint id
	std::vector<u32> colorStart;          // size numColors+1
	u32 count() const { return (u32)a.size(); }
};

// Device-side step counters (u32 words of World::dCounters), copied to pinned host memory twice per step.
enum
{
	CTR_NUM_PAIRS = 0,      // broadphase overlaps
	CTR_NUM_VALID = 1,      // candidate pairs that survive prune/classify (= manifold slots)
	CTR_NUM_MANIFOLDS = 2,  // slots with >= 1 contact
	CTR_EPA_COUNT = 3,      // GJK hits waiting for EPA
	CTR_NUM_COLORS = 4,
	CTR_FIRST_LARGE = 5,    // sorted position of the first "large" collider
	CTR_LAST_ROUND = 6,     // last colouring round that coloured anything
	CTR_OVERFLOW = 7,       // manifolds sent to the serial bucket because the round budget ran out
	CTR_CELL_SIZE = 8,      // float bits: largest extent of a collider riding on a rigid body
	CTR_NUM_ACTIVE = 9,     // append cursor of the active-manifold list
	CTR_NUM_CONTACTS = 10,  // sum of contact counts over the active manifolds
	CTR_PAIR_OVERFLOW = 11, // some collider has more broadphase partners than its slab holds
	CTR_FIRST_INACTIVE = 12,// sorted position of the first collider whose body is simulated by another GPU
	CTR_FLOW_STATUS = 13,   // cluster sweep: non-zero = the launch could not run or a lane gave up waiting (result invalid): the host redoes the step
	CTR_EPA_COUNT_HULL = 15,// GJK hits among the hull pairs (their EPA work list grows from the end of epaList)
	CTR_FLOW_PROBES = 14,   // unused (kept: the statistics layout has the word)
	CTR_BUCKET_START = 16,  // 65 words: first slot of narrowphase bucket key b (tA*6+tB); [64] unused
	CTR_KEY_START = 96,     // (MI_MAX_COLORS+1)*4 + 1 words: first schedule slot of key colour*4 + (4-count); last = numManifolds
	CTR_COLOR_BARRIER = 360,// 4 words: grid barrier of the fused colouring kernel (arrivals, 3 x manifolds left)
	CTR_SAP_AXIS = 368,     // 2 words: the reference sweep's sorting axis (collision_broad.cpp:443-444) for internal step k at [k & 1]: written by step k - 1 from its AABB centres, read by k_classify
	CTR_SAP_MASKED = 370,   // colliders with an empty AABB (bodies simulated by another GPU): not part of the axis statistic
	CTR_REGION_START = 384, // 9 words: first position in flowOrder of XCD region r; [8] = numManifolds
	CTR_REGION_CUTS = 400,  // 7 floats: region r holds bodies with cuts[r-1] <= x < cuts[r]
	CTR_REGION_RANGE = 408, // 2 floats: [lo, hi] of the histogram that produces the next cuts
	CTR_REGION_MINMAX = 410,// 2 words: running min / max of x (order-preserving integer encoding) for the next range
	CTR_EVENT_COUNT = 416,  // append cursor of the event ring (trigger enter/leave, collision begin/end), reset by mi_drain_events
	CTR_EVENT_OVERFLOW = 417,// bit 0: the event ring was full, events were dropped; bit 1: a pair-set table was full
	CTR_TERRAIN_BASE = 418, // first manifold slot of the terrain contacts (= number of pair manifold slots)
	CTR_TERRAIN_OVERFLOW = 419,// more terrain contacts than slots: contacts were dropped (the host fails the world)
	CTR_CL_NUM_TASKS = 424, // 5 words: cluster sweep: tasks of phase p (positions up to the last non-empty one)
	CTR_CL_STATUS = 429,    // cluster build: bit 0 = more tasks in a phase than the table holds, bit 1 = a task exceeds k_cl_color's tables
	CTR_CL_SHARED = 430,    // statistics: bodies handed between tasks (summed over tasks)
	CTR_CL_PHASE_COUNT = 431,// 5 words: statistics: manifolds per phase
	CTR_CL_BBOX = 436,      // 6 words: min xyz, max xyz of the simulated bodies' centres of gravity (order-preserving integer encoding)
	CTR_CL_LEFT = 372,      // 5 words: cluster build: manifolds left over by the curve phases (cursor of the list the component phase works on); statistics of the component phase: tasks, total weight, manifolds whose ends disagree, weight of the largest component sent to the rest task
	CTR_ACTIVE_BODIES = 377, // simulated bodies / their colliders + the static ones (lengths of the active lists, k_active_scan); more of the latter than the broadphase was launched for
	CTR_ACTIVE_COLS = 378,
	CTR_ACTIVE_OVERFLOW = 379,
	CTR_CELL_SIZE_USED = 380, // float bits: the cell size the current sorted order was built with (CTR_CELL_SIZE is reset for the next step's atomicMax before the last pair kernel runs)
	CTR_CL_SCRATCH = 371,   // cluster sweep: append cursor of the global row scratch (contacts that fit neither registers nor LDS), reset before every launch
	CTR_VALIDATE = 448,     // 2 words: non-finite values found by the debug guard (MI_PHYSICS_VALIDATE=1), first offender (stage << 28 | index)
	CTR_CL_REMAIN = 442,    // 6 words: manifolds still unassigned when partition phase p starts ([0] unused: all active ones)
	CTR_WORDS = 512,
};
// Cluster sweep (k_cluster.hip): up to CL_MAX_PARTS partition phases + the rest phase; task key = phase * CL_MAX_TASKS + task.
#define MI_REPLAY_WIDTH 8u // lanes of the reference's SIMD batches (constraints.cpp: CONSTRAINT_SIMD_WIDTH with AVX)
#define CL_MAX_PARTS 4u
#define CL_MAX_PHASES (CL_MAX_PARTS + 1u)
#define CL_MAX_TASKS 512u
#define CL_BODY_STRIDE 4096u
#define CL_MAX_JOINT_CLASSES 64u   // (type, colour) classes of joints the cluster sweep can run
#define CL_TASK_MAX_JOINTS 512u    // joints per task (one lane each)
#define MI_NUM_SCHEDULE_KEYS ((MI_MAX_COLORS + 1) * 4)

struct World
{
	int device = 0;
	hipStream_t stream = nullptr;
	int lastError = 0; std::string lastErrorText;

	// ---- host mirrors (add API) ----
	struct HBody { float pos[3], rot[4]; float localCOG[3], invMass, invInertia[9]; float gravityFactor, linDamp, angDamp; float v[3], w[3], force[3], torque[3]; std::vector<u32> colliders; bool removed = false; };
	struct HCollider { float shape[10]; float restitution, friction, density; u32 type, body; float spos[3], srot[4]; u32 zoneType, zoneIndex; }; // zoneType: 0 none, 2 force field, 3 trigger (physics_object_type, physics.h:49-57)
	struct HField { float force[3]; float pos[3], rot[4]; u32 hasTransform, numColliders; };   // force_field_component (physics.h:182-185) + the entity's transform
	struct HTrigger { float pos[3], rot[4]; u32 numColliders; };                                 // trigger_component (physics.h:200-203); the callback becomes mi_drain_events
	std::vector<HField> fields; std::vector<HTrigger> triggers;
	// heightmap_collider_component (heightmap_collider.h:127-152): chunksPerDim x chunksPerDim chunks of 129 x 129 uint16 heights
	u32 terrainChunksPerDim = 0, terrainSlotsPerCollider = 8, terrainMinSlots = 8192; float terrainChunkSize = 0.f, terrainAmplitude = 1.f, terrainMinCorner[3] = { 0.f, 0.f, 0.f }, terrainMaterial[3] = { 0.f, 0.f, 0.f };
	std::vector<uint16_t> hTerrainHeights; std::vector<u32> hTerrainValid;
	DevBuf<uint16_t> terrainHeights; DevBuf<u32> terrainValid, terrainCounts, terrainOffsets;
	u32 terrainSlotCap() const { return terrainChunksPerDim ? std::max(terrainMinSlots, terrainSlotsPerCollider * (u32)colliders.size()) : 0u; }
	u32 prevTruePairs = 0;                // broadphase overlaps of the last step (prevNumPairs counts the terrain slots too)
	bool prevSlabOverflow = false;        // CTR_PAIR_OVERFLOW of the last step
	hipEvent_t countersEvent = nullptr;   // behind the step's asynchronous read of the counters
	// cloth_component (cloth.h:5-60): parameters + host mirror of the particle state (authoritative until the first step; refreshed by downloadCloths)
	struct HClothConstraint { u32 a, b; float restDistance, inverseMassSum; u32 color; };
	struct HCloth
	{
		float width, height, totalMass, stiffness, damping, gravityFactor, oldTotalMass, oldStiffness; u32 gridX, gridY;
		std::vector<float> pos, prev, vel, invMass;          // 3 n, 3 n, 3 n, n
		std::vector<HClothConstraint> constraints;            // sorted by colour (stable)
	};
	std::vector<HCloth> cloths;
	bool clothsDirty = true, clothStateOnDevice = false; // dirty: the host mirror changed, upload before the next launch; onDevice: the device copy is newer than the mirror
	u32 clothIterations[3] = { 0, 1, 0 };  // physics_settings::numCloth{Velocity,Position,Drift}Iterations of the last mi_step (mi_set_cloth_iterations for mi_step_internal)
	DevBuf<float> clothPlanes; u32 clothStride = 0; // 10 planes of clothStride floats: position xyz, velocity xyz, previous position xyz, inverse mass
	DevBuf<uint2> clothAB; DevBuf<float2> clothRestIms; DevBuf<float4> clothTemp; DevBuf<uint8_t> clothDescs; DevBuf<u32> clothList;
	u32 numSmallCloths = 0, maxSmallClothParticles = 0; // cloths that fit the LDS variant come first in clothList
	void uploadCloths(); void downloadCloths();
	bool fieldsDirty = true;              // a force changed: re-upload the per-field world-space forces
	bool collisionBeginEvents = false, collisionEndEvents = false;
	std::vector<HBody> bodies;
	std::vector<HCollider> colliders;
	struct HHull { std::vector<float> vertices; std::vector<u32> triangles; float aabbMin[3], aabbMax[3]; }; // bounding_hull_geometry (bounding_volumes.h:208-218)
	std::vector<HHull> hulls;
	JointSet joints[MI_JOINT_TYPES];
	bool topologyDirty = true;   // bodies/colliders added since last upload
	bool jointsDirty = true;
	bool stateOnDevice = false;  // device holds the authoritative pose/velocity

	// ---- device buffers ----
	u32 nb = 0, nc = 0;
	DevBuf<float4> pose, pose0, poseLerp, vel, bprops, force, cog, invIw;
	DevBuf<ColliderRec> colLocal, colWorld;
	DevBuf<float4> colStaticPose, aabbMin, aabbMax;
	DevBuf<float4> hullVerts, hullInfo;   // all hull vertices (xyz); per geometry {aabbMin.xyz, firstVertex}, {aabbMax.xyz, vertexCount}
	DevBuf<uint8_t> aliveMask;            // per body: 0 = deleted (mi_delete_body); ANDed into every simulate mask handed in
	DevBuf<uint8_t> simMask;              // per body: 1 = simulated here (owned or ghost), 0 = lives on another GPU's slab
	// spatial slab (mi_slab_*): ownership code per body, stamp of the last refresh of a ghost, this rank's interval
	DevBuf<uint8_t> slabCode; DevBuf<u32> slabFresh; u32 slabRank = 0, slabSize = 0, slabAxis = 0, slabStamp = 0; float slabLo = 0.f, slabHi = 0.f, slabMargin = 0.f;
	// broadphase
	DevBuf<u32> cellCount, cellBase;      // colliders per cell bucket (+ 'large', 'simulated elsewhere'), first / end position of every bucket in the sorted order
	DevBuf<u32> hashKey, sortIdx, cellStart /* {first, end} per bucket */, largeFlag, largeScan, largeList, pairCount, pairOffset;
	DevBuf<float4> sBox; // sorted colliders: {min.xyz, collider index} {max.xyz, cell tag} per position
	// active lists (k_bodies.hip): the simulated bodies, their colliders + the static ones, ascending; rebuilt when the simulate mask may have changed
	DevBuf<u32> actBodies, actCols, actBlockCount, actBlockBase, colBody; bool activeDirty = true; u32 estActiveBodies = 0, estActiveCols = 0, pairBound = 0, sapBlocks = 0;
	DevBuf<double> sapPartial;            // per workgroup of k_build_colliders: sum of the AABB centres (3), of their squares (3), colliders counted (1)
	DevBuf<uint2> pairs, pairSlab;
	u32 hashTableSize = 0;
	// narrowphase
	DevBuf<u32> pairKey, pairKeySorted; DevBuf<uint2> pairsSorted;
	DevBuf<ManifoldRec> manifolds;
	// colouring / solver
	DevBuf<u64> bodyMask, claim; DevBuf<u32> mColor, mKey, mKeySorted, mIdx, mOrder;
	DevBuf<uint4> actIds;                 // active manifolds: (bodyA, bodyB, count, slot)
	DevBuf<u32> epaList; DevBuf<float4> gjkSimplex; // GJK hits -> EPA work list (9 float4 per hit)
	DevBuf<float4> rowPlanes, rowShared; DevBuf<float2> rowLambda; DevBuf<uint4> rowIds;
	DevBuf<u64> flow;                     // cluster sweep: 32-byte hand-over record per (phase, shared body): two tagged 16-byte halves
	// cluster sweep (k_cluster.hip)
	bool validate = false;                // MI_PHYSICS_VALIDATE=1 / mi_enable_validation: NaN / Inf guard after every stage (the reference's VALIDATE macros, physics.cpp:807-926)
	// replay of the reference's batch order (mi_debug_set_replay / MI_PHYSICS_REPLAY=1; a test facility: the whole contact sweep is ONE workgroup)
	bool replayReferenceOrder = false; DevBuf<u32> replayEntries; std::vector<u32> replayHost; u32 replayBatches = 0;
	u32 scheduleReferenceBatches(const std::vector<uint4>& ids, u32 numPositions);
	bool useCluster = true;               // MI_PHYSICS_NO_CLUSTER=1: launch-per-colour sweep only
	bool lastStepCluster = false, backupVelocities = false;
	u32 clusterPredictDiv = 4, clusterPollSleep = 1, clusterBlocksLimit = 0, clusterFailStreak = 0, clusterParts = 2, clusterTaskWeight = 64u * 1000u, clusterTaskWeightLater = 64u * 500u, clusterShift[CL_MAX_PARTS][3] = { { 0, 0, 0 }, { 13, 9, 15 }, { 27, 21, 31 }, { 7, 29, 5 } }; // MI_CLUSTER_PARTS / _TASK / _SHIFT
	bool clusterPartsFixed = false;       // MI_CLUSTER_PARTS given: no adaptation
	bool clusterSortDue = true; u32 clusterSortAge = 0, clusterSortInterval = 8, clusterSortBodies = 0, clusterSortedParts = 0; // body order along the curves: refreshed every few steps (MI_CLUSTER_SORT_INTERVAL)
	u32 clusterLdsBytes = 0, clusterBlocks = 0, clusterCooldown = 0;
	DevBuf<u32> clKeys, clKeysSorted, clVals, clSorted, clRank, clWsum, clCum, clPhaseMask, clTaskKey, clTaskPos, clPre, clLocal, clEntry, clTaskCount, clTaskStart, clBodyList, clSharedSlot; // clEntry: the contact schedule of every task (at 4 x its first manifold position): manifold position | contact << 12
	DevBuf<uint8_t> clTasks;
	DevBuf<float4> clRowScratch;
	DevBuf<u32> clJointBodyMask; bool clJointListsValid = false; // per body: 1 if a joint of the sweep touches it (phase-0 bit of its phase mask); the joints' task lists of the last refresh are still good
	DevBuf<u32> clCompLabel, clLeftList; bool useComponents = true; // the component phase (MI_CLUSTER_NO_COMPONENTS=1: curve phases + rest task only)
	bool compIdle = false;                    // the last step's curve phases left nothing over (and its rest task was empty): this step skips the component phase's launches (what is left over goes to the rest task)
	DevBuf<u32> clChunk; u32 clChunkParts = 0, clChunkJointVersion = ~0u; bool clChunkWithJoints = false, useChunkCache = true; u32 chunkHeadroomPercent = 10, chunkCachedPhases = CL_MAX_PARTS; // chunk of every body per phase, kept between re-sorts (MI_CLUSTER_NO_CHUNK_CACHE=1: the full partition pipeline every step)
	// joints inside the cluster sweep: island representative per body (jointed bodies must share a task), the joints of all types in
	// (type, colour) order {type | class << 8, index in the type's colour-sorted arrays, body a, body b}, and the per-step lists
	DevBuf<u32> clRep, clJointTask, clJointPos, clJointCount, clJointStart, clJointList; DevBuf<uint4> clJointTable; DevBuf<uint2> clTaskJoints; DevBuf<u32> clJointClassStart;
	bool useClusterJoints = true;         // MI_CLUSTER_NO_JOINTS=1: joints keep their per-colour launches (one cluster launch per iteration then)
	u32 clNumJoints = 0, clNumJointClasses = 0; bool clJointsInCluster = false; // false: more (type, colour) classes than the kernel's table: joints keep their launches
	DevBuf<u64> flowTrace;                // developer timeline (mi_debug_flow_trace): 32 x u64 per slot, allocated on request only
	u32 flowEpoch = 0;
	// Safety net of the persistent kernels: if one gives up waiting (only possible when the GPU is shared with another persistent
	// kernel), the step's velocity integration is skipped on the device and the host redoes solve + integration with the launch
	// sweep from the saved pre-solve velocities, at the next point where it synchronises anyway.
	DevBuf<float4> velBackup; bool flowPending = false; float pendingDt = 0.f; u32 pendingIters = 0, flowTestAbortStep = ~0u;
	void recoverFlow(); int resolvePendingFlow();
	// force fields, triggers, events
	DevBuf<float4> fieldForce;            // per field: world-space force (localized fields only; global ones are summed on the host into globalForce)
	DevBuf<u32> fieldMask; u32 fieldWords = 0; // per body: bit f set = inside localized field f this step (set by k_zone_overlap, consumed + cleared by k_apply_fields)
	float globalForce[3] = { 0.f, 0.f, 0.f }; bool anyGlobalForce = false;
	DevBuf<u64> triggerSet[2], collisionSet[2]; u32 triggerSetSize = 0, collisionSetSize = 0, triggerCur = 0, collisionCur = 0;
	DevBuf<uint8_t> eventRing; u32 eventCap = 1u << 16;
	std::vector<u64> restoredTriggerKeys, restoredCollisionKeys; // mi_world_restore: the snapshot's previous-step sets, entered when the first step sizes the tables
	std::vector<mi_event> pendingEvents;  // drained from the device, not yet handed to the caller
	void uploadFields(); void ensureEventBuffers(u32 numPairs);
	DevBuf<uint8_t> tempStorage;
	DevBuf<u32> sortHist;                 // counting sort: per-tile bucket histograms + their scan
	DevBuf<u32> dCounters; u32* hCounters = nullptr; // CTR_WORDS words each
	size_t pairCap = 0, rowCap = 0;

	// settings snapshot for the running step
	u32 iterations = 30;
	u32 coloringRounds = 24;     // adaptive: last useful round of the previous step + margin (launch-per-round colouring only)
	DevBuf<u64> colorHash[2]; u32 colorHashCur = 0, colorHashSize = 0, stepsSinceFullColoring = 0, fullColoringInterval = 16; // warm-started colouring (MI_PHYSICS_NO_WARM_COLORING=1: from scratch every step)
	bool useWarmColoring = true, forceFullColoring = true;
	bool useFusedColoring = true; u32 colorMaxBlocks = 0; // all colouring rounds in one launch with a grid barrier (MI_PHYSICS_NO_FUSED_COLORING=1: one launch per round)
	u32 lastNumManifolds = 0;    // sizes the colouring-round launches of the next step
	mi_stats stats = {};
	// Stage timing: HIP events of the last STAGE_RING timed steps, read back without stalling every step (mi_get_stats harvests them)
	static const u32 STAGE_RING = 32;
	std::vector<hipEvent_t> stageEvents;  // STAGE_RING x 6
	u32 ringHead = 0, ringPending = 0, accTimed = 0; double accMs[5] = { 0, 0, 0, 0, 0 };
	bool timeStages = false;
	void harvestTiming();
	// Counts of the steps since the last mi_get_stats (the host learns a step's counts at its next synchronisation)
	double sumContacts = 0, sumManifolds = 0, sumColors = 0, sumPairs = 0, sumProbes = 0; u32 sumSteps = 0, countedStep = 0, prevNumPairs = 0;
	void countPreviousStep(); void refreshCounters();

	// The N-iteration solver sweep (joint colours + contact colours per iteration) replayed as one hipGraph.  Launch arguments are
	// step-invariant (ranges live in dCounters), so a graph is rebuilt only when the colour count, a colour's size class, the joint
	// schedule or a buffer address changes.
	struct SolveGraph
	{
		hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr;
		u32 numColors = 0, firstTail = 0, iterations = 0; bool serial = false; u32 jointVersion = 0; u64 bufferVersion = 0;
		u32 gridBlocks[MI_MAX_COLORS] = {};
	} solveGraph;
	u32 jointVersion = 0; u64 bufferVersion = 0;
	bool useGraph = false;

	World(int dev);
	~World();
	void fail(int code, const std::string& what);
	void upload();
	void uploadJoints();
	void downloadState();
	int stepInternal(float dt, u32 iterations);
	int step(float* timer, const mi_physics_settings* s, float dt);
};

// ---- launchers (one per stage; each defined next to its kernels) --------------------------------------------------
void launch_active_lists(World& w);                        // (re)builds the lists of simulated bodies / colliders if the simulate mask may have changed
void launch_restore_velocities(World& w);                  // velocities of the simulated bodies <- velBackup
u32 active_grid(u32 estimate, u32 total);
void launch_build_colliders(World& w);
void launch_broadphase_count(World& w);                    // grid build + pair count + scan; leaves numPairs in dCounters
void launch_broadphase_write(World& w, u32 numPairs, bool slabOverflow);       // slabOverflow: some collider has more partners than its slab holds (CTR_PAIR_OVERFLOW)
void launch_narrowphase(World& w, u32 numPairs);            // numPairs may exceed the device's pair count (a launch sized before the host knows it)
void launch_zone_overlap(World& w, u32 numPairs);           // force-field / trigger overlap tests on the classified pairs (once per step)
void launch_integrate_forces(World& w, float dt);
void launch_heightmap(World& w, u32 numPairs, u32 slotCap);  // terrain contacts appended after the pair manifolds (physics.cpp:1236-1249)
void launch_cloth(World& w, float dt);                      // cloth_component::applyWindForce + simulate for every cloth (physics.cpp:1354-1358)
u32 cloth_lds_particle_limit();
void launch_apply_fields(World& w);                        // localized + global force fields -> force accumulators (before the force integration)
void launch_trigger_events(World& w);                      // leave events + table hand-over of the trigger overlap set (after the narrowphase)
void launch_collision_events(World& w, u32 numPairs);      // begin / end events of this step's manifolds (after the force integration)
void launch_coloring(World& w, u32 numPairs);
void launch_contact_init(World& w, u32 numPairs, float dt);
void launch_solve_replay(World& w, u32 numBatches);
void launch_solve_contacts_iteration(World& w, const u32* gridBlocks, u32 numColors, u32 firstTail, bool serialBucket);
bool cluster_available(World& w);                          // sets up the cluster kernels' LDS budget once; false = this device cannot run them
void launch_active_list(World& w, u32 numPairs);           // manifolds with contacts -> actIds (no colours)
void launch_cluster_build(World& w, u32 numPairs);         // body order, tasks, local colouring, final slot order (k_cluster.hip)
void launch_cluster_solve(World& w, u32 itBegin, u32 itEnd);
bool cluster_solves_joints(const World& w);                // the cluster sweep of this step runs the joints too (one launch for all iterations)
u32 flow_num_regions(const World& w);
void flow_choose_regions(World& w);
void launch_integrate_velocities(World& w, float dt);
void launch_joint_init(World& w, float dt);
void launch_joint_solve_iteration(World& w);
void launch_and_mask(World& w);
void launch_slab_classify(World& w);
void launch_slab_pack(World& w, void* left, void* right, u32 capacity);
void launch_slab_unpack(World& w, const void* left, const void* right, u32 capacity);
void launch_validate(World& w, u32 stage, u32 numPairs); // stage 0: world colliders + AABBs, 1: contacts, 2: body update records, 3: poses + velocities after the step
void launch_copy_pose0(World& w);
void launch_lerp_pose(World& w, float t);
size_t primitives_temp_bytes(size_t maxItems);
void csort_pairs_u32(World& w, const u32* keys, u32* keysOut, const u32* vals, u32* valsOut, u32 n, u32 numBuckets); // stable, keys < numBuckets <= 272
void csort_pairs_u64(World& w, const u32* keys, u32* keysOut, const u64* vals, u64* valsOut, u32 n, u32 numBuckets);
