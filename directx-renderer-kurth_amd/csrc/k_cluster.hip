// Contact sweep around LDS-resident clusters ("K11-cluster"): the production contact solver.
//
// Why.  A Gauss-Seidel sweep over a proper colouring has (colours x iterations) ~ 25 x 30 dependent phases per step.  Across the
// chip a phase boundary costs a launch (~5 us) or a tagged hand-over through L2 / the fabric (~2.5 us); inside ONE workgroup it
// costs a barrier over LDS (~0.25 us with the step's work).  So the world is cut into spatial clusters that one 512-lane workgroup
// each solves entirely out of LDS and registers for all iterations, and only what a cut crosses goes through memory:
//
//   phase 0..P-1 ("curve phases", default P = 2): the bodies are ordered along a Morton curve (each phase its own, shifted, curve)
//       and the curve is chunked by weight into tasks of ~1000 (later phases ~500) contacts; a manifold whose two bodies fall into
//       the same chunk is INTERIOR to that task.  Phase p only looks at what phases < p left over.
//   phase P ("component phase"): what the curves leave over is not cut again; its connected components are dealt whole to the tasks
//       of one more phase.  A rest task (phase CL_MAX_PARTS) only takes what that cannot place (nothing, in practice).
//
// Tasks of one phase share no body, so they run concurrently, one workgroup each; a workgroup runs its (at most two per phase)
// tasks in phase order, iteration after iteration.  Inside a task the CONTACTS are coloured locally (k_cl_color: a manifold with K
// contacts takes K consecutive colours) and swept colour by colour with a workgroup barrier in between, four lanes (a quad) per
// contact row: lane q owns one of vA, wA, vB, wB in LDS and the row's vectors for it, the row velocity is summed inside the quad with
// two DPP adds.  Body velocities live in LDS for the whole launch, the rows of the workgroup's first task in registers
// (CLQ_SETS x CLQ_QUADS contacts), all other rows in LDS in the same lane-private format, global scratch only beyond that.  A body
// touched in more than one phase is handed from task to task through tagged 2 x 16-byte records (sc1 store / sc1 poll,
// MI355X_MICROARCH.md "tagged granules"): with d = number of phases that touch the body, the task of phase p is its r-th user,
// r = popcount(phaseMask & ((1 << p) - 1)), waits for turn epoch + it * d + r and publishes + 1.  Every wait points to a strictly
// earlier (iteration, phase): no cycles.
//
// The result is a Gauss-Seidel sweep in the sequential order (phase, task, local colour, position) — the order
// mi_debug_read_schedule reports and the CPU oracle follows — with bit-identical arithmetic to the launch-per-colour sweep.
#include "world.h"
#include "solver_rows.h"
#include "joint_solve.h"

void prim_sort_pairs_u32(World& w, const u32* kin, u32* kout, const u32* vin, u32* vout, u32 n, u32 bits);
void prim_exclusive_scan_u32(World& w, const u32* in, u32* out, u32 n);

#define CL_LANES 1024u                    // k_cl_color
#define CL_TASKS_PER_PHASE 2u              // tasks of one phase a workgroup may run (LDS holds the bodies and meta of all its tasks)
#define CL_MAX_LOCAL_TASKS 8u              // tasks of all phases per workgroup
#define CLS_LANES 512u                    // k_cl_solve: 8 waves = 128 quads (four lanes work on one contact row) ...
#define CLQ_QUADS (CLS_LANES / 4u)
#define CLQ_SETS 7u                       // ... each keeping this many contact rows in registers (19 VGPRs per row and lane).  Eight sets reach the 256 VGPRs a launch of
                                          // 2 waves per SIMD allows only with 12-21 registers spilled into the colour loop: 7 sets (236 VGPRs, none spilled) solve 8 % faster
#define CLQ_REG_CONTACTS (CLQ_QUADS * CLQ_SETS) // contacts of a workgroup's first task that live in registers (worlds whose joints run inside the sweep: CLQ_SETS_JOINTS sets, the joint solves need the registers)
#define CLQ_SETS_JOINTS 4u
#define CL_WEIGHT_REG_LIMIT (64u * 1250u)  // chunk weight up to which a task's INTERIOR contacts (70 - 85 % of what its bodies own) fit the register sets, give or take what LDS holds
#define CL_UNASSIGNED 0xFFFFFFFFu
#define CL_WEIGHT_MANIFOLD 64u            // weight of a manifold on the curve ...
#define CL_WEIGHT_EXTRA 64u               // ... plus this per contact beyond the first (the sweep's unit is the contact)
#define CL_TASK_MAX_MANIFOLDS 2048u       // hard limits of k_cl_color's LDS tables (a task normally holds <= taskManifolds + one body's degree)
#define CL_TASK_MAX_BODIES 4095u
#define CL_HASH_SIZE 8192u
#define CL_LOCAL_STATIC 0xFFFFu           // local body index of the static dummy body
#define CL_SERIAL_COLOR 64u
#define CL_REST_CAP 1024u                 // once no more than this many manifolds are unassigned, they all go to the rest task (it must fit k_cl_color's tables)
#define CL_SPIN_LIMIT (1u << 22)          // polls before a lane gives up: only reached when the workgroups are not all resident

MI_DEV u32 clOrderedBits(float f) { u32 b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
MI_DEV float clOrderedFloat(u32 o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }
MI_DEV u32 clSpread10(u32 x) { x &= 0x3FFu; x = (x | (x << 16)) & 0x030000FFu; x = (x | (x << 8)) & 0x0300F00Fu; x = (x | (x << 4)) & 0x030C30C3u; x = (x | (x << 2)) & 0x09249249u; return x; }
MI_DEV u32 clHash(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// ---------------------------------------------------------------------------------------------------------------
// Body order: bounding box of the centres of gravity, Morton keys per phase, radix sort, ranks.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cl_bbox(u32 nb, const float4* __restrict__ cog, const uint8_t* __restrict__ simMask, u32* __restrict__ counters)
{
	float mn[3] = { MI_FLT_MAX, MI_FLT_MAX, MI_FLT_MAX }, mx[3] = { -MI_FLT_MAX, -MI_FLT_MAX, -MI_FLT_MAX };
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
	{
		if (!simMask[i]) continue;
		float4 c = cog[i];
		if (!(c.x == c.x && c.y == c.y && c.z == c.z)) continue;
		mn[0] = fminf(mn[0], c.x); mn[1] = fminf(mn[1], c.y); mn[2] = fminf(mn[2], c.z);
		mx[0] = fmaxf(mx[0], c.x); mx[1] = fmaxf(mx[1], c.y); mx[2] = fmaxf(mx[2], c.z);
	}
	for (int k = 0; k < 3; ++k)
		for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
	__shared__ float sMn[4][3], sMx[4][3]; // one atomic pair per workgroup and axis: same-address atomics from all over the chip serialise
	if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 3; ++k) { sMn[threadIdx.x >> 6][k] = mn[k]; sMx[threadIdx.x >> 6][k] = mx[k]; }
	__syncthreads();
	if (threadIdx.x < 3u)
	{
		u32 k = threadIdx.x;
		float a = fminf(fminf(sMn[0][k], sMn[1][k]), fminf(sMn[2][k], sMn[3][k])), b = fmaxf(fmaxf(sMx[0][k], sMx[1][k]), fmaxf(sMx[2][k], sMx[3][k]));
		if (a <= b) { atomicMin(&counters[CTR_CL_BBOX + k], clOrderedBits(a)); atomicMax(&counters[CTR_CL_BBOX + 3 + k], clOrderedBits(b)); }
	}
}

struct ClShifts { u32 s[CL_MAX_PARTS][3]; };

__global__ void __launch_bounds__(256) k_cl_keys(u32 nb, u32 numParts, ClShifts shifts, u32 maxShift, const float4* __restrict__ cog, const uint8_t* __restrict__ simMask,
	const u32* __restrict__ counters, u32* __restrict__ keys, u32* __restrict__ vals)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	float lo[3], ext = 0.f;
	for (int k = 0; k < 3; ++k)
	{
		lo[k] = clOrderedFloat(counters[CTR_CL_BBOX + k]);
		float hi = clOrderedFloat(counters[CTR_CL_BBOX + 3 + k]);
		ext = fmaxf(ext, hi - lo[k]);
	}
	float scale = (ext > 0.f) ? (float)(1023u - maxShift) / ext : 0.f; // one cell size for the three axes
	float4 c = cog[i];
	bool sim = simMask[i] != 0 && c.x == c.x && c.y == c.y && c.z == c.z;
	int q[3] = { (int)((c.x - lo[0]) * scale), (int)((c.y - lo[1]) * scale), (int)((c.z - lo[2]) * scale) };
	for (int k = 0; k < 3; ++k) q[k] = q[k] < 0 ? 0 : (q[k] > (int)(1023u - maxShift) ? (int)(1023u - maxShift) : q[k]);
	for (u32 p = 0; p < numParts; ++p)
	{
		u32 key = clSpread10((u32)q[0] + shifts.s[p][0]) | (clSpread10((u32)q[1] + shifts.s[p][1]) << 1) | (clSpread10((u32)q[2] + shifts.s[p][2]) << 2);
		keys[(size_t)p * nb + i] = sim ? key : 0x3FFFFFFFu; // bodies simulated elsewhere sort last; no manifold refers to them
		vals[(size_t)p * nb + i] = i;
	}
}

__global__ void __launch_bounds__(256) k_cl_ranks(u32 nb, u32 numParts, const u32* __restrict__ sortedBodies, u32* __restrict__ rank)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	for (u32 p = 0; p < numParts; ++p) rank[(size_t)p * (nb + 1) + sortedBodies[(size_t)p * nb + i]] = i;
	if (i == 0) for (u32 p = 0; p < numParts; ++p) rank[(size_t)p * (nb + 1) + nb] = 0xFFFFFFFFu; // the static dummy owns nothing
}

// ---------------------------------------------------------------------------------------------------------------
// Assignment.  wsum[r] = weight of the not-yet-assigned manifolds OWNED by the body of rank r (owner = the dynamic body of the
// manifold that comes first on this phase's curve); cum = exclusive scan; task of a body = cum / taskWeight; a manifold whose
// dynamic bodies agree on the task is interior to it.  Whatever a task holds is owned by its bodies, so a task's weight is below
// taskWeight + one body's weight.
// ---------------------------------------------------------------------------------------------------------------
// Bid of a manifold for its bodies in a colouring round: lowest wins.  Manifolds with more contacts bid lower, so they are coloured
// first and gather in the low colours: a colour's sweep time is that of its longest manifold, and this keeps the 2-4-contact ones
// (20 % of a mixed pile) out of most colours.  Then a pseudo-random priority (hash of the narrowphase slot and the round), then
// the position inside the task, which makes the bid unique.
// Where phase p's task 0 goes: task t of phase p belongs to workgroup (clPhaseOffset + t) % G.  The first phase starts at workgroup 0;
// behind it the phases are placed LAST PHASE FIRST (rest task, then the last partition phase, ...): the workgroups the first phase
// leaves free run their task from registers, and the few tasks of the last phases — every iteration's critical path runs through
// them — get those places before the second phase's many tasks do.
MI_DEV u32 clPhaseOffset(const u32* counters, u32 p)
{
	u32 off = p ? counters[CTR_CL_NUM_TASKS] : 0u;
	for (u32 q = CL_MAX_PHASES - 1u; q > p && p; --q) off += counters[CTR_CL_NUM_TASKS + q];
	return off;
}
// A phase may not have more tasks than the solve launch has workgroups (task t of a phase runs on workgroup (offset + t) % G, all
// of them resident): when the pile outgrows "G tasks of the configured weight", the chunks grow instead.  cum[nb] = total weight.
// And a world of jointed islands that would leave most workgroups without a first-phase task — 256 ragdolls — is spread over the
// launch in smaller tasks (fewer joints and colours per task; islands are never cut, so smaller tasks cost no extra hand-overs;
// config 4: 28 -> 165 tasks, 0.583 ms at 1000, 0.544 at 400, 0.516 at 150, 0.522 at 100), down to CL_WEIGHT_SPREAD_MIN.  Only the first
// phase (its weight argument carries CL_WEIGHT_ISLANDS).  Contact-only worlds keep their chunks: smaller ones cut more manifolds
// (config 2, 10 k spheres: no gain at 300, a fourth phase at 150).
#define CL_WEIGHT_ISLANDS 0x80000000u
#define CL_WEIGHT_SPREAD_MIN (64u * 150u)
MI_DEV u32 clEffectiveWeight(u32 taskWeightArg, u32 totalWeight, u32 maxTasks)
{
	const u32 taskWeight = taskWeightArg & ~CL_WEIGHT_ISLANDS;
	u32 need = totalWeight / maxTasks + 1u;                                   // one task per workgroup ...
	if (need <= taskWeight)
	{
		if ((taskWeightArg & CL_WEIGHT_ISLANDS) && 2u * need <= taskWeight)
		{
			const u32 spread = need + need / 2u;
			return spread > CL_WEIGHT_SPREAD_MIN ? spread : min(CL_WEIGHT_SPREAD_MIN, taskWeight);
		}
		return taskWeight;
	}
	if (need <= CL_WEIGHT_REG_LIMIT) return need;                             // ... as long as such a task still fits the lanes' registers,
	u32 need2 = totalWeight / (CL_TASKS_PER_PHASE * maxTasks) + 1u;             // then up to CL_TASKS_PER_PHASE per workgroup (the later ones run from LDS)
	return need2 > CL_WEIGHT_REG_LIMIT ? need2 : CL_WEIGHT_REG_LIMIT;
}
MI_DEV u32 clBid(u32 slot, u32 count, u32 round, u32 i) { return (((4u - count) & 3u) << 22) | ((clHash(slot * 2654435761u + round) & 0x3FFu) << 12) | (i & 0xFFFu); } // 24 bits
#define CL_REMAIN_SUBS 64u // the 'still unassigned after phase p' count is kept in 64 partial counters: ~2000 workgroups adding to ONE word queue up behind each other
#define CL_SUBCOUNTERS 8u // a task's append cursor is split in 8 (by workgroup) so that ~650 returning atomics do not queue on one address

// Everything the assignment accumulates into, cleared in one launch.
__global__ void __launch_bounds__(256) k_cl_clear(u32 nb1, u32* __restrict__ wsum, u32* __restrict__ phaseMask, u32* __restrict__ taskCount, u32* __restrict__ jointCount, u32* __restrict__ counters, u32* __restrict__ compLabel,
	const u32* __restrict__ jointBodyMask, u32 keepJointLists)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nb1) compLabel[i] = i;               // components of what the curve phases leave over: every body its own
	if (i < 5u) counters[CTR_CL_LEFT + i] = 0;
	if (i == 0u) counters[CTR_CL_SCRATCH] = 0;   // append cursor of the solve launch's row scratch
	if (i < CL_MAX_TASKS && !keepJointLists) jointCount[i] = 0; // (between two refreshes the joints' tasks do not change: their lists are kept, see launch_cluster_build)
	if (i < CL_MAX_PARTS * nb1) wsum[i] = 0;
	if (i < nb1) phaseMask[i] = (keepJointLists && jointBodyMask) ? jointBodyMask[i] : 0u; // (kept joint lists: their bodies' first-phase bit, which the joint assignment sets otherwise)
	if (i < CL_MAX_PHASES * CL_MAX_TASKS * CL_SUBCOUNTERS + 6u * CL_REMAIN_SUBS) taskCount[i] = 0; // (+ the split 'still unassigned' counters behind the task counters)
	if (i < 7u) counters[CTR_CL_STATUS + i] = 0;  // status, shared bodies, manifolds per phase
	if (i < 6u) counters[CTR_CL_REMAIN + i] = 0;
}

MI_DEV u32 clWeight(u32 count) { return CL_WEIGHT_MANIFOLD + (count - 1u) * CL_WEIGHT_EXTRA; }

// rep: island representative per body (bodies connected by joints share one; the body itself otherwise; the dummy maps to itself).
// In phase 0 a body counts where its representative is on the curve, so that an island is never cut.
#define CL_WEIGHT_JOINT (6u * CL_WEIGHT_MANIFOLD) // a joint's solve costs several contact rows: at most ~160 joints per task
__global__ void __launch_bounds__(256) k_cl_weights0(const u32* __restrict__ counters, u32 nb, const uint4* __restrict__ actIds, const u32* __restrict__ rank0, const u32* __restrict__ rep,
	u32* __restrict__ wsum, u32* __restrict__ taskKey)
{
	u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= counters[CTR_NUM_ACTIVE]) return;
	uint4 ids = actIds[j];
	u32 ra = rank0[rep ? rep[ids.x] : ids.x], rb = rank0[rep ? rep[ids.y] : ids.y]; // the dummy's rank is 0xFFFFFFFF
	atomicAdd(&wsum[min(ra, rb)], clWeight(ids.z));
	taskKey[j] = CL_UNASSIGNED;
}
__global__ void __launch_bounds__(256) k_cl_joint_weights(u32 numJoints, const uint4* __restrict__ table, const u32* __restrict__ rank0, const u32* __restrict__ rep, u32* __restrict__ wsum)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= numJoints) return;
	atomicAdd(&wsum[rank0[rep[table[i].z]]], CL_WEIGHT_JOINT);
}
// After phase 0's scan: every joint goes to the task of its island.
__global__ void __launch_bounds__(256) k_cl_joint_assign(u32 numJoints, u32 nb, u32 taskWeight, u32 maxTasks, const uint4* __restrict__ table, const u32* __restrict__ rank0, const u32* __restrict__ rep, const u32* __restrict__ cum,
	u32* __restrict__ jointTask, u32* __restrict__ jointPos, u32* __restrict__ jointCount, u32* __restrict__ phaseMask, u32* __restrict__ status)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= numJoints) return;
	uint4 e = table[i];
	u32 t = cum[rank0[rep[e.z]]] / clEffectiveWeight(taskWeight, cum[nb], maxTasks);
	if (t >= CL_MAX_TASKS) { atomicOr(status, 1u); t = CL_MAX_TASKS - 1u; }
	jointTask[i] = t;
	jointPos[i] = atomicAdd(&jointCount[t], 1u);
	atomicOr(&phaseMask[e.z], 1u); atomicOr(&phaseMask[e.w], 1u);
}
__global__ void __launch_bounds__(256) k_cl_joint_scatter(u32 numJoints, const u32* __restrict__ jointTask, const u32* __restrict__ jointPos, const u32* __restrict__ jointStart, u32* __restrict__ jointList)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < numJoints) jointList[jointStart[jointTask[i]] + jointPos[i]] = i;
}

// Phase p: assign what is interior; what is left adds its weight to the next phase's curve, or (last partition) goes to the rest task.
__global__ void __launch_bounds__(256) k_cl_assign(u32* counters, u32 nb, u32 phase, u32 numParts, u32 taskWeight, u32 maxTasks, const uint4* __restrict__ actIds,
	const u32* __restrict__ rank, const u32* __restrict__ cum, const u32* __restrict__ rankNext, u32* __restrict__ wsumNext,
	u32* __restrict__ taskKey, u32* __restrict__ taskPos, u32* __restrict__ taskCount, u32* __restrict__ phaseMask, u32* __restrict__ status, const u32* __restrict__ rep, u32* __restrict__ leftList, u32 leftCap)
{
	u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	taskWeight = clEffectiveWeight(taskWeight, cum[nb], maxTasks);
	u32* remainSub = taskCount + CL_MAX_PHASES * CL_MAX_TASKS * CL_SUBCOUNTERS;
	__shared__ u32 sEntering;
	if (threadIdx.x < 64u)
	{
		u32 v = phase ? remainSub[phase * CL_REMAIN_SUBS + threadIdx.x] : 0u;
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if (threadIdx.x == 0) sEntering = phase ? v : counters[CTR_NUM_ACTIVE];
	}
	__syncthreads();
	const u32 entering = sEntering;
	const bool dumpAll = entering <= CL_REST_CAP && !rep; // few enough left: one task takes them all, later partitions stay empty (with joints, phase 0 keeps the contacts of an island next to its joints)
	bool pending = j < counters[CTR_NUM_ACTIVE] && taskKey[j] == CL_UNASSIGNED;
	u32 key = CL_UNASSIGNED;
	uint4 ids = make_uint4(0, 0, 0, 0);
	bool da = false, db = false;
	if (pending)
	{
		ids = actIds[j];
		da = ids.x < nb; db = ids.y < nb;
		if (dumpAll) key = CL_MAX_PARTS * CL_MAX_TASKS;
		else
		{
			u32 ta = da ? cum[rank[rep ? rep[ids.x] : ids.x]] / taskWeight : 0u, tb = db ? cum[rank[rep ? rep[ids.y] : ids.y]] / taskWeight : 0u; // rep: phase 0 with joints only
			if (!da) ta = tb;
			if (!db) tb = ta;
			if (ta == tb)
			{
				if (ta >= CL_MAX_TASKS) { atomicOr(status, 1u); ta = CL_MAX_TASKS - 1u; }
				key = phase * CL_MAX_TASKS + ta;
			}
			else if (phase + 1u == numParts && !leftList) key = CL_MAX_PARTS * CL_MAX_TASKS; // the rest task
		}
	}
	// what the last curve phase leaves goes to the component phase (k_cl_components), through a list
	const bool toList = pending && key == CL_UNASSIGNED && phase + 1u == numParts && leftList;
	if (toList)
	{
		const u64 m = __ballot(1);
		const u32 lane = threadIdx.x & 63u, leader = (u32)__ffsll((long long)m) - 1u;
		u32 base = 0;
		if (lane == leader) base = atomicAdd(&counters[CTR_CL_LEFT], (u32)__popcll(m));
		base = __shfl(base, leader) + (u32)__popcll(m & ((1ull << lane) - 1ull));
		if (base < leftCap) leftList[base] = j;
	}
	bool left = pending && key == CL_UNASSIGNED;
	u32 numLeft = (u32)__syncthreads_count(left); // one atomic per workgroup
	if (threadIdx.x == 0 && numLeft) atomicAdd(&remainSub[(phase + 1u) * CL_REMAIN_SUBS + (blockIdx.x & (CL_REMAIN_SUBS - 1u))], numLeft);
	if (!pending) return;
	if (key != CL_UNASSIGNED)
	{
		u32 ph = key / CL_MAX_TASKS;
		taskKey[j] = key;
		// append position: one atomic per (wave, task) instead of one per manifold.  The active list follows the narrowphase slots,
		// i.e. the broadphase's cell order, so a wave's manifolds belong to very few tasks; returning atomics on one address are
		// served one after the other (~0.2 us each), and a task used to get ~80 of them per sub-counter.
		{
			// groups of equal keys first (ballots only), then ALL the groups' leaders issue their atomics together: one round trip
			// to the memory-side atomic unit per wave, not one per distinct key
			u64 todo = __ballot(1), mine = 0;
			const u32 lane = threadIdx.x & 63u;
			while (todo)
			{
				u32 leader = (u32)__ffsll((long long)todo) - 1u;
				u32 k0 = __shfl(key, leader);
				u64 same = __ballot(key == k0) & todo;
				if (key == k0) mine = same;
				todo &= ~same;
			}
			const u32 myLeader = (u32)__ffsll((long long)mine) - 1u;
			u32 base = 0;
			if (lane == myLeader) base = atomicAdd(&taskCount[key * CL_SUBCOUNTERS + (blockIdx.x & (CL_SUBCOUNTERS - 1u))], (u32)__popcll(mine));
			base = __shfl(base, myLeader);
			u32 pos = base + (u32)__popcll(mine & ((1ull << lane) - 1ull));
			taskPos[j] = pos;
		}
		// (most bodies have the bit already from another manifold of theirs: look before the atomic)
		if (da && !(__hip_atomic_load(&phaseMask[ids.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << ph))) atomicOr(&phaseMask[ids.x], 1u << ph);
		if (db && !(__hip_atomic_load(&phaseMask[ids.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << ph))) atomicOr(&phaseMask[ids.y], 1u << ph);
	}
	else if (!toList)
	{
		u32 ra = rankNext[ids.x], rb = rankNext[ids.y];
		atomicAdd(&wsumNext[min(ra, rb)], clWeight(ids.z));
	}
}

// ---- the partition cached between re-sorts -------------------------------------------------------------------------------------
// Bodies move a fraction of their size per step and the pile's contacts change by well under a per cent per step, so the chunk
// boundaries of a phase (which chunk a body's curve position belongs to) are computed with the full pipeline — weights, scans,
// one assignment pass per phase — only on the steps that also re-sort the bodies along the curves; in between, the stored chunk of
// every body per phase decides where a manifold goes, in ONE pass without scans.  Any partition is valid; a stale one only lets the
// tasks' sizes drift by the few per cent the pile changes in those steps (the refresh chunks are cut 4 % short for that).
__global__ void __launch_bounds__(256) k_cl_store_chunks(u32 nb, u32 taskWeight, u32 maxTasks, const u32* __restrict__ rank, const u32* __restrict__ cum, const u32* __restrict__ rep, u32* __restrict__ chunk)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	u32 t = cum[rank[rep ? rep[i] : i]] / clEffectiveWeight(taskWeight, cum[nb], maxTasks);
	chunk[i] = min(t, CL_MAX_TASKS - 1u);
}
// numCached: the phases placed from the stored chunks (1: the first phase only — the later phases, a quarter of the manifolds, keep
// the per-step pipeline on what is left, which keeps their tasks at the size their fast path needs).  What the cached phases leave
// goes on to phase numCached (its weight onto that phase's curve), or to the rest task when there is none.
__global__ void __launch_bounds__(256) k_cl_assign_cached(u32* counters, u32 nb, u32 numParts, u32 numCached, u32 withJoints, const uint4* __restrict__ actIds, const u32* __restrict__ chunk,
	const u32* __restrict__ rankNext, u32* __restrict__ wsumNext, u32* __restrict__ taskKey, u32* __restrict__ taskPos, u32* __restrict__ taskCount, u32* __restrict__ phaseMask, u32* __restrict__ leftList, u32 leftCap)
{
	const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	const u32 numActive = counters[CTR_NUM_ACTIVE];
	const bool live = j < numActive;
	const bool dumpAll = numActive <= CL_REST_CAP && !withJoints;
	u32* remainSub = taskCount + CL_MAX_PHASES * CL_MAX_TASKS * CL_SUBCOUNTERS;
	u32 key = CL_MAX_PARTS * CL_MAX_TASKS, phase = numParts; // the rest task unless a phase takes it
	uint4 ids = make_uint4(0, 0, 0, 0);
	bool da = false, db = false;
	if (live)
	{
		ids = actIds[j];
		da = ids.x < nb; db = ids.y < nb;
		if (!dumpAll)
		{
			// (the first two phases' chunks are requested together: a manifold the first phase cuts does not wait a second round trip)
			const u32* c1 = chunk + (size_t)(nb + 1u);
			u32 ta = da ? chunk[ids.x] : 0u, tb = db ? chunk[ids.y] : 0u, ta1 = (da && numCached > 1u) ? c1[ids.x] : 0u, tb1 = (db && numCached > 1u) ? c1[ids.y] : 0u;
			for (u32 p = 0; p < numCached; ++p)
			{
				if (p == 1u) { ta = ta1; tb = tb1; }
				else if (p > 1u) { const u32* c = chunk + (size_t)p * (nb + 1u); ta = da ? c[ids.x] : 0u; tb = db ? c[ids.y] : 0u; }
				if (!da) ta = tb;
				if (!db) tb = ta;
				if (ta == tb) { key = p * CL_MAX_TASKS + ta; phase = p; break; }
			}
		}
	}
	// manifolds still unassigned when phase q + 1 starts (statistics; the host adapts the number of phases from them)
	const bool goesOn = live && phase == numParts && numCached < numParts && !dumpAll; // left by the cached phases, with a pipeline phase to go to
	for (u32 q = 0; q < numCached; ++q)
	{
		u32 numLeft = (u32)__syncthreads_count(live && phase > q);
		if (threadIdx.x == 0 && numLeft) atomicAdd(&remainSub[(q + 1u) * CL_REMAIN_SUBS + (blockIdx.x & (CL_REMAIN_SUBS - 1u))], numLeft);
	}
	const bool toList = live && phase == numParts && numCached == numParts && !dumpAll && leftList; // left by ALL curve phases: the component phase takes it
	if (toList)
	{
		const u64 m = __ballot(1);
		const u32 lane = threadIdx.x & 63u, leader = (u32)__ffsll((long long)m) - 1u;
		u32 base = 0;
		if (lane == leader) base = atomicAdd(&counters[CTR_CL_LEFT], (u32)__popcll(m));
		base = __shfl(base, leader) + (u32)__popcll(m & ((1ull << lane) - 1ull));
		if (base < leftCap) leftList[base] = j;
		taskKey[j] = CL_UNASSIGNED;
	}
	if (!live || toList) return;
	if (goesOn)
	{
		taskKey[j] = CL_UNASSIGNED;
		u32 ra = rankNext[ids.x], rb = rankNext[ids.y];
		atomicAdd(&wsumNext[min(ra, rb)], clWeight(ids.z));
		return;
	}
	taskKey[j] = key;
	{
		u64 todo = __ballot(1), mine = 0;
		const u32 lane = threadIdx.x & 63u;
		while (todo)
		{
			u32 leader = (u32)__ffsll((long long)todo) - 1u;
			u32 k0 = __shfl(key, leader);
			u64 same = __ballot(key == k0) & todo;
			if (key == k0) mine = same;
			todo &= ~same;
		}
		const u32 myLeader = (u32)__ffsll((long long)mine) - 1u;
		u32 base = 0;
		if (lane == myLeader) base = atomicAdd(&taskCount[key * CL_SUBCOUNTERS + (blockIdx.x & (CL_SUBCOUNTERS - 1u))], (u32)__popcll(mine));
		base = __shfl(base, myLeader);
		taskPos[j] = base + (u32)__popcll(mine & ((1ull << lane) - 1ull));
	}
	const u32 ph = key / CL_MAX_TASKS;
	if (da) atomicOr(&phaseMask[ids.x], 1u << ph); // (results unused: the wave does not wait for them; a load-then-or would)
	if (db) atomicOr(&phaseMask[ids.y], 1u << ph);
}
__global__ void __launch_bounds__(256) k_cl_joint_assign_cached(u32 numJoints, const uint4* __restrict__ table, const u32* __restrict__ chunk0, u32* __restrict__ jointTask, u32* __restrict__ jointPos,
	u32* __restrict__ jointCount, u32* __restrict__ phaseMask)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= numJoints) return;
	uint4 e = table[i];
	u32 t = chunk0[e.z]; // (the chunk of the island's representative: k_cl_store_chunks)
	jointTask[i] = t;
	jointPos[i] = atomicAdd(&jointCount[t], 1u);
	atomicOr(&phaseMask[e.z], 1u); atomicOr(&phaseMask[e.w], 1u);
}

// ---- the component phase ---------------------------------------------------------------------------------------------------------
// What the curve phases leave over (a few per cent of the manifolds: those cut by every curve's chunk borders, in small clumps where
// the borders cross) is not cut again: its connected components (bodies joined by left-over manifolds) are found and whole components
// are dealt to the tasks of ONE more phase, so nothing is left for a further phase and the rest task stays empty (each phase costs a
// hand-over and its slowest task's colours in EVERY iteration).  Union-find on the global label array (label[b] = a body of b's
// component with a smaller or equal id; k_cl_clear set label[b] = b), a fixed number of rounds — a component that has not
// converged by then only sends the manifolds whose ends still disagree to the rest task.  Components are
// dealt to tasks by a hash of their label (the clumps are tens of manifolds against tasks of hundreds: the load evens out), a
// component too large for a task is sent to the rest task.
#define CL_COMP_ROUNDS 4u
#define CL_COMP_MAX_WEIGHT (64u * 1400u) // a component heavier than this cannot be a task's (k_cl_color's tables): rest task
#define CL_COMP_BLOCKS 64u               // workgroups of the component kernels (they stride over the list: its length is only known on the device)
#define CL_LD(P_) __hip_atomic_load((P_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
MI_DEV u32 clFind(u32* label, u32 x) { u32 r = CL_LD(&label[x]); for (u32 h = 0; h < 16u; ++h) { u32 up = CL_LD(&label[r]); if (up == r) break; r = up; } return r; }
// One round: every left-over manifold hooks the larger of its ends' roots under the smaller one, then shortens its ends' paths.  Rounds
// are separate launches; inside a round hooks and shortcuts of different manifolds interleave freely: a label always names a body of
// the same component with a smaller or equal id, so whatever the interleaving the labels only ever merge what belongs together, and
// a link lost to a concurrent hook is found again by the next round (the manifold that made it is looked at in every round).
__global__ void __launch_bounds__(256) k_cl_comp_round(const u32* __restrict__ counters, u32 nb, u32 leftCap, const u32* __restrict__ leftList, const uint4* __restrict__ actIds, u32* label)
{
	const u32 n = min(counters[CTR_CL_LEFT], leftCap);
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		uint4 ids = actIds[leftList[i]];
		if (ids.x >= nb || ids.y >= nb) continue;
		u32 ra = clFind(label, ids.x), rb = clFind(label, ids.y);
		if (ra != rb) atomicMin(&label[max(ra, rb)], min(ra, rb));
		atomicMin(&label[ids.x], clFind(label, ids.x));
		atomicMin(&label[ids.y], clFind(label, ids.y));
	}
}
// Weight of every component (at its label) and of the lot (counters[CTR_CL_LEFT + 2], zeroed by k_cl_clear).
__global__ void __launch_bounds__(256) k_cl_comp_weights(u32* __restrict__ counters, u32 nb, u32 leftCap, const u32* __restrict__ leftList, const uint4* __restrict__ actIds, u32* label, u32* __restrict__ compWeight)
{
	const u32 n = min(counters[CTR_CL_LEFT], leftCap);
	u32 mine = 0;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		uint4 ids = actIds[leftList[i]];
		u32 b = ids.x < nb ? ids.x : ids.y;
		u32 w = clWeight(ids.z);
		atomicAdd(&compWeight[clFind(label, b)], w);
		mine += w;
	}
	for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
	if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(&counters[CTR_CL_LEFT + 2], mine);
}
// Deal the manifolds: both ends in one component of fitting size -> that component's task (hash of its label); otherwise the rest task.
__global__ void __launch_bounds__(256) k_cl_comp_assign(u32* __restrict__ counters, u32 nb, u32 phase, u32 taskWeight, u32 leftCap, const u32* __restrict__ leftList, const uint4* __restrict__ actIds, u32* label,
	const u32* __restrict__ compWeight, u32* __restrict__ taskKey, u32* __restrict__ taskPos, u32* __restrict__ taskCount, u32* __restrict__ phaseMask)
{
	const u32 n = min(counters[CTR_CL_LEFT], leftCap), total = counters[CTR_CL_LEFT + 2];
	const u32 numTasks = min(max(1u, (total + taskWeight - 1u) / taskWeight), CL_MAX_TASKS);
	if (blockIdx.x == 0 && threadIdx.x == 0) counters[CTR_CL_LEFT + 1] = numTasks;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		const u32 j = leftList[i];
		uint4 ids = actIds[j];
		const bool da = ids.x < nb, db = ids.y < nb;
		u32 la = da ? clFind(label, ids.x) : 0u, lb = db ? clFind(label, ids.y) : 0u;
		if (!da) la = lb;
		if (!db) lb = la;
		u32 key = CL_MAX_PARTS * CL_MAX_TASKS;
		if (la == lb && compWeight[la] <= CL_COMP_MAX_WEIGHT) key = phase * CL_MAX_TASKS + clHash(la * 2654435761u) % numTasks;
		else if (la != lb) atomicAdd(&counters[CTR_CL_LEFT + 3], 1u); else atomicMax(&counters[CTR_CL_LEFT + 4], compWeight[la]); // (statistics: not converged / too large)
		taskKey[j] = key;
		taskPos[j] = atomicAdd(&taskCount[key * CL_SUBCOUNTERS + ((j >> 8) & (CL_SUBCOUNTERS - 1u))], 1u); // (the sub-counter k_cl_scatter derives from j's position in ITS launch)
		const u32 ph = key / CL_MAX_TASKS;
		if (da) atomicOr(&phaseMask[ids.x], 1u << ph);
		if (db) atomicOr(&phaseMask[ids.y], 1u << ph);
	}
}
#undef CL_LD

// One workgroup: exclusive scan of the per-task counts -> first slot of every task; tasks per phase; end of schedule.
// Inclusive prefix sum over the 1024 lanes of a workgroup: shuffles inside a wave, one LDS step across the 16 waves (two barriers; a
// Hillis-Steele ladder through LDS was 20).
MI_DEV u32 clBlockInclusive1024(u32 v, u32* waveTotals /* [16], LDS */)
{
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (u32 o = 1; o < 64u; o <<= 1) { u32 up = __shfl_up(v, o); if (lane >= o) v += up; }
	if (lane == 63u) waveTotals[wave] = v;
	__syncthreads();
	u32 before = 0;
	for (u32 k = 0; k < 16u; ++k) before += (k < wave) ? waveTotals[k] : 0u;
	__syncthreads(); // (waveTotals is reused by the next call)
	return v + before;
}
__global__ void __launch_bounds__(1024) k_cl_offsets(u32* __restrict__ counters, u32 numParts, const u32* __restrict__ taskCount, u32* __restrict__ taskStart, const u32* __restrict__ jointCount, u32* __restrict__ jointStart)
{
	__shared__ u32 part[16];
	__shared__ u32 lastTask[CL_MAX_PHASES];
	const u32 total = CL_MAX_PHASES * CL_MAX_TASKS * CL_SUBCOUNTERS, per = (total + 1023u) / 1024u;
	u32 t = threadIdx.x;
	if (t < CL_MAX_PHASES) lastTask[t] = 0;
	__shared__ uint16_t cnt[CL_MAX_PHASES * CL_MAX_TASKS * CL_SUBCOUNTERS]; // the counters, read once with neighbouring lanes on neighbouring words (a sub-counter holds < 64 k)
	for (u32 e = t; e < total; e += 1024u) cnt[e] = (uint16_t)min(taskCount[e], 0xFFFFu);
	__syncthreads(); // (cnt, lastTask)
	u32 sum = 0;
	for (u32 k = 0; k < per; ++k) if (t * per + k < total) sum += cnt[t * per + k];
	u32 run = clBlockInclusive1024(sum, part) - sum;
	for (u32 k = 0; k < per; ++k)
	{
		u32 e = t * per + k;
		if (e >= total) break;
		u32 c = cnt[e], key = e / CL_SUBCOUNTERS;
		taskStart[e] = run; run += c;
		if (c) atomicMax(&lastTask[key / CL_MAX_TASKS], (key % CL_MAX_TASKS) + 1u);
	}
	if (t == 1023u) taskStart[total] = run;
	__syncthreads();
	// joints per phase-0 task (CL_MAX_TASKS <= 1024 entries: one per lane); a task may hold joints and no manifold
	{
		u32 jc = (jointCount && t < CL_MAX_TASKS) ? jointCount[t] : 0u;
		const u32 jIncl = jointCount ? clBlockInclusive1024(jc, part) : 0u; // (uniform branch)
		if (jointStart && t < CL_MAX_TASKS) { jointStart[t] = jIncl - jc; if (t == CL_MAX_TASKS - 1u) jointStart[CL_MAX_TASKS] = jIncl; }
		if (jc) atomicMax(&lastTask[0], t + 1u);
		__syncthreads();
	}
	const u32 totalManifolds = taskStart[total];
	if (t < CL_MAX_PHASES) counters[CTR_CL_NUM_TASKS + t] = lastTask[t];
	if (t < 6u) { u32 v = 0; for (u32 k = 0; k < CL_REMAIN_SUBS; ++k) v += taskCount[total + t * CL_REMAIN_SUBS + k]; counters[CTR_CL_REMAIN + t] = v; } // for the host's statistics / phase-count adaptation
	if (t == 0)
	{
		counters[CTR_NUM_MANIFOLDS] = totalManifolds;
		counters[CTR_NUM_COLORS] = 0; // k_cl_color: atomicMax of the local colour counts
		for (int k = 0; k < 3; ++k) { counters[CTR_CL_BBOX + k] = 0xFFFFFFFFu; counters[CTR_CL_BBOX + 3 + k] = 0u; } // consumed by k_cl_keys: ready for the next step
	}
}

__global__ void __launch_bounds__(256) k_cl_scatter(const u32* __restrict__ counters, const u32* __restrict__ taskKey, const u32* __restrict__ taskPos, const u32* __restrict__ taskStart, u32* __restrict__ pre)
{
	u32 j = blockIdx.x * blockDim.x + threadIdx.x; // same launch geometry as k_cl_assign: blockIdx selects the same sub-counter
	if (j >= counters[CTR_NUM_ACTIVE]) return;
	pre[taskStart[taskKey[j] * CL_SUBCOUNTERS + (blockIdx.x & (CL_SUBCOUNTERS - 1u))] + taskPos[j]] = j;
}

// ---------------------------------------------------------------------------------------------------------------
// Per task, one workgroup: local body table, local colouring, order by (colour, 4 - contacts), task header.
//   LDS: body hash (global id -> local index), per local body a 64-bit colour mask and a claim word, per manifold its two local
//   bodies, its key and its final position.
// Colouring = the rounds of k_color_round with LDS atomics: every uncoloured manifold bids for both bodies with a pseudo-random
// priority (deterministic: hash of its narrowphase slot and the round); who holds both takes the lowest colour free on both.
// Manifolds that find no colour below 64 form the task's serial tail (one per barrier).
// Local indices: bodies touched in more than one phase ("shared") first, then the task-private ones.
// ---------------------------------------------------------------------------------------------------------------
struct ClTask
{
	u32 first, count, numBodies, numShared, numColors, serialStart, numRows, sharedBase; // numRows: contacts; serialStart: first CONTACT position of the serial tail; sharedBase: first hand-over record of the task's shared bodies
	u32 colorStart[72]; // CONTACT position (relative to 4 * first in the contact tables) of the first contact of colour c; [numColors] = serialStart
};
static_assert(sizeof(ClTask) == 320, "task header");

__global__ void __launch_bounds__(1024) k_cl_color(u32* __restrict__ counters, u32 nb, const u32* __restrict__ taskStart, u32* __restrict__ pre, const uint4* __restrict__ actIds,
	const u32* __restrict__ phaseMask, ClTask* __restrict__ tasks, u32* __restrict__ bodyList, u32* __restrict__ mOrder, u32* __restrict__ mKeySorted, u32* __restrict__ mLocal, u32* __restrict__ cEntry, u32* __restrict__ sharedSlot,
	const u32* __restrict__ jointStart, const u32* __restrict__ jointList, const uint4* __restrict__ jointTable, uint2* __restrict__ taskJoints, u32* __restrict__ jointClassStart, u64* __restrict__ trace)
{
	extern __shared__ u32 clds[];
	u32* hKey = clds;                                   // [CL_HASH_SIZE] global id + 1, 0 = empty
	u32* hVal = hKey + CL_HASH_SIZE;                    // [CL_HASH_SIZE] local index
	u64* mask = (u64*)(hVal + CL_HASH_SIZE);            // [CL_TASK_MAX_BODIES + 1]
	u32* claim = (u32*)(mask + CL_TASK_MAX_BODIES + 1); // [CL_TASK_MAX_BODIES + 1]
	u32* mAB = claim + CL_TASK_MAX_BODIES + 1;          // [CL_TASK_MAX_MANIFOLDS] la | lb << 16
	u32* mKey = mAB + CL_TASK_MAX_MANIFOLDS;            // [..] colour * 4 + (4 - count); UNCOLORED while colouring
	u32* mPos = mKey + CL_TASK_MAX_MANIFOLDS;           // [..] final position
	u32* mCnt = mPos + CL_TASK_MAX_MANIFOLDS;           // [..] contact count by final position, then its exclusive scan of (count - 1)
	u32* mSlot = mCnt + CL_TASK_MAX_MANIFOLDS;          // [..] narrowphase slot | contacts << 28 (the colouring rounds' priorities hash it)
	u32* hist = mSlot + CL_TASK_MAX_MANIFOLDS;          // [264] per key, then cursors
	u32* jHist = hist + 264;                            // [CL_MAX_JOINT_CLASSES + 1] joints per (type, colour) class, then cursors
	u32* cHist = jHist + CL_MAX_JOINT_CLASSES + 1;      // [CL_SERIAL_COLOR + 2] contacts per colour (64 = the serial tail), then first positions, then cursors
	__shared__ u32 sNumShared, sNumPrivate, sMaxColor, sScan[16], sSharedBase;
	const u32 tid = threadIdx.x;
	// developer timeline (mi_debug_flow_trace): row 14 of the task's 16 rows = core-clock stamps of the stages below, [15] = colouring rounds
#define CL_STAMP(I_) if (trace && tid == 0 && key < CL_MAX_TASKS) trace[((size_t)key * 16u + 14u) * 32u + (I_)] = clock64();

	// Task t of phase p is built by workgroup (tasks of the earlier phases + t) % G — the rotation the solve launch uses — so that
	// the later phases' tasks go to the workgroups the first phase left idle first, and nobody builds more than
	// ceil(tasks / G) + 1 of them (by key order workgroup 0 built one task of EVERY phase: 4 x 35 us on the kernel's critical path).
	for (u32 ph = 0; ph < CL_MAX_PHASES; ++ph)
	{
	const u32 tasksInPhase = counters[CTR_CL_NUM_TASKS + ph];
	const u32 tFirst = (blockIdx.x + gridDim.x - (clPhaseOffset(counters, ph) % gridDim.x)) % gridDim.x;
	for (u32 key = ph * CL_MAX_TASKS + tFirst; key < ph * CL_MAX_TASKS + min(tasksInPhase, CL_MAX_TASKS); key += gridDim.x)
	{
		u32 first = taskStart[key * CL_SUBCOUNTERS], n = taskStart[(key + 1u) * CL_SUBCOUNTERS] - first;
		ClTask* T = tasks + key;
		// joints of the task (phase 0 only: an island lives in one phase-0 task)
		const u32 jFirst = (jointStart && key < CL_MAX_TASKS) ? jointStart[key] : 0u, nj = (jointStart && key < CL_MAX_TASKS) ? jointStart[key + 1u] - jFirst : 0u;
		if (jointStart && key < CL_MAX_TASKS && tid <= CL_MAX_JOINT_CLASSES) jointClassStart[(size_t)key * (CL_MAX_JOINT_CLASSES + 2u) + tid] = 0u; // (no joints: all classes empty)
		if (jointStart && key < CL_MAX_TASKS && tid == 0) jointClassStart[(size_t)key * (CL_MAX_JOINT_CLASSES + 2u) + CL_MAX_JOINT_CLASSES + 1u] = jFirst;
		if (!n && !nj) { if (tid == 0) { T->first = first; T->count = 0; T->numBodies = 0; T->numShared = 0; T->numColors = 0; T->serialStart = 0; T->numRows = 0; } continue; }
		if (n > CL_TASK_MAX_MANIFOLDS || nj > CL_TASK_MAX_JOINTS) { if (tid == 0) { atomicOr(&counters[CTR_CL_STATUS], 2u); T->first = first; T->count = 0; T->numBodies = 0; T->numShared = 0; T->numColors = 0; T->serialStart = 0; T->numRows = 0; } continue; }
		u32 phase = key / CL_MAX_TASKS;
		for (u32 h = tid; h < CL_HASH_SIZE; h += CL_LANES) hKey[h] = 0;
		for (u32 h = tid; h < 264u + CL_MAX_JOINT_CLASSES + 1u + CL_SERIAL_COLOR + 2u; h += CL_LANES) hist[h] = 0;
		if (tid == 0) { sNumShared = 0; sNumPrivate = 0; sMaxColor = 0; }
		__syncthreads();
		CL_STAMP(0)
		// 1. distinct dynamic bodies
		// (the manifold's ids are fetched once, through two dependent global loads, and parked in LDS for the sort and step 2)
		for (u32 i = tid; i < n; i += CL_LANES)
		{
			u32 pi = pre[first + i];
			uint4 ids = actIds[pi];
			mKey[i] = ids.x; mCnt[i] = ids.y; mAB[i] = pi; mSlot[i] = (ids.w & 0x0FFFFFFFu) | (ids.z << 28);
			for (u32 e = 0; e < 2; ++e)
			{
				u32 g = e ? ids.y : ids.x;
				if (g >= nb) continue;
				u32 h = clHash(g) & (CL_HASH_SIZE - 1u);
				for (;;)
				{
					u32 old = atomicCAS(&hKey[h], 0u, g + 1u);
					if (old == 0u || old == g + 1u) break;
					h = (h + 1u) & (CL_HASH_SIZE - 1u);
				}
			}
		}
		for (u32 i = tid; i < nj; i += CL_LANES) // the joints' bodies: a limb in the air has joints and no contact
		{
			uint4 e4 = jointTable[jointList[jFirst + i]];
			for (u32 e = 0; e < 2; ++e)
			{
				u32 g = e ? e4.w : e4.z;
				if (g >= nb) continue;
				u32 h = clHash(g) & (CL_HASH_SIZE - 1u);
				for (;;)
				{
					u32 old = atomicCAS(&hKey[h], 0u, g + 1u);
					if (old == 0u || old == g + 1u) break;
					h = (h + 1u) & (CL_HASH_SIZE - 1u);
				}
			}
		}
		__syncthreads();
		for (u32 h = tid; h < CL_HASH_SIZE; h += CL_LANES) // (uniform trip count: the ballots below see whole waves)
		{
			// local index = running count of the shared / private bodies: one LDS atomic per wave, not per body (same-address LDS
			// atomics are served one lane at a time)
			const bool has = hKey[h] != 0u;
			const bool shared = has && __popc(phaseMask[hKey[h] - 1u]) > 1;
			const u64 bs = __ballot(shared), bp = __ballot(has && !shared);
			const u32 lane = tid & 63u;
			u32 baseS = 0, baseP = 0;
			if (lane == 0u) { if (bs) baseS = atomicAdd(&sNumShared, (u32)__popcll(bs)); if (bp) baseP = atomicAdd(&sNumPrivate, (u32)__popcll(bp)); }
			baseS = __shfl(baseS, 0); baseP = __shfl(baseP, 0);
			const u64 below = (1ull << lane) - 1ull;
			if (has) hVal[h] = shared ? ((baseS + (u32)__popcll(bs & below)) | 0x80000000u) : baseP + (u32)__popcll(bp & below);
		}
		__syncthreads();
		const u32 numShared = sNumShared, numBodies = sNumShared + sNumPrivate;
		const bool tooMany = numBodies > CL_TASK_MAX_BODIES; // uniform
		if (tooMany) { if (tid == 0) { atomicOr(&counters[CTR_CL_STATUS], 2u); T->first = first; T->count = 0; T->numBodies = 0; T->numShared = 0; T->numColors = 0; T->serialStart = 0; T->numRows = 0; } __syncthreads(); continue; }
		// The task's shared bodies get a contiguous run of hand-over records (32 B each): its lanes publish them with coalesced stores, and
		// whoever uses a body next finds the record through sharedSlot[phase][body].  (The placement of the run depends on the order the
		// tasks get here; results do not.)
		if (tid == 0) sSharedBase = atomicAdd(&counters[CTR_CL_SHARED], numShared);
		__syncthreads();
		const u32 sharedBase = sSharedBase;
		for (u32 h = tid; h < CL_HASH_SIZE; h += CL_LANES)
			if (hKey[h])
			{
				u32 v = hVal[h];
				u32 l = (v & 0x80000000u) ? (v & 0x7FFFFFFFu) : numShared + v;
				hVal[h] = l;
				bodyList[(size_t)key * CL_BODY_STRIDE + l] = hKey[h] - 1u;
				if (l < numShared) sharedSlot[(size_t)phase * (nb + 1u) + (hKey[h] - 1u)] = sharedBase + l;
			}
		for (u32 l = tid; l <= numBodies; l += CL_LANES) { mask[l] = 0ull; claim[l] = 0xFFFFFFFFu; }
		__syncthreads();
		CL_STAMP(1)
		// 1b. The task's manifolds arrive in the order their append atomics landed.  The colouring below breaks bid ties by position, so
		// the positions are made a function of the inputs first: bitonic sort by {contact count, narrowphase slot} (unique per
		// manifold).  With that the whole schedule, and so every result, repeats from run to run (snapshot / restore continue
		// bit-identically).  Stages that exchange inside 128 consecutive elements stay inside one wave and need no workgroup barrier.
		{
			u32 m = 128u; while (m < n) m <<= 1;
			// one 64-bit word per element {key, arrival index} (a stage is then one LDS round trip: two reads, compare, two writes);
			// the words live in the colour-mask table, which the rounds need zeroed only afterwards
			u64* sk = mask;
			static_assert(CL_TASK_MAX_BODIES + 1u >= CL_TASK_MAX_MANIFOLDS, "the colour masks double as the sort's scratch");
			for (u32 i = tid; i < m; i += CL_LANES) sk[i] = ((u64)(i < n ? mSlot[i] : 0xFFFFFFFFu) << 32) | i;
			__syncthreads();
			for (u32 k = 2u; k <= m; k <<= 1)
				for (u32 j = k >> 1, lj = 31u - (u32)__clz(k >> 1); j > 0u; j >>= 1, --lj) // j = 1 << lj (no integer division in the index arithmetic)
				{
					if (tid < (m >> 1))
					{
						u32 a = ((tid >> lj) << (lj + 1u)) + (tid & (j - 1u)), b = a + j;
						bool up = (a & k) == 0u;
						u64 ka = sk[a], kb = sk[b];
						if ((ka > kb) == up) { sk[a] = kb; sk[b] = ka; }
					}
					if (j > 64u || (j == 1u && k >= 128u)) __syncthreads(); // the next stage (j / 2, or the next k's first) crosses the waves' 128-element blocks
					else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				}
			__syncthreads();
			for (u32 i = tid; i < n; i += CL_LANES) { u64 e = sk[i]; mSlot[i] = (u32)(e >> 32); mPos[i] = (u32)e; }
			__syncthreads();
			for (u32 l = tid; l <= numBodies; l += CL_LANES) mask[l] = 0ull;
			for (u32 i = tid; i < m; i += CL_LANES) mask[i] = 0ull; // (the scratch may reach beyond the bodies)
		}
		CL_STAMP(2)
		// 2. local ids of every manifold, in the sorted order (n <= 2 * CL_LANES: two per lane, gathered before anything is overwritten)
		{
			static_assert(CL_TASK_MAX_MANIFOLDS <= 2u * CL_LANES, "two manifolds per lane");
			u32 gx[2], gy[2], gp[2];
			for (u32 r = 0; r < 2u; ++r) { u32 i = tid + r * CL_LANES; if (i < n) { u32 o = mPos[i]; gx[r] = mKey[o]; gy[r] = mCnt[o]; gp[r] = mAB[o]; } }
			__syncthreads();
			for (u32 r = 0; r < 2u; ++r)
			{
				u32 i = tid + r * CL_LANES;
				if (i >= n) continue;
				u32 loc[2];
				for (u32 e = 0; e < 2; ++e)
				{
					u32 g = e ? gy[r] : gx[r];
					loc[e] = CL_LOCAL_STATIC;
					if (g >= nb) continue;
					u32 h = clHash(g) & (CL_HASH_SIZE - 1u);
					while (hKey[h] != g + 1u) h = (h + 1u) & (CL_HASH_SIZE - 1u);
					loc[e] = hVal[h];
				}
				mAB[i] = loc[0] | (loc[1] << 16);
				mKey[i] = 0xFFFFFFFFu;
				pre[first + i] = gp[r]; // (read again when the final order is written out)
			}
		}
		__syncthreads();
		// 2b. the joints: local ids of their bodies, ordered by (type, colour) class (counting sort: class sizes, offsets, cursors)
		if (nj)
		{
			for (u32 i = tid; i < nj; i += CL_LANES) atomicAdd(&jHist[min(jointTable[jointList[jFirst + i]].x >> 8, CL_MAX_JOINT_CLASSES - 1u)], 1u);
			__syncthreads();
			if (tid == 0)
			{
				u32 run = 0;
				for (u32 c = 0; c < CL_MAX_JOINT_CLASSES; ++c) { u32 v = jHist[c]; jHist[c] = run; jointClassStart[(size_t)key * (CL_MAX_JOINT_CLASSES + 2u) + c] = run; run += v; }
				jointClassStart[(size_t)key * (CL_MAX_JOINT_CLASSES + 2u) + CL_MAX_JOINT_CLASSES] = run;
			}
			__syncthreads();
			for (u32 i = tid; i < nj; i += CL_LANES)
			{
				u32 ji = jointList[jFirst + i];
				uint4 e4 = jointTable[ji];
				u32 loc[2];
				for (u32 e = 0; e < 2; ++e)
				{
					u32 g = e ? e4.w : e4.z;
					loc[e] = CL_LOCAL_STATIC;
					if (g >= nb) continue;
					u32 h = clHash(g) & (CL_HASH_SIZE - 1u);
					while (hKey[h] != g + 1u) h = (h + 1u) & (CL_HASH_SIZE - 1u);
					loc[e] = hVal[h];
				}
				u32 p = atomicAdd(&jHist[min(e4.x >> 8, CL_MAX_JOINT_CLASSES - 1u)], 1u); // (order inside a class is free: its joints share no body)
				taskJoints[jFirst + p] = make_uint2(ji, loc[0] | (loc[1] << 16));
			}
			__syncthreads();
		}
		CL_STAMP(3)
		// 3. colouring rounds.  A claim word holds {round, inverted bid}: a later round's bid beats any earlier one under atomicMax, so
		// the claims need no clearing between rounds: two barriers per round (bid | decide; the second one also tells whether anybody
		// is left).  No shared counters inside the rounds: a same-address LDS atomic from every lane costs more than the round itself.
		for (u32 l = tid; l < numBodies; l += CL_LANES) claim[l] = 0u;
		__syncthreads();
		// (a lane keeps its two manifolds' slot word, local body ids and colour in registers over the rounds: what it reads from LDS
		// in a round is the claims and the colour masks only)
		u32 rSlot[2], rAB[2], rKey[2];
		for (u32 r = 0; r < 2u; ++r) { u32 i = tid + r * CL_LANES; rKey[r] = 0u; rSlot[r] = 0u; rAB[r] = 0u; if (i < n) { rSlot[r] = mSlot[i]; rAB[r] = mAB[i]; rKey[r] = 0xFFFFFFFFu; } }
		for (u32 round = 0; ; ++round)
		{
			const bool lastRound = round >= 254u; // (the round tag has 8 bits: whoever is still uncoloured then goes to the serial tail)
			u32 bid[2];
			for (u32 r = 0; r < 2u; ++r)
			{
				if (rKey[r] != 0xFFFFFFFFu) continue;
				u32 sc = rSlot[r];
				bid[r] = ((round + 1u) << 24) | (0xFFFFFFu - clBid(sc & 0x0FFFFFFFu, sc >> 28, round, tid + r * CL_LANES));
				u32 la = rAB[r] & 0xFFFFu, lb = rAB[r] >> 16;
				if (la != CL_LOCAL_STATIC) atomicMax(&claim[la], bid[r]);
				if (lb != CL_LOCAL_STATIC) atomicMax(&claim[lb], bid[r]);
			}
			__syncthreads();
			u32 left = 0;
			for (u32 r = 0; r < 2u; ++r)
			{
				if (rKey[r] != 0xFFFFFFFFu) continue;
				u32 cnt = rSlot[r] >> 28;
				u32 la = rAB[r] & 0xFFFFu, lb = rAB[r] >> 16;
				bool won = (la == CL_LOCAL_STATIC || claim[la] == bid[r]) && (lb == CL_LOCAL_STATIC || claim[lb] == bid[r]);
				if (!won && !lastRound) { ++left; continue; }
				// A manifold of cnt contacts takes cnt CONSECUTIVE colours [c, c + cnt) on both bodies (the sweep's step is one contact row:
				// its contacts run in colours c, c + 1, ...), the lowest such run free on both: manifolds that share a body get disjoint
				// runs, so "by first colour" is still a sequential order of whole manifolds (what the schedule export reports).
				u64 used = (la != CL_LOCAL_STATIC ? mask[la] : 0ull) | (lb != CL_LOCAL_STATIC ? mask[lb] : 0ull);
				u64 fr = ~used;
				if (cnt > 1u) fr &= fr >> 1;
				if (cnt > 2u) fr &= fr >> 1;
				if (cnt > 3u) fr &= fr >> 1; // bit c set = colours c .. c + cnt - 1 all free (the shifts bring zeros in at the top: a run never passes colour 63)
				u32 c = (won && fr) ? (u32)__ffsll((long long)fr) - 1u : CL_SERIAL_COLOR;
				if (c < CL_SERIAL_COLOR)
				{
					const u64 run = ((cnt >= 64u ? 0ull : (1ull << cnt)) - 1ull) << c;
					if (la != CL_LOCAL_STATIC) mask[la] |= run; // the only winner on this body in this round
					if (lb != CL_LOCAL_STATIC) mask[lb] |= run;
				}
				rKey[r] = c * 4u + (4u - cnt);
			}
			if (!__syncthreads_or((int)left)) { if (trace && tid == 0 && key < CL_MAX_TASKS) trace[((size_t)key * 16u + 14u) * 32u + 15u] = round + 1u; break; }
		}
		for (u32 r = 0; r < 2u; ++r) { u32 i = tid + r * CL_LANES; if (i < n) mKey[i] = rKey[r]; }
		__syncthreads();
		CL_STAMP(4)
		// 4. manifold order by key = (first colour, contact count): histogram, scan by one wave, cursors.  This is the order the rows are
		// initialised in and the schedule export reports (manifold after manifold; manifolds that share a body have disjoint colour runs).
		for (u32 i = tid; i < n; i += CL_LANES)
		{
			const u32 k = mKey[i], c = k >> 2, cnt = 4u - (k & 3u);
			atomicAdd(&hist[k], 1u);
			// ... and the CONTACT histogram per colour (contact q of a manifold runs in colour c + q; the serial tail is class 64)
			if (c < CL_SERIAL_COLOR) { for (u32 q = 0; q < cnt; ++q) atomicAdd(&cHist[c + q], 1u); }
			else atomicAdd(&cHist[CL_SERIAL_COLOR], cnt);
		}
		__syncthreads();
		if (tid < 64u) // 260 keys, 5 per lane (65 colours x 4 counts): serial scan over 64 lanes
		{
			u32 base = tid * 5u, s = 0;
			u32 v[5];
			for (u32 k = 0; k < 5u; ++k) { v[k] = (base + k < 260u) ? hist[base + k] : 0u; s += v[k]; }
			u32 incl = s;
			for (int o = 1; o < 64; o <<= 1) { u32 up = __shfl_up(incl, o); if ((int)tid >= o) incl += up; }
			u32 run = incl - s;
			for (u32 k = 0; k < 5u; ++k) { if (base + k < 260u) hist[base + k] = run; run += v[k]; }
			// contacts per colour -> first contact position of every colour (lane = colour); number of colours in use
			const u32 cc = cHist[tid];
			u32 ci = cc;
			for (int o = 1; o < 64; o <<= 1) { u32 up = __shfl_up(ci, o); if ((int)tid >= o) ci += up; }
			const u32 coloured = (u32)__shfl(ci, 63), serial = cHist[CL_SERIAL_COLOR];
			u32 top = cc ? tid + 1u : 0u;
			for (int o = 32; o > 0; o >>= 1) top = max(top, (u32)__shfl_xor(top, o));
			cHist[tid] = ci - cc;
			if (tid == 0) { cHist[CL_SERIAL_COLOR] = coloured; cHist[CL_SERIAL_COLOR + 1u] = coloured + serial; sMaxColor = top; }
		}
		__syncthreads();
		const u32 numColors = sMaxColor, numContacts = cHist[CL_SERIAL_COLOR + 1u], serialStartC = cHist[CL_SERIAL_COLOR];
		if (tid <= CL_SERIAL_COLOR + 1u) T->colorStart[tid] = cHist[tid];
		if (tid == 0)
		{
			T->first = first; T->count = n; T->numBodies = numBodies; T->numShared = numShared; T->numColors = numColors; T->serialStart = serialStartC; T->numRows = numContacts;
			atomicMax(&counters[CTR_NUM_COLORS], numColors + (serialStartC < numContacts ? 1u : 0u));
			T->sharedBase = sharedBase;
			atomicAdd(&counters[CTR_CL_PHASE_COUNT + phase], n);
		}
		__syncthreads();
		// Positions inside a (colour, count) class: an atomic cursor (order inside a class is free, its manifolds share no body: the
		// RESULTS repeat from run to run, the memory order need not).
		// The serial tail IS order-dependent: its positions follow the (sorted) index.
		for (u32 i = tid; i < n; i += CL_LANES)
		{
			u32 k = mKey[i], p;
			if ((k >> 2) < CL_SERIAL_COLOR) p = atomicAdd(&hist[k], 1u);
			else { p = hist[k]; for (u32 j = 0; j < i; ++j) p += (mKey[j] == k) ? 1u : 0u; } // (rare: a body with more than 64 contacts in one task)
			mPos[i] = p;
		}
		__syncthreads();
		CL_STAMP(5)
		// 5. the contact schedule: contact q of a manifold of first colour c gets a position inside colour c + q (cursor: free order
		// inside a colour); the serial tail's contacts follow in manifold position order, a manifold's contacts in order.
		// Entry = manifold position inside the task | q << 12.
		for (u32 i = tid; i < n; i += CL_LANES)
		{
			const u32 k = mKey[i], c = k >> 2, cnt = 4u - (k & 3u), mp = mPos[i];
			if (c < CL_SERIAL_COLOR) { for (u32 q = 0; q < cnt; ++q) cEntry[(size_t)4u * first + atomicAdd(&cHist[c + q], 1u)] = mp | (q << 12); }
			else
			{
				u32 p = serialStartC;
				for (u32 j = 0; j < n; ++j) if ((mKey[j] >> 2) >= CL_SERIAL_COLOR && mPos[j] < mp) p += 4u - (mKey[j] & 3u);
				for (u32 q = 0; q < cnt; ++q) cEntry[(size_t)4u * first + p + q] = mp | (q << 12);
			}
		}
		CL_STAMP(6)
		for (u32 i = tid; i < n; i += CL_LANES)
		{
			u32 p = mPos[i];
			mOrder[first + p] = pre[first + i];
			mKeySorted[first + p] = mKey[i];
			mLocal[first + p] = mAB[i];
		}
		__syncthreads();
		CL_STAMP(7)
	}
	}
#undef CL_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// The sweep.
// ---------------------------------------------------------------------------------------------------------------
// FOUR lanes (a quad) work on one contact row: lane q owns one of the row's four body vectors x (q = 0: vA, 1: wA, 2: vB, 3: wB — one
// float4 of the body's LDS record) and its pieces of the row: dT / dN = what x is dotted with in the tangent / normal row velocity
// (body A's negated), aT / aN = what an impulse adds to x (inverse mass and sign folded in).  A row velocity is the sum of the
// four lanes' 3-term dots — two DPP adds inside the quad, every lane gets the bit-identical total (p0 + p1) + (p2 + p3) — the
// impulse update is done by all four lanes redundantly, each then updates its vector: solver_rows.h's solveRow, lane for lane.
// Measured against one lane per manifold (tests/micro/colorstep.hip): a colour step is ONE LDS access each way and ~27 vector
// instructions per lane instead of four and ~66 — 344 cycles against 707, 450 against 1 880 when all eight waves have work — and a
// manifold of k contacts is k such steps (its contacts run in consecutive colours) instead of one step of 700 + 465 (k - 1) cycles.
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct QuadRow { V3 dT, aT, dN, aN; float mT, mN, bias, friction, lamN, lamT; }; // 18 floats per lane
#define CLQ_ZERO_FLOAT4S (2u * CLQ_QUADS + 4u)
#define CLQ_ROW_FLOAT4S 14u // a contact row outside the registers (LDS, or the global scratch beyond LDS): per lane q three float4 {dT, aT.x} {aT.yz, dN.xy} {dN.z, aN} at 3 q, then {mT, mN, bias, friction}, {lamN, lamT, addresses of lanes 0 | 1 << 16, 2 | 3 << 16}

struct ClLocal // a task of this workgroup, in LDS
{
	u32 first, count, numBodies, numShared, numColors, serialStart, numContacts, phase, key, sharedBase, numJoints;
	u32 bodyOff;     // float4 index of the task's bodies (2 float4 each)
	u32 infoOff;     // u32 index of per-body {global id, turn info, hand-over record}
	u32 regContacts; // contacts [0, regContacts) live in the lanes' register sets (the workgroup's first task only)
	u32 rowOff;      // float4 index of the rows kept in LDS: contacts [regContacts, regContacts + rowCap)
	u32 rowCap;      // ... the rest, contacts [regContacts + rowCap, numContacts), in the global scratch from scratchBase on
	u32 scratchBase;
	u32 colorStart[68];
};

struct ClArgs
{
	u32* counters; const ClTask* tasks; const u32* bodyList; const u32* phaseMask; const u32* sharedSlot;
	const u32* mKeySorted; const u32* mLocal; const u32* cEntry;
	const float4* rowPlanes; const float4* rowShared; float2* rowLambda; float4* rowScratch; u32 scratchContacts;
	u32 predictDiv, pollSleep; // pacing of the hand-over polls (MI_CLUSTER_PREDICT_DIV / MI_CLUSTER_POLL_SLEEP)
	float4* vel; u64* flow; u64* trace; // trace: developer timeline (mi_debug_flow_trace), normally null
	size_t rowCap; u32 nb, flowBytes, epoch, itBegin, itEnd, ldsFloat4s;
	// joints run by the sweep (null / 0 when the world has none or they keep their own launches): per phase-0 task the class offsets
	// [CL_MAX_JOINT_CLASSES + 2] (last two: joint count, first entry), the class-sorted entries {table index, la | lb << 16}, the table
	// {type | class << 8, index in the type's arrays, body a, body b}, the per-type update records, world inverse inertia
	const u32* jointClassStart; const uint2* taskJoints; const uint4* jointTable; float* jointUpd[MI_JOINT_TYPES]; const float4* invIw; u32 numJointClasses;
};

MI_DEV float clQuadSum(float p)
{
	float q = p + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, false)); // quad_perm [1, 0, 3, 2]: lin + ang of one body
	return q + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x4E, 0xF, 0xF, false));   // quad_perm [2, 3, 0, 1]: body A's + body B's
}
// One contact row (friction, then normal: constraints.cpp:3404-3442) on this lane's body vector x.
MI_DEV void clSolveQuad(QuadRow& r, V3& x)
{
	float vt = clQuadSum(rowDot3(x, r.dT));
	float maxFriction = r.friction * r.lamN;
	float newT = rowClampSym(__builtin_fmaf(-r.mT, vt, r.lamT), maxFriction);
	float d = newT - r.lamT; r.lamT = newT;
	x = rowFma(d, r.aT, x);
	float vn = clQuadSum(rowDot3(x, r.dN));
	float newN = fmaxf(__builtin_fmaf(-r.mN, vn - r.bias, r.lamN), 0.f);
	d = newN - r.lamN; r.lamN = newN;
	x = rowFma(d, r.aN, x);
}
// Lane q's view of contact position p of task L, from the global row planes: its pieces of the row and the LDS address of its body
// vector (a static body: the all-zero record; its apply vectors are zero, so what is written back is the zero that was read).
MI_DEV void clBuildQuadRow(QuadRow& r, u32& addr, const ClLocal& L, const ClArgs& A, const float4* lds, u32 p, u32 q, u32 zeroRec, u32& slotOut, u32& kOut)
{
	const u32 e = A.cEntry[(size_t)4u * L.first + p], mp = e & 0xFFFu, k = e >> 12, slot = L.first + mp;
	const u32 ab = A.mLocal[slot], local = q < 2u ? (ab & 0xFFFFu) : (ab >> 16);
	const float4 sh = A.rowShared[slot];
	ContactRow row; loadRow(row, k, slot, A.rowCap, A.rowPlanes, A.rowLambda);
	const bool isStatic = local == CL_LOCAL_STATIC;
	addr = (isStatic ? zeroRec : L.bodyOff + 2u * local) + (q & 1u);
	const float invMass = isStatic ? 0.f : lds[L.bodyOff + 2u * local].w; // (.w of a body's first float4 = its inverse mass, constant over the launch)
	const V3 t = v3(row.p0.x, row.p0.y, row.p0.z), n = v3(sh.x, sh.y, sh.z);
	if (q == 0u) { r.dT = -t; r.aT = -(invMass * t); r.dN = -n; r.aN = -(invMass * n); }
	else if (q == 1u) { r.dT = -v3(row.p0.w, row.p1.x, row.p1.y); r.aT = -v3(row.p3.w, row.p4.x, row.p4.y); r.dN = -v3(row.p2.y, row.p2.z, row.p2.w); r.aN = -v3(row.p5.y, row.p5.z, row.p5.w); }
	else if (q == 2u) { r.dT = t; r.aT = invMass * t; r.dN = n; r.aN = invMass * n; }
	else { r.dT = v3(row.p1.z, row.p1.w, row.p2.x); r.aT = v3(row.p4.z, row.p4.w, row.p5.x); r.dN = v3(row.p3.x, row.p3.y, row.p3.z); r.aN = v3(row.p6.x, row.p6.y, row.p6.z); }
	r.mT = row.p7.x; r.mN = row.p6.w; r.bias = row.p7.y; r.friction = sh.w; r.lamN = row.lam.x; r.lamT = row.lam.y;
	slotOut = slot; kOut = k;
}
// Row storage outside the registers (P = the contact's CLQ_ROW_FLOAT4S float4): every lane stores its vectors, lane 0 the scalars and lane
// 0 / 2 the packed addresses of their pair of lanes (the addresses of a quad: two bodies x {linear, angular} = base and base + 1).
template <typename PTR> MI_DEV void clStoreQuadRow(PTR P, u32 q, const QuadRow& r, u32 addr)
{
	P[3u * q] = make_float4(r.dT.x, r.dT.y, r.dT.z, r.aT.x); P[3u * q + 1u] = make_float4(r.aT.y, r.aT.z, r.dN.x, r.dN.y); P[3u * q + 2u] = make_float4(r.dN.z, r.aN.x, r.aN.y, r.aN.z);
	if (q == 0u) P[12] = make_float4(r.mT, r.mN, r.bias, r.friction);
	const u32 other = (u32)__builtin_amdgcn_update_dpp(0, (int)addr, 0xB1, 0xF, 0xF, false); // the partner lane's address (quad_perm [1, 0, 3, 2])
	const u32 otherPair = (u32)__builtin_amdgcn_update_dpp(0, (int)(addr | (other << 16)), 0x4E, 0xF, 0xF, false); // lanes 2 | 3 << 16 seen from lane 0
	if (q == 0u) P[13] = make_float4(r.lamN, r.lamT, __uint_as_float(addr | (other << 16)), __uint_as_float(otherPair));
}
template <typename PTR> MI_DEV void clLoadQuadRow(QuadRow& r, u32& addr, PTR P, u32 q)
{
	float4 a = P[3u * q], b = P[3u * q + 1u], c = P[3u * q + 2u], d = P[12], e = P[13];
	r.dT = v3f4(a); r.aT = v3(a.w, b.x, b.y); r.dN = v3(b.z, b.w, c.x); r.aN = v3(c.y, c.z, c.w);
	r.mT = d.x; r.mN = d.y; r.bias = d.z; r.friction = d.w; r.lamN = e.x; r.lamT = e.y;
	const u32 pair = __float_as_uint(q < 2u ? e.z : e.w);
	addr = (q & 1u) ? (pair >> 16) : (pair & 0xFFFFu);
}

// JOINTS: the instantiation for worlds whose joints run inside the sweep (its extra registers and code stay out of the other one).
template <bool JOINTS> __global__ void __launch_bounds__(CLS_LANES) k_cl_solve(ClArgs A)
{
	extern __shared__ float4 lds[];
	__shared__ ClLocal sTask[CL_MAX_LOCAL_TASKS];
	__shared__ u32 sNumTasks, sAbort;
	const u32 tid = threadIdx.x, G = gridDim.x, quad = tid >> 2, q = tid & 3u;
	constexpr u32 SETS = JOINTS ? CLQ_SETS_JOINTS : CLQ_SETS;
	u32* status = A.counters + CTR_FLOW_STATUS;
	__amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A.flow, 0, A.flowBytes, 0x00020000);
	// The static body: an all-zero record.  Its apply vectors are zero, so a contact lane writes back the zero it read; every quad has its own copy
	// (CLQ_ZERO_FLOAT4S float4 at the end of LDS: a third of a pile's contacts touch the ground, and same-address writes of one wave
	// instruction are served one after the other).  The joints read a shared copy and write into a sink.
	const u32 zeroBase = A.ldsFloat4s - CLQ_ZERO_FLOAT4S, zeroRec = zeroBase + 2u * CLQ_QUADS, sinkRec = zeroRec + 2u;

	// ---- which tasks are mine, and where they live in LDS ----
	if (tid == 0)
	{
		if (A.trace) A.trace[(size_t)blockIdx.x * 16u * 32u + 15 * 32] = wall_clock64();
		u32 nT = 0, off = 0, used = 0; // used: float4s of LDS handed out
		bool bad = A.counters[CTR_CL_STATUS] != 0u;
		for (u32 p = 0; p < CL_MAX_PHASES; ++p)
		{
			u32 tasksInPhase = A.counters[CTR_CL_NUM_TASKS + p];
			if (tasksInPhase > CL_TASKS_PER_PHASE * G) { bad = true; atomicOr(status, 128u); }
			// task t of phase p runs on workgroup (clPhaseOffset + t) % G; a phase with more tasks than workgroups wraps around (its tasks
			// share no body, so a workgroup may run two of them one after the other)
			off = clPhaseOffset(A.counters, p);
			u32 t = (blockIdx.x + G - (off % G)) % G;
			for (; t < tasksInPhase && !bad; t += G)
			{
				u32 key = p * CL_MAX_TASKS + t;
				const ClTask* T = A.tasks + key;
				const u32 tj = (A.jointClassStart && p == 0u) ? A.jointClassStart[(size_t)key * (CL_MAX_JOINT_CLASSES + 2u) + CL_MAX_JOINT_CLASSES] : 0u; // joints of the task
				if (!T->count && !tj) continue;
				if (nT == CL_MAX_LOCAL_TASKS) { bad = true; atomicOr(status, 256u); break; }
				ClLocal& L = sTask[nT];
				L.first = T->first; L.count = T->count; L.numBodies = T->numBodies; L.numShared = T->numShared; L.numColors = T->numColors; L.serialStart = T->serialStart; L.numContacts = T->numRows;
				L.phase = p; L.key = key; L.sharedBase = T->sharedBase; L.numJoints = tj;
				if (tj > CLS_LANES || (tj && nT)) { bad = true; atomicOr(status, 2048u); } // one lane per joint; joints run with the workgroup's first task only
				for (u32 c = 0; c <= CL_SERIAL_COLOR + 1u; ++c) L.colorStart[c] = T->colorStart[c];
				L.colorStart[CL_SERIAL_COLOR + 2u] = L.colorStart[CL_SERIAL_COLOR + 1u]; L.colorStart[CL_SERIAL_COLOR + 3u] = L.colorStart[CL_SERIAL_COLOR + 1u]; // (the colour loop reads two entries ahead)
				L.bodyOff = used; used += 2u * L.numBodies;
				L.infoOff = used * 4u; used += (3u * L.numBodies + 3u) / 4u;
				L.regContacts = (nT == 0) ? min(L.numContacts, CLQ_QUADS * SETS) : 0u;
				L.rowOff = 0; L.rowCap = 0;
				++nT;
			}
		}
		if (used + CLQ_ZERO_FLOAT4S > A.ldsFloat4s) { bad = true; atomicOr(status, 512u); } // the bodies alone exceed LDS: cannot run this launch
		// rows beyond the register sets: whatever LDS is left, in task order
		for (u32 k = 0; k < nT && !bad; ++k)
		{
			ClLocal& L = sTask[k];
			u32 want = L.numContacts - L.regContacts;
			u32 left = A.ldsFloat4s - used - CLQ_ZERO_FLOAT4S; // (the end of LDS holds the static body's all-zero records and the sink)
			u32 fit = left / CLQ_ROW_FLOAT4S;
			u32 cap = want < fit ? want : fit;              // what does not fit goes to the global scratch (L2-resident, every step of such a task waits for it: the cluster build sizes the later phases' tasks so that this is rare)
			L.rowOff = used; L.rowCap = cap; used += CLQ_ROW_FLOAT4S * cap; L.scratchBase = 0;
			if (want > cap) { L.scratchBase = atomicAdd(&A.counters[CTR_CL_SCRATCH], want - cap); if (L.scratchBase + (want - cap) > A.scratchContacts) { bad = true; atomicOr(status, 1024u); break; } }
		}
		if (bad) atomicOr(status, 64u);
		sNumTasks = nT; sAbort = (bad || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1u : 0u;
	}
	__syncthreads();
	if (sAbort) return; // uniform: the launch cannot run (or somebody has given up already); the host redoes the step
	const u32 numTasks = sNumTasks;
	if (!numTasks) return;

	// ---- prologue: bodies, then the rows ----
	for (u32 k = 0; k < numTasks; ++k)
	{
		const ClLocal& L = sTask[k];
		u32* info = (u32*)lds + L.infoOff;
		for (u32 l = tid; l < L.numBodies; l += CLS_LANES)
		{
			u32 g = A.bodyList[(size_t)L.key * CL_BODY_STRIDE + l];
			u32 pm = A.phaseMask[g];
			u32 deg = __popc(pm), rank = __popc(pm & ((1u << L.phase) - 1u));
			// where the body comes from: the record of the phase that used it last (the last phase of the previous iteration for this
			// iteration's first user)
			u32 below = pm & ((1u << L.phase) - 1u);
			u32 prev = below ? 31u - (u32)__clz(below) : 31u - (u32)__clz(pm);
			info[3 * l] = g; info[3 * l + 1] = deg | (rank << 8);
			info[3 * l + 2] = (l < L.numShared) ? A.sharedSlot[(size_t)prev * (A.nb + 1u) + g] : 0u;
			lds[L.bodyOff + 2 * l] = A.vel[2 * g]; lds[L.bodyOff + 2 * l + 1] = A.vel[2 * g + 1]; // shared ones too: .w = invMass stays, the rest is replaced at every acquire
		}
	}
	for (u32 i = tid; i < CLQ_ZERO_FLOAT4S; i += CLS_LANES) lds[zeroBase + i] = make_float4(0.f, 0.f, 0.f, 0.f);
	__syncthreads();
	// the first task's contacts in registers: set s of this lane's quad holds contact position s * CLQ_QUADS + quad
	QuadRow rows[SETS]; u32 addr[SETS];
	{
		const ClLocal& L = sTask[0];
#pragma unroll
		for (u32 s = 0; s < SETS; ++s)
		{
			const u32 p = s * CLQ_QUADS + quad;
			rows[s].dT = rows[s].aT = rows[s].dN = rows[s].aN = v3(0.f, 0.f, 0.f); rows[s].mT = rows[s].mN = rows[s].bias = rows[s].friction = rows[s].lamN = rows[s].lamT = 0.f; addr[s] = sinkRec;
			if (p < L.regContacts) { u32 slot, kk; clBuildQuadRow(rows[s], addr[s], L, A, lds, p, q, zeroBase + 2u * quad, slot, kk); }
		}
	}
	// every other contact of the workgroup's tasks: its four lane rows in LDS while there is room
	for (u32 k = 0; k < numTasks; ++k)
	{
		const ClLocal& L = sTask[k];
		for (u32 i = quad; i < L.numContacts - L.regContacts; i += CLQ_QUADS)
		{
			QuadRow r; u32 a, slot, kk;
			clBuildQuadRow(r, a, L, A, lds, L.regContacts + i, q, zeroBase + 2u * quad, slot, kk);
			if (i < L.rowCap) clStoreQuadRow(lds + L.rowOff + i * CLQ_ROW_FLOAT4S, q, r, a);
			else clStoreQuadRow(A.rowScratch + (size_t)(L.scratchBase + i - L.rowCap) * CLQ_ROW_FLOAT4S, q, r, a);
		}
	}
	// The last colours of a task hold a handful of contacts (the busiest body's last rows).  Those of the first task whose positions all
	// fall into ONE block of 16 positions belong to one wave (16 quads of one register set): that wave runs them back to back, in
	// program order, without the workgroup barrier in between (LDS serves a wave's accesses in order).  tailStart0 = first such
	// colour, tailSet = their register set, myTail = this lane's colour among them (255: none).
	u32 tailStart0 = sTask[0].numColors, tailSet = 0, myTail = 255u;
	{
		const ClLocal& L = sTask[0];
		const u32 endPos = L.colorStart[L.numColors];
		if (L.numColors && endPos && endPos <= L.regContacts)
		{
			const u32 blk = (endPos - 1u) >> 4;
			while (tailStart0 > 0u && (L.colorStart[tailStart0 - 1u] >> 4) == blk) --tailStart0;
			if (L.numColors - tailStart0 < 2u) tailStart0 = L.numColors;
			else
			{
				tailSet = L.colorStart[tailStart0] / CLQ_QUADS;
				const u32 p = tailSet * CLQ_QUADS + quad;
				if (p >= L.colorStart[tailStart0] && p < endPos) { myTail = tailStart0; while (L.colorStart[myTail + 1u] <= p) ++myTail; }
			}
		}
		tailStart0 = __builtin_amdgcn_readfirstlane(tailStart0); tailSet = __builtin_amdgcn_readfirstlane(tailSet);
	}
	// this lane's joint (first task only, phase 0): class, update record, the two bodies as LDS addresses and as global ids (inverse inertia)
	const u32 bodyOff0 = sTask[0].bodyOff;
	u32 jClass = 0xFFFFFFFFu, jType = 0, jA = 0, jB = 0, jRdA = zeroRec, jWrA = sinkRec, jRdB = zeroRec, jWrB = sinkRec; float* jRec = nullptr;
	const u32 numJoints0 = (JOINTS && sTask[0].phase == 0u) ? sTask[0].numJoints : 0u;
	if (JOINTS && tid < numJoints0)
	{
		const u32* cs = A.jointClassStart + (size_t)sTask[0].key * (CL_MAX_JOINT_CLASSES + 2u);
		uint2 e = A.taskJoints[cs[CL_MAX_JOINT_CLASSES + 1u] + tid];
		uint4 t4 = A.jointTable[e.x];
		jType = t4.x & 0xFFu; jClass = t4.x >> 8; jA = t4.z; jB = t4.w;
		jRec = A.jointUpd[jType] + (size_t)t4.y * jointUpdateFloats(jType);
		u32 la = e.y & 0xFFFFu, lb = e.y >> 16;
		if (la != CL_LOCAL_STATIC) { jRdA = bodyOff0 + 2u * la; jWrA = jRdA; }
		if (lb != CL_LOCAL_STATIC) { jRdB = bodyOff0 + 2u * lb; jWrB = jRdB; }
	}
	__syncthreads();
	// developer timeline: 16 rows of 32 stamps per workgroup: row 3 k = "task k acquired its shared bodies" in iteration (column), row
	// 3 k + 1 = "task k's colours done"; rows 5-6: core-clock stamp after every colour of iteration 10 of the first task, rows 7-8: the
	// colours' sizes; row 15: [0] kernel start, [1] prologue done, [2 + 4 k ..] task k's size, colours, shared bodies, phase
	u64* trace = A.trace ? A.trace + (size_t)blockIdx.x * 16u * 32u : nullptr;
	if (trace && tid == 0)
	{
		trace[15 * 32 + 1] = wall_clock64();
		for (u32 k = 0; k < numTasks && k < 5u; ++k) { trace[15 * 32 + 2 + 4 * k] = sTask[k].count; trace[15 * 32 + 3 + 4 * k] = sTask[k].numColors; trace[15 * 32 + 4 + 4 * k] = sTask[k].numShared; trace[15 * 32 + 5 + 4 * k] = sTask[k].phase | (sTask[k].numBodies << 8) | ((u64)sTask[k].numContacts << 32); }
	}

	// One step = the contacts at positions [cs, end) of a task (one colour, or one contact of the serial tail).  Positions below the
	// task's regContacts: register set position / CLQ_QUADS of quad position % CLQ_QUADS; beyond: rows from LDS / the scratch.  The
	// colour loops below are written SET BY SET (a colour's positions are consecutive, so the colours that begin in set S are a run
	// of the loop; one that reaches beyond set S + 1 is cut into two steps), so that the code of a step names its one or two
	// register sets statically: no dispatch, and nothing of the other sets passes through the loop.
#define CLQ_SOLVE_SET(S_, CS_, END_) if ((S_) < SETS) { const u32 pos = (S_) * CLQ_QUADS + quad; if (pos >= (CS_) && pos < (END_)) { float4 b = lds[addr[(S_) < SETS ? (S_) : 0u]]; V3 x = v3f4(b); clSolveQuad(rows[(S_) < SETS ? (S_) : 0u], x); lds[addr[(S_) < SETS ? (S_) : 0u]] = make_float4(x.x, x.y, x.z, b.w); } }
#define CLQ_SOLVE_ROWS(CS_, END_) \
	for (u32 p = max((CS_), regC) + quad; p < (END_); p += CLQ_QUADS) \
	{ \
		const u32 i = p - regC; \
		QuadRow r; u32 a; \
		float4* P = lds + rowOff + i * CLQ_ROW_FLOAT4S; \
		float4* S = A.rowScratch + (size_t)(scratchBase + i - rowCap) * CLQ_ROW_FLOAT4S; \
		if (i < rowCap) clLoadQuadRow(r, a, P, q); else clLoadQuadRow(r, a, S, q); \
		float4 b = lds[a]; V3 x = v3f4(b); clSolveQuad(r, x); lds[a] = make_float4(x.x, x.y, x.z, b.w); \
		if (q == 0u) { if (i < rowCap) ((float2*)(P + 13))[0] = make_float2(r.lamN, r.lamT); else ((float2*)(S + 13))[0] = make_float2(r.lamN, r.lamT); } \
	}
#define CLQ_ADVANCE(END_) \
	__syncthreads(); \
	if (stamp && c < 63u && (END_) == csNext) { trace[5 * 32 + 1 + c] = clock64(); trace[7 * 32 + c] = csNext - L.colorStart[c]; } \
	if ((END_) == csNext) { csCur = csNext; csNext = __builtin_amdgcn_readfirstlane(csAfter); ++c; } else csCur = (END_);
#define CLQ_SET_LOOP(S_) \
	if ((S_) < SETS) while (c < mainColors && csCur < min(regC, ((S_) + 1u) * CLQ_QUADS)) \
	{ \
		const u32 csAfter = L.colorStart[c + 2u]; /* (requested now, needed at the next colour: the LDS round trip hides behind this step) */ \
		const u32 end = min(csNext, ((S_) + 2u) * CLQ_QUADS), endReg = min(end, regC); \
		CLQ_SOLVE_SET(S_, csCur, endReg) \
		if (end > ((S_) + 1u) * CLQ_QUADS) { CLQ_SOLVE_SET((S_) + 1u, csCur, endReg) if (end > regC) { CLQ_SOLVE_ROWS(csCur, end) } } \
		CLQ_ADVANCE(end) \
	}
	static_assert(CLQ_SETS <= 8u, "the colour loop below names eight sets");

	// ---- iterations ----
	bool aborted = false;
	u64 lastPublish = 0; u32 lastWait = 0; // when this workgroup last handed its bodies on, and how long (10 ns ticks) the bodies of its first task then took to come back
	for (u32 it = A.itBegin; it < A.itEnd && !aborted; ++it)
	{
		for (u32 k = 0; k < numTasks; ++k)
		{
			const ClLocal& L = sTask[k];
			const u32* info = (const u32*)lds + L.infoOff;
			// acquire the bodies other phases also touch.  The sweep is periodic: a task's bodies come back about one iteration
			// period after they came back last time, so the workgroup sleeps through most of the previous wait before it polls (the
			// polls are uncached loads through the fabric: 50k lanes polling all the time slow every hand-over down); then every lane
			// polls the tagged halves of up to two bodies per pass, all loads in flight together, and fetches the second half
			// (stored before the first) once the tag has arrived.
			{
				if (k == 0 && lastWait > 64u && it > A.itBegin + 1u)
				{
					u64 until = lastPublish + (u64)(lastWait - lastWait / A.predictDiv);
					while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
				}
				const u32 rel = it - A.itBegin;
				for (u32 base = 0; base < L.numShared; base += 2u * CLS_LANES)
				{
					u32 gid[2], want[2]; bool pend[2]; bool any = false;
#pragma unroll
					for (u32 qq = 0; qq < 2; ++qq)
					{
						u32 l = base + qq * CLS_LANES + tid;
						pend[qq] = false; gid[qq] = 0; want[qq] = 0;
						if (l >= L.numShared) continue;
						u32 ti = info[3 * l + 1], deg = ti & 0xFFu, rank = ti >> 8;
						if (rel == 0u && rank == 0u) continue; // first user of the launch: the prologue's copy of vel is current
						gid[qq] = info[3 * l + 2]; want[qq] = A.epoch + rel * deg + rank; pend[qq] = true; any = true;
					}
					u32 spins = 0;
					while (any)
					{
						u32x4 h0[2], h1[2];
						asm volatile("" ::: "memory");
#pragma unroll
						for (u32 qq = 0; qq < 2; ++qq)
							if (pend[qq]) { h0[qq] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gid[qq] * 32u, 0, 16); h1[qq] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gid[qq] * 32u + 16u, 0, 16); }
						any = false;
#pragma unroll
						for (u32 qq = 0; qq < 2; ++qq)
						{
							if (!pend[qq]) continue;
							if (h0[qq].w == want[qq] && h1[qq].w == want[qq]) // each half carries its own tag
							{
								u32 l = base + qq * CLS_LANES + tid;
								float invMass = lds[L.bodyOff + 2 * l].w; // constant over the launch
								lds[L.bodyOff + 2 * l] = make_float4(__uint_as_float(h0[qq].x), __uint_as_float(h0[qq].y), __uint_as_float(h0[qq].z), invMass);
								lds[L.bodyOff + 2 * l + 1] = make_float4(__uint_as_float(h1[qq].x), __uint_as_float(h1[qq].y), __uint_as_float(h1[qq].z), 0.f);
								pend[qq] = false;
							}
							any = any || pend[qq];
						}
						if (any)
						{
							if (++spins > CL_SPIN_LIMIT) { atomicOr(status, 1u); sAbort = 1u; break; }
							if ((spins & 63u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { sAbort = 1u; break; }
							if (A.pollSleep == 1u) __builtin_amdgcn_s_sleep(1); else if (A.pollSleep >= 2u) __builtin_amdgcn_s_sleep(4);
						}
					}
				}
			}
			__syncthreads();
			if (sAbort) { aborted = true; break; }
			if (k == 0 && it > A.itBegin) lastWait = (u32)(wall_clock64() - lastPublish);
			if (trace && tid == 0 && it - A.itBegin < 32u && k < 5u) trace[(3 * k) * 32 + (it - A.itBegin)] = wall_clock64();
			// joints first (constraints.cpp:3748-3772: all joint types, then the contacts): one (type, colour) class per step
			if (JOINTS && k == 0 && numJoints0)
			{
				const u32* cs = A.jointClassStart + (size_t)L.key * (CL_MAX_JOINT_CLASSES + 2u);
				for (u32 c = 0; c < A.numJointClasses; ++c)
				{
					if (cs[c + 1u] == cs[c]) continue; // (uniform: the class has no joint in this task)
					if (jClass == c)
					{
						Vel v; float4 a0 = lds[jRdA], a1 = lds[jRdA + 1], b0 = lds[jRdB], b1 = lds[jRdB + 1];
						v.vA = v3f4(a0); v.wA = v3f4(a1); v.vB = v3f4(b0); v.wB = v3f4(b1); v.invMassA = a0.w; v.invMassB = b0.w;
						M3 IA = ldInvI(A.invIw, jA), IB = ldInvI(A.invIw, jB);
						jointSolve(jType, jRec, v, IA, IB);
						lds[jWrA] = make_float4(v.vA.x, v.vA.y, v.vA.z, v.invMassA); lds[jWrA + 1] = make_float4(v.wA.x, v.wA.y, v.wA.z, 0.f);
						lds[jWrB] = make_float4(v.vB.x, v.vB.y, v.vB.z, v.invMassB); lds[jWrB + 1] = make_float4(v.wB.x, v.wB.y, v.wB.z, 0.f);
					}
					__syncthreads();
				}
			}
			// the contacts, colour by colour (a colour = one row solve per quad), then the serial tail one contact per step
			{
				const u32 numColors = __builtin_amdgcn_readfirstlane(L.numColors), serialStart = __builtin_amdgcn_readfirstlane(L.serialStart), numContacts = __builtin_amdgcn_readfirstlane(L.numContacts);
				// (what a step needs of the task record, in scalar registers: read from LDS once per turn, not behind every barrier)
				const u32 regC = __builtin_amdgcn_readfirstlane(L.regContacts), rowOff = __builtin_amdgcn_readfirstlane(L.rowOff), rowCap = __builtin_amdgcn_readfirstlane(L.rowCap), scratchBase = __builtin_amdgcn_readfirstlane(L.scratchBase);
				const bool stamp = trace && tid == 0 && it == A.itBegin + 10u && k == 0;
				if (stamp) trace[5 * 32] = clock64();
				const u32 mainColors = (k == 0u) ? tailStart0 : numColors;
				u32 c = 0, csCur = __builtin_amdgcn_readfirstlane(L.colorStart[0]), csNext = __builtin_amdgcn_readfirstlane(L.colorStart[1]);
				CLQ_SET_LOOP(0) CLQ_SET_LOOP(1) CLQ_SET_LOOP(2) CLQ_SET_LOOP(3) CLQ_SET_LOOP(4) CLQ_SET_LOOP(5) CLQ_SET_LOOP(6) CLQ_SET_LOOP(7)
				while (c < mainColors) // colours that lie entirely beyond the register sets (a task that is not the workgroup's first, or larger than the sets)
				{
					const u32 csAfter = L.colorStart[c + 2u];
					CLQ_SOLVE_ROWS(csCur, csNext)
					CLQ_ADVANCE(csNext)
				}
				if (mainColors < numColors) // the first task's trailing colours: one wave, no workgroup barrier in between
				{
#define CLQ_TAIL(S_) case S_: if ((S_) < SETS && myTail != 255u) for (u32 ct = mainColors; ct < numColors; ++ct) { if (myTail == ct) { float4 b = lds[addr[(S_) < SETS ? (S_) : 0u]]; V3 x = v3f4(b); clSolveQuad(rows[(S_) < SETS ? (S_) : 0u], x); lds[addr[(S_) < SETS ? (S_) : 0u]] = make_float4(x.x, x.y, x.z, b.w); } __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); } break;
					switch (tailSet) { CLQ_TAIL(0) CLQ_TAIL(1) CLQ_TAIL(2) CLQ_TAIL(3) CLQ_TAIL(4) CLQ_TAIL(5) CLQ_TAIL(6) CLQ_TAIL(7) default: break; }
#undef CLQ_TAIL
					__syncthreads();
					if (stamp) for (u32 ct = mainColors; ct < numColors && ct < 63u; ++ct) { trace[5 * 32 + 1 + ct] = clock64(); trace[7 * 32 + ct] = L.colorStart[ct + 1u] - L.colorStart[ct]; }
				}
				for (u32 sp = serialStart; sp < numContacts; ++sp) // the serial tail (manifolds that found no colour run below 64): one contact per step
				{
					if (sp < regC)
					{
						switch (sp / CLQ_QUADS)
						{
							case 0: CLQ_SOLVE_SET(0u, sp, sp + 1u) break; case 1: CLQ_SOLVE_SET(1u, sp, sp + 1u) break; case 2: CLQ_SOLVE_SET(2u, sp, sp + 1u) break; case 3: CLQ_SOLVE_SET(3u, sp, sp + 1u) break;
							case 4: CLQ_SOLVE_SET(4u, sp, sp + 1u) break; case 5: CLQ_SOLVE_SET(5u, sp, sp + 1u) break; case 6: CLQ_SOLVE_SET(6u, sp, sp + 1u) break; default: CLQ_SOLVE_SET(7u, sp, sp + 1u) break;
						}
					}
					else { CLQ_SOLVE_ROWS(sp, sp + 1u) }
					__syncthreads();
				}
			}
			if (trace && tid == 0 && it - A.itBegin < 32u && k < 5u) trace[(3 * k + 1) * 32 + (it - A.itBegin)] = wall_clock64();
			// hand the shared bodies on
			for (u32 l = tid; l < L.numShared; l += CLS_LANES)
			{
				u32 g = info[3 * l], ti = info[3 * l + 1];
				u32 deg = ti & 0xFFu, rank = ti >> 8;
				u32 want = A.epoch + (it - A.itBegin) * deg + rank;
				u32 rec = (L.sharedBase + l) * 32u; // consecutive lanes, consecutive records: the write-through stores coalesce
				float4 b0 = lds[L.bodyOff + 2 * l], b1 = lds[L.bodyOff + 2 * l + 1];
				if (it + 1u == A.itEnd && rank + 1u == deg) { A.vel[2 * g] = b0; A.vel[2 * g + 1] = make_float4(b1.x, b1.y, b1.z, 0.f); } // last user of the launch
				else
				{
					u32x4 h1 = { __float_as_uint(b1.x), __float_as_uint(b1.y), __float_as_uint(b1.z), want + 1u };
					u32x4 h0 = { __float_as_uint(b0.x), __float_as_uint(b0.y), __float_as_uint(b0.z), want + 1u };
					__builtin_amdgcn_raw_buffer_store_b128(h1, rsrc, rec + 16u, 0, 16);
					__builtin_amdgcn_raw_buffer_store_b128(h0, rsrc, rec, 0, 16);
				}
			}
			if (k + 1u == numTasks) lastPublish = wall_clock64();
		}
	}
	if (aborted) return; // the host redoes the step (World::recoverSolve)

	// ---- epilogue: task-private bodies and the accumulated impulses go home ----
	for (u32 k = 0; k < numTasks; ++k)
	{
		const ClLocal& L = sTask[k];
		const u32* info = (const u32*)lds + L.infoOff;
		for (u32 l = L.numShared + tid; l < L.numBodies; l += CLS_LANES)
		{
			u32 g = info[3 * l];
			float4 b1 = lds[L.bodyOff + 2 * l + 1];
			A.vel[2 * g] = lds[L.bodyOff + 2 * l]; A.vel[2 * g + 1] = make_float4(b1.x, b1.y, b1.z, 0.f);
		}
		if (q == 0u) // (a quad's four lanes hold the same impulses)
		{
			if (k == 0)
			{
#pragma unroll
				for (u32 s = 0; s < SETS; ++s)
				{
					const u32 p = s * CLQ_QUADS + quad;
					if (p < L.regContacts) { const u32 e = A.cEntry[(size_t)4u * L.first + p]; A.rowLambda[(size_t)(e >> 12) * A.rowCap + L.first + (e & 0xFFFu)] = make_float2(rows[s].lamN, rows[s].lamT); }
				}
			}
			for (u32 i = quad; i < L.numContacts - L.regContacts; i += CLQ_QUADS)
			{
				const u32 e = A.cEntry[(size_t)4u * L.first + L.regContacts + i];
				const float4 lam = i < L.rowCap ? lds[L.rowOff + i * CLQ_ROW_FLOAT4S + 13u] : A.rowScratch[(size_t)(L.scratchBase + i - L.rowCap) * CLQ_ROW_FLOAT4S + 13u];
				A.rowLambda[(size_t)(e >> 12) * A.rowCap + L.first + (e & 0xFFFu)] = make_float2(lam.x, lam.y);
			}
		}
	}
#undef CLQ_SET_LOOP
#undef CLQ_ADVANCE
#undef CLQ_SOLVE_ROWS
#undef CLQ_SOLVE_SET
}

// ---------------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------------
static size_t clColorLdsBytes()
{
	return sizeof(u32) * (2 * CL_HASH_SIZE + (CL_TASK_MAX_BODIES + 1) + 5 * CL_TASK_MAX_MANIFOLDS + 264 + CL_MAX_JOINT_CLASSES + 1 + CL_SERIAL_COLOR + 2) + sizeof(u64) * (CL_TASK_MAX_BODIES + 1);
}

bool cluster_solves_joints(const World& w) { return w.clJointsInCluster && w.useClusterJoints; }

bool cluster_available(World& w)
{
	if (w.clusterLdsBytes) return w.clusterLdsBytes != ~0u;
	int maxLds = 0, cus = 0;
	MI_CHECK(hipDeviceGetAttribute(&maxLds, hipDeviceAttributeMaxSharedMemoryPerBlock, w.device));
	MI_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w.device));
	hipFuncAttributes fa = {};
	MI_CHECK(hipFuncGetAttributes(&fa, (const void*)k_cl_solve<true>));
	size_t dyn = (maxLds > 0 ? (size_t)maxLds : 65536) - fa.sharedSizeBytes - 256;
	dyn &= ~(size_t)15;
	if (hipFuncSetAttribute((const void*)k_cl_solve<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess
		|| hipFuncSetAttribute((const void*)k_cl_solve<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess
		|| hipFuncSetAttribute((const void*)k_cl_color, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clColorLdsBytes()) != hipSuccess)
	{
		(void)hipGetLastError();
		w.clusterLdsBytes = ~0u;
		return false;
	}
	w.clusterLdsBytes = (u32)dyn; w.clusterBlocks = (u32)std::max(1, cus);
	if (w.clusterBlocksLimit) w.clusterBlocks = std::min(w.clusterBlocks, w.clusterBlocksLimit); // MI_CLUSTER_BLOCKS (tests: a small launch, so that phases wrap around)
	return true;
}

// Everything between "manifolds exist" and "rows can be initialised": order of the bodies, tasks, local colouring, final slot order.
void launch_cluster_build(World& w, u32 numPairs)
{
	if (!numPairs) return;
	u32 nb = w.nb;
	size_t nb1 = (size_t)nb + 1;
	{ const u32 P = CL_MAX_PARTS;
	w.clKeys.ensure((size_t)P * nb, w.stream); w.clKeysSorted.ensure((size_t)P * nb, w.stream); w.clVals.ensure((size_t)P * nb, w.stream); w.clSorted.ensure((size_t)P * nb, w.stream);
	w.clRank.ensure((size_t)P * nb1, w.stream); w.clSharedSlot.ensure((size_t)CL_MAX_PHASES * nb1, w.stream); w.clWsum.ensure(P * nb1, w.stream); w.clCum.ensure(nb1, w.stream); w.clPhaseMask.ensure(nb1, w.stream); }
	w.clTaskKey.ensure(w.pairCap, w.stream); w.clTaskPos.ensure(w.pairCap, w.stream); w.clPre.ensure(w.pairCap, w.stream); w.clLocal.ensure(w.pairCap, w.stream); w.clEntry.ensure(4 * w.pairCap, w.stream);
	const u32 totalKeys = CL_MAX_PHASES * CL_MAX_TASKS;
	w.clTaskCount.ensure(totalKeys * CL_SUBCOUNTERS + 6u * CL_REMAIN_SUBS, w.stream); w.clTaskStart.ensure(totalKeys * CL_SUBCOUNTERS + 1, w.stream);
	w.clTasks.ensure((size_t)totalKeys * sizeof(ClTask), w.stream); w.clBodyList.ensure((size_t)totalKeys * CL_BODY_STRIDE, w.stream);
	if (w.lastError) return;

	dim3 bgrid((nb + 255) / 256), block(256), mgrid((numPairs + 255) / 256);
	// active manifolds (k_active_list of the colouring: also counts contacts); no warm colours, no global colour masks
	launch_active_list(w, numPairs);
	// Body order along the phases' curves.  Any order is correct, this one makes the clusters compact; bodies move a fraction of
	// their size per step, so the order is refreshed every few steps only (four radix sorts of all bodies), at once when bodies were
	// added and after a snapshot was taken or restored (so that a restored world and its original keep making the same choices).
	const u32 P = CL_MAX_PARTS;
	const u32 parts = w.useComponents ? std::min(w.clusterParts, CL_MAX_PARTS - 1u) : w.clusterParts; // curve phases (the component phase comes on top)
	// the curves in use and the one behind them (phase p attributes what it leaves over along curve p + 1) are sorted; all of them the first
	// time and when the body count changed, so that every rank array holds valid positions (each sort is ~80 us at 100 k bodies)
	const u32 sortParts = (w.clusterSortBodies != nb) ? P : std::min(P, parts + 1u);
	bool refresh = false; // this step re-sorts the bodies and re-cuts the chunks; the steps in between reuse the stored chunks
	if (w.clusterSortDue || w.clusterSortAge >= w.clusterSortInterval || w.clusterSortBodies != nb || w.clusterSortedParts < sortParts)
	{
		refresh = true;
		ClShifts sh; u32 maxShift = 0;
		for (u32 p = 0; p < CL_MAX_PARTS; ++p) for (u32 k = 0; k < 3; ++k) { sh.s[p][k] = w.clusterShift[p][k]; maxShift = std::max(maxShift, sh.s[p][k]); }
		hipLaunchKernelGGL(k_cl_bbox, dim3(std::min<u32>(bgrid.x, 64u)), block, 0, w.stream, nb, w.cog.p, w.simMask.p, w.dCounters.p);
		hipLaunchKernelGGL(k_cl_keys, bgrid, block, 0, w.stream, nb, P, sh, maxShift, w.cog.p, w.simMask.p, w.dCounters.p, w.clKeys.p, w.clVals.p);
		for (u32 p = 0; p < sortParts; ++p)
			prim_sort_pairs_u32(w, w.clKeys.p + (size_t)p * nb, w.clKeysSorted.p + (size_t)p * nb, w.clVals.p + (size_t)p * nb, w.clSorted.p + (size_t)p * nb, nb, 30);
		hipLaunchKernelGGL(k_cl_ranks, bgrid, block, 0, w.stream, nb, sortParts, w.clSorted.p, w.clRank.p);
		w.clusterSortDue = false; w.clusterSortAge = 0; w.clusterSortBodies = nb; w.clusterSortedParts = sortParts; // (curves sorted by THIS refresh: one more phase in use than that forces the next refresh at once, so the order never depends on an older sort)
	}
	w.clusterSortAge++;
	// tasks
	u32 clearItems = std::max<u32>((u32)(CL_MAX_PARTS * nb1), totalKeys * CL_SUBCOUNTERS + 6u * CL_REMAIN_SUBS);
	const bool withJoints = cluster_solves_joints(w);
	const u32* rep = withJoints ? w.clRep.p : nullptr;
	const u32 nj = withJoints ? w.clNumJoints : 0u;
	w.clJointCount.ensure(CL_MAX_TASKS, w.stream); w.clJointStart.ensure(CL_MAX_TASKS + 1, w.stream);
	w.clJointTask.ensure(std::max(nj, 1u), w.stream); w.clJointPos.ensure(std::max(nj, 1u), w.stream); w.clJointList.ensure(std::max(nj, 1u), w.stream); w.clTaskJoints.ensure(std::max(nj, 1u), w.stream);
	w.clJointClassStart.ensure((size_t)CL_MAX_TASKS * (CL_MAX_JOINT_CLASSES + 2u), w.stream);
	if (w.lastError) return;
	w.clCompLabel.ensure(nb1, w.stream); w.clLeftList.ensure(w.pairCap, w.stream);
	if (w.lastError) return;
	// A world whose curve phases left nothing over in the last step (ragdolls standing apart: every island interior to its task) skips
	// the component phase's six launches; what the curves do leave over in this step then goes to the rest task, as without the
	// component phase, and the next step runs the components again (World::countPreviousStep).
	u32* leftList = (w.useComponents && !w.compIdle) ? w.clLeftList.p : nullptr; const u32 leftCap = (u32)w.pairCap;
	const u32 firstFlags = withJoints ? CL_WEIGHT_ISLANDS : 0u; // (goes with the first phase's weight)
	const u32 maxTasks = std::min<u32>(CL_MAX_TASKS / CL_TASKS_PER_PHASE, w.clusterBlocks) - std::min<u32>(8u, w.clusterBlocks / 8u); // per phase, with a margin for the chunks' rounding
	w.clChunk.ensure((size_t)CL_MAX_PARTS * nb1, w.stream);
	if (w.lastError) return;
	if (!w.useChunkCache || w.clChunkParts < parts || w.clChunkJointVersion != w.jointVersion || w.clChunkWithJoints != withJoints) refresh = true;
	// The joints' tasks follow their islands' chunks, which change at a refresh only: in between, the task lists of the joints (task,
	// position, counts, the scattered list) are kept as the refresh step built them — two launches less per step for a ragdoll world.
	const bool keepJointLists = !refresh && nj != 0u && w.clJointListsValid;
	hipLaunchKernelGGL(k_cl_clear, dim3((clearItems + 255) / 256), block, 0, w.stream, (u32)nb1, w.clWsum.p, w.clPhaseMask.p, w.clTaskCount.p, w.clJointCount.p, w.dCounters.p, w.clCompLabel.p,
		nj ? w.clJointBodyMask.p : (const u32*)nullptr, keepJointLists ? 1u : 0u);
	if (refresh)
	{
		// (with the cache on, the chunks are cut 4 % short: the pile may grow until the next refresh)
		const u32 weight0 = w.useChunkCache ? w.clusterTaskWeight - (u32)((u64)w.clusterTaskWeight * w.chunkHeadroomPercent / 100u) : w.clusterTaskWeight, weightLater = (w.useChunkCache && w.chunkCachedPhases > 1u) ? w.clusterTaskWeightLater - (u32)((u64)w.clusterTaskWeightLater * w.chunkHeadroomPercent / 100u) : w.clusterTaskWeightLater;
		hipLaunchKernelGGL(k_cl_weights0, mgrid, block, 0, w.stream, w.dCounters.p, nb, w.actIds.p, w.clRank.p, rep, w.clWsum.p, w.clTaskKey.p);
		if (nj) hipLaunchKernelGGL(k_cl_joint_weights, dim3((nj + 255) / 256), block, 0, w.stream, nj, w.clJointTable.p, w.clRank.p, rep, w.clWsum.p);
		for (u32 p = 0; p < parts; ++p)
		{
			u32* wsum = w.clWsum.p + (size_t)p * nb1; u32* wsumNext = w.clWsum.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1;
			prim_exclusive_scan_u32(w, wsum, w.clCum.p, nb + 1);
			hipLaunchKernelGGL(k_cl_assign, mgrid, block, 0, w.stream, w.dCounters.p, nb, p, parts, p ? weightLater : (weight0 | firstFlags), maxTasks, w.actIds.p, w.clRank.p + (size_t)p * nb1, w.clCum.p,
				w.clRank.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1, wsumNext, w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p, w.dCounters.p + CTR_CL_STATUS, p == 0 ? rep : nullptr, leftList, leftCap);
			if (p == 0 && nj) // (cum still holds phase 0's scan)
				hipLaunchKernelGGL(k_cl_joint_assign, dim3((nj + 255) / 256), block, 0, w.stream, nj, nb, weight0 | firstFlags, maxTasks, w.clJointTable.p, w.clRank.p, rep, w.clCum.p, w.clJointTask.p, w.clJointPos.p, w.clJointCount.p, w.clPhaseMask.p, w.dCounters.p + CTR_CL_STATUS);
			if (w.useChunkCache)
				hipLaunchKernelGGL(k_cl_store_chunks, bgrid, block, 0, w.stream, nb, p ? weightLater : (weight0 | firstFlags), maxTasks, w.clRank.p + (size_t)p * nb1, w.clCum.p, p == 0 ? rep : nullptr, w.clChunk.p + (size_t)p * nb1);
		}
		w.clChunkParts = parts; w.clChunkJointVersion = w.jointVersion; w.clChunkWithJoints = withJoints;
	}
	else
	{
		const u32 cached = std::min(parts, w.chunkCachedPhases);
		hipLaunchKernelGGL(k_cl_assign_cached, mgrid, block, 0, w.stream, w.dCounters.p, nb, parts, cached, withJoints ? 1u : 0u, w.actIds.p, w.clChunk.p,
			w.clRank.p + (size_t)std::min(cached, CL_MAX_PARTS - 1) * nb1, w.clWsum.p + (size_t)std::min(cached, CL_MAX_PARTS - 1) * nb1, w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p, leftList, leftCap);
		if (nj && !keepJointLists) hipLaunchKernelGGL(k_cl_joint_assign_cached, dim3((nj + 255) / 256), block, 0, w.stream, nj, w.clJointTable.p, w.clChunk.p, w.clJointTask.p, w.clJointPos.p, w.clJointCount.p, w.clPhaseMask.p);
		for (u32 p = cached; p < parts; ++p) // the later phases: the per-step pipeline on what is left
		{
			u32* wsum = w.clWsum.p + (size_t)p * nb1; u32* wsumNext = w.clWsum.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1;
			prim_exclusive_scan_u32(w, wsum, w.clCum.p, nb + 1);
			hipLaunchKernelGGL(k_cl_assign, mgrid, block, 0, w.stream, w.dCounters.p, nb, p, parts, p ? w.clusterTaskWeightLater : (w.clusterTaskWeight | firstFlags), maxTasks, w.actIds.p, w.clRank.p + (size_t)p * nb1, w.clCum.p,
				w.clRank.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1, wsumNext, w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p, w.dCounters.p + CTR_CL_STATUS, (const u32*)nullptr, leftList, leftCap);
		}
	}
	// what the curve phases left over: whole connected components to the tasks of one more phase (index = parts)
	if (leftList && parts < CL_MAX_PARTS)
	{
		u32* compWeight = w.clWsum.p + (size_t)(CL_MAX_PARTS - 1) * nb1; // (the last curve's weight sums are not in use: zeroed by k_cl_clear)
		for (u32 r = 0; r < CL_COMP_ROUNDS; ++r) hipLaunchKernelGGL(k_cl_comp_round, dim3(CL_COMP_BLOCKS), block, 0, w.stream, w.dCounters.p, nb, leftCap, w.clLeftList.p, w.actIds.p, w.clCompLabel.p);
		hipLaunchKernelGGL(k_cl_comp_weights, dim3(CL_COMP_BLOCKS), block, 0, w.stream, w.dCounters.p, nb, leftCap, w.clLeftList.p, w.actIds.p, w.clCompLabel.p, compWeight);
		hipLaunchKernelGGL(k_cl_comp_assign, dim3(CL_COMP_BLOCKS), block, 0, w.stream, w.dCounters.p, nb, parts, w.clusterTaskWeightLater, leftCap, w.clLeftList.p, w.actIds.p, w.clCompLabel.p, compWeight,
			w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p);
	}
	hipLaunchKernelGGL(k_cl_offsets, dim3(1), dim3(1024), 0, w.stream, w.dCounters.p, parts, w.clTaskCount.p, w.clTaskStart.p, nj ? w.clJointCount.p : (u32*)nullptr, nj ? w.clJointStart.p : (u32*)nullptr);
	if (nj && !keepJointLists) hipLaunchKernelGGL(k_cl_joint_scatter, dim3((nj + 255) / 256), block, 0, w.stream, nj, w.clJointTask.p, w.clJointPos.p, w.clJointStart.p, w.clJointList.p);
	w.clJointListsValid = nj != 0u && w.useChunkCache;
	hipLaunchKernelGGL(k_cl_scatter, mgrid, block, 0, w.stream, w.dCounters.p, w.clTaskKey.p, w.clTaskPos.p, w.clTaskStart.p, w.clPre.p);
	hipLaunchKernelGGL(k_cl_color, dim3(w.clusterBlocks), dim3(CL_LANES), clColorLdsBytes(), w.stream, w.dCounters.p, nb, w.clTaskStart.p, w.clPre.p, w.actIds.p,
		w.clPhaseMask.p, (ClTask*)w.clTasks.p, w.clBodyList.p, w.mOrder.p, w.mKeySorted.p, w.clLocal.p, w.clEntry.p, w.clSharedSlot.p,
		nj ? w.clJointStart.p : (const u32*)nullptr, w.clJointList.p, w.clJointTable.p, w.clTaskJoints.p, w.clJointClassStart.p, w.flowTrace.p);
}

// Iterations [itBegin, itEnd) of the contact sweep in one launch.
void launch_cluster_solve(World& w, u32 itBegin, u32 itEnd)
{
	if (itBegin >= itEnd) return;
	size_t words = (size_t)(w.nb + 1) * CL_MAX_PHASES * 4; // one 32-byte hand-over record per (phase, body) at most
	if (w.flow.cap < words) { w.flow.ensure(words, w.stream); w.flowEpoch = 0; if (w.lastError) return; } // (a failed allocation leaves the old, smaller buffer: nothing may be launched over it)
	if (w.flowEpoch == 0 || w.flowEpoch >= 0xFFFEu) // first use or the turn counter about to wrap: no stale record may ever match
	{
		MI_CHECK(hipMemsetAsync(w.flow.p, 0, sizeof(u64) * words, w.stream));
		w.flowEpoch = 0;
	}
	w.flowEpoch++;
	if (w.flowTestAbortStep == w.stats.numInternalSteps) // tests: pretend a lane timed out; everybody drains without solving
	{
		u32 one = 16u;
		MI_CHECK(hipMemcpyAsync(w.dCounters.p + CTR_FLOW_STATUS, &one, sizeof(u32), hipMemcpyHostToDevice, w.stream));
		MI_CHECK(hipStreamSynchronize(w.stream));
	}
	const size_t scratchContacts = std::min<size_t>(2 * w.pairCap, 512u * 1024u); // rows that fit neither the registers nor LDS (224 B each)
	w.clRowScratch.ensure(scratchContacts * CLQ_ROW_FLOAT4S, w.stream);
	if (w.lastError) return;
	if (itBegin) MI_CHECK(hipMemsetAsync(w.dCounters.p + CTR_CL_SCRATCH, 0, sizeof(u32), w.stream)); // (the step's first launch finds it cleared by k_cl_clear)
	ClArgs A;
	A.rowScratch = w.clRowScratch.p; A.scratchContacts = (u32)scratchContacts;
	A.counters = w.dCounters.p; A.tasks = (const ClTask*)w.clTasks.p; A.bodyList = w.clBodyList.p; A.phaseMask = w.clPhaseMask.p; A.sharedSlot = w.clSharedSlot.p;
	A.mKeySorted = w.mKeySorted.p; A.mLocal = w.clLocal.p; A.cEntry = w.clEntry.p;
	A.rowPlanes = w.rowPlanes.p; A.rowShared = w.rowShared.p; A.rowLambda = w.rowLambda.p; A.vel = w.vel.p; A.flow = w.flow.p; A.trace = w.flowTrace.p; A.predictDiv = w.clusterPredictDiv; A.pollSleep = w.clusterPollSleep;
	A.rowCap = w.rowCap; A.nb = w.nb; A.flowBytes = (u32)(words * sizeof(u64)); A.epoch = w.flowEpoch << 16; A.itBegin = itBegin; A.itEnd = itEnd;
	A.ldsFloat4s = w.clusterLdsBytes / 16u;
	const bool withJoints = cluster_solves_joints(w);
	A.jointClassStart = withJoints ? w.clJointClassStart.p : nullptr; A.taskJoints = w.clTaskJoints.p; A.jointTable = w.clJointTable.p; A.invIw = w.invIw.p; A.numJointClasses = withJoints ? w.clNumJointClasses : 0u;
	for (u32 t = 0; t < MI_JOINT_TYPES; ++t) A.jointUpd[t] = w.joints[t].dUpdate.p;
	if (withJoints) hipLaunchKernelGGL(k_cl_solve<true>, dim3(w.clusterBlocks), dim3(CLS_LANES), w.clusterLdsBytes, w.stream, A);
	else hipLaunchKernelGGL(k_cl_solve<false>, dim3(w.clusterBlocks), dim3(CLS_LANES), w.clusterLdsBytes, w.stream, A);
}
