// Contact sweep around LDS-resident clusters ("K11-cluster"): the production contact solver.
//
// Why.  A Gauss-Seidel sweep over a proper colouring has (colours x iterations) ~ 20 x 30 = 600 dependent phases per step.  Across the
// chip a phase boundary costs a launch (~5 us) or a tagged hand-over through L2 / the fabric (~2 us); inside ONE workgroup it
// costs a barrier over LDS (~0.3 us).  So the world is cut into spatial clusters that one 1024-lane workgroup each solves
// entirely out of LDS for all iterations, and only what a cut crosses goes through memory:
//
//   phase 0..P-1 ("partitions"): the bodies are ordered along a Morton curve (each phase its own, shifted, curve) and the curve is
//       chunked by weight into tasks of <= ~960 manifolds; a manifold whose two bodies fall into the same chunk is INTERIOR to
//       that task.  Phase p only looks at what phases < p left over, so its chunks cover ever larger volumes and swallow the
//       earlier phases' cut surfaces.
//   phase P ("rest"): whatever is cut in every partition (a few hundred manifolds at 200k) forms one last task.
//
// Tasks of one phase share no body, so they run concurrently, one workgroup each; a workgroup runs its (at most one per phase)
// tasks in phase order, iteration after iteration.  Inside a task the manifolds are coloured locally (in LDS, by the task's
// workgroup: k_cl_color) and swept colour by colour with a workgroup barrier in between; body velocities live in LDS for the whole
// launch, the first contact row of the first task's manifolds in registers, all other rows in LDS (global memory only when the
// LDS budget is exceeded).  A body touched in more than one phase is handed from task to task through the tagged 2 x 16-byte
// records of the old dataflow sweep (sc1 store / sc1 poll, MI355X_MICROARCH.md "tagged granules"): with d = number of phases that
// touch the body, the task of phase p is its r-th user, r = popcount(phaseMask & ((1 << p) - 1)), waits for turn
// epoch + it * d + r and publishes + 1.  Every wait points to a strictly earlier (iteration, phase): no cycles.
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
#define CLQ_SETS 10u                      // ... each keeping this many contact rows in registers (19 VGPRs per row and lane; 256 VGPRs per lane at 2 waves per SIMD)
#define CLQ_REG_CONTACTS (CLQ_QUADS * CLQ_SETS) // contacts of a workgroup's first task that live in registers
#define CL_WEIGHT_REG_LIMIT (64u * (CLQ_REG_CONTACTS - 30u))  // chunk weight up to which a task's contacts (almost always) fit the register sets
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
MI_DEV u32 clEffectiveWeight(u32 taskWeight, u32 totalWeight, u32 maxTasks)
{
	u32 need = totalWeight / maxTasks + 1u;                                   // one task per workgroup ...
	if (need <= taskWeight) return taskWeight;
	if (need <= CL_WEIGHT_REG_LIMIT) return need;                             // ... as long as such a task still fits the lanes' registers,
	u32 need2 = totalWeight / (CL_TASKS_PER_PHASE * maxTasks) + 1u;             // then up to CL_TASKS_PER_PHASE per workgroup (the later ones run from LDS)
	return need2 > CL_WEIGHT_REG_LIMIT ? need2 : CL_WEIGHT_REG_LIMIT;
}
MI_DEV u32 clBid(u32 slot, u32 count, u32 round, u32 i) { return (((4u - count) & 3u) << 22) | ((clHash(slot * 2654435761u + round) & 0x3FFu) << 12) | (i & 0xFFFu); } // 24 bits
#define CL_REMAIN_SUBS 64u // the 'still unassigned after phase p' count is kept in 64 partial counters: ~2000 workgroups adding to ONE word queue up behind each other
#define CL_SUBCOUNTERS 8u // a task's append cursor is split in 8 (by workgroup) so that ~650 returning atomics do not queue on one address

// Everything the assignment accumulates into, cleared in one launch.
__global__ void __launch_bounds__(256) k_cl_clear(u32 nb1, u32* __restrict__ wsum, u32* __restrict__ phaseMask, u32* __restrict__ taskCount, u32* __restrict__ jointCount, u32* __restrict__ counters)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < CL_MAX_TASKS) jointCount[i] = 0;
	if (i < CL_MAX_PARTS * nb1) wsum[i] = 0;
	if (i < nb1) phaseMask[i] = 0;
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
	u32* __restrict__ taskKey, u32* __restrict__ taskPos, u32* __restrict__ taskCount, u32* __restrict__ phaseMask, u32* __restrict__ status, const u32* __restrict__ rep)
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
			else if (phase + 1u == numParts) key = CL_MAX_PARTS * CL_MAX_TASKS; // the rest task
		}
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
	else
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
	const u32* __restrict__ rankNext, u32* __restrict__ wsumNext, u32* __restrict__ taskKey, u32* __restrict__ taskPos, u32* __restrict__ taskCount, u32* __restrict__ phaseMask)
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
			for (u32 p = 0; p < numCached; ++p)
			{
				const u32* c = chunk + (size_t)p * (nb + 1u);
				u32 ta = da ? c[ids.x] : 0u, tb = db ? c[ids.y] : 0u;
				if (!da) ta = tb;
				if (!db) tb = ta;
				if (ta == tb) { key = p * CL_MAX_TASKS + ta; phase = p; break; }
			}
	}
	// manifolds still unassigned when phase q + 1 starts (statistics; the host adapts the number of phases from them)
	const bool goesOn = live && phase == numParts && numCached < numParts && !dumpAll; // left by the cached phases, with a pipeline phase to go to
	for (u32 q = 0; q < numCached; ++q)
	{
		u32 numLeft = (u32)__syncthreads_count(live && phase > q);
		if (threadIdx.x == 0 && numLeft) atomicAdd(&remainSub[(q + 1u) * CL_REMAIN_SUBS + (blockIdx.x & (CL_REMAIN_SUBS - 1u))], numLeft);
	}
	if (!live) return;
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
	if (da && !(__hip_atomic_load(&phaseMask[ids.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << ph))) atomicOr(&phaseMask[ids.x], 1u << ph);
	if (db && !(__hip_atomic_load(&phaseMask[ids.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << ph))) atomicOr(&phaseMask[ids.y], 1u << ph);
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

// One workgroup: exclusive scan of the per-task counts -> first slot of every task; tasks per phase; end of schedule.
__global__ void __launch_bounds__(1024) k_cl_offsets(u32* __restrict__ counters, u32 numParts, const u32* __restrict__ taskCount, u32* __restrict__ taskStart, const u32* __restrict__ jointCount, u32* __restrict__ jointStart)
{
	__shared__ u32 part[1024];
	__shared__ u32 lastTask[CL_MAX_PHASES];
	const u32 total = CL_MAX_PHASES * CL_MAX_TASKS * CL_SUBCOUNTERS, per = (total + 1023u) / 1024u;
	u32 t = threadIdx.x;
	if (t < CL_MAX_PHASES) lastTask[t] = 0;
	u32 sum = 0;
	for (u32 k = 0; k < per; ++k) if (t * per + k < total) sum += taskCount[t * per + k];
	part[t] = sum;
	__syncthreads();
	for (u32 o = 1; o < 1024u; o <<= 1) { u32 v = (t >= o) ? part[t - o] : 0u; __syncthreads(); part[t] += v; __syncthreads(); }
	u32 run = part[t] - sum;
	for (u32 k = 0; k < per; ++k)
	{
		u32 e = t * per + k;
		if (e >= total) break;
		u32 c = taskCount[e], key = e / CL_SUBCOUNTERS;
		taskStart[e] = run; run += c;
		if (c) atomicMax(&lastTask[key / CL_MAX_TASKS], (key % CL_MAX_TASKS) + 1u);
	}
	if (t == 1023u) taskStart[total] = run;
	__syncthreads();
	// joints per phase-0 task (CL_MAX_TASKS <= 1024 entries: one per lane); a task may hold joints and no manifold
	{
		u32 jc = (jointCount && t < CL_MAX_TASKS) ? jointCount[t] : 0u;
		part[t] = jc;
		__syncthreads();
		for (u32 o = 1; o < 1024u; o <<= 1) { u32 v = (t >= o) ? part[t - o] : 0u; __syncthreads(); part[t] += v; __syncthreads(); }
		if (jointStart && t < CL_MAX_TASKS) { jointStart[t] = part[t] - jc; if (t == CL_MAX_TASKS - 1u) jointStart[CL_MAX_TASKS] = part[t]; }
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
	const u32* __restrict__ phaseMask, ClTask* __restrict__ tasks, u32* __restrict__ bodyList, u32* __restrict__ bodyUsers, u32* __restrict__ mOrder, u32* __restrict__ mKeySorted, u32* __restrict__ mLocal, u32* __restrict__ mExtra, u32* __restrict__ mRank, u32* __restrict__ sharedSlot,
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
		for (u32 h = tid; h < 264u + CL_MAX_JOINT_CLASSES + 1u; h += CL_LANES) hist[h] = 0;
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
		// 4. order by key: histogram, scan by one wave, cursors
		for (u32 i = tid; i < n; i += CL_LANES) atomicAdd(&hist[mKey[i]], 1u);
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
			// number of colours in use = highest non-empty colour below the serial class + 1
			u32 top = 0;
			for (u32 k = 0; k < 5u; ++k) if (v[k] && (base + k) / 4u < CL_SERIAL_COLOR) top = (base + k) / 4u + 1u;
			for (int o = 32; o > 0; o >>= 1) top = max(top, (u32)__shfl_xor(top, o));
			if (tid == 0) sMaxColor = top;
		}
		__syncthreads();
		const u32 numColors = sMaxColor;
		if (tid <= CL_SERIAL_COLOR) T->colorStart[tid] = hist[tid * 4u];
		if (tid == 0)
		{
			T->first = first; T->count = n; T->numBodies = numBodies; T->numShared = numShared; T->numColors = numColors; T->serialStart = hist[CL_SERIAL_COLOR * 4u];
			T->colorStart[CL_SERIAL_COLOR + 1u] = n;
			atomicMax(&counters[CTR_NUM_COLORS], numColors + (hist[CL_SERIAL_COLOR * 4u] < n ? 1u : 0u));
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
			else { p = hist[k]; for (u32 j = 0; j < i; ++j) p += (mKey[j] == k) ? 1u : 0u; } // (rare: a body with more than 64 users in one task)
			mPos[i] = p; mCnt[p] = (4u - (k & 3u));
		}
		__syncthreads();
		CL_STAMP(5)
		// 5. extra-row offsets: exclusive scan of (count - 1) over the final positions (2 per lane)
		{
			u32 p0 = 2u * tid, e0 = (p0 < n) ? mCnt[p0] - 1u : 0u, e1 = (p0 + 1u < n) ? mCnt[p0 + 1u] - 1u : 0u;
			u32 s = e0 + e1, incl = s;
			for (int o = 1; o < 64; o <<= 1) { u32 up = __shfl_up(incl, o); if ((int)(tid & 63u) >= o) incl += up; }
			if ((tid & 63u) == 63u) sScan[tid >> 6] = incl;
			__syncthreads();
			u32 waveBase = 0;
			for (u32 wv = 0; wv < (tid >> 6); ++wv) waveBase += sScan[wv];
			u32 excl = waveBase + incl - s;
			__syncthreads();
			if (p0 < n) mCnt[p0] = excl;
			if (p0 + 1u < n) mCnt[p0 + 1u] = excl + e0;
			if (tid == CL_LANES - 1u) T->numRows = n + waveBase + incl;
		}
		__syncthreads();
		CL_STAMP(6)
		// 6. rank of every manifold among the users of each of its bodies, in position order (the hand-over turn numbers).
		// Coloured manifolds: a colour occurs once per body, so the rank is the number of lower colours in the body's mask.  The serial
		// tail (no colour left below 64: bodies with more than 64 users in this task) is walked by one lane in position order.
		u32* inv = hKey;   // serial manifolds by position (the hash is no longer needed)
		u32* rankL = hVal; // per manifold: rA | dA << 8 | rB << 16 | dB << 24 (d filled in below)
		const u32 serialStart = hist[CL_SERIAL_COLOR * 4u - 1u]; // cursors have run: cursor of the last coloured class = first serial position
		for (u32 l = tid; l < numBodies; l += CL_LANES) claim[l] = 0; // serial users per body
		for (u32 i = tid; i < n; i += CL_LANES) if ((mKey[i] >> 2) >= CL_SERIAL_COLOR) inv[mPos[i] - serialStart] = i;
		__syncthreads();
		if (tid == 0)
			for (u32 p = serialStart; p < n; ++p)
			{
				u32 i = inv[p - serialStart], ab = mAB[i], la = ab & 0xFFFFu, lb = ab >> 16, r = 0;
				if (la != CL_LOCAL_STATIC) { r |= ((u32)__popcll(mask[la]) + claim[la]) & 0xFFu; claim[la]++; }
				if (lb != CL_LOCAL_STATIC) { r |= (((u32)__popcll(mask[lb]) + claim[lb]) & 0xFFu) << 16; claim[lb]++; }
				rankL[i] = r;
			}
		__syncthreads();
		bool tooBusy = false;
		for (u32 l = tid; l < numBodies; l += CL_LANES)
		{
			u32 users = (u32)__popcll(mask[l]) + claim[l];
			if (users > 250u) tooBusy = true; // the turn arithmetic keeps ranks and user counts in 8 bits
			bodyUsers[(size_t)key * CL_BODY_STRIDE + l] = users;
		}
		if (tooBusy) atomicOr(&counters[CTR_CL_STATUS], 2u);
		for (u32 i = tid; i < n; i += CL_LANES)
		{
			u32 p = mPos[i], ab = mAB[i], la = ab & 0xFFFFu, lb = ab >> 16, c = mKey[i] >> 2;
			u32 r = (c >= CL_SERIAL_COLOR) ? rankL[i] : 0u;
			u64 lower = (c >= CL_SERIAL_COLOR) ? 0ull : ((1ull << c) - 1ull);
			if (la != CL_LOCAL_STATIC) { if (c < CL_SERIAL_COLOR) r |= (u32)__popcll(mask[la] & lower); r |= ((((u32)__popcll(mask[la]) + claim[la])) & 0xFFu) << 8; }
			if (lb != CL_LOCAL_STATIC) { if (c < CL_SERIAL_COLOR) r |= (u32)__popcll(mask[lb] & lower) << 16; r |= ((((u32)__popcll(mask[lb]) + claim[lb])) & 0xFFu) << 24; }
			mOrder[first + p] = pre[first + i];
			mKeySorted[first + p] = mKey[i];
			mLocal[first + p] = ab;
			mExtra[first + p] = mCnt[p];
			mRank[first + p] = r;
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
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct ClLocal // a task of this workgroup, in LDS
{
	u32 first, count, numBodies, numShared, numColors, serialStart, phase, key, sharedBase, numJoints;
	u32 bodyOff;   // float4 index of the task's bodies (2 float4 each)
	u32 infoOff;   // u32 index (in float4 units * 4) of per-body {global id, turn info}
	u32 metaOff;   // float4 index of the per-manifold meta (not for the register task): {la|lb<<16, key|extra<<10, -, -}, {n.xyz, friction}
	u32 rowOff;    // float4 index of the row planes
	u32 rowCap;    // rows of this task that live in LDS
	u32 inRegs;    // the first contact row of every manifold sits in its lane's registers
	u32 colorStart[66];
};

// Row r of a task's LDS row region: 8 float4 planes, plane-major (consecutive rows -> consecutive addresses); plane 7 = {mT, bias, lambdaN, lambdaT}.
MI_DEV void clLoadRowLds(ContactRow& r, const float4* lds, u32 rowOff, u32 rowCap, u32 row)
{
	const float4* P = lds + rowOff + row;
	r.p0 = P[0]; r.p1 = P[rowCap]; r.p2 = P[2 * rowCap]; r.p3 = P[3 * rowCap]; r.p4 = P[4 * rowCap]; r.p5 = P[5 * rowCap]; r.p6 = P[6 * rowCap];
	float4 q = P[7 * rowCap]; r.p7 = make_float2(q.x, q.y); r.lam = make_float2(q.z, q.w);
}
MI_DEV void clStoreRowLds(float4* lds, u32 rowOff, u32 rowCap, u32 row, const ContactRow& r)
{
	float4* P = lds + rowOff + row;
	P[0] = r.p0; P[rowCap] = r.p1; P[2 * rowCap] = r.p2; P[3 * rowCap] = r.p3; P[4 * rowCap] = r.p4; P[5 * rowCap] = r.p5; P[6 * rowCap] = r.p6;
	P[7 * rowCap] = make_float4(r.p7.x, r.p7.y, r.lam.x, r.lam.y);
}
MI_DEV void clStoreLambdaLds(float4* lds, u32 rowOff, u32 rowCap, u32 row, float2 lam) { ((float2*)(lds + rowOff + 7 * rowCap + row))[1] = lam; }
MI_DEV float2 clLoadLambdaLds(const float4* lds, u32 rowOff, u32 rowCap, u32 row) { return ((const float2*)(lds + rowOff + 7 * rowCap + row))[1]; }

struct ClArgs
{
	u32* counters; const ClTask* tasks; const u32* bodyList; const u32* bodyUsers; const u32* phaseMask; const u32* sharedSlot;
	const u32* mKeySorted; const u32* mLocal; const u32* mExtra; const u32* mRank;
	const float4* rowPlanes; const float4* rowShared; float2* rowLambda;
	u32 predictDiv, pollSleep; // pacing of the hand-over polls (MI_CLUSTER_PREDICT_DIV / MI_CLUSTER_POLL_SLEEP)
	float4* vel; u64* flow; u64* trace; // trace: developer timeline (mi_debug_flow_trace), normally null
	size_t rowCap; u32 nb, flowBytes, epoch, itBegin, itEnd, ldsFloat4s;
	// joints run by the sweep (null / 0 when the world has none or they keep their own launches): per phase-0 task the class offsets
	// [CL_MAX_JOINT_CLASSES + 2] (last two: joint count, first entry), the class-sorted entries {table index, la | lb << 16}, the table
	// {type | class << 8, index in the type's arrays, body a, body b}, the per-type update records, world inverse inertia
	const u32* jointClassStart; const uint2* taskJoints; const uint4* jointTable; float* jointUpd[MI_JOINT_TYPES]; const float4* invIw; u32 numJointClasses;
};

// One manifold: both bodies from LDS, its rows (registers / LDS / global memory), both bodies back.
template <bool REG> MI_DEV void clSolveManifold(float4* lds, const ClLocal& L, const ClArgs& A, u32 pos, u32 ab, u32 keyExtra, float4 sh, ContactRow& r0)
{
	u32 la = ab & 0xFFFFu, lb = ab >> 16;
	u32 count = 4u - (keyExtra & 3u), extra = keyExtra >> 10;
	float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, b0 = a0, b1 = a0;
	if (la != CL_LOCAL_STATIC) { a0 = lds[L.bodyOff + 2 * la]; a1 = lds[L.bodyOff + 2 * la + 1]; }
	if (lb != CL_LOCAL_STATIC) { b0 = lds[L.bodyOff + 2 * lb]; b1 = lds[L.bodyOff + 2 * lb + 1]; }
	V3 vA = v3f4(a0), wA = v3f4(a1), vB = v3f4(b0), wB = v3f4(b1);
	float invMassA = a0.w, invMassB = b0.w;
	V3 n = v3(sh.x, sh.y, sh.z);
	float friction = sh.w;
	u32 slot = L.first + pos;
	u32 rowBase = REG ? extra : pos + extra; // LDS row of contact 1 (register task) or contact 0
	for (u32 k = 0; k < count; ++k)
	{
		if (REG && k == 0) { solveRow(r0, n, friction, invMassA, invMassB, vA, wA, vB, wB); continue; }
		u32 row = rowBase + (REG ? k - 1u : k);
		ContactRow cur;
		if (row < L.rowCap)
		{
			clLoadRowLds(cur, lds, L.rowOff, L.rowCap, row);
			solveRow(cur, n, friction, invMassA, invMassB, vA, wA, vB, wB);
			clStoreLambdaLds(lds, L.rowOff, L.rowCap, row, cur.lam);
		}
		else // beyond the LDS budget: streamed from global memory every iteration
		{
			loadRow(cur, k, slot, A.rowCap, A.rowPlanes, A.rowLambda);
			solveRow(cur, n, friction, invMassA, invMassB, vA, wA, vB, wB);
			A.rowLambda[(size_t)k * A.rowCap + slot] = cur.lam;
		}
	}
	if (la != CL_LOCAL_STATIC) { lds[L.bodyOff + 2 * la] = make_float4(vA.x, vA.y, vA.z, invMassA); lds[L.bodyOff + 2 * la + 1] = make_float4(wA.x, wA.y, wA.z, 0.f); }
	if (lb != CL_LOCAL_STATIC) { lds[L.bodyOff + 2 * lb] = make_float4(vB.x, vB.y, vB.z, invMassB); lds[L.bodyOff + 2 * lb + 1] = make_float4(wB.x, wB.y, wB.z, 0.f); }
}

// The same for a manifold of the register task, with everything address-like resolved beforehand: rd / wr are the float4 indices of
// the two bodies (a static body reads the all-zero record and writes into a sink, so there is no branch around the LDS traffic),
// rowOff / rowCap / slot come in registers instead of being re-read from the task record in LDS behind every barrier
// (each such read is a dependent LDS round trip of ~100 cycles on the sweep's critical path).
MI_DEV void clSolveReg(float4* lds, const ClArgs& A, u32 rdA, u32 wrA, u32 rdB, u32 wrB, u32 keyExtra, float4 sh, ContactRow& r0, u32 rowOff, u32 rowCap, u32 slot)
{
	float4 a0 = lds[rdA], a1 = lds[rdA + 1], b0 = lds[rdB], b1 = lds[rdB + 1];
	V3 vA = v3f4(a0), wA = v3f4(a1), vB = v3f4(b0), wB = v3f4(b1);
	float invMassA = a0.w, invMassB = b0.w;
	V3 n = v3(sh.x, sh.y, sh.z);
	float friction = sh.w;
	solveRow(r0, n, friction, invMassA, invMassB, vA, wA, vB, wB);
	u32 count = 4u - (keyExtra & 3u), extra = keyExtra >> 10;
	for (u32 k = 1; k < count; ++k)
	{
		u32 row = extra + k - 1u;
		ContactRow cur;
		if (row < rowCap)
		{
			clLoadRowLds(cur, lds, rowOff, rowCap, row);
			solveRow(cur, n, friction, invMassA, invMassB, vA, wA, vB, wB);
			clStoreLambdaLds(lds, rowOff, rowCap, row, cur.lam);
		}
		else // beyond the LDS budget: streamed from global memory every iteration
		{
			loadRow(cur, k, slot, A.rowCap, A.rowPlanes, A.rowLambda);
			solveRow(cur, n, friction, invMassA, invMassB, vA, wA, vB, wB);
			A.rowLambda[(size_t)k * A.rowCap + slot] = cur.lam;
		}
	}
	lds[wrA] = make_float4(vA.x, vA.y, vA.z, invMassA); lds[wrA + 1] = make_float4(wA.x, wA.y, wA.z, 0.f);
	lds[wrB] = make_float4(vB.x, vB.y, vB.z, invMassB); lds[wrB + 1] = make_float4(wB.x, wB.y, wB.z, 0.f);
}

// A manifold of the workgroup's SECOND task: its rows all live in LDS (or, beyond the budget, in global memory), but what a lane needs to
// find them — body addresses, key, shared normal — sits in its registers like the first task's, so a colour step is ONE LDS round trip
// (bodies and first row together) instead of the generic path's chain task record -> meta -> bodies / rows.
MI_DEV void clSolveLds(float4* lds, const ClArgs& A, u32 rdA, u32 wrA, u32 rdB, u32 wrB, u32 keyExtra, float4 sh, u32 rowBase, u32 rowOff, u32 rowCap, u32 slot)
{
	float4 a0 = lds[rdA], a1 = lds[rdA + 1], b0 = lds[rdB], b1 = lds[rdB + 1];
	V3 vA = v3f4(a0), wA = v3f4(a1), vB = v3f4(b0), wB = v3f4(b1);
	float invMassA = a0.w, invMassB = b0.w;
	V3 n = v3(sh.x, sh.y, sh.z);
	float friction = sh.w;
	u32 count = 4u - (keyExtra & 3u);
	for (u32 k = 0; k < count; ++k)
	{
		u32 row = rowBase + k;
		ContactRow cur;
		if (row < rowCap)
		{
			clLoadRowLds(cur, lds, rowOff, rowCap, row);
			solveRow(cur, n, friction, invMassA, invMassB, vA, wA, vB, wB);
			clStoreLambdaLds(lds, rowOff, rowCap, row, cur.lam);
		}
		else // beyond the LDS budget: streamed from global memory every iteration
		{
			loadRow(cur, k, slot, A.rowCap, A.rowPlanes, A.rowLambda);
			solveRow(cur, n, friction, invMassA, invMassB, vA, wA, vB, wB);
			A.rowLambda[(size_t)k * A.rowCap + slot] = cur.lam;
		}
	}
	lds[wrA] = make_float4(vA.x, vA.y, vA.z, invMassA); lds[wrA + 1] = make_float4(wA.x, wA.y, wA.z, 0.f);
	lds[wrB] = make_float4(vB.x, vB.y, vB.z, invMassB); lds[wrB + 1] = make_float4(wB.x, wB.y, wB.z, 0.f);
}

// JOINTS: the instantiation for worlds whose joints run inside the sweep (its extra registers and code stay out of the other one).
template <bool JOINTS> __global__ void __launch_bounds__(CLS_LANES) k_cl_solve(ClArgs A)
{
	extern __shared__ float4 lds[];
	__shared__ ClLocal sTask[CL_MAX_LOCAL_TASKS];
	__shared__ u32 sNumTasks, sAbort;
	const u32 tid = threadIdx.x, G = gridDim.x;
	u32* status = A.counters + CTR_FLOW_STATUS;
	__amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A.flow, 0, A.flowBytes, 0x00020000);

	// ---- which tasks are mine, and where they live in LDS ----
	if (tid == 0)
	{
		if (A.trace) A.trace[(size_t)blockIdx.x * 16u * 32u + 15 * 32] = wall_clock64();
		u32 nT = 0, off = 0, used = 0; // used: float4s of LDS handed out
		bool bad = A.counters[CTR_CL_STATUS] != 0u;
		for (u32 p = 0; p < CL_MAX_PHASES; ++p)
		{
			u32 tasksInPhase = A.counters[CTR_CL_NUM_TASKS + p];
			if (tasksInPhase > CL_TASKS_PER_PHASE * G) bad = true;
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
				if (nT == CL_MAX_LOCAL_TASKS) { bad = true; break; }
				ClLocal& L = sTask[nT];
				L.first = T->first; L.count = T->count; L.numBodies = T->numBodies; L.numShared = T->numShared; L.numColors = T->numColors; L.serialStart = T->serialStart;
				L.phase = p; L.key = key; L.sharedBase = T->sharedBase; L.numJoints = tj;
				if (tj > CLS_LANES || (tj && nT)) bad = true; // one lane per joint; joints run with the workgroup's first task only
				for (u32 c = 0; c <= CL_SERIAL_COLOR + 1u; ++c) L.colorStart[c] = T->colorStart[c];
				L.bodyOff = used; used += 2u * L.numBodies;
				L.infoOff = used * 4u; used += (3u * L.numBodies + 3u) / 4u;
				L.inRegs = (nT == 0 && L.count <= CLS_LANES * CLS_R) ? 1u : 0u;
				L.metaOff = used; if (!L.inRegs) used += 2u * L.count;
				L.rowOff = 0; L.rowCap = 0;
				++nT;
			}
		}
		// rows: whatever LDS is left, in task order (7 plane-rows of 16 B per row: 6 float4 + 1 float2 rounded up)
		for (u32 k = 0; k < nT; ++k)
		{
			ClLocal& L = sTask[k];
			u32 want = A.tasks[L.key].numRows - (L.inRegs ? L.count : 0u);
			u32 left = (used + 4u < A.ldsFloat4s) ? A.ldsFloat4s - used - 4u : 0u; // the last four float4 are the static body's all-zero record and the sink its writes go to
			u32 fit = left / 8u;                          // 8 float4 per row
			u32 cap = want < fit ? want : fit;
			L.rowOff = used; L.rowCap = cap; used += 8u * cap;
		}
		if (used + 4u > A.ldsFloat4s) bad = true; // bodies + meta alone exceed LDS: cannot run this launch
		if (bad) atomicOr(status, 64u);
		sNumTasks = nT; sAbort = (bad || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1u : 0u;
	}
	__syncthreads();
	if (sAbort) return; // uniform: the launch cannot run (or somebody has given up already); the host redoes the step
	const u32 numTasks = sNumTasks;
	if (!numTasks) return;

	// ---- prologue: bodies, meta, rows ----
	static_assert(CLS_R == 2u, "two named register sets below");
	u32 regAB0 = 0, regAB1 = 0, regKE0 = 0xFFFFFFFFu, regKE1 = 0xFFFFFFFFu; float4 regSh0 = make_float4(0.f, 0.f, 0.f, 0.f), regSh1 = regSh0; ContactRow regRow0 = {}, regRow1 = {}, noRow = {};
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
		if (L.inRegs)
		{
#define CL_LOAD_REG(R_, AB_, KE_, SH_, ROW_) \
			{ \
				u32 i = tid + (R_) * CLS_LANES; \
				if (i < L.count) \
				{ \
					u32 slot = L.first + i; \
					u32 key = A.mKeySorted[slot], extra = A.mExtra[slot], count = 4u - (key & 3u); \
					AB_ = A.mLocal[slot]; KE_ = key | (extra << 10); SH_ = A.rowShared[slot]; \
					loadRow(ROW_, 0, slot, A.rowCap, A.rowPlanes, A.rowLambda); \
					for (u32 kk = 1; kk < count; ++kk) \
					{ \
						u32 row = extra + kk - 1u; \
						if (row >= L.rowCap) continue; \
						ContactRow cur; \
						loadRow(cur, kk, slot, A.rowCap, A.rowPlanes, A.rowLambda); \
						clStoreRowLds(lds, L.rowOff, L.rowCap, row, cur); \
					} \
				} \
			}
			CL_LOAD_REG(0u, regAB0, regKE0, regSh0, regRow0)
			CL_LOAD_REG(1u, regAB1, regKE1, regSh1, regRow1)
#undef CL_LOAD_REG
		}
		else for (u32 i = tid; i < L.count; i += CLS_LANES)
		{
			u32 slot = L.first + i;
			u32 key = A.mKeySorted[slot], extra = A.mExtra[slot], count = 4u - (key & 3u);
			lds[L.metaOff + 2 * i] = make_float4(__uint_as_float(A.mLocal[slot]), __uint_as_float(key | (extra << 10)), 0.f, 0.f);
			lds[L.metaOff + 2 * i + 1] = A.rowShared[slot];
			for (u32 kk = 0; kk < count; ++kk)
			{
				u32 row = i + extra + kk;
				if (row >= L.rowCap) continue;
				ContactRow cur;
				loadRow(cur, kk, slot, A.rowCap, A.rowPlanes, A.rowLambda);
				clStoreRowLds(lds, L.rowOff, L.rowCap, row, cur);
			}
		}
	}
	// the register task's bodies as LDS addresses (static body: read the zero record, write into the sink)
	const u32 zeroRec = A.ldsFloat4s - 4u, sinkRec = A.ldsFloat4s - 2u;
	if (tid < 2u) lds[zeroRec + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
	const u32 bodyOff0 = sTask[0].bodyOff;
	const u32 rdA0 = (regAB0 & 0xFFFFu) == CL_LOCAL_STATIC ? zeroRec : bodyOff0 + 2u * (regAB0 & 0xFFFFu), wrA0 = (regAB0 & 0xFFFFu) == CL_LOCAL_STATIC ? sinkRec : rdA0;
	const u32 rdB0 = (regAB0 >> 16) == CL_LOCAL_STATIC ? zeroRec : bodyOff0 + 2u * (regAB0 >> 16), wrB0 = (regAB0 >> 16) == CL_LOCAL_STATIC ? sinkRec : rdB0;
	const u32 rdA1 = (regAB1 & 0xFFFFu) == CL_LOCAL_STATIC ? zeroRec : bodyOff0 + 2u * (regAB1 & 0xFFFFu), wrA1 = (regAB1 & 0xFFFFu) == CL_LOCAL_STATIC ? sinkRec : rdA1;
	const u32 rdB1 = (regAB1 >> 16) == CL_LOCAL_STATIC ? zeroRec : bodyOff0 + 2u * (regAB1 >> 16), wrB1 = (regAB1 >> 16) == CL_LOCAL_STATIC ? sinkRec : rdB1;
	const u32 rowOff0 = __builtin_amdgcn_readfirstlane(sTask[0].rowOff), rowCap0 = __builtin_amdgcn_readfirstlane(sTask[0].rowCap), first0 = __builtin_amdgcn_readfirstlane(sTask[0].first);
	// the workgroup's second task, if it has at most one manifold per lane: this lane's manifold of it (position = lane)
	const bool second = !JOINTS && numTasks > 1u && !sTask[1].inRegs && sTask[1].count <= CLS_LANES;
	u32 s2KE = 0xFFFFFFFFu, s2rdA = zeroRec, s2wrA = sinkRec, s2rdB = zeroRec, s2wrB = sinkRec, s2rowBase = 0; float4 s2Sh = make_float4(0.f, 0.f, 0.f, 0.f);
	if (second && tid < sTask[1].count)
	{
		const u32 slot = sTask[1].first + tid, ab = A.mLocal[slot], key = A.mKeySorted[slot], extra = A.mExtra[slot], off1 = sTask[1].bodyOff;
		s2KE = key | (extra << 10); s2Sh = A.rowShared[slot]; s2rowBase = tid + extra;
		if ((ab & 0xFFFFu) != CL_LOCAL_STATIC) { s2rdA = off1 + 2u * (ab & 0xFFFFu); s2wrA = s2rdA; }
		if ((ab >> 16) != CL_LOCAL_STATIC) { s2rdB = off1 + 2u * (ab >> 16); s2wrB = s2rdB; }
	}
	const u32 rowOff1 = second ? __builtin_amdgcn_readfirstlane(sTask[1].rowOff) : 0u, rowCap1 = second ? __builtin_amdgcn_readfirstlane(sTask[1].rowCap) : 0u, first1 = second ? __builtin_amdgcn_readfirstlane(sTask[1].first) : 0u;
	// first of the register task's trailing colours that one wave runs without barriers (see the colour loop)
	u32 tailStart0 = sTask[0].numColors;
	if (sTask[0].inRegs && sTask[0].serialStart > 0u) { const u32 lastBlock = (sTask[0].serialStart - 1u) >> 6; while (tailStart0 > 0u && (sTask[0].colorStart[tailStart0 - 1u] >> 6) == lastBlock) --tailStart0; }
	tailStart0 = __builtin_amdgcn_readfirstlane(tailStart0);
	u32 tailStart1 = second ? sTask[1].numColors : 0u;
	if (second && sTask[1].serialStart > 0u) { const u32 lastBlock = (sTask[1].serialStart - 1u) >> 6; while (tailStart1 > 0u && (sTask[1].colorStart[tailStart1 - 1u] >> 6) == lastBlock) --tailStart1; }
	tailStart1 = __builtin_amdgcn_readfirstlane(tailStart1);
	// this lane's joint (first task only, phase 0): class, update record, the two bodies as LDS addresses and as global ids (inverse inertia)
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
		for (u32 k = 0; k < numTasks && k < 5u; ++k) { trace[15 * 32 + 2 + 4 * k] = sTask[k].count; trace[15 * 32 + 3 + 4 * k] = sTask[k].numColors; trace[15 * 32 + 4 + 4 * k] = sTask[k].numShared; trace[15 * 32 + 5 + 4 * k] = sTask[k].phase | (sTask[k].numBodies << 8) | ((u64)sTask[k].rowCap << 32); }
	}

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
			// polls the tagged first halves of up to four bodies per pass, all loads in flight together, and fetches the second half
			// (stored before the first) once the tag has arrived.
			{
				if (k == 0 && lastWait > 64u && it > A.itBegin + 1u)
				{
					u64 until = lastPublish + (u64)(lastWait - lastWait / A.predictDiv);
					while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
				}
				const u32 rel = it - A.itBegin;
				for (u32 base = 0; base < L.numShared; base += 4u * CLS_LANES)
				{
					u32 gid[4], want[4]; bool pend[4]; bool any = false;
#pragma unroll
					for (u32 q = 0; q < 4; ++q)
					{
						u32 l = base + q * CLS_LANES + tid;
						pend[q] = false; gid[q] = 0; want[q] = 0;
						if (l >= L.numShared) continue;
						u32 ti = info[3 * l + 1], deg = ti & 0xFFu, rank = ti >> 8;
						if (rel == 0u && rank == 0u) continue; // first user of the launch: the prologue's copy of vel is current
						gid[q] = info[3 * l + 2]; want[q] = A.epoch + rel * deg + rank; pend[q] = true; any = true;
					}
					u32 spins = 0;
					while (any)
					{
						u32x4 h0[4], h1[4];
						asm volatile("" ::: "memory");
#pragma unroll
						for (u32 q = 0; q < 4; ++q)
							if (pend[q]) { h0[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gid[q] * 32u, 0, 16); h1[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, gid[q] * 32u + 16u, 0, 16); }
						any = false;
#pragma unroll
						for (u32 q = 0; q < 4; ++q)
						{
							if (!pend[q]) continue;
							if (h0[q].w == want[q] && h1[q].w == want[q]) // each half carries its own tag
							{
								u32 l = base + q * CLS_LANES + tid;
								float invMass = lds[L.bodyOff + 2 * l].w; // constant over the launch
								lds[L.bodyOff + 2 * l] = make_float4(__uint_as_float(h0[q].x), __uint_as_float(h0[q].y), __uint_as_float(h0[q].z), invMass);
								lds[L.bodyOff + 2 * l + 1] = make_float4(__uint_as_float(h1[q].x), __uint_as_float(h1[q].y), __uint_as_float(h1[q].z), 0.f);
								pend[q] = false;
							}
							any = any || pend[q];
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
			// colours
			if (L.inRegs)
			{
				const u32 col0 = (regKE0 & 0x3FFu) >> 2, col1 = (regKE1 & 0x3FFu) >> 2; // 255 = no manifold: matches no colour
				const u32 numColors = __builtin_amdgcn_readfirstlane(L.numColors), serialStart = __builtin_amdgcn_readfirstlane(L.serialStart), taskCount = __builtin_amdgcn_readfirstlane(L.count);
				const bool stamp = trace && tid == 0 && it == A.itBegin + 10u && k == 0;
				if (stamp) trace[5 * 32] = clock64();
				// The last colours hold a handful of manifolds (the busiest body's last users).  Those whose positions all fall into
				// ONE 64-position block belong to one wave (and one register set): that wave runs them back to back, in program
				// order, without the workgroup barrier in between (LDS serves a wave's accesses in order).
				const u32 tailStart = tailStart0;
				for (u32 c = 0; c < tailStart; ++c)
				{
					if (col0 == c) clSolveReg(lds, A, rdA0, wrA0, rdB0, wrB0, regKE0, regSh0, regRow0, rowOff0, rowCap0, first0 + tid);
					if (col1 == c) clSolveReg(lds, A, rdA1, wrA1, rdB1, wrB1, regKE1, regSh1, regRow1, rowOff0, rowCap0, first0 + tid + CLS_LANES);
					__syncthreads();
					if (stamp) { trace[5 * 32 + 1 + c] = clock64(); trace[7 * 32 + c] = L.colorStart[c + 1] - L.colorStart[c]; }
				}
				if (tailStart < numColors)
				{
					if (col0 >= tailStart && col0 < numColors)
						for (u32 c = tailStart; c < numColors; ++c)
						{
							if (col0 == c) clSolveReg(lds, A, rdA0, wrA0, rdB0, wrB0, regKE0, regSh0, regRow0, rowOff0, rowCap0, first0 + tid);
							__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
						}
					if (col1 >= tailStart && col1 < numColors)
						for (u32 c = tailStart; c < numColors; ++c)
						{
							if (col1 == c) clSolveReg(lds, A, rdA1, wrA1, rdB1, wrB1, regKE1, regSh1, regRow1, rowOff0, rowCap0, first0 + tid + CLS_LANES);
							__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
						}
					__syncthreads();
					if (stamp) for (u32 c = tailStart; c < numColors; ++c) { trace[5 * 32 + 1 + c] = clock64(); trace[7 * 32 + c] = L.colorStart[c + 1] - L.colorStart[c]; }
				}
				for (u32 sp = serialStart; sp < taskCount; ++sp) // the serial tail (manifolds that found no colour below 64): one per step
				{
					if (tid == sp) clSolveReg(lds, A, rdA0, wrA0, rdB0, wrB0, regKE0, regSh0, regRow0, rowOff0, rowCap0, first0 + tid);
					if (tid + CLS_LANES == sp) clSolveReg(lds, A, rdA1, wrA1, rdB1, wrB1, regKE1, regSh1, regRow1, rowOff0, rowCap0, first0 + tid + CLS_LANES);
					__syncthreads();
				}
			}
			else if (second && k == 1u)
			{
				const u32 col = (s2KE & 0x3FFu) >> 2; // 255 = no manifold
				const u32 numColors = __builtin_amdgcn_readfirstlane(L.numColors), serialStart = __builtin_amdgcn_readfirstlane(L.serialStart), taskCount = __builtin_amdgcn_readfirstlane(L.count);
				for (u32 c = 0; c < tailStart1; ++c)
				{
					if (col == c) clSolveLds(lds, A, s2rdA, s2wrA, s2rdB, s2wrB, s2KE, s2Sh, s2rowBase, rowOff1, rowCap1, first1 + tid);
					__syncthreads();
				}
				if (tailStart1 < numColors) // the trailing colours inside one wave, as in the register task
				{
					if (col >= tailStart1 && col < numColors)
						for (u32 c = tailStart1; c < numColors; ++c)
						{
							if (col == c) clSolveLds(lds, A, s2rdA, s2wrA, s2rdB, s2wrB, s2KE, s2Sh, s2rowBase, rowOff1, rowCap1, first1 + tid);
							__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
						}
					__syncthreads();
				}
				for (u32 sp = serialStart; sp < taskCount; ++sp) // the serial tail: one manifold per step
				{
					if (tid == sp) clSolveLds(lds, A, s2rdA, s2wrA, s2rdB, s2wrB, s2KE, s2Sh, s2rowBase, rowOff1, rowCap1, first1 + tid);
					__syncthreads();
				}
			}
			else
			{
				for (u32 c = 0; c < L.numColors; ++c)
				{
					for (u32 i = L.colorStart[c] + tid; i < L.colorStart[c + 1]; i += CLS_LANES)
					{
						float4 m0 = lds[L.metaOff + 2 * i], sh = lds[L.metaOff + 2 * i + 1];
						clSolveManifold<false>(lds, L, A, i, __float_as_uint(m0.x), __float_as_uint(m0.y), sh, noRow);
					}
					__syncthreads();
				}
				for (u32 s = L.serialStart; s < L.count; ++s)
				{
					if (tid == 0)
					{
						float4 m0 = lds[L.metaOff + 2 * s], sh = lds[L.metaOff + 2 * s + 1];
						clSolveManifold<false>(lds, L, A, s, __float_as_uint(m0.x), __float_as_uint(m0.y), sh, noRow);
					}
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
		if (L.inRegs)
		{
#define CL_STORE_REG(R_, KE_, ROW_) \
			{ \
				u32 i = tid + (R_) * CLS_LANES; \
				if (i < L.count) \
				{ \
					u32 slot = L.first + i, count = 4u - (KE_ & 3u), extra = KE_ >> 10; \
					A.rowLambda[slot] = ROW_.lam; \
					for (u32 kk = 1; kk < count; ++kk) \
					{ \
						u32 row = extra + kk - 1u; \
						if (row < L.rowCap) A.rowLambda[(size_t)kk * A.rowCap + slot] = clLoadLambdaLds(lds, L.rowOff, L.rowCap, row); \
					} \
				} \
			}
			CL_STORE_REG(0u, regKE0, regRow0)
			CL_STORE_REG(1u, regKE1, regRow1)
#undef CL_STORE_REG
		}
		else for (u32 i = tid; i < L.count; i += CLS_LANES)
		{
			u32 slot = L.first + i;
			u32 keyExtra = __float_as_uint(lds[L.metaOff + 2 * i].y);
			u32 count = 4u - (keyExtra & 3u), extra = keyExtra >> 10;
			for (u32 kk = 0; kk < count; ++kk)
			{
				u32 row = i + extra + kk;
				if (row < L.rowCap) A.rowLambda[(size_t)kk * A.rowCap + slot] = clLoadLambdaLds(lds, L.rowOff, L.rowCap, row);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------------
static size_t clColorLdsBytes()
{
	return sizeof(u32) * (2 * CL_HASH_SIZE + (CL_TASK_MAX_BODIES + 1) + 5 * CL_TASK_MAX_MANIFOLDS + 264 + CL_MAX_JOINT_CLASSES + 1) + sizeof(u64) * (CL_TASK_MAX_BODIES + 1);
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
	w.clTaskKey.ensure(w.pairCap, w.stream); w.clTaskPos.ensure(w.pairCap, w.stream); w.clPre.ensure(w.pairCap, w.stream); w.clLocal.ensure(w.pairCap, w.stream); w.clExtra.ensure(w.pairCap, w.stream); w.clRankInfo.ensure(w.pairCap, w.stream);
	const u32 totalKeys = CL_MAX_PHASES * CL_MAX_TASKS;
	w.clTaskCount.ensure(totalKeys * CL_SUBCOUNTERS + 6u * CL_REMAIN_SUBS, w.stream); w.clTaskStart.ensure(totalKeys * CL_SUBCOUNTERS + 1, w.stream);
	w.clTasks.ensure((size_t)totalKeys * sizeof(ClTask), w.stream); w.clBodyList.ensure((size_t)totalKeys * CL_BODY_STRIDE, w.stream); w.clBodyUsers.ensure((size_t)totalKeys * CL_BODY_STRIDE, w.stream);
	if (w.lastError) return;

	dim3 bgrid((nb + 255) / 256), block(256), mgrid((numPairs + 255) / 256);
	// active manifolds (k_active_list of the colouring: also counts contacts); no warm colours, no global colour masks
	launch_active_list(w, numPairs);
	// Body order along the phases' curves.  Any order is correct, this one makes the clusters compact; bodies move a fraction of
	// their size per step, so the order is refreshed every few steps only (four radix sorts of all bodies), at once when bodies were
	// added and after a snapshot was taken or restored (so that a restored world and its original keep making the same choices).
	const u32 P = CL_MAX_PARTS; // all curves, whatever the number of partition phases in use: that number adapts from step to step
	bool refresh = false; // this step re-sorts the bodies and re-cuts the chunks; the steps in between reuse the stored chunks
	if (w.clusterSortDue || w.clusterSortAge >= w.clusterSortInterval || w.clusterSortBodies != nb)
	{
		refresh = true;
		ClShifts sh; u32 maxShift = 0;
		for (u32 p = 0; p < CL_MAX_PARTS; ++p) for (u32 k = 0; k < 3; ++k) { sh.s[p][k] = w.clusterShift[p][k]; maxShift = std::max(maxShift, sh.s[p][k]); }
		hipLaunchKernelGGL(k_cl_bbox, dim3(std::min<u32>(bgrid.x, 64u)), block, 0, w.stream, nb, w.cog.p, w.simMask.p, w.dCounters.p);
		hipLaunchKernelGGL(k_cl_keys, bgrid, block, 0, w.stream, nb, P, sh, maxShift, w.cog.p, w.simMask.p, w.dCounters.p, w.clKeys.p, w.clVals.p);
		for (u32 p = 0; p < P; ++p)
			prim_sort_pairs_u32(w, w.clKeys.p + (size_t)p * nb, w.clKeysSorted.p + (size_t)p * nb, w.clVals.p + (size_t)p * nb, w.clSorted.p + (size_t)p * nb, nb, 30);
		hipLaunchKernelGGL(k_cl_ranks, bgrid, block, 0, w.stream, nb, P, w.clSorted.p, w.clRank.p);
		w.clusterSortDue = false; w.clusterSortAge = 0; w.clusterSortBodies = nb;
	}
	w.clusterSortAge++;
	// tasks
	const u32 parts = w.clusterParts;
	u32 clearItems = std::max<u32>((u32)(CL_MAX_PARTS * nb1), totalKeys * CL_SUBCOUNTERS + 6u * CL_REMAIN_SUBS);
	const bool withJoints = cluster_solves_joints(w);
	const u32* rep = withJoints ? w.clRep.p : nullptr;
	const u32 nj = withJoints ? w.clNumJoints : 0u;
	w.clJointCount.ensure(CL_MAX_TASKS, w.stream); w.clJointStart.ensure(CL_MAX_TASKS + 1, w.stream);
	w.clJointTask.ensure(std::max(nj, 1u), w.stream); w.clJointPos.ensure(std::max(nj, 1u), w.stream); w.clJointList.ensure(std::max(nj, 1u), w.stream); w.clTaskJoints.ensure(std::max(nj, 1u), w.stream);
	w.clJointClassStart.ensure((size_t)CL_MAX_TASKS * (CL_MAX_JOINT_CLASSES + 2u), w.stream);
	if (w.lastError) return;
	hipLaunchKernelGGL(k_cl_clear, dim3((clearItems + 255) / 256), block, 0, w.stream, (u32)nb1, w.clWsum.p, w.clPhaseMask.p, w.clTaskCount.p, w.clJointCount.p, w.dCounters.p);
	const u32 maxTasks = std::min<u32>(CL_MAX_TASKS / CL_TASKS_PER_PHASE, w.clusterBlocks) - std::min<u32>(8u, w.clusterBlocks / 8u); // per phase, with a margin for the chunks' rounding
	w.clChunk.ensure((size_t)CL_MAX_PARTS * nb1, w.stream);
	if (w.lastError) return;
	if (!w.useChunkCache || w.clChunkParts < parts || w.clChunkJointVersion != w.jointVersion || w.clChunkWithJoints != withJoints) refresh = true;
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
			hipLaunchKernelGGL(k_cl_assign, mgrid, block, 0, w.stream, w.dCounters.p, nb, p, parts, p ? weightLater : weight0, maxTasks, w.actIds.p, w.clRank.p + (size_t)p * nb1, w.clCum.p,
				w.clRank.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1, wsumNext, w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p, w.dCounters.p + CTR_CL_STATUS, p == 0 ? rep : nullptr);
			if (p == 0 && nj) // (cum still holds phase 0's scan)
				hipLaunchKernelGGL(k_cl_joint_assign, dim3((nj + 255) / 256), block, 0, w.stream, nj, nb, weight0, maxTasks, w.clJointTable.p, w.clRank.p, rep, w.clCum.p, w.clJointTask.p, w.clJointPos.p, w.clJointCount.p, w.clPhaseMask.p, w.dCounters.p + CTR_CL_STATUS);
			if (w.useChunkCache)
				hipLaunchKernelGGL(k_cl_store_chunks, bgrid, block, 0, w.stream, nb, p ? weightLater : weight0, maxTasks, w.clRank.p + (size_t)p * nb1, w.clCum.p, p == 0 ? rep : nullptr, w.clChunk.p + (size_t)p * nb1);
		}
		w.clChunkParts = parts; w.clChunkJointVersion = w.jointVersion; w.clChunkWithJoints = withJoints;
	}
	else
	{
		const u32 cached = std::min(parts, w.chunkCachedPhases);
		hipLaunchKernelGGL(k_cl_assign_cached, mgrid, block, 0, w.stream, w.dCounters.p, nb, parts, cached, withJoints ? 1u : 0u, w.actIds.p, w.clChunk.p,
			w.clRank.p + (size_t)std::min(cached, CL_MAX_PARTS - 1) * nb1, w.clWsum.p + (size_t)std::min(cached, CL_MAX_PARTS - 1) * nb1, w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p);
		if (nj) hipLaunchKernelGGL(k_cl_joint_assign_cached, dim3((nj + 255) / 256), block, 0, w.stream, nj, w.clJointTable.p, w.clChunk.p, w.clJointTask.p, w.clJointPos.p, w.clJointCount.p, w.clPhaseMask.p);
		for (u32 p = cached; p < parts; ++p) // the later phases: the per-step pipeline on what is left
		{
			u32* wsum = w.clWsum.p + (size_t)p * nb1; u32* wsumNext = w.clWsum.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1;
			prim_exclusive_scan_u32(w, wsum, w.clCum.p, nb + 1);
			hipLaunchKernelGGL(k_cl_assign, mgrid, block, 0, w.stream, w.dCounters.p, nb, p, parts, p ? w.clusterTaskWeightLater : w.clusterTaskWeight, maxTasks, w.actIds.p, w.clRank.p + (size_t)p * nb1, w.clCum.p,
				w.clRank.p + (size_t)std::min(p + 1, CL_MAX_PARTS - 1) * nb1, wsumNext, w.clTaskKey.p, w.clTaskPos.p, w.clTaskCount.p, w.clPhaseMask.p, w.dCounters.p + CTR_CL_STATUS, (const u32*)nullptr);
		}
	}
	hipLaunchKernelGGL(k_cl_offsets, dim3(1), dim3(1024), 0, w.stream, w.dCounters.p, parts, w.clTaskCount.p, w.clTaskStart.p, nj ? w.clJointCount.p : (u32*)nullptr, nj ? w.clJointStart.p : (u32*)nullptr);
	if (nj) hipLaunchKernelGGL(k_cl_joint_scatter, dim3((nj + 255) / 256), block, 0, w.stream, nj, w.clJointTask.p, w.clJointPos.p, w.clJointStart.p, w.clJointList.p);
	hipLaunchKernelGGL(k_cl_scatter, mgrid, block, 0, w.stream, w.dCounters.p, w.clTaskKey.p, w.clTaskPos.p, w.clTaskStart.p, w.clPre.p);
	hipLaunchKernelGGL(k_cl_color, dim3(w.clusterBlocks), dim3(CL_LANES), clColorLdsBytes(), w.stream, w.dCounters.p, nb, w.clTaskStart.p, w.clPre.p, w.actIds.p,
		w.clPhaseMask.p, (ClTask*)w.clTasks.p, w.clBodyList.p, w.clBodyUsers.p, w.mOrder.p, w.mKeySorted.p, w.clLocal.p, w.clExtra.p, w.clRankInfo.p, w.clSharedSlot.p,
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
	ClArgs A;
	A.counters = w.dCounters.p; A.tasks = (const ClTask*)w.clTasks.p; A.bodyList = w.clBodyList.p; A.bodyUsers = w.clBodyUsers.p; A.phaseMask = w.clPhaseMask.p; A.sharedSlot = w.clSharedSlot.p;
	A.mKeySorted = w.mKeySorted.p; A.mLocal = w.clLocal.p; A.mExtra = w.clExtra.p; A.mRank = w.clRankInfo.p;
	A.rowPlanes = w.rowPlanes.p; A.rowShared = w.rowShared.p; A.rowLambda = w.rowLambda.p; A.vel = w.vel.p; A.flow = w.flow.p; A.trace = w.flowTrace.p; A.predictDiv = w.clusterPredictDiv; A.pollSleep = w.clusterPollSleep;
	A.rowCap = w.rowCap; A.nb = w.nb; A.flowBytes = (u32)(words * sizeof(u64)); A.epoch = w.flowEpoch << 16; A.itBegin = itBegin; A.itEnd = itEnd;
	A.ldsFloat4s = w.clusterLdsBytes / 16u;
	const bool withJoints = cluster_solves_joints(w);
	A.jointClassStart = withJoints ? w.clJointClassStart.p : nullptr; A.taskJoints = w.clTaskJoints.p; A.jointTable = w.clJointTable.p; A.invIw = w.invIw.p; A.numJointClasses = withJoints ? w.clNumJointClasses : 0u;
	for (u32 t = 0; t < MI_JOINT_TYPES; ++t) A.jointUpd[t] = w.joints[t].dUpdate.p;
	if (withJoints) hipLaunchKernelGGL(k_cl_solve<true>, dim3(w.clusterBlocks), dim3(CLS_LANES), w.clusterLdsBytes, w.stream, A);
	else hipLaunchKernelGGL(k_cl_solve<false>, dim3(w.clusterBlocks), dim3(CLS_LANES), w.clusterLdsBytes, w.stream, A);
}
