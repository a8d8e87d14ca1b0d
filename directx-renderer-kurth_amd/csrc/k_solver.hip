// Contact constraint pipeline on the GPU: graph colouring of contact manifolds (replaces the reference's serial greedy 8-lane
// scheduler, constraints.cpp:51-184), contact row initialisation (constraints.cpp:3307-3379 / 3451-3616) and the projected
// Gauss-Seidel sweep (constraints.cpp:3381-3449 / 3618-3709).
//
// MI355X design.  The unit of scheduling is the MANIFOLD (<= 4 contacts between one body pair): one lane owns a manifold, keeps
// both bodies' velocities in registers across its contacts and touches each body once per sweep (2 x 32-B gather + scatter), so
// box stacks need 4x fewer colours and body traffic than per-contact colouring.  Manifolds of one colour share no dynamic body
// => a colour is one fully parallel launch; colours run in order.  Rows are SoA float4 planes indexed by the manifold's position
// in (colour, contact-count) order, so every wave reads 1 KiB contiguous per plane (16 B/lane), lanes that still have a k-th
// contact are contiguous, and a lane knows its contact count from its slot index alone — every row load of a sweep is issued
// before the first dependent gather returns.  No atomics on velocities; the static dummy body (index numBodies) is never written.
#include "world.h"
#include <rocprim/rocprim.hpp>

void prim_sort_pairs_u32(World& w, const u32* kin, u32* kout, const u32* vin, u32* vout, u32 n, u32 bits);

#define UNCOLORED 0xFFFFFFFFu
#define KEY_INACTIVE (MI_NUM_SCHEDULE_KEYS + 3u) // sorts behind every schedule key; 264 buckets
MI_DEV u32 hash32(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// later rounds and (pseudo-random) higher priority => smaller key; the manifold slot makes keys unique and order-independent
MI_DEV u64 claimKey(u32 round, u32 slot) { return ((u64)(0xFFFFu - round) << 48) | ((u64)(hash32(slot * 2654435761u + round) & 0xFFFFFFu) << 24) | (u64)(slot & 0xFFFFFFu); }

// ---------------------------------------------------------------------------------------------------------------
// Active list: manifolds with at least one contact, appended with one wave-aggregated atomic per wave.  The list order is
// arbitrary; nothing downstream depends on it (claims are keyed by slot, a colour's members are mutually independent).
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// Warm start of the colouring.  Most manifolds persist from step to step; a persisting manifold keeps last step's colour (two
// persisting manifolds on one body had different colours then, so they still have), and only the new ones go through the claim
// rounds — 3-6 rounds instead of ~24.  Last step's colours live in a hash table keyed by the collider pair (open addressing,
// 64-bit entries {valid:1, colliderLo:28, colliderHi:28, colour:6}, two tables used alternately so that one can be cleared while the other is
// read).  Kept colours are never lowered, so the palette slowly spreads: every 16th step (and whenever the colour count nears the
// 64-colour limit, after a snapshot, or on request) the world is coloured from scratch.
// ---------------------------------------------------------------------------------------------------------------
#define COLOR_HASH_EMPTY 0ull
MI_DEV u64 colorHashKey(u32 ia, u32 ib) { u32 lo = min(ia, ib), hi = max(ia, ib); return ((u64)(lo & 0x0FFFFFFFu) << 34) | ((u64)(hi & 0x0FFFFFFFu) << 6); } // colour goes into the low 6 bits
MI_DEV u32 colorHashSlot(u64 key, u32 mask) { u64 k = key >> 6; k ^= k >> 29; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 32; return (u32)k & mask; }
// entry = key | colour, with bit 63 set so that a valid entry is never COLOR_HASH_EMPTY (colliders < 2^28)
MI_DEV u32 colorHashLookup(const u64* __restrict__ table, u32 mask, u64 key)
{
	u32 h = colorHashSlot(key, mask);
	for (u32 probe = 0; probe < 64; ++probe)
	{
		u64 e = table[(h + probe) & mask];
		if (e == COLOR_HASH_EMPTY) return UNCOLORED;
		if ((e & ~0x3Full) == (key | (1ull << 63))) return (u32)(e & 0x3Full);
	}
	return UNCOLORED;
}
__global__ void __launch_bounds__(256) k_color_store(const u32* __restrict__ counters, const uint4* __restrict__ actIds, const u32* __restrict__ mColor,
	const u64* __restrict__ pairSorted, u64* __restrict__ table, u32 mask)
{
	u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= counters[CTR_NUM_ACTIVE]) return;
	u32 c = mColor[j];
	if (c >= MI_MAX_COLORS) return;
	u64 packed = pairSorted[actIds[j].w];
	u64 key = colorHashKey((u32)packed, (u32)(packed >> 32)) | (1ull << 63);
	u32 h = colorHashSlot(key & ~(1ull << 63), mask);
	for (u32 probe = 0; probe < 64; ++probe) // a table four times the manifold count: 64 probes practically always suffice; a miss only costs a recolouring
	{
		unsigned long long old = atomicCAS((unsigned long long*)&table[(h + probe) & mask], COLOR_HASH_EMPTY, key | c);
		if (old == COLOR_HASH_EMPTY) return;
	}
}

__global__ void __launch_bounds__(1024) k_active_list(u32* __restrict__ counters, const ManifoldRec* __restrict__ manifolds, uint4* __restrict__ actIds, u32* __restrict__ mColor,
	const u64* __restrict__ pairSorted, const u64* __restrict__ warmTable, u32 warmMask, u32 nb, u64* __restrict__ bodyMask)
{
	// ONE pair of global atomics per 1024-lane workgroup: same-address atomics from all over the chip serialise (two per wave cost
	// 100 us at 500k candidate pairs).  Waves reserve their ranges in an LDS counter, lane 0 of the workgroup reserves the global range.
	__shared__ u32 sCount, sContacts, sBase;
	u32 m = blockIdx.x * blockDim.x + threadIdx.x;
	if (m == 0) // state of the colouring that follows (nobody else touches these words in this kernel)
	{
		counters[CTR_LAST_ROUND] = 0; counters[CTR_OVERFLOW] = 0;
		for (u32 i = 0; i < 4; ++i) counters[CTR_COLOR_BARRIER + i] = 0;
	}
	if (threadIdx.x == 0) { sCount = 0; sContacts = 0; }
	__syncthreads();
	uint4 ids = make_uint4(0, 0, 0, 0);
	if (m < counters[CTR_NUM_VALID]) ids = manifolds[m].ids;
	u32 total = ids.z; // contacts of this wave
	for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
	bool active = ids.z != 0;
	u64 mask = __ballot(active);
	u32 lane = threadIdx.x & 63u;
	u32 waveBase = 0;
	if (lane == 0 && mask) { waveBase = atomicAdd(&sCount, (u32)__popcll(mask)); atomicAdd(&sContacts, total); }
	waveBase = __shfl(waveBase, 0);
	__syncthreads();
	if (threadIdx.x == 0 && sCount) { sBase = atomicAdd(&counters[CTR_NUM_ACTIVE], sCount); atomicAdd(&counters[CTR_NUM_CONTACTS], sContacts); }
	__syncthreads();
	if (!active) return;
	u32 j = sBase + waveBase + (u32)__popcll(mask & ((1ull << lane) - 1ull));
	actIds[j] = make_uint4(ids.x, ids.y, ids.z, m);
	u32 c = UNCOLORED;
	if (warmTable) // this manifold existed last step: keep its colour
	{
		u64 packed = pairSorted[m];
		c = colorHashLookup(warmTable, warmMask, colorHashKey((u32)packed, (u32)(packed >> 32)));
		if (c != UNCOLORED)
		{
			if (ids.x < nb) atomicOr((unsigned long long*)&bodyMask[ids.x], 1ull << c);
			if (ids.y < nb) atomicOr((unsigned long long*)&bodyMask[ids.y], 1ull << c);
		}
	}
	mColor[j] = c;
}

// ---------------------------------------------------------------------------------------------------------------
// Colouring: Luby-style rounds.  Round r: every uncoloured manifold checks whether it won BOTH of its bodies in round r-1
// (then takes the lowest colour free at both bodies); otherwise it bids again.  Bids are 64-bit atomicMin into a per-body
// slot, double-buffered by round parity; min() is order-independent, so the colouring is deterministic.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_color_round(u32* __restrict__ counters, u32 nb, u32 round, u32 lastRound, const uint4* __restrict__ actIds,
	u32* __restrict__ mColor, u64* __restrict__ bodyMask, u64* __restrict__ claim)
{
	u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= counters[CTR_NUM_ACTIVE]) return;
	if (mColor[j] != UNCOLORED) return;
	uint4 ids = actIds[j];
	u32 a = ids.x, b = ids.y, slot = ids.w;
	bool da = a < nb, db = b < nb;
	if (round > 0)
	{
		const u64* prev = claim + (size_t)((round - 1) & 1) * nb;
		u64 key = claimKey(round - 1, slot);
		bool won = (!da || prev[a] == key) && (!db || prev[b] == key);
		if (won)
		{
			u64 used = (da ? bodyMask[a] : 0ull) | (db ? bodyMask[b] : 0ull);
			u64 freeMask = ~used;
			u32 c;
			if (freeMask == 0ull) { c = MI_SERIAL_COLOR; }
			else
			{
				c = (u32)__ffsll((long long)freeMask) - 1;
				if (da) bodyMask[a] |= (1ull << c);
				if (db) bodyMask[b] |= (1ull << c);
			}
			mColor[j] = c;
			atomicMax(&counters[CTR_LAST_ROUND], round); // drives the adaptive round budget
			return;
		}
	}
	if (round == lastRound) // out of rounds: the serial bucket keeps the step correct; the host raises the budget
	{
		mColor[j] = MI_SERIAL_COLOR;
		atomicAdd(&counters[CTR_OVERFLOW], 1u);
		return;
	}
	u64* cur = claim + (size_t)(round & 1) * nb;
	u64 key = claimKey(round, slot);
	if (da) atomicMin((unsigned long long*)&cur[a], (unsigned long long)key);
	if (db) atomicMin((unsigned long long*)&cur[b], (unsigned long long)key);
}

__global__ void __launch_bounds__(256) k_color_keys(const u32* __restrict__ counters, u32 numPairs, const uint4* __restrict__ actIds, const u32* __restrict__ mColor,
	u32* __restrict__ mKey, u32* __restrict__ mIdx)
{
	u32 j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= numPairs) return;
	u32 key = KEY_INACTIVE;
	if (j < counters[CTR_NUM_ACTIVE])
	{
		u32 c = mColor[j];
		if (c <= MI_SERIAL_COLOR) key = c * 4 + (4 - actIds[j].z);
	}
	mKey[j] = key;
	mIdx[j] = j;
}

__global__ void __launch_bounds__(512) k_color_offsets(u32* __restrict__ counters, u32 numPairs, const u32* __restrict__ keySorted)
{
	u32 k = threadIdx.x; // schedule key colour*4 + (4-count); k = MI_NUM_SCHEDULE_KEYS -> end of schedule = numManifolds
	__shared__ u32 starts[MI_NUM_SCHEDULE_KEYS + 1];
	if (k <= MI_NUM_SCHEDULE_KEYS)
	{
		u32 target = (k < MI_NUM_SCHEDULE_KEYS) ? k : KEY_INACTIVE;
		u32 lo = 0, hi = numPairs;
		while (lo < hi) { u32 mid = (lo + hi) >> 1; if (keySorted[mid] < target) lo = mid + 1; else hi = mid; }
		starts[k] = lo;
		counters[CTR_KEY_START + k] = lo;
	}
	__syncthreads();
	if (k == 0)
	{
		u32 n = 0;
		for (u32 c = 0; c < MI_MAX_COLORS; ++c) if (starts[(c + 1) * 4] > starts[c * 4]) n = c + 1;
		counters[CTR_NUM_COLORS] = n;
		counters[CTR_NUM_MANIFOLDS] = starts[MI_NUM_SCHEDULE_KEYS];
		counters[CTR_FLOW_STATUS] = 0; counters[CTR_FLOW_PROBES] = 0;
	}
}

void launch_active_list(World& w, u32 numPairs)
{
	if (!numPairs) return;
	hipLaunchKernelGGL(k_active_list, dim3((numPairs + 1023) / 1024), dim3(1024), 0, w.stream, w.dCounters.p, w.manifolds.p, w.actIds.p, w.mColor.p,
		(const u64*)w.pairsSorted.p, (const u64*)nullptr, 0u, w.nb, w.bodyMask.p);
}

void launch_coloring(World& w, u32 numPairs)
{
	if (!numPairs) return;
	dim3 grid((numPairs + 255) / 256), block(256);
	u32 nb = w.nb;
	// warm start: last step's colours by collider pair, unless it is time for a colouring from scratch
	u32 tableSize = 1024; while (tableSize < 4u * std::max<u32>(numPairs, w.lastNumManifolds)) tableSize <<= 1;
	bool sizeChanged = w.colorHash[0].cap < tableSize;
	for (int t = 0; t < 2; ++t) if (w.colorHash[t].cap < tableSize) w.colorHash[t].ensure(tableSize, w.stream);
	if (sizeChanged) w.colorHashSize = 0;
	bool warm = w.useWarmColoring && w.colorHashSize == tableSize && !w.forceFullColoring && w.stepsSinceFullColoring < w.fullColoringInterval && w.nc < (1u << 28)
		&& w.hCounters[CTR_NUM_COLORS] < 48;
	w.stepsSinceFullColoring = warm ? w.stepsSinceFullColoring + 1 : 0;
	w.forceFullColoring = false;
	const u64* readTable = warm ? w.colorHash[w.colorHashCur].p : nullptr;
	hipLaunchKernelGGL(k_active_list, dim3((numPairs + 1023) / 1024), dim3(1024), 0, w.stream, w.dCounters.p, w.manifolds.p, w.actIds.p, w.mColor.p,
		(const u64*)w.pairsSorted.p, readTable, tableSize - 1, nb, w.bodyMask.p);
	// the active count is not known on the host yet: size the round launches by last step's count (+25 %), never above numPairs
	u32 est = w.lastNumManifolds ? std::min<u32>(numPairs, w.lastNumManifolds + w.lastNumManifolds / 4 + 1024) : numPairs;
	dim3 rgrid((est + 255) / 256);
	u32 rounds = w.coloringRounds; // one launch per round: no grid barrier, nothing that needs the whole chip
	for (u32 r = 0; r <= rounds; ++r)
		hipLaunchKernelGGL(k_color_round, (r == rounds) ? grid : rgrid, block, 0, w.stream, w.dCounters.p, nb, r, rounds, w.actIds.p, w.mColor.p, w.bodyMask.p, w.claim.p);
	if (w.useWarmColoring) // remember this step's colours for the next one (in the other table)
	{
		u32 other = w.colorHashCur ^ 1u;
		MI_CHECK(hipMemsetAsync(w.colorHash[other].p, 0, sizeof(u64) * tableSize, w.stream));
		hipLaunchKernelGGL(k_color_store, grid, block, 0, w.stream, w.dCounters.p, w.actIds.p, w.mColor.p, (const u64*)w.pairsSorted.p, w.colorHash[other].p, tableSize - 1);
		w.colorHashCur = other; w.colorHashSize = tableSize;
	}
	hipLaunchKernelGGL(k_color_keys, grid, block, 0, w.stream, w.dCounters.p, numPairs, w.actIds.p, w.mColor.p, w.mKey.p, w.mIdx.p);
	csort_pairs_u32(w, w.mKey.p, w.mKeySorted.p, w.mIdx.p, w.mOrder.p, numPairs, KEY_INACTIVE + 1);
	hipLaunchKernelGGL(k_color_offsets, dim3(1), dim3(512), 0, w.stream, w.dCounters.p, numPairs, w.mKeySorted.p);
}

// ---------------------------------------------------------------------------------------------------------------
// K10: contact rows (constraints.cpp:3307-3379).  Per manifold: 96-B manifold gather + 2 x (cog 16 + invI 48 + vel 32) body gathers;
// writes MI_ROW_PLANES float4 planes + lambda per contact (layout: solver_rows.h), one shared float4 (normal, friction) and the id quad.
// ---------------------------------------------------------------------------------------------------------------
MI_DEV M3 loadInvI(const float4* __restrict__ invIw, u32 i)
{
	float4 c0 = invIw[3 * i], c1 = invIw[3 * i + 1], c2 = invIw[3 * i + 2];
	M3 I; I.m00 = c0.x; I.m10 = c0.y; I.m20 = c0.z; I.m01 = c1.x; I.m11 = c1.y; I.m21 = c1.z; I.m02 = c2.x; I.m12 = c2.y; I.m22 = c2.z;
	return I;
}

__global__ void __launch_bounds__(256) k_contact_init(const u32* __restrict__ counters, float dt, size_t rowCap, const u32* __restrict__ mOrder, const uint4* __restrict__ actIds,
	const ManifoldRec* __restrict__ manifolds, const float4* __restrict__ cog, const float4* __restrict__ invIw, const float4* __restrict__ vel,
	float4* __restrict__ rowPlanes, float4* __restrict__ rowShared, float2* __restrict__ rowLambda, uint4* __restrict__ rowIds)
{
	u32 s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= counters[CTR_NUM_MANIFOLDS]) return;
	u32 m = actIds[mOrder[s]].w;
	ManifoldRec man = manifolds[m];
	u32 a = man.ids.x, b = man.ids.y, count = man.ids.z;
	float4 ca = cog[a], cb = cog[b];
	V3 posA = v3f4(ca), posB = v3f4(cb);
	float invMassA = ca.w, invMassB = cb.w;
	M3 IA = loadInvI(invIw, a), IB = loadInvI(invIw, b);
	V3 vA = v3f4(vel[2 * a]), wA = v3f4(vel[2 * a + 1]), vB = v3f4(vel[2 * b]), wB = v3f4(vel[2 * b + 1]);
	V3 n = v3f4(man.nf);
	u32 fr = __float_as_uint(man.nf.w);
	float friction = (float)(fr >> 16) / (float)0xFFFF;
	float restitution = (float)(fr & 0xFFFF) / (float)0xFFFF;
	float invDt = 1.f / dt;

	rowShared[s] = make_float4(n.x, n.y, n.z, friction);
	rowIds[s] = make_uint4(a, b, count, m);

#pragma unroll
	for (u32 k = 0; k < MI_MAX_CONTACTS_PER_MANIFOLD; ++k) // (unrolled: man.p[k] stays in registers; a count-bounded loop indexes it dynamically, i.e. through scratch memory)
	{
		if (k >= count) break;
		V3 point = v3f4(man.p[k]);
		float depth = man.p[k].w;
		V3 rA = point - posA, rB = point - posB;
		V3 anchorVelocityA = vA + cross(wA, rA);
		V3 anchorVelocityB = vB + cross(wB, rB);
		V3 rel = anchorVelocityB - anchorVelocityA;
		V3 t = noz(rel - dot(n, rel) * n);

		V3 crAt = cross(rA, t), crBt = cross(rB, t);
		float invT = invMassA + dot(crAt, IA * crAt) + invMassB + dot(crBt, IB * crBt);
		float mT = (invT != 0.f) ? (1.f / invT) : 0.f;
		V3 JtA = IA * crAt, JtB = IB * crBt;

		V3 crAn = cross(rA, n), crBn = cross(rB, n);
		float invN = invMassA + dot(crAn, IA * crAn) + invMassB + dot(crBn, IB * crBn);
		float mN = (invN != 0.f) ? (1.f / invN) : 0.f;
		float bias = 0.f;
		if (dt > 1e-5f)
		{
			float vRel = dot(n, rel);
			const float slop = -0.001f;
			if (-depth < slop && vRel < 0.f) { bias = -restitution * vRel - 0.1f * (-depth - slop) * invDt; }
		}
		V3 JnA = IA * crAn, JnB = IB * crBn;

		float4* P = rowPlanes + (size_t)(k * MI_ROW_PLANES) * rowCap + s;
		P[0 * rowCap] = make_float4(t.x, t.y, t.z, crAt.x);
		P[1 * rowCap] = make_float4(crAt.y, crAt.z, crBt.x, crBt.y);
		P[2 * rowCap] = make_float4(crBt.z, crAn.x, crAn.y, crAn.z);
		P[3 * rowCap] = make_float4(crBn.x, crBn.y, crBn.z, JtA.x);
		P[4 * rowCap] = make_float4(JtA.y, JtA.z, JtB.x, JtB.y);
		P[5 * rowCap] = make_float4(JtB.z, JnA.x, JnA.y, JnA.z);
		P[6 * rowCap] = make_float4(JnB.x, JnB.y, JnB.z, mN);
		P[7 * rowCap] = make_float4(mT, bias, 0.f, 0.f);
		rowLambda[(size_t)k * rowCap + s] = make_float2(0.f, 0.f);
	}
}

void launch_contact_init(World& w, u32 numPairs, float dt)
{
	if (!numPairs) return;
	hipLaunchKernelGGL(k_contact_init, dim3((numPairs + 255) / 256), dim3(256), 0, w.stream, w.dCounters.p, dt, w.rowCap, w.mOrder.p, w.actIds.p, w.manifolds.p,
		w.cog.p, w.invIw.p, w.vel.p, w.rowPlanes.p, w.rowShared.p, w.rowLambda.p, w.rowIds.p);
}

// ---------------------------------------------------------------------------------------------------------------
// K11: the sweep.  One lane = one manifold; friction row then normal row per contact (A.3 of SURVEY).
// ---------------------------------------------------------------------------------------------------------------
#include "solver_rows.h"

// `count` comes from the slot index (the schedule is sorted by contact count inside a colour), so all row loads are issued
// up front, in parallel with the id -> body gather chain.
MI_DEV void solveManifold(u32 s, u32 count, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes, const float4* __restrict__ rowShared, float2* __restrict__ rowLambda,
	const uint4* __restrict__ rowIds, float4* __restrict__ vel)
{
	uint4 ids = rowIds[s];
	float4 sh = rowShared[s];
	ContactRow r0, r1, r2, r3;
	loadRow(r0, 0, s, rowCap, rowPlanes, rowLambda);
	if (count > 1) loadRow(r1, 1, s, rowCap, rowPlanes, rowLambda);
	if (count > 2) loadRow(r2, 2, s, rowCap, rowPlanes, rowLambda);
	if (count > 3) loadRow(r3, 3, s, rowCap, rowPlanes, rowLambda);
	u32 a = ids.x, b = ids.y;
	float4 la = vel[2 * a], aa = vel[2 * a + 1], lb = vel[2 * b], ab = vel[2 * b + 1];
	V3 n = v3(sh.x, sh.y, sh.z);
	float friction = sh.w;
	V3 vA = v3f4(la), wA = v3f4(aa), vB = v3f4(lb), wB = v3f4(ab);
	float invMassA = la.w, invMassB = lb.w;

	solveRow(r0, n, friction, invMassA, invMassB, vA, wA, vB, wB);
	rowLambda[s] = r0.lam;
	if (count > 1) { solveRow(r1, n, friction, invMassA, invMassB, vA, wA, vB, wB); rowLambda[rowCap + s] = r1.lam; }
	if (count > 2) { solveRow(r2, n, friction, invMassA, invMassB, vA, wA, vB, wB); rowLambda[2 * rowCap + s] = r2.lam; }
	if (count > 3) { solveRow(r3, n, friction, invMassA, invMassB, vA, wA, vB, wB); rowLambda[3 * rowCap + s] = r3.lam; }

	if (a < nb) { vel[2 * a] = make_float4(vA.x, vA.y, vA.z, invMassA); vel[2 * a + 1] = make_float4(wA.x, wA.y, wA.z, 0.f); }
	if (b < nb) { vel[2 * b] = make_float4(vB.x, vB.y, vB.z, invMassB); vel[2 * b + 1] = make_float4(wB.x, wB.y, wB.z, 0.f); }
}

// One colour of the schedule.  The slot range and the contact-count boundaries are read from the device counters (written by
// k_color_offsets), so the launch carries no per-step arguments and the whole 30-iteration sweep replays as one hipGraph.
// Slots below b3/b2/b1 hold manifolds with 4/>=3/>=2 contacts.  Grid-stride: a stale (smaller) grid stays correct.
__global__ void __launch_bounds__(256) k_solve_color(u32 color, const u32* __restrict__ counters, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes,
	const float4* __restrict__ rowShared, float2* __restrict__ rowLambda, const uint4* __restrict__ rowIds, float4* __restrict__ vel)
{
	const u32* k = counters + CTR_KEY_START + 4 * color;
	u32 start = k[0], b3 = k[1], b2 = k[2], b1 = k[3], end = k[4];
	for (u32 s = start + blockIdx.x * blockDim.x + threadIdx.x; s < end; s += gridDim.x * blockDim.x)
	{
		u32 count = 1u + (s < b1) + (s < b2) + (s < b3);
		solveManifold(s, count, nb, rowCap, rowPlanes, rowShared, rowLambda, rowIds, vel);
	}
}

// Overflow bucket: bodies with more than 64 simultaneously touching partners.  Sequential, one lane, in slot order.
__global__ void k_solve_serial(const u32* __restrict__ counters, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes, const float4* __restrict__ rowShared,
	float2* __restrict__ rowLambda, const uint4* __restrict__ rowIds, float4* __restrict__ vel)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	u32 start = counters[CTR_KEY_START + 4 * MI_SERIAL_COLOR], end = counters[CTR_KEY_START + 4 * MI_SERIAL_COLOR + 4];
	for (u32 s = start; s < end; ++s) solveManifold(s, rowIds[s].z, nb, rowCap, rowPlanes, rowShared, rowLambda, rowIds, vel);
}

// The small colours at the end of the schedule (greedy colouring leaves a geometric tail: at config 3 the last 11 of 21 colours
// hold 2.6 % of the manifolds) are swept by ONE workgroup in ONE launch, colour after colour with a workgroup barrier in between:
// a kernel boundary costs ~4.7 us on this chip (per-XCD L2 write-back + invalidate), a barrier inside a CU a few hundred ns.
__global__ void __launch_bounds__(1024) k_solve_tail(u32 firstColor, u32 numColors, const u32* __restrict__ counters, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes,
	const float4* __restrict__ rowShared, float2* __restrict__ rowLambda, const uint4* __restrict__ rowIds, float4* __restrict__ vel)
{
	for (u32 color = firstColor; color < numColors; ++color)
	{
		const u32* k = counters + CTR_KEY_START + 4 * color;
		u32 start = k[0], b3 = k[1], b2 = k[2], b1 = k[3], end = k[4];
		for (u32 s = start + threadIdx.x; s < end; s += blockDim.x)
		{
			u32 count = 1u + (s < b1) + (s < b2) + (s < b3);
			solveManifold(s, count, nb, rowCap, rowPlanes, rowShared, rowLambda, rowIds, vel);
		}
		__syncthreads(); // workgroup-scope release/acquire: the next colour sees this colour's velocity writes (same CU, same L1)
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Replay of the REFERENCE's own contact order (debug facility, mi_debug_set_replay): the host runs the reference's greedy batch
// scheduler over this step's contacts (World::scheduleReferenceBatches, constraints.cpp:51-184) and this kernel sweeps the batches
// one after the other — the order solveCollisionVelocityConstraintsSIMD executes them in (constraints.cpp:3618-3709) — with the
// lanes of a batch (disjoint bodies) side by side.  Contact-granular: entry = schedule position | contact index << 28.
// One workgroup, a barrier between batches (same CU, same L1: the next batch sees this batch's velocity writes, as in k_solve_tail).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_solve_replay(u32 numBatches, const u32* __restrict__ entries, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes,
	const float4* __restrict__ rowShared, float2* __restrict__ rowLambda, const uint4* __restrict__ rowIds, float4* __restrict__ vel)
{
	for (u32 bi = 0; bi < numBatches; ++bi)
	{
		u32 e = threadIdx.x < MI_REPLAY_WIDTH ? entries[bi * MI_REPLAY_WIDTH + threadIdx.x] : 0xFFFFFFFFu;
		if (e != 0xFFFFFFFFu)
		{
			u32 s = e & 0x0FFFFFFFu, k = e >> 28;
			uint4 ids = rowIds[s];
			float4 sh = rowShared[s];
			ContactRow r;
			loadRow(r, k, s, rowCap, rowPlanes, rowLambda);
			u32 a = ids.x, b = ids.y;
			float4 la = vel[2 * a], aa = vel[2 * a + 1], lb = vel[2 * b], ab = vel[2 * b + 1];
			V3 vA = v3f4(la), wA = v3f4(aa), vB = v3f4(lb), wB = v3f4(ab);
			solveRow(r, v3(sh.x, sh.y, sh.z), sh.w, la.w, lb.w, vA, wA, vB, wB);
			rowLambda[(size_t)k * rowCap + s] = r.lam;
			if (a < nb) { vel[2 * a] = make_float4(vA.x, vA.y, vA.z, la.w); vel[2 * a + 1] = make_float4(wA.x, wA.y, wA.z, 0.f); }
			if (b < nb) { vel[2 * b] = make_float4(vB.x, vB.y, vB.z, lb.w); vel[2 * b + 1] = make_float4(wB.x, wB.y, wB.z, 0.f); }
		}
		__syncthreads();
	}
}
void launch_solve_replay(World& w, u32 numBatches)
{
	if (!numBatches) return;
	hipLaunchKernelGGL(k_solve_replay, dim3(1), dim3(64), 0, w.stream, numBatches, w.replayEntries.p, w.nb, w.rowCap, w.rowPlanes.p, w.rowShared.p, w.rowLambda.p, w.rowIds.p, w.vel.p);
}

// One Gauss-Seidel iteration over all contact colours; colour c < firstTail is launched with gridBlocks[c] blocks (0 = skip),
// colours [firstTail, numColors) go to the single-workgroup tail kernel.
void launch_solve_contacts_iteration(World& w, const u32* gridBlocks, u32 numColors, u32 firstTail, bool serialBucket)
{
	for (u32 c = 0; c < numColors && c < firstTail; ++c)
	{
		if (!gridBlocks[c]) continue;
		hipLaunchKernelGGL(k_solve_color, dim3(gridBlocks[c]), dim3(256), 0, w.stream, c, w.dCounters.p, w.nb, w.rowCap, w.rowPlanes.p, w.rowShared.p,
			w.rowLambda.p, w.rowIds.p, w.vel.p);
	}
	if (firstTail < numColors)
		hipLaunchKernelGGL(k_solve_tail, dim3(1), dim3(1024), 0, w.stream, firstTail, numColors, w.dCounters.p, w.nb, w.rowCap, w.rowPlanes.p, w.rowShared.p,
			w.rowLambda.p, w.rowIds.p, w.vel.p);
	if (serialBucket)
		hipLaunchKernelGGL(k_solve_serial, dim3(1), dim3(64), 0, w.stream, w.dCounters.p, w.nb, w.rowCap, w.rowPlanes.p, w.rowShared.p, w.rowLambda.p, w.rowIds.p, w.vel.p);
}
