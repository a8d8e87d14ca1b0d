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

// All colouring rounds in ONE launch: the same rounds as k_color_round (identical claims, identical colours), separated by a grid
// barrier (one agent-scope atomic per workgroup + a poll, ~2 us) instead of a kernel boundary (~10 us per round at 200k manifolds).
// Lanes keep their manifolds across rounds (grid-stride with a fixed mapping), so mColor stays private to its lane; the per-body
// words other workgroups read (claims, colour masks) go through agent-scope atomics.  The loop ends as soon as a round leaves
// nothing uncoloured.  All workgroups must be resident (the launcher sizes the grid accordingly); the barrier spin is bounded.
#define COLOR_BARRIER_SPINS (1u << 22)
__global__ void __launch_bounds__(1024) k_color_all(u32* counters, u32 nb, u32 maxRounds, const uint4* __restrict__ actIds, u32* __restrict__ mColor,
	u64* bodyMask, u64* claim)
{
	u32* bar = counters + CTR_COLOR_BARRIER;        // [0] arrivals (monotonic over the launch), [1..3] uncoloured manifolds left after round r % 3
	const u32 n = counters[CTR_NUM_ACTIVE];
	const u32 T = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
	__shared__ u32 sLeft, sStop;
	u32 lastUseful = 0;
	for (u32 round = 0; round <= maxRounds; ++round)
	{
		if (threadIdx.x == 0) sLeft = 0;
		__syncthreads();
		u32 left = 0;
		for (u32 j = tid; j < n; j += T)
		{
			if (mColor[j] != UNCOLORED) continue;
			uint4 ids = actIds[j];
			u32 a = ids.x, b = ids.y, slot = ids.w;
			bool da = a < nb, db = b < nb;
			bool colored = false;
			if (round > 0)
			{
				const u64* prev = claim + (size_t)((round - 1) & 1) * nb;
				u64 key = claimKey(round - 1, slot);
				bool won = (!da || __hip_atomic_load(&prev[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key) && (!db || __hip_atomic_load(&prev[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key);
				if (won)
				{
					u64 used = (da ? __hip_atomic_load(&bodyMask[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull) | (db ? __hip_atomic_load(&bodyMask[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull);
					u64 freeMask = ~used;
					u32 c;
					if (freeMask == 0ull) { c = MI_SERIAL_COLOR; }
					else
					{
						c = (u32)__ffsll((long long)freeMask) - 1;
						if (da) atomicOr((unsigned long long*)&bodyMask[a], 1ull << c); // the only winner on this body in this round
						if (db) atomicOr((unsigned long long*)&bodyMask[b], 1ull << c);
					}
					mColor[j] = c;
					lastUseful = round;
					colored = true;
				}
			}
			if (!colored)
			{
				if (round == maxRounds) { mColor[j] = MI_SERIAL_COLOR; atomicAdd(&counters[CTR_OVERFLOW], 1u); }
				else
				{
					u64* cur = claim + (size_t)(round & 1) * nb;
					u64 key = claimKey(round, slot);
					if (da) atomicMin((unsigned long long*)&cur[a], (unsigned long long)key);
					if (db) atomicMin((unsigned long long*)&cur[b], (unsigned long long)key);
					++left;
				}
			}
		}
		for (int o = 32; o > 0; o >>= 1) left += __shfl_xor(left, o);
		if ((threadIdx.x & 63u) == 0u && left) atomicAdd(&sLeft, left);
		__syncthreads();
		if (threadIdx.x == 0)
		{
			if (sLeft) atomicAdd(&bar[1 + round % 3], sLeft);
			if (blockIdx.x == 0) __hip_atomic_store(&bar[1 + (round + 1) % 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // free since round - 2
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			atomicAdd(&bar[0], 1u);
			u32 target = (round + 1) * gridDim.x, spins = 0;
			while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
			{
				__builtin_amdgcn_s_sleep(2);
				if (++spins > COLOR_BARRIER_SPINS) { atomicOr(&counters[CTR_FLOW_STATUS], 8u); break; }
			}
			sStop = (__hip_atomic_load(&bar[1 + round % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u || spins > COLOR_BARRIER_SPINS) ? 1u : 0u;
		}
		__syncthreads();
		if (sStop) break;
	}
	if (lastUseful) atomicMax(&counters[CTR_LAST_ROUND], lastUseful);
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

// ---------------------------------------------------------------------------------------------------------------
// XCD regions for the dataflow sweep.  The 8 XCDs have private L2s: a hand-over between two workgroups of ONE XCD can stay in that
// L2 (plain store, L1-bypassing load: 0.33 us, no fabric traffic), one across XCDs must be written through (sc1: 0.65 us and two
// fabric transactions per 16 bytes).  So the world is cut into 8 slabs along x holding equal contact work, every manifold is run
// by a workgroup of the XCD that serves its slab, and only bodies touched from two slabs use the write-through records.
// Cuts come from a 2048-bin histogram of the bodies' x (weighted by last step's degree), one step behind: any cuts are correct,
// they only balance the load.
// ---------------------------------------------------------------------------------------------------------------
#define REGION_BINS 2048u
MI_DEV u32 orderedBits(float f) { u32 b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
MI_DEV float orderedFloat(u32 o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__global__ void __launch_bounds__(256) k_region_hist(u32 nb, const float4* __restrict__ cog, const u64* __restrict__ bodyMask, u32* __restrict__ counters, u32* __restrict__ hist)
{
	__shared__ u32 bins[REGION_BINS];
	for (u32 i = threadIdx.x; i < REGION_BINS; i += blockDim.x) bins[i] = 0;
	__syncthreads();
	float lo = __uint_as_float(counters[CTR_REGION_RANGE]), hi = __uint_as_float(counters[CTR_REGION_RANGE + 1]);
	float scale = (hi > lo) ? (float)REGION_BINS / (hi - lo) : 0.f;
	float mn = MI_FLT_MAX, mx = -MI_FLT_MAX;
	for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
	{
		float x = cog[i].x;
		if (!(x == x)) continue;
		mn = fminf(mn, x); mx = fmaxf(mx, x);
		int bin = (int)((x - lo) * scale);
		bin = bin < 0 ? 0 : (bin >= (int)REGION_BINS ? (int)REGION_BINS - 1 : bin);
		atomicAdd(&bins[bin], 1u + (u32)__popcll(bodyMask[i]));
	}
	for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
	if ((threadIdx.x & 63u) == 0u && mn <= mx) { atomicMin(&counters[CTR_REGION_MINMAX], orderedBits(mn)); atomicMax(&counters[CTR_REGION_MINMAX + 1], orderedBits(mx)); }
	__syncthreads();
	for (u32 i = threadIdx.x; i < REGION_BINS; i += blockDim.x) if (bins[i]) atomicAdd(&hist[i], bins[i]);
}

__global__ void __launch_bounds__(1024) k_region_cuts(u32* __restrict__ counters, u32* __restrict__ hist)
{
	__shared__ u32 scan[REGION_BINS];
	u32 t = threadIdx.x;
	scan[t] = hist[t]; scan[t + 1024] = hist[t + 1024];
	hist[t] = 0; hist[t + 1024] = 0;
	__syncthreads();
	if (t == 0) // 2048 adds on one lane: 2 us, once per step
	{
		u32 run = 0;
		for (u32 i = 0; i < REGION_BINS; ++i) { run += scan[i]; scan[i] = run; }
	}
	__syncthreads();
	u32 total = scan[REGION_BINS - 1];
	float lo = __uint_as_float(counters[CTR_REGION_RANGE]), hi = __uint_as_float(counters[CTR_REGION_RANGE + 1]);
	float width = (hi - lo) / (float)REGION_BINS;
	if (t < 7)
	{
		u32 target = (u32)(((u64)total * (t + 1)) / 8u);
		u32 l = 0, h = REGION_BINS;
		while (l < h) { u32 mid = (l + h) >> 1; if (scan[mid] < target) l = mid + 1; else h = mid; }
		counters[CTR_REGION_CUTS + t] = __float_as_uint(total ? lo + width * (float)(l + 1) : MI_FLT_MAX); // cut after bin l
	}
	__syncthreads();
	if (t == 0)
	{
		u32 mnb = counters[CTR_REGION_MINMAX], mxb = counters[CTR_REGION_MINMAX + 1];
		if (mnb <= mxb)
		{
			float mn = orderedFloat(mnb), mx = orderedFloat(mxb);
			float pad = 0.01f * (mx - mn) + 0.5f;
			counters[CTR_REGION_RANGE] = __float_as_uint(mn - pad); counters[CTR_REGION_RANGE + 1] = __float_as_uint(mx + pad);
		}
		counters[CTR_REGION_MINMAX] = 0xFFFFFFFFu; counters[CTR_REGION_MINMAX + 1] = 0u;
	}
}

__global__ void k_region_offsets(u32* __restrict__ counters, const u32* __restrict__ regionSorted, u32 n)
{
	u32 r = threadIdx.x; // 0..8
	if (r > 8) return;
	u32 lo = 0, hi = n;
	while (lo < hi) { u32 mid = (lo + hi) >> 1; if (regionSorted[mid] < r) lo = mid + 1; else hi = mid; }
	counters[CTR_REGION_START + r] = lo;
}

// Worlds below this many manifolds run the dataflow sweep without regions (too few workgroups to populate every XCD).
// Decided once per step (before the colouring, from last step's manifold count) and used by every stage of that step.
u32 flow_num_regions(const World& w) { return w.flowRegions; }
void flow_choose_regions(World& w) { w.flowRegions = (w.useFlow && w.useFlowRegions && w.lastNumManifolds >= 16384u && w.lastNumManifolds <= w.flowMaxManifolds) ? 8u : 1u; }

static void launch_region_cuts(World& w)
{
	u32 nb = w.nb;
	w.regionHist.ensure(REGION_BINS, w.stream);
	int passes = 1;
	if (!w.regionsReady)
	{
		MI_CHECK(hipMemsetAsync(w.regionHist.p, 0, sizeof(u32) * REGION_BINS, w.stream));
		u32 init[4] = { 0u, 0u, 0xFFFFFFFFu, 0u }; // empty range, empty min/max
		MI_CHECK(hipMemcpyAsync(w.dCounters.p + CTR_REGION_RANGE, init, sizeof(init), hipMemcpyHostToDevice, w.stream));
		MI_CHECK(hipMemsetAsync(w.dCounters.p + CTR_FLOW_CENSUS, 0, sizeof(u32) * 16, w.stream));
		MI_CHECK(hipStreamSynchronize(w.stream));
		w.regionsReady = true; passes = 2; // the first pass only finds the range
	}
	for (int p = 0; p < passes; ++p)
	{
		hipLaunchKernelGGL(k_region_hist, dim3(std::min<u32>((nb + 255) / 256, 256u)), dim3(256), 0, w.stream, nb, w.cog.p, w.bodyMask.p, w.dCounters.p, w.regionHist.p);
		hipLaunchKernelGGL(k_region_cuts, dim3(1), dim3(1024), 0, w.stream, w.dCounters.p, w.regionHist.p);
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
	w.regMask.ensure(nb + 1, w.stream);
	if (flow_num_regions(w) > 1)
	{
		launch_region_cuts(w); // reads last step's bodyMask (degrees) before it is cleared
		MI_CHECK(hipMemsetAsync(w.regMask.p, 0, sizeof(u32) * (nb + 1), w.stream));
	}
	else w.regionsReady = false;
	if (flow_num_regions(w) > 1) // otherwise k_integrate_forces has cleared both while it was touching every body anyway
	{
		MI_CHECK(hipMemsetAsync(w.bodyMask.p, 0, sizeof(u64) * (nb + 1), w.stream));
		MI_CHECK(hipMemsetAsync(w.claim.p, 0xFF, sizeof(u64) * 2 * (nb + 1), w.stream));
	}
	// warm start: last step's colours by collider pair, unless it is time for a colouring from scratch
	u32 tableSize = 1024; while (tableSize < 4u * std::max<u32>(numPairs, w.lastNumManifolds)) tableSize <<= 1;
	bool sizeChanged = w.colorHash[0].cap < tableSize;
	for (int t = 0; t < 2; ++t) if (w.colorHash[t].cap < tableSize) w.colorHash[t].ensure(tableSize, w.stream);
	if (sizeChanged) w.colorHashSize = 0;
	bool warm = w.useWarmColoring && w.colorHashSize == tableSize && !w.forceFullColoring && w.stepsSinceFullColoring < w.fullColoringInterval && w.nc < (1u << 28)
		&& w.hCounters[CTR_NUM_COLORS] < 48 && flow_num_regions(w) == 1;
	w.stepsSinceFullColoring = warm ? w.stepsSinceFullColoring + 1 : 0;
	w.forceFullColoring = false;
	const u64* readTable = warm ? w.colorHash[w.colorHashCur].p : nullptr;
	hipLaunchKernelGGL(k_active_list, dim3((numPairs + 1023) / 1024), dim3(1024), 0, w.stream, w.dCounters.p, w.manifolds.p, w.actIds.p, w.mColor.p,
		(const u64*)w.pairsSorted.p, readTable, tableSize - 1, nb, w.bodyMask.p);
	// the active count is not known on the host yet: size the round launches by last step's count (+25 %), never above numPairs
	u32 est = w.lastNumManifolds ? std::min<u32>(numPairs, w.lastNumManifolds + w.lastNumManifolds / 4 + 1024) : numPairs;
	dim3 rgrid((est + 255) / 256);
	if (w.useFusedColoring)
	{
		// all rounds in one launch; every workgroup must be resident for the grid barrier (small kernel: 8 blocks per CU fit)
		if (!w.colorMaxBlocks)
		{
			int perCU = 0, cus = 0;
			MI_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_color_all, 1024, 0));
			MI_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w.device));
			w.colorMaxBlocks = (u32)std::max(1, std::min(perCU, 1)) * (u32)std::max(1, cus); // one 1024-lane workgroup per CU: few barrier arrivals
		}
		// two manifolds per lane: half the barrier arrivals per round, and the second manifold's loads hide behind the first's (measured: -35 us at 120k manifolds)
		u32 blocks = std::min<u32>(std::max(1u, (est + 2047) / 2048), w.colorMaxBlocks);
		hipLaunchKernelGGL(k_color_all, dim3(blocks), dim3(1024), 0, w.stream, w.dCounters.p, nb, 1024u, w.actIds.p, w.mColor.p, w.bodyMask.p, w.claim.p);
	}
	else
	{
		u32 rounds = w.coloringRounds;
		for (u32 r = 0; r <= rounds; ++r)
			hipLaunchKernelGGL(k_color_round, (r == rounds) ? grid : rgrid, block, 0, w.stream, w.dCounters.p, nb, r, rounds, w.actIds.p, w.mColor.p, w.bodyMask.p, w.claim.p);
	}
	if (w.useWarmColoring) // remember this step's colours for the next one (in the other table)
	{
		u32 other = w.colorHashCur ^ 1u;
		MI_CHECK(hipMemsetAsync(w.colorHash[other].p, 0, sizeof(u64) * tableSize, w.stream));
		hipLaunchKernelGGL(k_color_store, rgrid.x >= grid.x ? grid : grid, block, 0, w.stream, w.dCounters.p, w.actIds.p, w.mColor.p, (const u64*)w.pairsSorted.p, w.colorHash[other].p, tableSize - 1);
		w.colorHashCur = other; w.colorHashSize = tableSize;
	}
	hipLaunchKernelGGL(k_color_keys, grid, block, 0, w.stream, w.dCounters.p, numPairs, w.actIds.p, w.mColor.p, w.mKey.p, w.mIdx.p);
	csort_pairs_u32(w, w.mKey.p, w.mKeySorted.p, w.mIdx.p, w.mOrder.p, numPairs, KEY_INACTIVE + 1);
	hipLaunchKernelGGL(k_color_offsets, dim3(1), dim3(512), 0, w.stream, w.dCounters.p, numPairs, w.mKeySorted.p);
}

// ---------------------------------------------------------------------------------------------------------------
// K10: contact rows.  Per manifold: 96-B manifold gather + 2 x (cog 16 + invI 48 + vel 32) body gathers; writes 6 float4 planes +
// lambda per contact, one shared float4 (normal, friction) and the id quad.  Plane p of contact k, slot s: rowPlanes[(k*6+p)*rowCap + s].
//   p0 = rA.xyz rB.x | p1 = rB.yz t.xy | p2 = t.z JnA.xyz | p3 = JtA.xyz JnB.x | p4 = JnB.yz JtB.xy | p5 = JtB.z mN mT bias
// ---------------------------------------------------------------------------------------------------------------
MI_DEV M3 loadInvI(const float4* __restrict__ invIw, u32 i)
{
	float4 c0 = invIw[3 * i], c1 = invIw[3 * i + 1], c2 = invIw[3 * i + 2];
	M3 I; I.m00 = c0.x; I.m10 = c0.y; I.m20 = c0.z; I.m01 = c1.x; I.m11 = c1.y; I.m21 = c1.z; I.m02 = c2.x; I.m12 = c2.y; I.m22 = c2.z;
	return I;
}

__global__ void __launch_bounds__(256) k_contact_init(const u32* __restrict__ counters, float dt, size_t rowCap, const u32* __restrict__ mOrder, const uint4* __restrict__ actIds,
	const ManifoldRec* __restrict__ manifolds, const float4* __restrict__ cog, const float4* __restrict__ invIw, const float4* __restrict__ vel,
	float4* __restrict__ rowPlanes, float4* __restrict__ rowShared, float2* __restrict__ rowLambda, uint4* __restrict__ rowIds,
	u32 nb, u32 numRegions, u32* __restrict__ regMask, u32* __restrict__ mRegion)
{
	u32 s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= counters[CTR_NUM_MANIFOLDS]) return;
	u32 m = actIds[mOrder[s]].w;
	ManifoldRec man = manifolds[m];
	u32 a = man.ids.x, b = man.ids.y, count = man.ids.z;
	float4 ca = cog[a], cb = cog[b];
	if (numRegions > 1) // XCD region of the manifold = slab of its first dynamic body; each body collects the regions it is touched from
	{
		float x = (a < nb) ? ca.x : cb.x;
		u32 region = 0;
		for (u32 i = 0; i < 7; ++i) region += (x >= __uint_as_float(counters[CTR_REGION_CUTS + i])) ? 1u : 0u;
		mRegion[s] = region;
		if (a < nb) atomicOr(&regMask[a], 1u << region);
		if (b < nb) atomicOr(&regMask[b], 1u << region);
	}
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

	for (u32 k = 0; k < count; ++k)
	{
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
		P[0 * rowCap] = make_float4(rA.x, rA.y, rA.z, rB.x);
		P[1 * rowCap] = make_float4(rB.y, rB.z, t.x, t.y);
		P[2 * rowCap] = make_float4(t.z, JnA.x, JnA.y, JnA.z);
		P[3 * rowCap] = make_float4(JtA.x, JtA.y, JtA.z, JnB.x);
		P[4 * rowCap] = make_float4(JnB.y, JnB.z, JtB.x, JtB.y);
		P[5 * rowCap] = make_float4(JtB.z, mN, mT, bias);
		rowLambda[(size_t)k * rowCap + s] = make_float2(0.f, 0.f);
	}
}

void launch_contact_init(World& w, u32 numPairs, float dt)
{
	if (!numPairs) return;
	w.mRegion.ensure(w.pairCap, w.stream); w.mRegionSorted.ensure(w.pairCap, w.stream); w.flowOrder.ensure(w.pairCap, w.stream);
	hipLaunchKernelGGL(k_contact_init, dim3((numPairs + 255) / 256), dim3(256), 0, w.stream, w.dCounters.p, dt, w.rowCap, w.mOrder.p, w.actIds.p, w.manifolds.p,
		w.cog.p, w.invIw.p, w.vel.p, w.rowPlanes.p, w.rowShared.p, w.rowLambda.p, w.rowIds.p, w.nb, flow_num_regions(w), w.regMask.p, w.mRegion.p);
}

// Region-major order of the schedule slots (stable: colour order is kept inside a region) + the region boundaries.
void launch_flow_regions(World& w, u32 numManifolds)
{
	if (flow_num_regions(w) <= 1 || !numManifolds) return;
	prim_sort_pairs_u32(w, w.mRegion.p, w.mRegionSorted.p, w.mIdx.p, w.flowOrder.p, numManifolds, 3);
	hipLaunchKernelGGL(k_region_offsets, dim3(1), dim3(64), 0, w.stream, w.dCounters.p, w.mRegionSorted.p, numManifolds);
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
// K11-flow: the SAME schedule (per body: its manifolds in colour order, iteration after iteration) executed as a dataflow instead
// of one launch per colour.  One resident lane per manifold; a lane may solve its manifold in iteration `it` as soon as both of
// its bodies have been handed over by their previous user.  The hand-over goes through a 64-byte record per body in global
// memory: six 64-bit words {fp32 value, turn}, written and polled with relaxed agent-scope atomics (coherent across the 8 XCDs;
// measured 0.65-0.8 us per hop against ~4.8 us for a dependent launch).  Turn numbering: a body with `d` manifolds is used
// d times per iteration; the manifold of colour c is its r-th user, r = popcount(bodyMask & ((1 << c) - 1)) — the colouring is
// proper, so every colour occurs at most once per body.  User r of iteration i waits for turn epoch + i*d + r and publishes
// epoch + i*d + r + 1.  The first user of a launch reads the body from `vel`, the last one writes it back there.  Every wait points
// to a strictly smaller (iteration, colour): no cycles, so with all lanes resident the kernel always drains; a spin limit and a
// device-wide abort flag make it drain even if residency is violated (result then invalid, the host falls back to the launch
// sweep).  Results are bit-identical to the launch-per-colour sweep: same per-body order, same arithmetic.
// ---------------------------------------------------------------------------------------------------------------
#define FLOW_SPIN_LIMIT (1u << 24) // polls before a lane gives up (seconds): only a true dead-lock (two persistent kernels sharing the GPU) gets here
#define FLOW_HOP_TICKS 50u // 0.5 us of the 100 MHz wall clock: expected time of one hand-over, paces the polling of far-away lanes

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// Body record: 64 B (one line per body); half 0 = {v.xyz, turn} at +0, half 1 = {w.xyz, turn} at +16.  Each half is ONE 16-byte
// sc1 (agent-scope, write-through) store and ONE 16-byte sc1 load, and carries its own turn tag, so no ordering between the halves
// and no release/acquire fence is needed (MI355X_MICROARCH.md, inter-workgroup visibility: tagged granules).
// A poll is issued for both bodies at once (all four 16-byte loads in flight together: one L2 round trip per polling trip) and
// evaluated afterwards.
struct FlowPoll { u32x4 h0, h1; };
// `eager`: fetch both halves with every poll (lowest latency; small and mid worlds).  Otherwise only the tagged first half is polled
// and the second one fetched after a match: half the L2 requests, which is what bounds the sweep beyond ~130k polling lanes.
MI_DEV void flowPollIssue(FlowPoll& p, __amdgpu_buffer_rsrc_t rsrc, u32 body, bool eager)
{
	asm volatile("" ::: "memory"); // a poll must be re-issued every time
	p.h0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, body * 64u, 0, 16);
	if (eager) p.h1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, body * 64u + 16u, 0, 16);
}
MI_DEV bool flowPollCheck(FlowPoll& p, __amdgpu_buffer_rsrc_t rsrc, u32 body, bool eager, u32 want, u32 rank, V3& v, V3& w, u32& behind)
{
	u32 d = want - p.h0.w;
	behind = (d > 4096u) ? rank : d;  // a stale record from an earlier launch: nobody has used the body in this launch yet
	behind = behind > 8u ? 8u : behind;
	if (p.h0.w != want) return false;
	if (!eager) p.h1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, body * 64u + 16u, 0, 16); // stored before half 0; re-polled if not there yet
	if (p.h1.w != want) return false;
	v = v3(__uint_as_float(p.h0.x), __uint_as_float(p.h0.y), __uint_as_float(p.h0.z));
	w = v3(__uint_as_float(p.h1.x), __uint_as_float(p.h1.y), __uint_as_float(p.h1.z));
	return true;
}
// `local`: every user of the body runs on this XCD, so the record may stay in this XCD's L2 (plain store; the polls bypass L1 and hit
// L2).  Otherwise write-through (sc1), the only store flavour another XCD's loads can observe.
MI_DEV void flowStore(__amdgpu_buffer_rsrc_t rsrc, u32 body, u32 turn, V3 v, V3 w, bool local)
{
	u32x4 h1 = { __float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), turn };
	u32x4 h0 = { __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), turn };
	if (local)
	{
		__builtin_amdgcn_raw_buffer_store_b128(h1, rsrc, body * 64u + 16u, 0, 0);
		__builtin_amdgcn_raw_buffer_store_b128(h0, rsrc, body * 64u, 0, 0);
	}
	else
	{
		__builtin_amdgcn_raw_buffer_store_b128(h1, rsrc, body * 64u + 16u, 0, 16);
		__builtin_amdgcn_raw_buffer_store_b128(h0, rsrc, body * 64u, 0, 16);
	}
}

// Everything about one manifold that does not change over the iterations of a launch.
struct FlowItem
{
	u32 s, a, b, count, degA, rankA, degB, rankB;
	bool da, db, localA, localB;
	float invMassA, invMassB, friction;
	V3 n;
	ContactRow r0; // first contact's row; lambda accumulates here
};

MI_DEV void flowItemLoad(FlowItem& m, u32 s, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes, const float4* __restrict__ rowShared, const float2* __restrict__ rowLambda,
	const uint4* __restrict__ rowIds, const u32* __restrict__ keySorted, const u64* __restrict__ bodyMask, const float4* vel, u32 numRegions, const u32* __restrict__ regMask, u64 colourMask)
{
	u32 key = keySorted[s];
	u32 color = key >> 2;
	uint4 ids = rowIds[s];
	float4 sh = rowShared[s];
	loadRow(m.r0, 0, s, rowCap, rowPlanes, rowLambda);
	m.s = s; m.count = 4u - (key & 3u); m.a = ids.x; m.b = ids.y;
	m.da = m.a < nb; m.db = m.b < nb;
	u64 lower = (1ull << color) - 1ull;
	u64 maskA = (m.da ? bodyMask[m.a] : 0ull) & colourMask, maskB = (m.db ? bodyMask[m.b] : 0ull) & colourMask; // users inside this launch's colour range
	m.degA = __popcll(maskA); m.rankA = __popcll(maskA & lower); m.degB = __popcll(maskB); m.rankB = __popcll(maskB & lower);
	m.invMassA = vel[2 * m.a].w; m.invMassB = vel[2 * m.b].w; // invMass rides in vel[2i].w and is constant
	m.localA = numRegions > 1 && m.da && __popc(regMask[m.a]) == 1; // touched from one region only = from this XCD only
	m.localB = numRegions > 1 && m.db && __popc(regMask[m.b]) == 1;
	m.n = v3(sh.x, sh.y, sh.z); m.friction = sh.w;
}

// One use of the manifold (iteration `it`): wait for both bodies, solve, hand both bodies on.  Returns the number of polls.
MI_DEV u32 flowTrip(FlowItem& m, u32 it, u32 itBegin, u32 itEnd, u32 epoch, __amdgpu_buffer_rsrc_t rsrc, size_t rowCap, const float4* __restrict__ rowPlanes,
	float2* __restrict__ rowLambda, float4* vel, u32* status, u32 hopTicks, u32 backoffCap, u64 notBefore, u64& readyAt, bool eager)
{
	const u32 a = m.a, b = m.b;
	u32 wantA = epoch + (it - itBegin) * m.degA + m.rankA, wantB = epoch + (it - itBegin) * m.degB + m.rankB;
	bool needA = m.da && !(it == itBegin && m.rankA == 0), needB = m.db && !(it == itBegin && m.rankB == 0);
	bool lastA = m.da && it + 1 == itEnd && m.rankA + 1 == m.degA, lastB = m.db && it + 1 == itEnd && m.rankB + 1 == m.degB;
	V3 vA = v3s(0.f), wA = v3s(0.f), vB = v3s(0.f), wB = v3s(0.f);
	if (!needA) { vA = v3f4(vel[2 * a]); wA = v3f4(vel[2 * a + 1]); } // first user of the launch (or the static dummy body)
	if (!needB) { vB = v3f4(vel[2 * b]); wB = v3f4(vel[2 * b + 1]); }

	bool done = false;
	u32 spins = 0, probes = 0;
	u64 nextA = notBefore, nextB = notBefore;
	u32 backA = 0, backB = 0;
	while (!done)
	{
		u64 now = wall_clock64();
		bool probed = false;
		bool pollA = needA && now >= nextA, pollB = needB && now >= nextB;
		FlowPoll pa, pb;
		if (pollA) flowPollIssue(pa, rsrc, a, eager);
		if (pollB) flowPollIssue(pb, rsrc, b, eager);
		if (pollA)
		{
			u32 behind;
			if (flowPollCheck(pa, rsrc, a, eager, wantA, m.rankA, vA, wA, behind)) needA = false;
			else
			{
				backA = backA ? (backA * 2u > backoffCap ? backoffCap : backA * 2u) : 8u;
				u32 wait = (behind > 1u) ? (behind - 1u) * hopTicks : 0u;
				nextA = now + (u64)(wait > backA ? wait : backA);
			}
			probed = true; ++probes;
		}
		if (pollB)
		{
			u32 behind;
			if (flowPollCheck(pb, rsrc, b, eager, wantB, m.rankB, vB, wB, behind)) needB = false;
			else
			{
				backB = backB ? (backB * 2u > backoffCap ? backoffCap : backB * 2u) : 8u;
				u32 wait = (behind > 1u) ? (behind - 1u) * hopTicks : 0u;
				nextB = now + (u64)(wait > backB ? wait : backB);
			}
			probed = true; ++probes;
		}
		if (!needA && !needB)
		{
			readyAt = now;
			solveRow(m.r0, m.n, m.friction, m.invMassA, m.invMassB, vA, wA, vB, wB);
			for (u32 k = 1; k < m.count; ++k) // further contacts (20 % of the manifolds have any) are streamed from L2 one by one
			{
				ContactRow cur;
				loadRow(cur, k, m.s, rowCap, rowPlanes, rowLambda);
				solveRow(cur, m.n, m.friction, m.invMassA, m.invMassB, vA, wA, vB, wB);
				rowLambda[(size_t)k * rowCap + m.s] = cur.lam;
			}
			if (m.da)
			{
				if (lastA) { vel[2 * a] = make_float4(vA.x, vA.y, vA.z, m.invMassA); vel[2 * a + 1] = make_float4(wA.x, wA.y, wA.z, 0.f); }
				else flowStore(rsrc, a, wantA + 1, vA, wA, m.localA);
			}
			if (m.db)
			{
				if (lastB) { vel[2 * b] = make_float4(vB.x, vB.y, vB.z, m.invMassB); vel[2 * b + 1] = make_float4(wB.x, wB.y, wB.z, 0.f); }
				else flowStore(rsrc, b, wantB + 1, vB, wB, m.localB);
			}
			done = true;
		}
		else if (probed)
		{
			++spins;
			if (spins > FLOW_SPIN_LIMIT) { atomicOr(status, 1u); done = true; }
			else if ((spins & 127u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) done = true;
		}
		if (!__any(probed)) __builtin_amdgcn_s_sleep(2); // every waiting lane of this wave is pacing itself: yield the issue slots
	}
	return probes;
}

MI_DEV u32 xccId() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu; } // HW_REG_XCC_ID: the XCD this wave runs on

// With regions (numRegions == 8) a workgroup first finds out which XCD it runs on and how many workgroups of this launch share that
// XCD (census: one atomic per workgroup, then everybody waits until all have registered — all are resident), and takes every
// nX-th 256-slot chunk of that XCD's region.  Placement is read from the hardware, not assumed from blockIdx.
// A lane that owns exactly one manifold keeps ids, ranks, masses and the first contact's row in registers for the whole launch;
// with more than one it strides iteration-major (slot-major would dead-lock: a lane's later slot can feed its earlier one).
// BLOCKS_PER_CU bounds the registers: 3 -> 168 VGPRs (a few spilled dwords outside the polling loop), 196k resident lanes; 2 -> no
// spills, 131k lanes.  Forcing 4 (128 VGPRs) spills INTO the polling loop: every poll then waits for a scratch reload, 2.5x slower.
template <int BLOCKS_PER_CU>
__global__ void __launch_bounds__(256, BLOCKS_PER_CU) k_solve_flow(u32* counters, u32 nb, size_t rowCap, const float4* __restrict__ rowPlanes,
	const float4* __restrict__ rowShared, float2* __restrict__ rowLambda, const uint4* __restrict__ rowIds, const u32* __restrict__ keySorted,
	const u64* __restrict__ bodyMask, float4* vel, u64* flow, u32 flowBytes, u32 epoch, u32 itBegin, u32 itEnd, u32 hopTicks, u32 backoffCap, u32 predictFrac,
	u32 numRegions, const u32* __restrict__ flowOrder, const u32* __restrict__ regMask, u64* __restrict__ trace, u32 firstColor, u32 eagerPolls)
{
	const bool eager = eagerPolls != 0u;
	u32* status = counters + CTR_FLOW_STATUS;
	u32* census = counters + CTR_FLOW_CENSUS;
	const u32 numM = counters[CTR_NUM_MANIFOLDS];
	const u32 slotBegin = counters[CTR_KEY_START + 4 * firstColor]; // colours below firstColor were swept by launches (hybrid sweep)
	const u64 colourMask = ~((1ull << firstColor) - 1ull);
	__amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(flow, 0, flowBytes, 0x00020000);
	__shared__ u32 sInfo[4];
	u32 first = slotBegin + blockIdx.x * 256u, stride = gridDim.x * 256u, begin = slotBegin, end = numM;
	if (numRegions > 1)
	{
		if (threadIdx.x == 0)
		{
			u32 xcc = xccId();
			u32 idx = atomicAdd(&census[xcc], 1u);
			atomicAdd(&census[8], 1u);
			u32 spins = 0;
			while (__hip_atomic_load(&census[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
			{
				__builtin_amdgcn_s_sleep(8);
				if (++spins > FLOW_SPIN_LIMIT) { atomicOr(status, 2u); break; }
			}
			u32 nX = __hip_atomic_load(&census[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (u32 r = 0; r < 8; ++r) // a region nobody serves would stall everyone: give up at once, the host falls back
				if (counters[CTR_REGION_START + r + 1] > counters[CTR_REGION_START + r] && __hip_atomic_load(&census[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(status, 4u);
			sInfo[0] = xcc; sInfo[1] = idx; sInfo[2] = nX ? nX : 1u;
		}
		__syncthreads();
		u32 xcc = sInfo[0];
		begin = counters[CTR_REGION_START + xcc]; end = counters[CTR_REGION_START + xcc + 1];
		first = begin + sInfo[1] * 256u; stride = sInfo[2] * 256u;
	}
	const bool single = (end - begin) <= stride;
	u32 probes = 0;
	// The host may have launched without looking at the schedule: manifolds in the serial bucket (more than 64 colours) have no rank
	// in their bodies' colour masks, so this kernel cannot run them — give up at once, the host redoes the step with launches.
	if (counters[CTR_KEY_START + 4 * MI_SERIAL_COLOR + 4] != counters[CTR_KEY_START + 4 * MI_SERIAL_COLOR]) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(status, 32u); return; }
	bool aborted = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
	if (!aborted)
	{
		if (single)
		{
			u32 pos = first + threadIdx.x;
			if (pos < end)
			{
				u32 slot = (numRegions > 1) ? flowOrder[pos] : pos;
				FlowItem m;
				flowItemLoad(m, slot, nb, rowCap, rowPlanes, rowShared, rowLambda, rowIds, keySorted, bodyMask, vel, numRegions, regMask, colourMask);
				// The sweep is periodic: a manifold's inputs arrive one iteration period after they arrived last time.  The lane sleeps
				// through the first predictFrac/256 of that period and polls only then (every poll is an L2 request; 100k lanes polling
				// all the time make a poll round take ~10 us).  The period is re-measured every iteration, so it can shrink and grow.
				u64 readyPrev = 0, readyNow = 0; u32 period = 0;
				for (u32 it = itBegin; it < itEnd; ++it)
				{
					u64 notBefore = (period && predictFrac) ? readyPrev + (((u64)period * predictFrac) >> 8) : 0ull;
					probes += flowTrip(m, it, itBegin, itEnd, epoch, rsrc, rowCap, rowPlanes, rowLambda, vel, status, hopTicks, backoffCap, notBefore, readyNow, eager);
					period = readyPrev ? (u32)(readyNow - readyPrev) : 0u;
					readyPrev = readyNow;
					if (trace) { trace[(size_t)slot * 32 + (it & 31u)] = readyNow; if (it == itBegin + 10u) trace[(size_t)slot * 32 + 31u] = notBefore; } // developer timeline: when this manifold saw its inputs complete in iteration it (per lane: the wave reconverges later); for iteration 10 also until when it slept
				}
				rowLambda[slot] = m.r0.lam;
			}
		}
		else
		{
			for (u32 it = itBegin; it < itEnd; ++it)
				for (u32 pos = first + threadIdx.x; pos < end; pos += stride)
				{
					u32 slot = (numRegions > 1) ? flowOrder[pos] : pos;
					FlowItem m;
					flowItemLoad(m, slot, nb, rowCap, rowPlanes, rowShared, rowLambda, rowIds, keySorted, bodyMask, vel, numRegions, regMask, colourMask);
					u64 readyNow;
					probes += flowTrip(m, it, itBegin, itEnd, epoch, rsrc, rowCap, rowPlanes, rowLambda, vel, status, hopTicks, backoffCap, 0ull, readyNow, eager);
					rowLambda[slot] = m.r0.lam;
				}
		}
	}
	for (int o = 32; o > 0; o >>= 1) probes += __shfl_xor(probes, o);
	if ((threadIdx.x & 63u) == 0u && probes) atomicAdd(status + 1, probes); // CTR_FLOW_PROBES: polling statistics
	if (numRegions > 1)
	{
		__syncthreads();
		if (threadIdx.x == 0 && atomicAdd(&census[9], 1u) == gridDim.x - 1u) // last workgroup out clears the census for the next launch
			for (u32 i = 0; i < 10; ++i) __hip_atomic_store(&census[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// Resident-lane budget of the dataflow kernel on this device (all lanes must be resident at once).
static u32 flowMaxBlocks(World& w, int variant) // variant 0: 3 blocks per CU, 1: 2 blocks per CU
{
	if (w.flowMaxBlocks[variant]) return w.flowMaxBlocks[variant];
	int perCU = 0, cus = 0;
	if (variant == 0) MI_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_solve_flow<3>, 256, 0));
	else MI_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_solve_flow<2>, 256, 0));
	MI_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w.device));
	w.flowMaxBlocks[variant] = (u32)std::max(1, perCU) * (u32)std::max(1, cus);
	return w.flowMaxBlocks[variant];
}

// Iterations [itBegin, itEnd) of the contact sweep in one launch.  numManifolds is the host's copy of CTR_NUM_MANIFOLDS.
void launch_solve_flow(World& w, u32 numManifolds, u32 itBegin, u32 itEnd, u32 firstColor)
{
	numManifolds -= std::min(numManifolds, w.hCounters[CTR_KEY_START + 4 * firstColor]); // manifolds of this launch
	if (!numManifolds || itBegin >= itEnd) return;
	size_t words = (size_t)(w.nb + 1) * 8;
	if (w.flow.cap < words) { w.flow.ensure(words, w.stream); w.flowEpoch = 0; }
	if (w.flowEpoch == 0 || w.flowEpoch >= 0xFFFFEu) // first use or turn counter about to wrap: no stale record may ever match
	{
		MI_CHECK(hipMemsetAsync(w.flow.p, 0, sizeof(u64) * (size_t)(w.nb + 1) * 8, w.stream));
		w.flowEpoch = 0;
	}
	w.flowEpoch++;
	if (w.flowTestAbortStep == w.stats.numInternalSteps) // tests: pretend a lane timed out; everybody drains without solving
	{
		u32 one = 16u;
		MI_CHECK(hipMemcpyAsync(w.dCounters.p + CTR_FLOW_STATUS, &one, sizeof(u32), hipMemcpyHostToDevice, w.stream));
		MI_CHECK(hipStreamSynchronize(w.stream));
	}
	u32 regions = firstColor ? 1u : flow_num_regions(w);
	// One lane per manifold whenever the 3-blocks-per-CU build can hold them all; beyond that lanes take several manifolds each and
	// the spill-free 2-blocks-per-CU build measured faster.  With regions every XCD must be able to hold its whole region: full grid.
	u32 need = (numManifolds + 255) / 256;
	int variant = (need <= flowMaxBlocks(w, 0)) ? 0 : 1;
	u32 blocks = (regions > 1) ? flowMaxBlocks(w, variant) : std::min<u32>(need, flowMaxBlocks(w, variant));
	auto kernel = variant == 0 ? k_solve_flow<3> : k_solve_flow<2>;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, w.stream, w.dCounters.p, w.nb, w.rowCap, w.rowPlanes.p, w.rowShared.p, w.rowLambda.p, w.rowIds.p,
		w.mKeySorted.p, w.bodyMask.p, w.vel.p, w.flow.p, (u32)(words * sizeof(u64)), w.flowEpoch << 12, itBegin, itEnd, numManifolds <= w.flowEagerMax ? w.flowHopTicks : w.flowHopTicksLarge, w.flowBackoffCap, w.flowPredictFrac,
		regions, w.flowOrder.p, w.regMask.p, w.flowTrace.p, firstColor, numManifolds <= w.flowEagerMax ? 1u : 0u);
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
