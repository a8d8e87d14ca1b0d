// Broadphase: the set of collider pairs whose world AABBs overlap inclusively (reference aabbVsAABB,
// bounding_volumes.h:352-358) — the same SET the reference's sort-and-sweep (collision_broad.cpp:297-447) reports.
// MI355X design: a hashed uniform grid instead of a serial sweep.  Cell edge = largest extent of any collider attached to a
// rigid body (so an AABB spans at most 2 cells per axis and all partners live in the 27 neighbouring cells); colliders larger
// than a cell (static ground planes) go to a short "large" list that every collider tests directly.  Colliders are radix-sorted
// by cell hash (rocPRIM), cell ranges come from boundary detection, and pairs are produced by a count pass + exclusive scan +
// write pass so the pair list has a deterministic order without atomics.
#include "world.h"
#include <rocprim/rocprim.hpp>

#define CELL_BIAS (1 << 20)
#define CELL_MASK ((1u << 21) - 1u)
#define EMPTY_CELL 0xFFFFFFFFu

MI_DEV u64 packCell(i32 ix, i32 iy, i32 iz) { return ((u64)(u32)(ix & CELL_MASK)) | ((u64)(u32)(iy & CELL_MASK) << 21) | ((u64)(u32)(iz & CELL_MASK) << 42); }
MI_DEV u32 hashCell(u64 k, u32 mask)
{
	k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 27; k *= 0x94d049bb133111ebull; k ^= k >> 31; // splitmix64 finaliser
	return (u32)k & mask;
}
MI_DEV i32 cellCoord(float v, float invCell)
{
	float c = floorf(v * invCell);
	c = fminf(fmaxf(c, -(float)(CELL_BIAS - 2)), (float)(CELL_BIAS - 2));
	return (i32)c + CELL_BIAS;
}
// What a candidate in a visited bucket is checked against (buckets are shared by the cells that hash alike): the low 10 bits of the
// three cell coordinates.  Two different cells among a collider's 27 neighbours differ in these bits; a far cell that agrees in all
// of them is more than a thousand cells away and fails the box test.
MI_DEV u32 cellTag(i32 ix, i32 iy, i32 iz) { return ((u32)ix & 0x3FFu) | (((u32)iy & 0x3FFu) << 10) | (((u32)iz & 0x3FFu) << 20); }
MI_DEV bool aabbOverlap(float4 amin, float4 amax, float4 bmin, float4 bmax)
{
	if (amax.x < bmin.x || amin.x > bmax.x) return false;
	if (amax.y < bmin.y || amin.y > bmax.y) return false;
	if (amax.z < bmin.z || amin.z > bmax.z) return false;
	return true;
}

__global__ void __launch_bounds__(256) k_cell_assign(const u32* __restrict__ activeCols, u32 hashMask, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
	u32* __restrict__ counters, u32* __restrict__ hashKey, u32* __restrict__ cellCount)
{
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid == 0) { counters[CTR_FIRST_LARGE] = 0xFFFFFFFFu; counters[CTR_FIRST_INACTIVE] = 0xFFFFFFFFu; counters[CTR_PAIR_OVERFLOW] = 0u; } // "none" until k_cell_rank / k_pairs say otherwise
	const u32 n = counters[CTR_ACTIVE_COLS];
	float maxExtent = fmaxf(__uint_as_float(counters[CTR_CELL_SIZE]), 1e-3f);
	float cell = maxExtent * 1.001f;
	float invCell = 1.f / cell;
	for (u32 a = gid; a < n; a += gridDim.x * blockDim.x)
	{
		const u32 i = activeCols[a];
		float4 mn = aabbMin[i], mx = aabbMax[i];
		float e = fmaxf(fmaxf(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z);
		u32 h;
		if (mn.x > mx.x) { h = hashMask + 2; }    // empty AABB (body simulated elsewhere): sorts last, never visited
		else if (e > maxExtent) { h = hashMask + 1; } // large: sorts behind every grid cell
		else { h = hashCell(packCell(cellCoord(mn.x, invCell), cellCoord(mn.y, invCell), cellCoord(mn.z, invCell)), hashMask); }
		hashKey[i] = h;
		atomicAdd(&cellCount[h], 1u); // bucket sizes (cleared by k_build_colliders): the order by bucket is a counting sort, see k_cell_place
	}
}

// Order of the colliders by cell bucket = what a stable sort by hash key would give (bucket after bucket, inside a bucket by
// collider index), without a sort: bucket sizes (k_cell_assign) -> exclusive scan = first position of every bucket -> every collider
// takes a slot of its bucket in arrival order (atomic cursor: cellBase[h] ends up at the bucket's END) -> k_cell_rank puts the few
// colliders of a bucket in index order, so that the order, and with it the pair list, repeats from run to run.  (rocPRIM's merge
// sort of the 100k keys was 12 launches, 70 us per step.)  The bucket of the colliders simulated elsewhere keeps arrival order: nobody visits it.
__global__ void __launch_bounds__(256) k_cell_place(const u32* __restrict__ counters, const u32* __restrict__ activeCols, const u32* __restrict__ hashKey, u32* __restrict__ cellBase, u32* __restrict__ tmpIdx)
{
	const u32 n = counters[CTR_ACTIVE_COLS];
	for (u32 a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) { const u32 i = activeCols[a]; tmpIdx[atomicAdd(&cellBase[hashKey[i]], 1u)] = i; }
}
// ... and, knowing its position, writes the collider's sorted record there and, as the first of its bucket, the bucket's range
// (what a separate gather pass over the sorted indices did before: one launch and two index arrays fewer).
__global__ void __launch_bounds__(256) k_cell_rank(u32* __restrict__ counters, u32 hashMask, const u32* __restrict__ hashKey, const u32* __restrict__ cellBase, const u32* __restrict__ cellCount, const u32* __restrict__ tmpIdx,
	const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax, float4* __restrict__ sBox, u32* __restrict__ cellRange)
{
	const u32 n = counters[CTR_ACTIVE_COLS];
	const float cell = fmaxf(__uint_as_float(counters[CTR_CELL_SIZE]), 1e-3f) * 1.001f;
	const float invCell = 1.f / cell;
	if (blockIdx.x == 0 && threadIdx.x == 0) counters[CTR_CELL_SIZE_USED] = counters[CTR_CELL_SIZE]; // for the pair kernels (the second one runs after the reset)
	for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
	{
		const u32 i = tmpIdx[t], h = hashKey[i], end = cellBase[h], start = end - cellCount[h];
		u32 rank = t - start;
		if (h != hashMask + 2u) { rank = 0; for (u32 u = start; u < end; ++u) rank += tmpIdx[u] < i ? 1u : 0u; }
		const u32 pos = start + rank;
		float4 mn = aabbMin[i], mx = aabbMax[i];
		mx.w = __uint_as_float(cellTag(cellCoord(mn.x, invCell), cellCoord(mn.y, invCell), cellCoord(mn.z, invCell)));
		mn.w = __uint_as_float(i);
		sBox[2 * pos] = mn; sBox[2 * pos + 1] = mx; // one 32-byte record per sorted position: a candidate test is ONE cache line (three arrays were three)
		if (t == start) // (one thread per non-empty bucket)
		{
			if (h <= hashMask) { cellRange[2 * h] = start; cellRange[2 * h + 1] = end; }
			else counters[h == hashMask + 1 ? CTR_FIRST_LARGE : CTR_FIRST_INACTIVE] = start;
		}
	}
}

// Overlapping partners of the collider at sorted position t.  Every pair is produced exactly once, as (A = this collider,
// B = partner): partners in the 13 "forward" neighbour cells, partners sorted before t in the own cell, and every large collider.
// SIXTEEN lanes work on one collider: lane g < 14 visits neighbour cell 13 + g, lane 14 the list of large colliders (a large
// collider itself: lane 0 tests the large colliders before it).  Each lane counts its hits, a 16-lane prefix sum places them —
// cell after cell, candidate after candidate, exactly the order one lane visiting everything would produce — and a second visit
// writes them.  (One lane per collider left 6 waves per CU chasing 14 dependent hash -> range -> AABB chains each: 150 us at 100k.)
// MODE_SLAB  : the first PAIR_SLAB partners go to a per-collider slab, the full count to pairCount.
// MODE_WRITE : ONLY the colliders with more than PAIR_SLAB partners (the slab pass sets CTR_PAIR_OVERFLOW) repeat the visit,
//              writing directly at pairOffset[t]; same order, so the pair list is identical.  Everybody else is packed from the slabs.
#define PAIR_SLAB 32
#define PAIR_LANES 16
enum { MODE_SLAB = 0, MODE_WRITE = 1 };
template <int MODE>
__global__ void __launch_bounds__(256) k_pairs(u32 nc, u32 hashMask, const float4* __restrict__ sBox, const uint2* __restrict__ cellRange, u32* __restrict__ counters,
	u32* __restrict__ pairCount, const u32* __restrict__ pairOffset, uint2* __restrict__ out, u32 pairCap)
{
	u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
	u32 t = gid / PAIR_LANES, g = gid % PAIR_LANES;
	bool valid = t < nc;                                   // nc = the launch's bound on the sorted positions (>= the active colliders, or the host repeats the broadphase)
	u32 nEnd = min(counters[CTR_FIRST_INACTIVE], min(nc, counters[CTR_ACTIVE_COLS])); // colliders behind this position have empty AABBs
	u32 firstLarge = min(counters[CTR_FIRST_LARGE], nEnd);
	bool live = valid && t < nEnd;
	if (MODE == MODE_WRITE) live = live && pairCount[t] > PAIR_SLAB; // everybody else is complete in its slab
	float4 amin = make_float4(0.f, 0.f, 0.f, 0.f), amax = amin;
	u32 me = 0;
	// this lane's candidate range [s, e) and what a candidate must match
	u32 s = 0, e = 0; u32 ntag = 0; bool checkKey = false;
	if (live)
	{
		amin = sBox[2 * t]; amax = sBox[2 * t + 1];
		me = __float_as_uint(amin.w);
		if (t >= firstLarge) { if (g == 0) { s = firstLarge; e = t; } }
		else if (g == 14) { s = firstLarge; e = nEnd; }
		else if (g < 14)
		{
			const float invCell = 1.f / (fmaxf(__uint_as_float(counters[CTR_CELL_SIZE_USED]), 1e-3f) * 1.001f); // (as k_cell_rank computes it)
			i32 ix = cellCoord(amin.x, invCell), iy = cellCoord(amin.y, invCell), iz = cellCoord(amin.z, invCell);
			i32 o = 13 + (i32)g; // offsets (dz,dy,dx) >= (0,0,0) in lexicographic order: own cell first, then the forward half
			i32 dz = o / 9 - 1, dy = (o / 3) % 3 - 1, dx = o % 3 - 1;
			ntag = cellTag(ix + dx, iy + dy, iz + dz);
			u32 h = hashCell(packCell(ix + dx, iy + dy, iz + dz), hashMask);
			uint2 range = cellRange[h];
			if (range.x != EMPTY_CELL)
			{
				s = range.x; e = range.y;
				if (g == 0) e = min(e, t); // own cell: only partners sorted before me
				checkKey = true;           // other cells may share the hash bucket
			}
		}
	}
	u32 n = 0;
	u32 hit0 = 0, hit1 = 0, hit2 = 0, hit3 = 0; // the first four partners of this lane stay in registers: the write pass then needs no second visit
	// (four candidates per turn, their boxes requested together — eight cost more in registers than they save: a wave's time is the longest lane's chain of dependent loads —
	// 78 load instructions per wave, 82 % of its cycles parked on them, when every candidate waited for the one before it)
	for (u32 u = s; u < e; u += 4u)
	{
		float4 bmin[4], bmax[4];
#pragma unroll
		for (u32 k = 0; k < 4u; ++k) if (u + k < e) { bmin[k] = sBox[2 * (u + k)]; bmax[k] = sBox[2 * (u + k) + 1]; }
#pragma unroll
		for (u32 k = 0; k < 4u; ++k)
		{
			if (u + k >= e) continue;
			if (checkKey && __float_as_uint(bmax[k].w) != ntag) continue;
			if (aabbOverlap(amin, amax, bmin[k], bmax[k]))
			{
				u32 partner = __float_as_uint(bmin[k].w);
				if (n == 0) hit0 = partner; else if (n == 1) hit1 = partner; else if (n == 2) hit2 = partner; else if (n == 3) hit3 = partner;
				++n;
			}
		}
	}
	// exclusive prefix over the 16 lanes of the group (the groups of a wave are aligned to 16 lanes)
	u32 incl = n;
	for (u32 d = 1; d < PAIR_LANES; d <<= 1) { u32 v = __shfl_up(incl, d, PAIR_LANES); if (g >= d) incl += v; }
	u32 total = __shfl(incl, PAIR_LANES - 1, PAIR_LANES);
	u32 pos = incl - n;
	if (MODE == MODE_SLAB && valid && g == 0) { pairCount[t] = live ? total : 0; if (total > PAIR_SLAB) counters[CTR_PAIR_OVERFLOW] = 1; }
	if (!n) return;
	size_t base = (MODE == MODE_WRITE) ? (size_t)pairOffset[t] : (size_t)t * PAIR_SLAB;
	u32 room = (MODE == MODE_WRITE) ? 0xFFFFFFFFu : PAIR_SLAB;
	if (n <= 4)
	{
		for (u32 k = 0; k < n; ++k)
		{
			u32 partner = k == 0 ? hit0 : (k == 1 ? hit1 : (k == 2 ? hit2 : hit3));
			if (pos + k < room && (MODE == MODE_SLAB || base + pos + k < pairCap)) out[base + pos + k] = make_uint2(me, partner);
		}
		return;
	}
	for (u32 u = s; u < e; ++u)
	{
		float4 bmin = sBox[2 * u], bmax = sBox[2 * u + 1];
		if (checkKey && __float_as_uint(bmax.w) != ntag) continue;
		if (!aabbOverlap(amin, amax, bmin, bmax)) continue;
		if (pos < room && (MODE == MODE_SLAB || base + pos < pairCap)) out[base + pos] = make_uint2(me, __float_as_uint(bmin.w));
		++pos;
	}
}

// Packs the per-collider slabs into the dense, deterministic pair list (order: sorted position, then traversal order).
__global__ void __launch_bounds__(256) k_pairs_pack(u32 nc, const u32* __restrict__ pairCount, const u32* __restrict__ pairOffset, const uint2* __restrict__ slab,
	uint2* __restrict__ pairs, u32 pairCap)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x; // one lane per slab entry
	u32 t = i / PAIR_SLAB, k = i % PAIR_SLAB;
	if (t >= nc || k >= pairCount[t]) return;
	u32 dst = pairOffset[t] + k;
	if (dst < pairCap) pairs[dst] = slab[i];
}

// One wave.  Besides the pair count: the sorting axis the reference's sweep would use in the NEXT step = the axis of largest variance
// of this step's AABB centres (collision_broad.cpp:443-444: variance = s2 - s * s / numColliders; x over y, x over z, y over z on
// ties), from the per-workgroup sums of k_build_colliders added up in a fixed order.  It goes to the word of step + 1's parity, so
// that a step whose start is run twice (World::stepInternal after a recovered cluster sweep) leaves this step's own axis alone.
__global__ void k_finish_pair_count(u32 launched, const u32* __restrict__ pairCount, const u32* __restrict__ pairOffset, u32* __restrict__ counters, const double* __restrict__ sapPartial, u32 sapBlocks, u32 stepParity)
{
	double acc[7] = { 0., 0., 0., 0., 0., 0., 0. };
	for (u32 b = threadIdx.x; b < sapBlocks; b += 64u) for (int k = 0; k < 7; ++k) acc[k] += sapPartial[(size_t)b * 7 + k];
	for (int o = 32; o > 0; o >>= 1) for (int k = 0; k < 7; ++k) acc[k] += __shfl_xor(acc[k], o);
	if (threadIdx.x == 0 && blockIdx.x == 0)
	{
		const u32 na = counters[CTR_ACTIVE_COLS], nc = min(na, launched);
		counters[CTR_ACTIVE_OVERFLOW] = na > launched ? 1u : 0u; // more active colliders than the pair kernels and scans were launched for: the host repeats the broadphase
		counters[CTR_NUM_PAIRS] = nc ? pairOffset[nc - 1] + pairCount[nc - 1] : 0;
		counters[CTR_CELL_SIZE] = 0; // nobody reads the cell size after the pair traversal: ready for the next step's atomicMax
		const double n = acc[6] > 0. ? acc[6] : 1.;
		const double vx = acc[3] - acc[0] * acc[0] / n, vy = acc[4] - acc[1] * acc[1] / n, vz = acc[5] - acc[2] * acc[2] / n;
		counters[CTR_SAP_AXIS + (stepParity ^ 1u)] = (vx > vy) ? ((vx > vz) ? 0u : 2u) : ((vy > vz) ? 1u : 2u);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// rocPRIM wrappers (radix sort / exclusive scan).  Temp storage grows on demand, outside the steady state.
// ---------------------------------------------------------------------------------------------------------------
static void ensureTemp(World& w, size_t bytes)
{
	if (bytes > w.tempStorage.cap) w.tempStorage.ensure(bytes + bytes / 2 + 4096, w.stream);
}
// (Measured: forcing rocPRIM's one-sweep radix sort instead of the merge sort it picks below a million items is slower here: 100 k
// cell hashes of 18 bits, broadphase stage 0.33 ms against 0.29 ms.)
void prim_sort_pairs_u32(World& w, const u32* kin, u32* kout, const u32* vin, u32* vout, u32 n, u32 bits)
{
	if (!n) return;
	size_t bytes = 0;
	MI_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, w.stream));
	ensureTemp(w, bytes);
	MI_CHECK(rocprim::radix_sort_pairs(w.tempStorage.p, bytes, kin, kout, vin, vout, n, 0, bits, w.stream));
}
void prim_sort_pairs_u32_u64(World& w, const u32* kin, u32* kout, const u64* vin, u64* vout, u32 n, u32 bits)
{
	if (!n) return;
	size_t bytes = 0;
	MI_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, w.stream));
	ensureTemp(w, bytes);
	MI_CHECK(rocprim::radix_sort_pairs(w.tempStorage.p, bytes, kin, kout, vin, vout, n, 0, bits, w.stream));
}
void prim_exclusive_scan_u32(World& w, const u32* in, u32* out, u32 n)
{
	if (!n) return;
	size_t bytes = 0;
	MI_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n, rocprim::plus<u32>(), w.stream));
	ensureTemp(w, bytes);
	MI_CHECK(rocprim::exclusive_scan(w.tempStorage.p, bytes, in, out, 0u, n, rocprim::plus<u32>(), w.stream));
}

static u32 log2ceil(u32 v) { u32 b = 0; while ((1u << b) < v) ++b; return b; }

void launch_broadphase_count(World& w)
{
	if (!w.nc) return;
	// kernels that walk the active colliders stride over whatever the device's count is; the pair kernels, their slabs and the scan
	// are laid out for `bound` sorted positions (the last known count + 12 %): if the count has outgrown it the step repeats this
	// function (World::stepInternal looks at CTR_ACTIVE_OVERFLOW)
	const u32 bound = (u32)std::min<u64>(w.nc, (u64)w.estActiveCols + w.estActiveCols / 8u + 2048u);
	w.pairBound = bound;
	dim3 grid(std::max(1u, (bound + 255u) / 256u)), block(256);
	u32 H = w.hashTableSize, mask = H - 1;
	// cell size (max extent) and the cleared cell table come out of k_build_colliders
	hipLaunchKernelGGL(k_cell_assign, grid, block, 0, w.stream, w.actCols.p, mask, w.aabbMin.p, w.aabbMax.p, w.dCounters.p, w.hashKey.p, w.cellCount.p);
	prim_exclusive_scan_u32(w, w.cellCount.p, w.cellBase.p, H + 3);
	hipLaunchKernelGGL(k_cell_place, grid, block, 0, w.stream, w.dCounters.p, w.actCols.p, w.hashKey.p, w.cellBase.p, w.sortIdx.p);
	hipLaunchKernelGGL(k_cell_rank, grid, block, 0, w.stream, w.dCounters.p, mask, w.hashKey.p, w.cellBase.p, w.cellCount.p, w.sortIdx.p, w.aabbMin.p, w.aabbMax.p, w.sBox.p, w.cellStart.p);
	w.pairSlab.ensure((size_t)bound * PAIR_SLAB, w.stream);
	if (w.lastError) return;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pairs<MODE_SLAB>), dim3((u32)(((size_t)bound * PAIR_LANES + 255) / 256)), block, 0, w.stream, bound, mask, w.sBox.p, (const uint2*)w.cellStart.p, w.dCounters.p,
		w.pairCount.p, w.pairOffset.p, w.pairSlab.p, 0u);
	prim_exclusive_scan_u32(w, w.pairCount.p, w.pairOffset.p, bound);
	hipLaunchKernelGGL(k_finish_pair_count, dim3(1), dim3(64), 0, w.stream, bound, w.pairCount.p, w.pairOffset.p, w.dCounters.p, w.sapPartial.p, w.sapBlocks, w.stats.numInternalSteps & 1u);
}

void launch_broadphase_write(World& w, u32 numPairs, bool slabOverflow)
{
	const u32 nc = w.pairBound;
	if (!nc || !numPairs) return;
	hipLaunchKernelGGL(k_pairs_pack, dim3((u32)(((size_t)nc * PAIR_SLAB + 255) / 256)), dim3(256), 0, w.stream, nc, w.pairCount.p, w.pairOffset.p, w.pairSlab.p, w.pairs.p, (u32)w.pairCap);
	if (slabOverflow) // some colliders have more than PAIR_SLAB partners: those (only) repeat their traversal, writing in place
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pairs<MODE_WRITE>), dim3((u32)(((size_t)nc * PAIR_LANES + 255) / 256)), dim3(256), 0, w.stream, nc, w.hashTableSize - 1, w.sBox.p, (const uint2*)w.cellStart.p, w.dCounters.p, w.pairCount.p, w.pairOffset.p, w.pairs.p, (u32)w.pairCap);
}
