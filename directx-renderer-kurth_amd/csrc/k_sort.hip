// Stable counting sort of (small key, value) pairs — replaces rocPRIM's radix_sort_pairs for the two big few-bit sorts of a step
// (candidate pairs by narrowphase bucket: 6 bits; manifolds by schedule key: 9 bits).  At 500k items rocPRIM dispatches a merge
// sort: one block-sort plus ~10 merge launches, ~110 us per call (profiles/r01_*: k_sort_config / k_merge_config).  Here: one
// histogram launch, one scan, one scatter launch.
//
// A tile is 512 consecutive items and belongs to ONE wave, which walks it in 8 rounds of 64 items: stability needs no cross-wave
// ordering, only the rank of a lane among the lanes of its round holding the same key (ballot per key bit) and a per-wave LDS cursor
// per bucket.  Tile histograms are laid out bucket-major, so one exclusive scan over (bucket, tile) yields every tile's start offset
// in every bucket.
#include "world.h"
#include <rocprim/rocprim.hpp>

void prim_exclusive_scan_u32(World& w, const u32* in, u32* out, u32 n);

#define CSORT_TILE 512u
#define CSORT_WAVES 4u
#define CSORT_MAX_BUCKETS 272u

__global__ void __launch_bounds__(64 * CSORT_WAVES) k_csort_hist(const u32* __restrict__ keys, u32 n, u32 numBuckets, u32 numTiles, u32* __restrict__ tileHist)
{
	__shared__ u32 hist[CSORT_WAVES][CSORT_MAX_BUCKETS];
	u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	u32 tile = blockIdx.x * CSORT_WAVES + wave;
	u32* h = hist[wave];
	for (u32 b = lane; b < numBuckets; b += 64) h[b] = 0;
	__builtin_amdgcn_wave_barrier();
	if (tile < numTiles)
	{
		for (u32 r = 0; r < CSORT_TILE / 64; ++r)
		{
			u32 i = tile * CSORT_TILE + r * 64 + lane;
			if (i < n) atomicAdd(&h[min(keys[i], numBuckets - 1)], 1u);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	if (tile < numTiles)
		for (u32 b = lane; b < numBuckets; b += 64) tileHist[(size_t)b * numTiles + tile] = h[b];
}

template <typename V, int BITS>
__global__ void __launch_bounds__(64 * CSORT_WAVES) k_csort_scatter(const u32* __restrict__ keys, const V* __restrict__ vals, u32 n, u32 numBuckets, u32 numTiles,
	const u32* __restrict__ tileOffset, u32* __restrict__ keysOut, V* __restrict__ valsOut)
{
	__shared__ u32 cursor[CSORT_WAVES][CSORT_MAX_BUCKETS];
	u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	u32 tile = blockIdx.x * CSORT_WAVES + wave;
	if (tile >= numTiles) return;
	u32* cur = cursor[wave];
	for (u32 b = lane; b < numBuckets; b += 64) cur[b] = tileOffset[(size_t)b * numTiles + tile];
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	for (u32 r = 0; r < CSORT_TILE / 64; ++r)
	{
		u32 i = tile * CSORT_TILE + r * 64 + lane;
		bool valid = i < n;
		u32 key = valid ? min(keys[i], numBuckets - 1) : 0u;
		V val = valid ? vals[i] : V(0);
		u64 same = __ballot(valid);                 // lanes of this round holding my key
		for (int bit = 0; bit < BITS; ++bit)
		{
			bool set = (key >> bit) & 1u;
			u64 b = __ballot(set);
			same &= set ? b : ~b;
		}
		u32 rank = (u32)__popcll(same & ((1ull << lane) - 1ull));
		u32 base = valid ? cur[key] : 0u;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
		if (valid)
		{
			keysOut[base + rank] = keys[i];
			valsOut[base + rank] = val;
			if (rank + 1 == (u32)__popcll(same)) cur[key] = base + rank + 1; // last lane of the group moves the cursor
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	}
}

// keys must be < numBuckets <= 272 (larger keys are clamped into the last bucket but written back unchanged).
template <typename V>
static void csort(World& w, const u32* keys, u32* keysOut, const V* vals, V* valsOut, u32 n, u32 numBuckets)
{
	if (!n) return;
	u32 numTiles = (n + CSORT_TILE - 1) / CSORT_TILE;
	size_t cells = (size_t)numBuckets * numTiles;
	w.sortHist.ensure(2 * cells, w.stream);
	u32 blocks = (numTiles + CSORT_WAVES - 1) / CSORT_WAVES;
	hipLaunchKernelGGL(k_csort_hist, dim3(blocks), dim3(64 * CSORT_WAVES), 0, w.stream, keys, n, numBuckets, numTiles, w.sortHist.p);
	prim_exclusive_scan_u32(w, w.sortHist.p, w.sortHist.p + cells, (u32)cells);
	if (numBuckets <= 64)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_csort_scatter<V, 6>), dim3(blocks), dim3(64 * CSORT_WAVES), 0, w.stream, keys, vals, n, numBuckets, numTiles, w.sortHist.p + cells, keysOut, valsOut);
	else
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_csort_scatter<V, 9>), dim3(blocks), dim3(64 * CSORT_WAVES), 0, w.stream, keys, vals, n, numBuckets, numTiles, w.sortHist.p + cells, keysOut, valsOut);
}

void csort_pairs_u32(World& w, const u32* keys, u32* keysOut, const u32* vals, u32* valsOut, u32 n, u32 numBuckets) { csort<u32>(w, keys, keysOut, vals, valsOut, n, numBuckets); }
void csort_pairs_u64(World& w, const u32* keys, u32* keysOut, const u64* vals, u64* valsOut, u32 n, u32 numBuckets) { csort<u64>(w, keys, keysOut, vals, valsOut, n, numBuckets); }
