// Micro-benchmark (developer tool, not part of the product): latency of one producer->consumer hand-over between two workgroups
// through global memory with agent-scope relaxed atomics, same XCD vs different XCD, and between two waves through LDS.
// Build: hipcc -O3 --offload-arch=gfx950 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; } // HW_REG_XCC_ID

// words: 6 x u64 {payload32, turn32}.  Block A writes turn 2k+1, block B answers 2k+2.
template <int WORDS>
__global__ void k_pingpong(unsigned long long* rec, int partner, int rounds, long long* out, uint32_t* xcc, int limit)
{
	uint32_t me = blockIdx.x;
	if (threadIdx.x == 0) xcc[me] = xcc_id();
	if (me != 0 && me != (uint32_t)partner) return;
	if (threadIdx.x != 0) return;
	bool first = me == 0;
	long long t0 = wall_clock64();
	uint32_t turn = 0;
	bool failed = false;
	for (int r = 0; r < rounds && !failed; ++r)
	{
		if (first)
		{
			++turn; // 2r+1
			for (int w = 0; w < WORDS; ++w) __hip_atomic_store(rec + w, ((unsigned long long)turn << 32) | (uint32_t)(r + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			++turn; // wait for 2r+2
			int spin = 0;
			for (;;)
			{
				bool ok = true;
				for (int w = 0; w < WORDS; ++w) { unsigned long long v = __hip_atomic_load(rec + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && (uint32_t)(v >> 32) == turn; }
				if (ok) break;
				if (++spin > limit) { failed = true; break; }
			}
		}
		else
		{
			++turn; // wait for 2r+1
			int spin = 0;
			for (;;)
			{
				bool ok = true;
				for (int w = 0; w < WORDS; ++w) { unsigned long long v = __hip_atomic_load(rec + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = ok && (uint32_t)(v >> 32) == turn; }
				if (ok) break;
				if (++spin > limit) { failed = true; break; }
			}
			++turn;
			for (int w = 0; w < WORDS; ++w) __hip_atomic_store(rec + w, ((unsigned long long)turn << 32) | (uint32_t)(r + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	long long t1 = wall_clock64();
	if (first) { out[0] = t1 - t0; out[1] = failed ? 1 : 0; }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 32-byte record as two tagged 16-byte halves; STORE_AUX 0 = plain store (stays in this XCD's L2), 16 = sc1 (write-through); loads are sc1
template <int STORE_AUX>
__global__ void k_pingpong_b128(uint32_t* rec, uint32_t bytes, int partner, int rounds, long long* out, int limit)
{
	uint32_t me = blockIdx.x;
	if (me != 0 && me != (uint32_t)partner) return;
	if (threadIdx.x != 0) return;
	__amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, bytes, 0x00020000);
	bool first = me == 0;
	long long t0 = wall_clock64();
	uint32_t turn = 0; bool failed = false; uint32_t torn = 0;
	for (int r = 0; r < rounds && !failed; ++r)
	{
		for (int phase = 0; phase < 2; ++phase)
		{
			++turn;
			bool writer = (phase == 0) == first;
			if (writer)
			{
				u32x4 h1 = { turn * 3u, turn * 5u, turn * 7u, turn }, h0 = { turn * 11u, turn * 13u, turn * 17u, turn };
				__builtin_amdgcn_raw_buffer_store_b128(h1, rsrc, 16, 0, STORE_AUX);
				__builtin_amdgcn_raw_buffer_store_b128(h0, rsrc, 0, 0, STORE_AUX);
			}
			else
			{
				int spin = 0;
				for (;;)
				{
					asm volatile("" ::: "memory");
					u32x4 h0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 0, 0, 16), h1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 16, 0, 16);
					if (h0.w == turn && h1.w == turn) { if (h0.x != turn * 11u || h0.z != turn * 17u || h1.x != turn * 3u || h1.z != turn * 7u) ++torn; break; }
					if (++spin > limit) { failed = true; break; }
				}
			}
		}
	}
	long long t1 = wall_clock64();
	if (first) { out[0] = t1 - t0; out[1] = failed ? 1 : 0; out[2] = torn; }
}

// two waves of one workgroup through LDS
__global__ void k_pingpong_lds(int rounds, long long* out, int limit)
{
	__shared__ volatile uint32_t data[8];
	__shared__ volatile uint32_t turnW;
	if (threadIdx.x == 0) turnW = 0;
	__syncthreads();
	uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (lane != 0) return;
	long long t0 = wall_clock64();
	uint32_t turn = 0; bool failed = false;
	for (int r = 0; r < rounds && !failed; ++r)
	{
		if (wave == 0)
		{
			++turn; for (int w = 0; w < 6; ++w) data[w] = r + w; turnW = turn;
			++turn; int spin = 0; while (turnW != turn) { if (++spin > limit) { failed = true; break; } }
		}
		else
		{
			++turn; int spin = 0; while (turnW != turn) { if (++spin > limit) { failed = true; break; } }
			uint32_t s = 0; for (int w = 0; w < 6; ++w) s += data[w];
			++turn; data[0] = s; turnW = turn;
		}
	}
	long long t1 = wall_clock64();
	if (wave == 0) { out[0] = t1 - t0; out[1] = failed; }
}

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main()
{
	unsigned long long* rec; long long* out; uint32_t* xcc;
	CK(hipMalloc(&rec, 4096)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&xcc, 4096));
	const int rounds = 2000, limit = 2000000;
	long long h[2]; std::vector<uint32_t> hx(64);
	int freq = 0; CK(hipDeviceGetAttribute(&freq, hipDeviceAttributeWallClockRate, 0)); // kHz
	printf("wall clock rate %d kHz\n", freq);
	for (int partner : { 1, 2, 7, 8, 16, 9 })
	{
		CK(hipMemset(rec, 0, 4096));
		hipLaunchKernelGGL(k_pingpong<1>, dim3(32), dim3(64), 0, 0, rec, partner, rounds, out, xcc, limit);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcc, 32 * 4, hipMemcpyDeviceToHost));
		printf("1 word : block 0 (xcc %u) <-> block %2d (xcc %u): %.1f ns per one-way hop%s\n", hx[0], partner, hx[partner], (double)h[0] / freq * 1e6 / (2.0 * rounds), h[1] ? "  FAILED" : "");
		CK(hipMemset(rec, 0, 4096));
		hipLaunchKernelGGL(k_pingpong<6>, dim3(32), dim3(64), 0, 0, rec, partner, rounds, out, xcc, limit);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
		printf("6 words: block 0 (xcc %u) <-> block %2d (xcc %u): %.1f ns per one-way hop%s\n", hx[0], partner, hx[partner], (double)h[0] / freq * 1e6 / (2.0 * rounds), h[1] ? "  FAILED" : "");
	}
	for (int partner : { 1, 8, 16, 5 })
	{
		long long h3[3];
		CK(hipMemset(rec, 0, 4096));
		hipLaunchKernelGGL(k_pingpong_b128<16>, dim3(32), dim3(64), 0, 0, (uint32_t*)rec, 4096u, partner, rounds, out, limit);
		CK(hipDeviceSynchronize()); CK(hipMemcpy(h3, out, 24, hipMemcpyDeviceToHost));
		printf("2x16B sc1 store  : block 0 (xcc %u) <-> block %2d (xcc %u): %.1f ns per hop%s torn %lld\n", hx[0], partner, hx[partner], (double)h3[0] / freq * 1e6 / (2.0 * rounds), h3[1] ? "  FAILED" : "", h3[2]);
		CK(hipMemset(rec, 0, 4096));
		hipLaunchKernelGGL(k_pingpong_b128<0>, dim3(32), dim3(64), 0, 0, (uint32_t*)rec, 4096u, partner, rounds, out, 200000);
		CK(hipDeviceSynchronize()); CK(hipMemcpy(h3, out, 24, hipMemcpyDeviceToHost));
		printf("2x16B plain store: block 0 (xcc %u) <-> block %2d (xcc %u): %.1f ns per hop%s torn %lld\n", hx[0], partner, hx[partner], (double)h3[0] / freq * 1e6 / (2.0 * rounds), h3[1] ? "  FAILED (expected across XCDs)" : "", h3[2]);
	}
	hipLaunchKernelGGL(k_pingpong_lds, dim3(1), dim3(128), 0, 0, rounds, out, limit);
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
	printf("LDS, two waves of one workgroup: %.1f ns per one-way hop%s\n", (double)h[0] / freq * 1e6 / (2.0 * rounds), h[1] ? "  FAILED" : "");
	// dependent empty launches
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (int*)nullptr);
	CK(hipEventRecord(e0, 0));
	for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (int*)nullptr);
	CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	printf("empty dependent launches: %.2f us each\n", ms * 1e3 / 2000);
	return 0;
}
