// Micro-benchmark (developer tool, not part of the product): where the cycles of ONE colour step of the cluster sweep go.
// A 512-lane workgroup per CU, bodies in LDS, one contact row per lane in registers (csrc/solver_rows.h, the product's row solve),
// a loop of colour steps separated by workgroup barriers, timed with s_memtime.  Variants take the step apart.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I../../directx-renderer-kurth_amd/csrc colorstep.hip -o colorstep
#include "solver_rows.h"
#include <cstdio>
#include <vector>
void mi_set_error(hipError_t, const char*, int) {}

// Candidate row solve: both row velocities from the velocities before the friction impulse, as three independent 3-term chains each;
// the normal row's velocity is then corrected by the friction impulse through the precomputed coupling cTN = Jn M^-1 Jt^T.
MI_DEV float rowVelocity3(V3 d, V3 cA, V3 cB, V3 vA, V3 wA, V3 vB, V3 wB)
{
	V3 dv = vB - vA;
	float s1 = dv.z * d.z; s1 = __builtin_fmaf(dv.y, d.y, s1); s1 = __builtin_fmaf(dv.x, d.x, s1);
	float s2 = wB.z * cB.z; s2 = __builtin_fmaf(wB.y, cB.y, s2); s2 = __builtin_fmaf(wB.x, cB.x, s2);
	float s3 = wA.z * cA.z; s3 = __builtin_fmaf(wA.y, cA.y, s3); s3 = __builtin_fmaf(wA.x, cA.x, s3);
	return (s1 + s2) - s3;
}
MI_DEV void solveRow2(ContactRow& r, V3 n, float friction, float invMassA, float invMassB, V3& vA, V3& wA, V3& vB, V3& wB, float cTN)
{
	V3 t = v3(r.p0.x, r.p0.y, r.p0.z), cAt = v3(r.p0.w, r.p1.x, r.p1.y), cBt = v3(r.p1.z, r.p1.w, r.p2.x);
	V3 cAn = v3(r.p2.y, r.p2.z, r.p2.w), cBn = v3(r.p3.x, r.p3.y, r.p3.z);
	V3 JtA = v3(r.p3.w, r.p4.x, r.p4.y), JtB = v3(r.p4.z, r.p4.w, r.p5.x), JnA = v3(r.p5.y, r.p5.z, r.p5.w), JnB = v3(r.p6.x, r.p6.y, r.p6.z);
	float mN = r.p6.w, mT = r.p7.x, bias = r.p7.y;
	float impulseN = r.lam.x, impulseT = r.lam.y;
	float vt = rowVelocity3(t, cAt, cBt, vA, wA, vB, wB);
	float vn = rowVelocity3(n, cAn, cBn, vA, wA, vB, wB);
	float maxFriction = friction * impulseN;
	float newT = __builtin_amdgcn_fmed3f(impulseT - mT * vt, -maxFriction, maxFriction);
	float dT = newT - impulseT;
	vn = __builtin_fmaf(dT, cTN, vn);
	float newN = fmaxf(impulseN - mN * (vn - bias), 0.f);
	float dN = newN - impulseN;
	rowApply(dT, t, JtA, JtB, invMassA, invMassB, vA, wA, vB, wB);
	rowApply(dN, n, JnA, JnB, invMassA, invMassB, vA, wA, vB, wB);
	r.lam = make_float2(newN, newT);
}
#ifdef ROW2
#define solveRow(R_, N_, F_, MA_, MB_, VA_, WA_, VB_, WB_) solveRow2(R_, N_, F_, MA_, MB_, VA_, WA_, VB_, WB_, 0.013f)
#endif
enum { V_FULL = 0, V_NOBARRIER = 1, V_ALLWAVES = 2, V_BARRIER_ONLY = 3, V_LDS_ONLY = 4, V_NO_LDS = 5, V_FOUR = 6, V_FOUR_PREFETCH = 7, V_QUAD_ONE = 8, V_QUAD_ALL = 9, V_LDS_R = 10, V_LDS_W = 11, V_LDS_1 = 12, V_QUAD_RANDOM = 13, V_QUAD_SHARED_ZERO = 14, V_COUNT = 15 };
static const char* names[V_COUNT] = { "full: one wave solves, 8 waves barrier", "same wave every step, wave fence instead of the barrier", "all 8 waves solve every step + barrier", "barrier only",
	"LDS round trip + barrier, no arithmetic", "arithmetic + barrier, bodies stay in registers", "4-contact manifold: rows 1-3 from LDS one after the other", "4-contact manifold: all LDS rows requested up front",
	"QUAD: 4 lanes per contact (one body vector each), one wave active", "QUAD: all 8 waves active", "LDS: 4 x read b128 + barrier", "LDS: 4 x write b128 + barrier", "LDS: 1 read + 1 write b128 + barrier",
	"QUAD, all waves, body vectors at pseudo-random LDS addresses (bank conflicts)", "QUAD, all waves, random addresses, a third of the quads share ONE static record (same-address writes)" };

MI_DEV void ldRow(ContactRow& r, const float4* lds, u32 off, u32 cap, u32 row)
{
	const float4* P = lds + off + row;
	r.p0 = P[0]; r.p1 = P[cap]; r.p2 = P[2 * cap]; r.p3 = P[3 * cap]; r.p4 = P[4 * cap]; r.p5 = P[5 * cap]; r.p6 = P[6 * cap];
	float4 q = P[7 * cap]; r.p7 = make_float2(q.x, q.y); r.lam = make_float2(q.z, q.w);
}


// QUAD candidate: four lanes per contact row.  Lane q owns one of the four body vectors x (vA, wA, vB, wB: one float4 of LDS) and
// its pieces of the row: dT / dN (what its vector is dotted with, signs folded in), aT / aN (what an impulse adds to it, inverse
// mass and signs folded in).  The row velocity is the sum of the four lanes' 3-term partial dots (two DPP adds inside the quad:
// every lane gets the bit-identical total), the impulse update is done redundantly by the four lanes, each then updates its vector.
struct QuadRow { V3 dT, aT, dN, aN; float mT, mN, bias, lamN, lamT; };
MI_DEV float quadSum(float p)
{
	float q = p + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, false)); // quad_perm [1,0,3,2]
	return q + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
}
MI_DEV void solveQuad(QuadRow& r, float friction, V3& x)
{
	float pt = x.z * r.dT.z; pt = __builtin_fmaf(x.y, r.dT.y, pt); pt = __builtin_fmaf(x.x, r.dT.x, pt);
	float vt = quadSum(pt);
	float maxFriction = friction * r.lamN;
	float newT = __builtin_amdgcn_fmed3f(__builtin_fmaf(-r.mT, vt, r.lamT), -maxFriction, maxFriction);
	float d = newT - r.lamT; r.lamT = newT;
	x = v3(__builtin_fmaf(d, r.aT.x, x.x), __builtin_fmaf(d, r.aT.y, x.y), __builtin_fmaf(d, r.aT.z, x.z));
	float pn = x.z * r.dN.z; pn = __builtin_fmaf(x.y, r.dN.y, pn); pn = __builtin_fmaf(x.x, r.dN.x, pn);
	float vn = quadSum(pn);
	float newN = fmaxf(__builtin_fmaf(-r.mN, vn - r.bias, r.lamN), 0.f);
	d = newN - r.lamN; r.lamN = newN;
	x = v3(__builtin_fmaf(d, r.aN.x, x.x), __builtin_fmaf(d, r.aN.y, x.y), __builtin_fmaf(d, r.aN.z, x.z));
}

template <int VAR> __global__ void __launch_bounds__(512) k_step(int steps, unsigned long long* out, float4* sink)
{
	extern __shared__ float4 lds[];
	const u32 tid = threadIdx.x, wave = tid >> 6;
	const u32 bodies = 2048, rowOff = 2 * bodies, rowCap = 768;
	for (u32 i = tid; i < 2 * bodies + 8 * rowCap; i += 512) lds[i] = make_float4(0.001f * (i & 255), 0.002f * (i & 127), -0.001f * (i & 63), 0.5f);
	ContactRow r;
	r.p0 = make_float4(0.1f, 0.2f, 0.3f, 0.01f * tid); r.p1 = make_float4(0.3f, 0.1f, 0.2f, 0.1f); r.p2 = make_float4(0.1f, 0.2f, 0.3f, 0.4f); r.p3 = make_float4(0.5f, 0.1f, 0.2f, 0.1f);
	r.p4 = make_float4(0.1f, 0.2f, 0.1f, 0.2f); r.p5 = make_float4(0.2f, 0.1f, 0.2f, 0.1f); r.p6 = make_float4(0.1f, 0.1f, 0.1f, 0.7f); r.p7 = make_float2(0.6f, 0.01f); r.lam = make_float2(0.f, 0.f);
	const u32 rdA = 2 * (2 * tid), rdB = 2 * (2 * tid + 1);
	float4 sh = make_float4(0.f, 1.f, 0.f, 0.5f);
	V3 n = v3(sh.x, sh.y, sh.z);
	V3 kvA = v3(0.f, 0.f, 0.f), kwA = kvA, kvB = kvA, kwB = kvA; // V_NO_LDS: the bodies
	QuadRow qr; qr.dT = v3(0.1f, 0.2f, 0.3f + 0.001f * tid); qr.aT = v3(0.3f, 0.1f, 0.2f); qr.dN = v3(0.f, 1.f, 0.01f * (tid & 3)); qr.aN = v3(0.1f, 0.5f, 0.2f); qr.mT = 0.6f; qr.mN = 0.7f; qr.bias = 0.01f; qr.lamN = 0.f; qr.lamT = 0.f;
	const u32 qAddr = 2 * tid; // lane q of quad tid / 4: one float4 of the quad's two bodies
	__syncthreads();
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int s = 0; s < steps; ++s)
	{
		bool mine = (VAR == V_ALLWAVES) ? true : (VAR == V_NOBARRIER ? wave == 0 : wave == (u32)(s & 7));
		if (VAR == V_BARRIER_ONLY) mine = false;
		if (VAR == V_QUAD_ALL || VAR == V_QUAD_RANDOM || VAR == V_QUAD_SHARED_ZERO) mine = true;
		if (VAR == V_QUAD_RANDOM || VAR == V_QUAD_SHARED_ZERO)
		{
			u32 h = (tid >> 1) * 2654435761u + (u32)s * 40503u; h ^= h >> 15; u32 a = ((h % 2000u) * 2u) + (tid & 1u);
			if (VAR == V_QUAD_SHARED_ZERO && ((tid >> 2) % 3u) == 0u && (tid & 2u)) a = 4000u + (tid & 1u);
			float4 b = lds[a]; V3 x = v3f4(b); solveQuad(qr, sh.w, x); lds[a] = make_float4(x.x, x.y, x.z, b.w);
		}
		else if (VAR == V_QUAD_ONE || VAR == V_QUAD_ALL)
		{
			if (mine) { float4 b = lds[qAddr]; V3 x = v3f4(b); solveQuad(qr, sh.w, x); lds[qAddr] = make_float4(x.x, x.y, x.z, b.w); }
		}
		else if (VAR == V_LDS_R) { if (mine) { float4 a0 = lds[rdA], a1 = lds[rdA + 1], b0 = lds[rdB], b1 = lds[rdB + 1]; kvA.x += a0.x + a1.y + b0.z + b1.x; } }
		else if (VAR == V_LDS_W) { if (mine) { float4 f = make_float4(kvA.x, kvA.y, 1.f, 2.f); lds[rdA] = f; lds[rdA + 1] = f; lds[rdB] = f; lds[rdB + 1] = f; } }
		else if (VAR == V_LDS_1) { if (mine) { float4 b = lds[qAddr]; b.x += 1.f; lds[qAddr] = b; } }
		else if (mine)
		{
			if (VAR == V_NO_LDS) solveRow(r, n, sh.w, 0.5f, 0.5f, kvA, kwA, kvB, kwB);
			else
			{
				float4 a0 = lds[rdA], a1 = lds[rdA + 1], b0 = lds[rdB], b1 = lds[rdB + 1];
				V3 vA = v3f4(a0), wA = v3f4(a1), vB = v3f4(b0), wB = v3f4(b1);
				if (VAR != V_LDS_ONLY) solveRow(r, n, sh.w, a0.w, b0.w, vA, wA, vB, wB);
				if (VAR == V_FOUR)
					for (u32 k = 0; k < 3; ++k)
					{
						ContactRow cur; ldRow(cur, lds, rowOff, rowCap, 3 * (tid & 255u) + k);
						solveRow(cur, n, sh.w, a0.w, b0.w, vA, wA, vB, wB);
						((float2*)(lds + rowOff + 7 * rowCap + 3 * (tid & 255u) + k))[1] = cur.lam;
					}
				if (VAR == V_FOUR_PREFETCH)
				{
					ContactRow c0, c1, c2; ldRow(c0, lds, rowOff, rowCap, 3 * (tid & 255u)); ldRow(c1, lds, rowOff, rowCap, 3 * (tid & 255u) + 1); ldRow(c2, lds, rowOff, rowCap, 3 * (tid & 255u) + 2);
					solveRow(c0, n, sh.w, a0.w, b0.w, vA, wA, vB, wB); solveRow(c1, n, sh.w, a0.w, b0.w, vA, wA, vB, wB); solveRow(c2, n, sh.w, a0.w, b0.w, vA, wA, vB, wB);
					((float2*)(lds + rowOff + 7 * rowCap + 3 * (tid & 255u)))[1] = c0.lam; ((float2*)(lds + rowOff + 7 * rowCap + 3 * (tid & 255u) + 1))[1] = c1.lam; ((float2*)(lds + rowOff + 7 * rowCap + 3 * (tid & 255u) + 2))[1] = c2.lam;
				}
				lds[rdA] = make_float4(vA.x, vA.y, vA.z, a0.w); lds[rdA + 1] = make_float4(wA.x, wA.y, wA.z, 0.f);
				lds[rdB] = make_float4(vB.x, vB.y, vB.z, b0.w); lds[rdB + 1] = make_float4(wB.x, wB.y, wB.z, 0.f);
			}
		}
		if (VAR == V_NOBARRIER) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads();
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (tid == 0) out[blockIdx.x] = t1 - t0;
	if (sink) { sink[blockIdx.x * 512 + tid] = make_float4(r.lam.x + kvA.x + qr.lamN, r.lam.y + kwB.y + qr.lamT, lds[rdA].x, lds[rdB + 1].y); }
}

template <int VAR> static void run(int blocks, int steps, unsigned long long* dOut, float4* dSink)
{
	size_t ldsBytes = 16 * (2 * 2048 + 8 * 768);
	(void)hipFuncSetAttribute((const void*)k_step<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
	std::vector<unsigned long long> h(blocks);
	for (int rep = 0; rep < 3; ++rep)
	{
		hipLaunchKernelGGL(k_step<VAR>, dim3(blocks), dim3(512), ldsBytes, 0, steps, dOut, dSink);
		hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return; }
	}
	(void)hipMemcpy(h.data(), dOut, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
	double sum = 0; for (auto v : h) sum += (double)v;
	printf("  %-62s %8.1f cycles per step (%d workgroups)\n", names[VAR], sum / blocks / steps, blocks);
}

int main(int argc, char** argv)
{
	int steps = 4096;
	unsigned long long* dOut; float4* dSink;
	(void)hipMalloc(&dOut, sizeof(unsigned long long) * 256); (void)hipMalloc(&dSink, sizeof(float4) * 256 * 512);
	for (int blocks : { 1, 256 })
	{
		run<V_FULL>(blocks, steps, dOut, dSink); run<V_NOBARRIER>(blocks, steps, dOut, dSink); run<V_ALLWAVES>(blocks, steps, dOut, dSink); run<V_BARRIER_ONLY>(blocks, steps, dOut, dSink);
		run<V_LDS_ONLY>(blocks, steps, dOut, dSink); run<V_NO_LDS>(blocks, steps, dOut, dSink); run<V_FOUR>(blocks, steps, dOut, dSink); run<V_FOUR_PREFETCH>(blocks, steps, dOut, dSink);
		run<V_QUAD_ONE>(blocks, steps, dOut, dSink); run<V_QUAD_ALL>(blocks, steps, dOut, dSink); run<V_LDS_R>(blocks, steps, dOut, dSink); run<V_LDS_W>(blocks, steps, dOut, dSink); run<V_LDS_1>(blocks, steps, dOut, dSink); run<V_QUAD_RANDOM>(blocks, steps, dOut, dSink); run<V_QUAD_SHARED_ZERO>(blocks, steps, dOut, dSink);
	}
	return 0;
}
