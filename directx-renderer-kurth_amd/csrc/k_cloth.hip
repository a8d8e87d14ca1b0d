// Cloth (row N4 of SURVEY §8f) — reference src/physics/cloth.cpp:147-318, stepped after the rigid bodies (physics.cpp:1354-1358).
// One workgroup per cloth runs the whole step in one launch: wind + integration, the velocity / position / drift Gauss-Seidel
// iterations, damping.  Particles live in LDS (SoA: x, y, z of position and velocity + inverse mass, 28 B per particle) when the
// cloth fits, so an iteration touches HBM only for its read-only constraint records; larger cloths run the same code on their global
// planes.  The reference solves constraints in storage order; here they are sorted into 12 colours (constraint family x one parity
// bit of the grid coordinate) inside which no two constraints share a particle: a colour is solved in parallel, colours in sequence,
// which is the sequential sweep in colour order (the oracle restates exactly that order: oracle/ocloth.h).
#include "world.h"

#define GRAVITY -9.81f // reference physics.h:11
#define CLOTH_COLORS 12
#define CLOTH_BLOCK 512
struct ClothDesc { u32 firstParticle, numParticles, gridX, gridY, firstConstraint; u32 colorStart[CLOTH_COLORS + 1]; float gravityFactor, damping; };

MI_DEV V3 clothTriangleForce(V3 a, V3 b, V3 c, V3 force) // cloth.cpp:165-172
{
	V3 normal = cross(b - a, c - a);
	V3 f = normal * dot(normalize(normal), force);
	return f * (1.f / 3.f);
}

template <bool LDS>
__global__ void __launch_bounds__(CLOTH_BLOCK) k_cloth_simulate(const ClothDesc* __restrict__ descs, const u32* __restrict__ list, float* __restrict__ planes, size_t stride,
	const uint2* __restrict__ ab, const float2* __restrict__ restIms, float4* __restrict__ temp, float windX, float windY, float windZ,
	u32 velocityIterations, u32 positionIterations, u32 driftIterations, float dt)
{
	extern __shared__ float lds[];
	const ClothDesc d = descs[list[blockIdx.x]];
	const u32 n = d.numParticles, gx = d.gridX, gy = d.gridY, tid = threadIdx.x;
	float* gpx = planes + d.firstParticle; float* gpy = gpx + stride; float* gpz = gpy + stride;
	float* gvx = gpz + stride; float* gvy = gvx + stride; float* gvz = gvy + stride;
	float* qx = gvz + stride; float* qy = qx + stride; float* qz = qy + stride;           // prevPositions
	float* gim = qz + stride;
	float* px = LDS ? lds : gpx; float* py = LDS ? lds + n : gpy; float* pz = LDS ? lds + 2 * n : gpz;
	float* vx = LDS ? lds + 3 * n : gvx; float* vy = LDS ? lds + 4 * n : gvy; float* vz = LDS ? lds + 5 * n : gvz;
	const float* im = LDS ? lds + 6 * n : gim;
	const uint2* cab = ab + d.firstConstraint; const float2* crk = restIms + d.firstConstraint; float4* ctemp = temp + d.firstConstraint;
	const u32 numConstraints = d.colorStart[CLOTH_COLORS];

	for (u32 i = tid; i < n; i += CLOTH_BLOCK) // prevPosition = position (cloth.cpp:223)
	{
		qx[i] = gpx[i]; qy[i] = gpy[i]; qz[i] = gpz[i];
		if (LDS) { lds[3 * n + i] = gvx[i]; lds[4 * n + i] = gvy[i]; lds[5 * n + i] = gvz[i]; lds[6 * n + i] = gim[i]; }
	}
	__syncthreads();

	const V3 wind = v3(windX, windY, windZ);
	const float gravityVelocity = GRAVITY * dt * d.gravityFactor;
	for (u32 i = tid; i < n; i += CLOTH_BLOCK) // applyWindForce gathered per particle in the reference's quad order, then cloth.cpp:205-226
	{
		u32 x = i % gx, y = i / gx;
		auto P = [&](u32 xx, u32 yy) { u32 j = yy * gx + xx; return v3(qx[j], qy[j], qz[j]); };
		V3 F = v3s(0.f);
		if (x > 0 && y > 0) F += clothTriangleForce(P(x, y), P(x, y - 1), P(x - 1, y), wind);                                   // quad (x-1, y-1): this is its br
		if (x + 1 < gx && y > 0) { F += clothTriangleForce(P(x, y - 1), P(x, y), P(x + 1, y - 1), wind); F += clothTriangleForce(P(x + 1, y), P(x + 1, y - 1), P(x, y), wind); } // quad (x, y-1): bl
		if (x > 0 && y + 1 < gy) { F += clothTriangleForce(P(x - 1, y), P(x - 1, y + 1), P(x, y), wind); F += clothTriangleForce(P(x, y + 1), P(x, y), P(x - 1, y + 1), wind); } // quad (x-1, y): tr
		if (x + 1 < gx && y + 1 < gy) F += clothTriangleForce(P(x, y), P(x, y + 1), P(x + 1, y), wind);                         // quad (x, y): tl
		float invMass = im[i];
		V3 v = v3(vx[i], vy[i], vz[i]);
		if (invMass > 0.f) v.y += gravityVelocity;
		v += F * (invMass * dt);
		V3 p = v3(qx[i], qy[i], qz[i]);
		p += v * dt;
		px[i] = p.x; py[i] = p.y; pz[i] = p.z; vx[i] = v.x; vy[i] = v.y; vz[i] = v.z;
	}
	__syncthreads();
	const float invDt = (dt > 1e-5f) ? (1.f / dt) : 1.f;

	auto solvePositions = [&]() // cloth.cpp:301-318, colour by colour
	{
		for (u32 c = 0; c < CLOTH_COLORS; ++c)
		{
			for (u32 k = d.colorStart[c] + tid; k < d.colorStart[c + 1]; k += CLOTH_BLOCK)
			{
				uint2 e = cab[k]; float2 rk = crk[k];
				if (rk.y > 0.f)
				{
					V3 delta = v3(px[e.y], py[e.y], pz[e.y]) - v3(px[e.x], py[e.x], pz[e.x]);
					float len = sqlen(delta);
					float sqRest = rk.x * rk.x;
					if (sqRest + len > 1e-5f)
					{
						float kk = ((sqRest - len) / (rk.y * (sqRest + len)));
						V3 da = delta * (kk * im[e.x]), db = delta * (kk * im[e.y]);
						px[e.x] -= da.x; py[e.x] -= da.y; pz[e.x] -= da.z;
						px[e.y] += db.x; py[e.y] += db.y; pz[e.y] += db.z;
					}
				}
			}
			__syncthreads();
		}
	};

	if (velocityIterations > 0) // cloth.cpp:231-258
	{
		for (u32 k = tid; k < numConstraints; k += CLOTH_BLOCK)
		{
			uint2 e = cab[k];
			V3 g = v3(qx[e.y], qy[e.y], qz[e.y]) - v3(qx[e.x], qy[e.x], qz[e.x]);
			float ims = crk[k].y;
			ctemp[k] = make_float4(g.x, g.y, g.z, (ims == 0.f) ? 0.f : (1.f / (sqlen(g) * ims)));
		}
		__syncthreads();
		for (u32 it = 0; it < velocityIterations; ++it)
			for (u32 c = 0; c < CLOTH_COLORS; ++c)
			{
				for (u32 k = d.colorStart[c] + tid; k < d.colorStart[c + 1]; k += CLOTH_BLOCK) // cloth.cpp:289-299
				{
					uint2 e = cab[k]; float4 t = ctemp[k];
					V3 g = v3(t.x, t.y, t.z);
					float j = -dot(g, v3(vx[e.x], vy[e.x], vz[e.x]) - v3(vx[e.y], vy[e.y], vz[e.y])) * t.w;
					V3 da = g * (j * im[e.x]), db = g * (j * im[e.y]);
					vx[e.x] += da.x; vy[e.x] += da.y; vz[e.x] += da.z;
					vx[e.y] -= db.x; vy[e.y] -= db.y; vz[e.y] -= db.z;
				}
				__syncthreads();
			}
		for (u32 i = tid; i < n; i += CLOTH_BLOCK) { V3 p = v3(qx[i], qy[i], qz[i]) + v3(vx[i], vy[i], vz[i]) * dt; px[i] = p.x; py[i] = p.y; pz[i] = p.z; }
		__syncthreads();
	}
	if (positionIterations > 0) // cloth.cpp:261-272
	{
		for (u32 it = 0; it < positionIterations; ++it) solvePositions();
		for (u32 i = tid; i < n; i += CLOTH_BLOCK) { V3 v = (v3(px[i], py[i], pz[i]) - v3(qx[i], qy[i], qz[i])) * invDt; vx[i] = v.x; vy[i] = v.y; vz[i] = v.z; }
		__syncthreads();
	}
	if (driftIterations > 0) // cloth.cpp:275-291
	{
		for (u32 i = tid; i < n; i += CLOTH_BLOCK) { qx[i] = px[i]; qy[i] = py[i]; qz[i] = pz[i]; }
		__syncthreads();
		for (u32 it = 0; it < driftIterations; ++it) solvePositions();
		for (u32 i = tid; i < n; i += CLOTH_BLOCK) { V3 v = v3(vx[i], vy[i], vz[i]) + (v3(px[i], py[i], pz[i]) - v3(qx[i], qy[i], qz[i])) * invDt; vx[i] = v.x; vy[i] = v.y; vz[i] = v.z; }
		__syncthreads();
	}
	const float dampingFactor = 1.f / (1.f + dt * d.damping);
	for (u32 i = tid; i < n; i += CLOTH_BLOCK) // damping (cloth.cpp:294-298) + write-back
	{
		V3 v = v3(vx[i], vy[i], vz[i]) * dampingFactor;
		gvx[i] = v.x; gvy[i] = v.y; gvz[i] = v.z;
		if (LDS) { gpx[i] = px[i]; gpy[i] = py[i]; gpz[i] = pz[i]; }
	}
}

u32 cloth_lds_particle_limit() { return (64u * 1024u) / (7u * sizeof(float)); }

void launch_cloth(World& w, float dt)
{
	if (w.cloths.empty()) return;
	w.uploadCloths();
	const ClothDesc* descs = (const ClothDesc*)w.clothDescs.p;
	if (w.numSmallCloths)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cloth_simulate<true>), dim3(w.numSmallCloths), dim3(CLOTH_BLOCK), 7 * sizeof(float) * w.maxSmallClothParticles, w.stream, descs, w.clothList.p, w.clothPlanes.p,
			(size_t)w.clothStride, w.clothAB.p, w.clothRestIms.p, w.clothTemp.p, w.globalForce[0], w.globalForce[1], w.globalForce[2], w.clothIterations[0], w.clothIterations[1], w.clothIterations[2], dt);
	w.clothStateOnDevice = true; // the host mirror is stale from here on (World::downloadCloths)
	u32 numLarge = (u32)w.cloths.size() - w.numSmallCloths;
	if (numLarge)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cloth_simulate<false>), dim3(numLarge), dim3(CLOTH_BLOCK), 0, w.stream, descs, w.clothList.p + w.numSmallCloths, w.clothPlanes.p,
			(size_t)w.clothStride, w.clothAB.p, w.clothRestIms.p, w.clothTemp.p, w.globalForce[0], w.globalForce[1], w.globalForce[2], w.clothIterations[0], w.clothIterations[1], w.clothIterations[2], dt);
}
