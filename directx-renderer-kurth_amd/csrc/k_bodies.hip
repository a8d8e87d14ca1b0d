// Per-body and per-collider streaming kernels: world-space collider build (reference physics.cpp:631-756),
// applyGravityAndIntegrateForces (rigid_body.cpp:95-124), integrateVelocity (rigid_body.cpp:126-142) and the
// physics_transform0/1 copies + interpolation of physicsStep (physics.cpp:1375-1378, 1396-1402).
// All HBM-bound, one thread per element, float4 (16 B/lane) accesses.
#include "world.h"

#define GRAVITY -9.81f // reference physics.h:11

// ---------------------------------------------------------------------------------------------------------------
// K1: world-space colliders + AABBs.  Reads 64 B local collider + 32 B pose, writes 64 B world collider + 2x16 B AABB.
// ---------------------------------------------------------------------------------------------------------------
MI_DEV void growBox(V3& mn, V3& mx, V3 o) { mn = vmin(mn, o); mx = vmax(mx, o); }

// bounding_box::transformToAABB (bounding_volumes.cpp:58-70): 8 corners in this order.
MI_DEV void boxToAABB(V3 lo, V3 hi, Q4 rot, V3 tr, V3& mn, V3& mx)
{
	mn = v3s(MI_FLT_MAX); mx = v3s(-MI_FLT_MAX);
	growBox(mn, mx, rot * lo + tr);
	growBox(mn, mx, rot * v3(hi.x, lo.y, lo.z) + tr);
	growBox(mn, mx, rot * v3(lo.x, hi.y, lo.z) + tr);
	growBox(mn, mx, rot * v3(hi.x, hi.y, lo.z) + tr);
	growBox(mn, mx, rot * v3(lo.x, lo.y, hi.z) + tr);
	growBox(mn, mx, rot * v3(hi.x, lo.y, hi.z) + tr);
	growBox(mn, mx, rot * v3(lo.x, hi.y, hi.z) + tr);
	growBox(mn, mx, rot * hi + tr);
}

// Returns the largest AABB extent of the collider if it rides on a rigid body (the broadphase cell size is the maximum of those), else 0.
MI_DEV float buildCollider(u32 i, u32 nb, const ColliderRec* __restrict__ colLocal, const float4* __restrict__ pose,
	const float4* __restrict__ colStaticPose, const uint8_t* __restrict__ simMask, ColliderRec* __restrict__ colWorld, float4* __restrict__ aabbMin, float4* __restrict__ aabbMax,
	const float4* __restrict__ hullInfo)
{
	ColliderRec c = colLocal[i];
	u32 type = colType(c), body = colBody(c);
	if (body < nb && !simMask[body]) // body simulated by another GPU (spatial slabs): empty AABB, overlaps nothing
	{
		colWorld[i] = c;
		aabbMin[i] = make_float4(MI_FLT_MAX, MI_FLT_MAX, MI_FLT_MAX, 0.f);
		aabbMax[i] = make_float4(-MI_FLT_MAX, -MI_FLT_MAX, -MI_FLT_MAX, 0.f);
		return 0.f;
	}
	const float4* P = (body < nb) ? (pose + 2 * body) : (colStaticPose + 2 * i);
	V3 tpos = v3f4(P[0]);
	Q4 trot = q4f4(P[1]);
	V3 mn, mx;
	ColliderRec o = c;
	switch (type)
	{
		case MI_SPHERE:
		{
			V3 center = tpos + trot * v3(c.a.x, c.a.y, c.a.z);
			float r = c.a.w;
			mn = center - v3s(r); mx = center + v3s(r);
			o.a = make_float4(center.x, center.y, center.z, r);
		} break;
		case MI_CYLINDER: // tight extents, physics.cpp:699-720
		{
			V3 posA = trot * v3(c.a.x, c.a.y, c.a.z) + tpos;
			V3 posB = trot * v3(c.a.w, c.b.x, c.b.y) + tpos;
			float r = c.b.z;
			V3 a = posB - posA;
			float aa = dot(a, a);
			float x = 1.f - a.x * a.x / aa, y = 1.f - a.y * a.y / aa, z = 1.f - a.z * a.z / aa;
			x = sqrtf(fmaxf(0.f, x)); y = sqrtf(fmaxf(0.f, y)); z = sqrtf(fmaxf(0.f, z));
			V3 e = r * v3(x, y, z);
			mn = vmin(posA - e, posB - e); mx = vmax(posA + e, posB + e);
			o.a = make_float4(posA.x, posA.y, posA.z, posB.x);
			o.b = make_float4(posB.y, posB.z, r, 0.f);
		} break;
		case MI_CAPSULE:
		{
			V3 posA = trot * v3(c.a.x, c.a.y, c.a.z) + tpos;
			V3 posB = trot * v3(c.a.w, c.b.x, c.b.y) + tpos;
			float r = c.b.z;
			V3 r3 = v3s(r);
			mn = v3s(MI_FLT_MAX); mx = v3s(-MI_FLT_MAX);
			growBox(mn, mx, posA + r3); growBox(mn, mx, posA - r3); growBox(mn, mx, posB + r3); growBox(mn, mx, posB - r3);
			o.a = make_float4(posA.x, posA.y, posA.z, posB.x);
			o.b = make_float4(posB.y, posB.z, r, 0.f);
		} break;
		case MI_AABB:
		{
			V3 lo = v3(c.a.x, c.a.y, c.a.z), hi = v3(c.a.w, c.b.x, c.b.y);
			boxToAABB(lo, hi, trot, tpos, mn, mx);
			if (trot.x == 0.f && trot.y == 0.f && trot.z == 0.f && trot.w == 1.f)
			{
				o.a = make_float4(mn.x, mn.y, mn.z, mx.x);
				o.b = make_float4(mx.y, mx.z, 0.f, 0.f);
			}
			else // a rotated AABB becomes an OBB (physics.cpp:725-733)
			{
				V3 center = trot * ((lo + hi) * 0.5f) + tpos;
				V3 radius = (hi - lo) * 0.5f;
				o.a = make_float4(trot.x, trot.y, trot.z, trot.w);
				o.b = make_float4(center.x, center.y, center.z, radius.x);
				o.c.x = radius.y; o.c.y = radius.z;
				o.d.x = __uint_as_float((u32)MI_OBB);
			}
		} break;
		case MI_OBB:
		{
			Q4 q = q4f4(c.a);
			V3 center = v3(c.b.x, c.b.y, c.b.z);
			V3 radius = v3(c.b.w, c.c.x, c.c.y);
			Q4 wq = trot * q;                         // bounding_oriented_box::transformToAABB/OBB (bounding_volumes.cpp:127-142)
			V3 wc = trot * center + tpos;
			boxToAABB(-radius, radius, wq, wc, mn, mx);
			o.a = make_float4(wq.x, wq.y, wq.z, wq.w);
			o.b = make_float4(wc.x, wc.y, wc.z, radius.x);
		} break;
		case MI_HULL: // physics.cpp:742-753: compose the transforms, AABB = the geometry's local box transformed
		{
			Q4 hq = q4f4(c.a);
			V3 hp = v3(c.b.x, c.b.y, c.b.z);
			u32 g = (u32)c.b.w;
			Q4 rotation = trot * hq;
			V3 position = trot * hp + tpos;
			float4 lo = hullInfo[2 * g], hi = hullInfo[2 * g + 1];
			boxToAABB(v3f4(lo), v3f4(hi), rotation, position, mn, mx);
			o.a = make_float4(rotation.x, rotation.y, rotation.z, rotation.w);
			o.b = make_float4(position.x, position.y, position.z, c.b.w);
		} break;
		default: mn = v3s(0.f); mx = v3s(0.f); break;
	}
	colWorld[i] = o;
	aabbMin[i] = make_float4(mn.x, mn.y, mn.z, 0.f);
	aabbMax[i] = make_float4(mx.x, mx.y, mx.z, 0.f);
	return (body < nb) ? fmaxf(fmaxf(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z) : 0.f;
}

// One lane per collider; the kernel also produces the broadphase cell size (max extent; max() is order-independent, so the atomic is
// deterministic; CTR_CELL_SIZE was reset at the end of the previous broadphase), clears the grid's cell table, and leaves per
// workgroup the sums the reference's sweep picks its NEXT sorting axis from (sum of the AABB centres and of their squares,
// collision_broad.cpp:374-376; reduced in a fixed order by k_finish_pair_count, in double: the reference adds them up in float, one
// collider after the other, which no parallel sum reproduces bit for bit — the two can disagree on the axis only where two
// variances agree to within that float sum's rounding).
__global__ void __launch_bounds__(256) k_build_colliders(u32 nb, const u32* __restrict__ activeCols, const ColliderRec* __restrict__ colLocal, const float4* __restrict__ pose,
	const float4* __restrict__ colStaticPose, const uint8_t* __restrict__ simMask, ColliderRec* __restrict__ colWorld, float4* __restrict__ aabbMin, float4* __restrict__ aabbMax,
	u32* __restrict__ counters, u32* __restrict__ cellStart, u32* __restrict__ cellCount, u32 hashTableSize, const float4* __restrict__ hullInfo, double* __restrict__ sapPartial)
{
	const u32 gid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
	const u32 n = counters[CTR_ACTIVE_COLS]; // colliders of this world's simulated bodies + the static ones (k_active_*): the launch is sized from the last known count, the loop takes any
	float e = 0.f;
	double acc[7] = { 0., 0., 0., 0., 0., 0., 0. };
	for (u32 a = gid; a < n; a += stride)
	{
		const u32 i = activeCols[a];
		e = fmaxf(e, buildCollider(i, nb, colLocal, pose, colStaticPose, simMask, colWorld, aabbMin, aabbMax, hullInfo));
		float4 mn = aabbMin[i], mx = aabbMax[i];
		if (mn.x <= mx.x)
		{
			float cx = (mn.x + mx.x) * 0.5f, cy = (mn.y + mx.y) * 0.5f, cz = (mn.z + mx.z) * 0.5f; // bounding_box::getCenter
			acc[0] += cx; acc[1] += cy; acc[2] += cz; acc[3] += (double)cx * cx; acc[4] += (double)cy * cy; acc[5] += (double)cz * cz; acc[6] += 1.;
		}
	}
	for (int o = 32; o > 0; o >>= 1) { e = fmaxf(e, __shfl_xor(e, o)); for (int k = 0; k < 7; ++k) acc[k] += __shfl_xor(acc[k], o); }
	__shared__ u32 sMax; // one global atomic per workgroup (same-address atomics serialise)
	__shared__ double sAcc[4][7];
	if (threadIdx.x == 0) sMax = 0;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) { if (e > 0.f) atomicMax(&sMax, __float_as_uint(e)); for (int k = 0; k < 7; ++k) sAcc[threadIdx.x >> 6][k] = acc[k]; }
	__syncthreads();
	if (threadIdx.x == 0 && sMax) atomicMax(&counters[CTR_CELL_SIZE], sMax);
	if (threadIdx.x < 7) sapPartial[(size_t)blockIdx.x * 7 + threadIdx.x] = ((sAcc[0][threadIdx.x] + sAcc[1][threadIdx.x]) + sAcc[2][threadIdx.x]) + sAcc[3][threadIdx.x];
	for (u32 h = gid; h < hashTableSize + 3u; h += stride) { if (h < hashTableSize) cellStart[2 * h] = 0xFFFFFFFFu; cellCount[h] = 0u; } // (cellStart: {first, end} per bucket) // EMPTY_CELL; bucket sizes (+ the 'large' and 'simulated elsewhere' buckets)
}

u32 active_grid(u32 estimate, u32 total) { return (u32)((std::min<u64>(total, (u64)estimate + estimate / 8u + 2048u) + 255u) / 256u); } // workgroups of 256 for a kernel that strides over an active list

void launch_build_colliders(World& w)
{
	if (!w.nc) return;
	launch_active_lists(w);
	const u32 blocks = std::max(1u, active_grid(w.estActiveCols, w.nc));
	w.sapPartial.ensure((size_t)blocks * 7, w.stream);
	if (w.lastError) return;
	w.sapBlocks = blocks;
	hipLaunchKernelGGL(k_build_colliders, dim3(blocks), dim3(256), 0, w.stream, w.nb, w.actCols.p, w.colLocal.p, w.pose.p, w.colStaticPose.p,
		w.simMask.p, w.colWorld.p, w.aabbMin.p, w.aabbMax.p, w.dCounters.p, w.cellStart.p, w.cellCount.p, w.hashTableSize, w.hullInfo.p, w.sapPartial.p);
}

// ---------------------------------------------------------------------------------------------------------------
// Active lists.  A world may simulate a subset of its bodies (a spatial slab of a multi-GPU world simulates what it owns plus ghosts,
// a deleted body is switched off): the per-body and per-collider kernels of a step walk LISTS of the simulated bodies and of their
// colliders (+ the static colliders), so a slab's step costs what its own bodies cost, not what the whole world's would.  The lists
// are in ascending index order (so every later order, and with it every result, is the same as without them) and rebuilt when the
// simulate mask may have changed: 256 items per workgroup -> counts, one workgroup scans the counts, the first kernel's twin scatters.
// ---------------------------------------------------------------------------------------------------------------
MI_DEV bool activeItem(u32 item, u32 nb, u32 nc, const uint8_t* simMask, const u32* colBody)
{
	if (item < nb) return simMask[item] != 0;
	const u32 c = item - nb;
	if (c >= nc) return false;
	const u32 body = colBody[c];
	return body >= nb || simMask[body] != 0;
}
// items [0, nb) = bodies, [nbPad, nbPad + nc) = colliders (nbPad = nb rounded up to the workgroup size, so that a workgroup holds one kind)
__global__ void __launch_bounds__(256) k_active_count(u32 nb, u32 nbPad, u32 nc, const uint8_t* __restrict__ simMask, const u32* __restrict__ colBody, u32* __restrict__ blockCount)
{
	const u32 g = blockIdx.x * 256u + threadIdx.x;
	const bool on = g < nbPad ? (g < nb && simMask[g] != 0) : activeItem(nb + (g - nbPad), nb, nc, simMask, colBody);
	const u32 c = (u32)__syncthreads_count(on);
	if (threadIdx.x == 0) blockCount[blockIdx.x] = c;
}
__global__ void __launch_bounds__(1024) k_active_scan(u32 bodyBlocks, u32 totalBlocks, const u32* __restrict__ blockCount, u32* __restrict__ blockBase, u32* __restrict__ counters)
{
	__shared__ u32 part[1024];
	__shared__ u32 sBodies;
	const u32 t = threadIdx.x, per = (totalBlocks + 1023u) / 1024u;
	// two independent exclusive scans laid end to end: the body blocks, then the collider blocks
	u32 sum = 0;
	for (u32 k = 0; k < per; ++k) { u32 b = t * per + k; if (b < totalBlocks) sum += blockCount[b]; }
	part[t] = sum;
	__syncthreads();
	for (u32 o = 1; o < 1024u; o <<= 1) { u32 v = (t >= o) ? part[t - o] : 0u; __syncthreads(); part[t] += v; __syncthreads(); }
	u32 run = part[t] - sum;
	for (u32 k = 0; k < per; ++k) { u32 b = t * per + k; if (b < totalBlocks) { if (b == bodyBlocks) sBodies = run; blockBase[b] = run; run += blockCount[b]; } }
	__syncthreads();
	if (t == 1023u) { const u32 bodies = bodyBlocks < totalBlocks ? sBodies : part[1023]; counters[CTR_ACTIVE_BODIES] = bodies; counters[CTR_ACTIVE_COLS] = part[1023] - bodies; }
}
__global__ void __launch_bounds__(256) k_active_scatter(u32 nb, u32 nbPad, u32 nc, u32 bodyBlocks, const uint8_t* __restrict__ simMask, const u32* __restrict__ colBody, const u32* __restrict__ blockBase,
	u32* __restrict__ actBodies, u32* __restrict__ actCols)
{
	const u32 g = blockIdx.x * 256u + threadIdx.x;
	const bool isBody = g < nbPad;
	const bool on = isBody ? (g < nb && simMask[g] != 0) : activeItem(nb + (g - nbPad), nb, nc, simMask, colBody);
	__shared__ u32 waveCount[4];
	const u64 m = __ballot(on);
	const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	if (lane == 0) waveCount[wv] = (u32)__popcll(m);
	__syncthreads();
	u32 base = blockBase[blockIdx.x];
	for (u32 k = 0; k < wv; ++k) base += waveCount[k];
	if (!on) return;
	const u32 pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
	if (isBody) actBodies[pos] = g; else actCols[pos - blockBase[bodyBlocks]] = g - nbPad;
}
void launch_active_lists(World& w)
{
	if (!w.activeDirty) return;
	w.activeDirty = false;
	const u32 nb = w.nb, nc = w.nc, nbPad = (nb + 255u) / 256u * 256u, bodyBlocks = nbPad / 256u, total = bodyBlocks + (nc + 255u) / 256u;
	w.actBodies.ensure((size_t)nb + 1, w.stream); w.actCols.ensure((size_t)nc + 1, w.stream); w.actBlockCount.ensure(total + 1, w.stream); w.actBlockBase.ensure(total + 1, w.stream);
	if (w.lastError || !total) return;
	hipLaunchKernelGGL(k_active_count, dim3(total), dim3(256), 0, w.stream, nb, nbPad, nc, w.simMask.p, w.colBody.p, w.actBlockCount.p);
	hipLaunchKernelGGL(k_active_scan, dim3(1), dim3(1024), 0, w.stream, bodyBlocks, total, w.actBlockCount.p, w.actBlockBase.p, w.dCounters.p);
	hipLaunchKernelGGL(k_active_scatter, dim3(total), dim3(256), 0, w.stream, nb, nbPad, nc, bodyBlocks, w.simMask.p, w.colBody.p, w.actBlockBase.p, w.actBodies.p, w.actCols.p);
}

// ---------------------------------------------------------------------------------------------------------------
// K8: gravity + force integration, world inertia.  140 B read + 104 B write per body.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_integrate_forces(u32 nb, float dt, const u32* __restrict__ counters, const u32* __restrict__ actBodies, const float4* __restrict__ pose, const float4* __restrict__ bprops,
	const float4* __restrict__ force, float4* __restrict__ vel, float4* __restrict__ cog, float4* __restrict__ invIw,
	u64* __restrict__ bodyMask, u64* __restrict__ claim, float4* __restrict__ velBackup)
{
	const u32 n = counters[CTR_ACTIVE_BODIES];
	for (u32 a = blockIdx.x * blockDim.x + threadIdx.x; a <= n; a += gridDim.x * blockDim.x)
	{
		const u32 i = a == n ? nb : actBodies[a]; // the simulated bodies, then the static dummy
		if (bodyMask) { bodyMask[i] = 0ull; claim[i] = ~0ull; claim[(size_t)nb + i] = ~0ull; } // per-body state of the colouring that follows (saves three fill launches)
		if (i == nb) // static dummy (physics.cpp:1279)
		{
			float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
			vel[2 * i] = z; vel[2 * i + 1] = z; cog[i] = z; invIw[3 * i] = z; invIw[3 * i + 1] = z; invIw[3 * i + 2] = z;
			if (velBackup) { velBackup[2 * i] = z; velBackup[2 * i + 1] = z; }
			continue;
		}
		V3 pos = v3f4(pose[2 * i]);
		Q4 rot = q4f4(pose[2 * i + 1]);
		float4 p0 = bprops[5 * i], c0 = bprops[5 * i + 1], c1 = bprops[5 * i + 2], c2 = bprops[5 * i + 3], p4 = bprops[5 * i + 4];
		V3 localCOG = v3f4(p0); float invMass = p0.w;
		M3 I; I.m00 = c0.x; I.m10 = c0.y; I.m20 = c0.z; I.m01 = c1.x; I.m11 = c1.y; I.m21 = c1.z; I.m02 = c2.x; I.m12 = c2.y; I.m22 = c2.z;
		float gravityFactor = p4.x, linDamp = p4.y, angDamp = p4.z;

		V3 gpos = pos + rot * localCOG;
		M3 R = quaternionToMat3(rot);
		M3 Iw = R * I * mtranspose(R);

		V3 F = v3f4(force[2 * i]), T = v3f4(force[2 * i + 1]);
		if (invMass > 0.f) { F.y += (GRAVITY / invMass * gravityFactor); }
		V3 linAcc = F * invMass;
		V3 angAcc = Iw * T;
		float4 lv = vel[2 * i], av = vel[2 * i + 1];
		V3 v = v3f4(lv), wv = v3f4(av);
		v += linAcc * dt;
		wv += angAcc * dt;
		v *= 1.f / (1.f + dt * linDamp);
		wv *= 1.f / (1.f + dt * angDamp);

		vel[2 * i] = make_float4(v.x, v.y, v.z, invMass);
		vel[2 * i + 1] = make_float4(wv.x, wv.y, wv.z, 0.f);
		// the pre-solve velocities, kept in case the cluster sweep has to be redone (World::recoverFlow): written here, where every
		// simulated body's velocity passes through registers anyway (the bodies simulated elsewhere keep theirs: the restore walks the same list)
		if (velBackup) { velBackup[2 * i] = make_float4(v.x, v.y, v.z, invMass); velBackup[2 * i + 1] = make_float4(wv.x, wv.y, wv.z, 0.f); }
		cog[i] = make_float4(gpos.x, gpos.y, gpos.z, invMass);
		invIw[3 * i] = make_float4(Iw.m00, Iw.m10, Iw.m20, 0.f);
		invIw[3 * i + 1] = make_float4(Iw.m01, Iw.m11, Iw.m21, 0.f);
		invIw[3 * i + 2] = make_float4(Iw.m02, Iw.m12, Iw.m22, 0.f);
	}
}
__global__ void __launch_bounds__(256) k_restore_velocities(u32 nb, const u32* __restrict__ counters, const u32* __restrict__ actBodies, const float4* __restrict__ velBackup, float4* __restrict__ vel)
{
	const u32 n = counters[CTR_ACTIVE_BODIES];
	for (u32 a = blockIdx.x * blockDim.x + threadIdx.x; a <= n; a += gridDim.x * blockDim.x)
	{
		const u32 i = a == n ? nb : actBodies[a];
		vel[2 * i] = velBackup[2 * i]; vel[2 * i + 1] = velBackup[2 * i + 1];
	}
}
void launch_restore_velocities(World& w)
{
	hipLaunchKernelGGL(k_restore_velocities, dim3(std::max(1u, active_grid(w.estActiveBodies + 1, w.nb + 1))), dim3(256), 0, w.stream, w.nb, w.dCounters.p, w.actBodies.p, w.velBackup.p, w.vel.p);
}

void launch_integrate_forces(World& w, float dt)
{
	hipLaunchKernelGGL(k_integrate_forces, dim3(std::max(1u, active_grid(w.estActiveBodies + 1, w.nb + 1))), dim3(256), 0, w.stream, w.nb, dt, w.dCounters.p, w.actBodies.p, w.pose.p, w.bprops.p, w.force.p,
		w.vel.p, w.cog.p, w.invIw.p, w.bodyMask.p, w.claim.p, w.backupVelocities ? w.velBackup.p : nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// K13: velocity integration.  Reads cog(16) + vel(32) + rot(16) + localCOG(16), writes pose(32) + clears accumulators.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_integrate_velocities(u32 nb, float dt, const u32* __restrict__ counters, const u32* __restrict__ actBodies, float4* __restrict__ pose, const float4* __restrict__ bprops,
	const float4* __restrict__ vel, const float4* __restrict__ cog, float4* __restrict__ force, const u32* __restrict__ flowStatus)
{
	if (*flowStatus) return; // the cluster sweep gave up: velocities are invalid, the host redoes the solve and this integration (World::recoverFlow)
	const u32 n = counters[CTR_ACTIVE_BODIES];
	for (u32 a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x)
	{
		const u32 i = actBodies[a];
		Q4 grot = q4f4(pose[2 * i + 1]);
		V3 gpos = v3f4(cog[i]);
		V3 v = v3f4(vel[2 * i]), wv = v3f4(vel[2 * i + 1]);
		V3 localCOG = v3f4(bprops[5 * i]);
		Q4 deltaRot = q4(0.5f * wv.x, 0.5f * wv.y, 0.5f * wv.z, 0.f);
		deltaRot = deltaRot * grot;
		Q4 rotation = qnormalize(q4(grot.x + deltaRot.x * dt, grot.y + deltaRot.y * dt, grot.z + deltaRot.z * dt, grot.w + deltaRot.w * dt));
		V3 position = gpos + v * dt;
		V3 epos = position - rotation * localCOG;
		pose[2 * i] = make_float4(epos.x, epos.y, epos.z, 0.f);
		pose[2 * i + 1] = make_float4(rotation.x, rotation.y, rotation.z, rotation.w);
		float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
		force[2 * i] = z; force[2 * i + 1] = z;
	}
}

void launch_integrate_velocities(World& w, float dt)
{
	if (!w.nb) return;
	hipLaunchKernelGGL(k_integrate_velocities, dim3(std::max(1u, active_grid(w.estActiveBodies, w.nb))), dim3(256), 0, w.stream, w.nb, dt, w.dCounters.p, w.actBodies.p, w.pose.p, w.bprops.p, w.vel.p, w.cog.p, w.force.p, w.dCounters.p + CTR_FLOW_STATUS);
}

// ---------------------------------------------------------------------------------------------------------------
// physicsStep's transform0 <- transform1 copy and the lerp(transform0, transform1, t) write-back.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lerp_pose(u32 nb, float t, const float4* __restrict__ p0, const float4* __restrict__ p1, float4* __restrict__ out)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	float4 a = p0[2 * i], b = p1[2 * i];
	out[2 * i] = make_float4(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), 0.f);
	float4 l = p0[2 * i + 1], u = p1[2 * i + 1];
	Q4 q = qnormalize(q4(l.x + t * (u.x - l.x), l.y + t * (u.y - l.y), l.z + t * (u.z - l.z), l.w + t * (u.w - l.w)));
	out[2 * i + 1] = make_float4(q.x, q.y, q.z, q.w);
}

void launch_copy_pose0(World& w)
{
	if (!w.nb) return;
	MI_CHECK(hipMemcpyAsync(w.pose0.p, w.pose.p, sizeof(float4) * 2 * w.nb, hipMemcpyDeviceToDevice, w.stream));
}

void launch_lerp_pose(World& w, float t)
{
	if (!w.nb) return;
	hipLaunchKernelGGL(k_lerp_pose, dim3((w.nb + 255) / 256), dim3(256), 0, w.stream, w.nb, t, w.pose0.p, w.pose.p, w.poseLerp.p);
}

// simMask &= aliveMask (bodies deleted with mi_delete_body stay switched off under any caller-provided simulate mask)
__global__ void __launch_bounds__(256) k_and_mask(u32 nb, uint8_t* __restrict__ simMask, const uint8_t* __restrict__ aliveMask)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nb && !aliveMask[i]) simMask[i] = 0;
}
void launch_and_mask(World& w)
{
	if (w.nb) hipLaunchKernelGGL(k_and_mask, dim3((w.nb + 255) / 256), dim3(256), 0, w.stream, w.nb, w.simMask.p, w.aliveMask.p);
	w.activeDirty = true;
}

// ---------------------------------------------------------------------------------------------------------------
// Debug guard (MI_PHYSICS_VALIDATE=1): the reference's VALIDATE macros (physics.cpp:807-926, compiled out there with #if 0) print
// every NaN / Inf in the world-space colliders and boxes, the contacts, the body update records; here a kernel per stage counts
// them and remembers the first offender, and the step fails with MI_ERR_INVALID_STATE at its next synchronisation.
// ---------------------------------------------------------------------------------------------------------------
MI_DEV bool finite4(float4 v, bool w) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && (!w || isfinite(v.w)); }
// One element = `stride` float4 at a[i * stride]; the first `check` of them are tested, .w included for the first `checkW`.
__global__ void __launch_bounds__(256) k_validate(u32 stage, u32 n, const float4* __restrict__ a, u32 stride, u32 check, u32 checkW, const u32* __restrict__ countSrc, u32* __restrict__ counters)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (countSrc) n = min(n, *countSrc);
	if (i >= n) return;
	bool ok = true;
	for (u32 k = 0; k < check; ++k) ok = ok && finite4(a[(size_t)i * stride + k], k < checkW);
	if (!ok && atomicAdd(&counters[CTR_VALIDATE], 1u) == 0u) counters[CTR_VALIDATE + 1] = (stage << 28) | (i & 0x0FFFFFFFu);
}
void launch_validate(World& w, u32 stage, u32 numPairs)
{
	if (!w.validate) return;
	dim3 block(256);
	auto run = [&](u32 n, const void* p, u32 stride, u32 check, u32 checkW, const u32* countSrc) { if (n) hipLaunchKernelGGL(k_validate, dim3((n + 255) / 256), block, 0, w.stream, stage, n, (const float4*)p, stride, check, checkW, countSrc, w.dCounters.p); };
	if (stage == 0) // world-space colliders (4 float4, the last one holds integers) + boxes
	{
		run(w.nc, w.colWorld.p, 4, 3, 3, nullptr); run(w.nc, w.aabbMin.p, 1, 1, 0, nullptr); run(w.nc, w.aabbMax.p, 1, 1, 0, nullptr);
	}
	else if (stage == 1) run(numPairs, w.manifolds.p, 6, 5, 4, w.dCounters.p + CTR_NUM_VALID); // 4 contact points with depth, the normal (its .w holds packed material bits)
	else if (stage == 2) // rigid_body_global_state: centre of gravity + inverse mass, world inverse inertia, velocities
	{
		run(w.nb, w.cog.p, 1, 1, 1, nullptr); run(w.nb, w.invIw.p, 3, 3, 0, nullptr); run(w.nb, w.vel.p, 2, 2, 1, nullptr);
	}
	else if (stage == 3) { run(w.nb, w.pose.p, 2, 2, 2, nullptr); run(w.nb, w.vel.p, 2, 2, 1, nullptr); } // the step's result
}

// ---------------------------------------------------------------------------------------------------------------
// Spatial slabs: ghost-body halo on the device (mi_slab_*; design: DESIGN.md section 5).  Message = uint4 header {count, dropped, 0, 0}
// + records of 72 B {index, flag, pose 2 x float4, vel 2 x float4}, written as 18 dwords.
// ---------------------------------------------------------------------------------------------------------------
#define SLAB_RECORD_DWORDS 18u
MI_DEV float slabCoord(float4 p, u32 axis) { return axis == 0u ? p.x : (axis == 1u ? p.y : p.z); }

__global__ void __launch_bounds__(256) k_slab_classify(u32 nb, u32 axis, float lo, float hi, float margin, const float4* __restrict__ pose, const uint8_t* __restrict__ alive,
	uint8_t* __restrict__ code, uint8_t* __restrict__ simMask, u32* __restrict__ fresh)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	float x = slabCoord(pose[2 * i], axis);
	uint8_t c = MI_SLAB_INACTIVE;
	if (x >= lo && x < hi) c = MI_SLAB_OWNED;
	else if (x >= hi && x < hi + margin) c = MI_SLAB_GHOST_RIGHT; // what the neighbours would send in their first exchange
	else if (x < lo && x >= lo - margin) c = MI_SLAB_GHOST_LEFT;
	code[i] = c; fresh[i] = 0u;
	simMask[i] = (c != MI_SLAB_INACTIVE && alive[i]) ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_slab_pack(u32 nb, u32 axis, float lo, float hi, float margin, u32 capacity, u32 stamp, const float4* __restrict__ pose, const float4* __restrict__ vel,
	const uint8_t* __restrict__ alive, uint8_t* __restrict__ code, u32* __restrict__ fresh, u32* __restrict__ msgLeft, u32* __restrict__ msgRight)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb || code[i] != MI_SLAB_OWNED || !alive[i]) return;
	float x = slabCoord(pose[2 * i], axis);
	for (u32 side = 0; side < 2; ++side)
	{
		u32* msg = side ? msgRight : msgLeft;
		if (!msg) continue;
		bool band = side ? (x >= hi - margin) : (x < lo + margin);
		if (!band) continue;
		bool migrate = side ? (x >= hi) : (x < lo);
		u32 slot = atomicAdd(&msg[0], 1u);
		if (slot >= capacity) { atomicAdd(&msg[1], 1u); continue; } // dropped: the host sees it in the header it receives back / sends on
		u32* r = msg + 4u + slot * SLAB_RECORD_DWORDS;
		r[0] = i; r[1] = migrate ? MI_SLAB_MIGRATE : MI_SLAB_GHOST;
		float4 p0 = pose[2 * i], p1 = pose[2 * i + 1], v0 = vel[2 * i], v1 = vel[2 * i + 1];
		r[2] = __float_as_uint(p0.x); r[3] = __float_as_uint(p0.y); r[4] = __float_as_uint(p0.z); r[5] = __float_as_uint(p0.w);
		r[6] = __float_as_uint(p1.x); r[7] = __float_as_uint(p1.y); r[8] = __float_as_uint(p1.z); r[9] = __float_as_uint(p1.w);
		r[10] = __float_as_uint(v0.x); r[11] = __float_as_uint(v0.y); r[12] = __float_as_uint(v0.z); r[13] = __float_as_uint(v0.w);
		r[14] = __float_as_uint(v1.x); r[15] = __float_as_uint(v1.y); r[16] = __float_as_uint(v1.z); r[17] = __float_as_uint(v1.w);
		if (migrate) { code[i] = side ? MI_SLAB_GHOST_RIGHT : MI_SLAB_GHOST_LEFT; fresh[i] = stamp; } // its state here is the freshest there is: keep it as a ghost this step
	}
}

__global__ void __launch_bounds__(256) k_slab_apply(u32 nb, u32 capacity, u32 stamp, uint8_t ghostCode, const u32* __restrict__ msg, float4* __restrict__ pose, float4* __restrict__ pose0,
	float4* __restrict__ poseLerp, float4* __restrict__ vel, uint8_t* __restrict__ code, u32* __restrict__ fresh)
{
	u32 k = blockIdx.x * blockDim.x + threadIdx.x;
	u32 count = min(msg[0], capacity);
	if (k >= count) return;
	const u32* r = msg + 4u + k * SLAB_RECORD_DWORDS;
	u32 i = r[0];
	if (i >= nb) return; // a corrupt record must not become a wild write
	float4 p0 = make_float4(__uint_as_float(r[2]), __uint_as_float(r[3]), __uint_as_float(r[4]), __uint_as_float(r[5]));
	float4 p1 = make_float4(__uint_as_float(r[6]), __uint_as_float(r[7]), __uint_as_float(r[8]), __uint_as_float(r[9]));
	pose[2 * i] = p0; pose[2 * i + 1] = p1; pose0[2 * i] = p0; pose0[2 * i + 1] = p1; poseLerp[2 * i] = p0; poseLerp[2 * i + 1] = p1;
	vel[2 * i] = make_float4(__uint_as_float(r[10]), __uint_as_float(r[11]), __uint_as_float(r[12]), __uint_as_float(r[13]));
	vel[2 * i + 1] = make_float4(__uint_as_float(r[14]), __uint_as_float(r[15]), __uint_as_float(r[16]), __uint_as_float(r[17]));
	code[i] = (r[1] == MI_SLAB_MIGRATE) ? (uint8_t)MI_SLAB_OWNED : ghostCode;
	fresh[i] = stamp;
}

// Ghosts the neighbour did not send this step have left its band: inactive.  The simulate mask follows the codes.
__global__ void __launch_bounds__(256) k_slab_retire(u32 nb, u32 stamp, bool haveLeft, bool haveRight, const uint8_t* __restrict__ alive, uint8_t* __restrict__ code, const u32* __restrict__ fresh, uint8_t* __restrict__ simMask)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	uint8_t c = code[i];
	if (((c == MI_SLAB_GHOST_LEFT && haveLeft) || (c == MI_SLAB_GHOST_RIGHT && haveRight)) && fresh[i] != stamp) { c = MI_SLAB_INACTIVE; code[i] = c; }
	simMask[i] = (c != MI_SLAB_INACTIVE && alive[i]) ? 1 : 0;
}

void launch_slab_classify(World& w)
{
	if (!w.nb) return;
	hipLaunchKernelGGL(k_slab_classify, dim3((w.nb + 255) / 256), dim3(256), 0, w.stream, w.nb, w.slabAxis, w.slabLo, w.slabHi, w.slabMargin, w.pose.p, w.aliveMask.p, w.slabCode.p, w.simMask.p, w.slabFresh.p);
	w.activeDirty = true;
}
void launch_slab_pack(World& w, void* left, void* right, u32 capacity)
{
	if (!w.nb) return;
	if (left) MI_CHECK(hipMemsetAsync(left, 0, 16, w.stream));
	if (right) MI_CHECK(hipMemsetAsync(right, 0, 16, w.stream));
	hipLaunchKernelGGL(k_slab_pack, dim3((w.nb + 255) / 256), dim3(256), 0, w.stream, w.nb, w.slabAxis, w.slabLo, w.slabHi, w.slabMargin, capacity, w.slabStamp, w.pose.p, w.vel.p,
		w.aliveMask.p, w.slabCode.p, w.slabFresh.p, (u32*)left, (u32*)right);
}
void launch_slab_unpack(World& w, const void* left, const void* right, u32 capacity)
{
	if (!w.nb) return;
	dim3 grid((capacity + 255) / 256), block(256);
	if (left && capacity) hipLaunchKernelGGL(k_slab_apply, grid, block, 0, w.stream, w.nb, capacity, w.slabStamp, (uint8_t)MI_SLAB_GHOST_LEFT, (const u32*)left, w.pose.p, w.pose0.p, w.poseLerp.p, w.vel.p, w.slabCode.p, w.slabFresh.p);
	if (right && capacity) hipLaunchKernelGGL(k_slab_apply, grid, block, 0, w.stream, w.nb, capacity, w.slabStamp, (uint8_t)MI_SLAB_GHOST_RIGHT, (const u32*)right, w.pose.p, w.pose0.p, w.poseLerp.p, w.vel.p, w.slabCode.p, w.slabFresh.p);
	hipLaunchKernelGGL(k_slab_retire, dim3((w.nb + 255) / 256), block, 0, w.stream, w.nb, w.slabStamp, left != nullptr, right != nullptr, w.aliveMask.p, w.slabCode.p, w.slabFresh.p, w.simMask.p);
	w.activeDirty = true; // ghosts came and went
}
