// Heightmap terrain collision (row N4 of SURVEY §8f) — reference src/terrain/heightmap_collider.h:36-206, heightmap_collider.cpp:22-153,
// src/physics/heightmap_collision.cpp:6-618, called after the narrowphase (physics.cpp:1236-1249).
// The reference walks a min/max mip pyramid per collider and appends any number of contacts per collider.  Here: one lane per
// rigid-body collider visits the cells under its (y-extended) AABB directly — a cell passes the pyramid exactly when the cell itself
// passes, so the triangle set is the same — in two passes (count, exclusive scan, write), and every terrain contact becomes a
// one-contact manifold appended after the pair manifolds, so the colouring, the contact rows and the solver see nothing new.
// Emission order per collider: chunk z, chunk x, cell z, cell x, triangle; the "lowest point under the terrain" contact last.
#include "world.h"
#include <rocprim/rocprim.hpp>

#define TERRAIN_VERTS 129u
#define TERRAIN_SLOT 0x80000000u
#define KEY_ZONE 62u

struct TerrainParams { u32 chunksPerDim; float chunkSize, invChunkSize, chunkScale, heightScale, invAmplitudeScale, minX, minY, minZ, friction, restitution; };
struct TerrainContact { V3 point, normal; float depth; };

MI_DEV V3 closestPointSegmentT(V3 q, V3 a, V3 b) { V3 ab = b - a; float t = dot(q - a, ab) / sqlen(ab); t = clampf(t, 0.f, 1.f); return a + t * ab; } // bounding_volumes.h:365-371
MI_DEV float closestSegmentSegmentT(V3 l1a, V3 l1b, V3 l2a, V3 l2b, V3& c1, V3& c2) // bounding_volumes.cpp:1251-1315
{
	float s, t;
	V3 d1 = l1b - l1a, d2 = l2b - l2a, r = l1a - l2a;
	float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
	if (a <= MI_EPSILON && e <= MI_EPSILON) { c1 = l1a; c2 = l2a; return dot(c1 - c2, c1 - c2); }
	if (a <= MI_EPSILON) { s = 0.f; t = f / e; t = clampf(t, 0.f, 1.f); }
	else
	{
		float c = dot(d1, r);
		if (e <= MI_EPSILON) { t = 0.f; s = clampf(-c / a, 0.f, 1.f); }
		else
		{
			float b = dot(d1, d2);
			float denom = a * e - b * b;
			if (denom != 0.f) s = clampf((b * f - c * e) / denom, 0.f, 1.f); else s = 0.f;
			t = (b * s + f) / e;
			if (t < 0.f) { t = 0.f; s = clampf(-c / a, 0.f, 1.f); }
			else if (t > 1.f) { t = 1.f; s = clampf((b - c) / a, 0.f, 1.f); }
		}
	}
	c1 = l1a + d1 * s; c2 = l2a + d2 * t;
	return dot(c1 - c2, c1 - c2);
}
MI_DEV V3 closestPointTriangle(V3 p, V3 a, V3 b, V3 c) // bounding_volumes.cpp:1317-1367
{
	V3 ab = b - a, ac = c - a, ap = p - a;
	float d1 = dot(ab, ap), d2 = dot(ac, ap);
	if (d1 <= 0.f && d2 <= 0.f) return a;
	V3 bp = p - b;
	float d3 = dot(ab, bp), d4 = dot(ac, bp);
	if (d3 >= 0.f && d4 <= d3) return b;
	float vc = d1 * d4 - d3 * d2;
	if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 / (d1 - d3); return a + v * ab; }
	V3 cp = p - c;
	float d5 = dot(ab, cp), d6 = dot(ac, cp);
	if (d6 >= 0.f && d5 <= d6) return c;
	float vb = d5 * d2 - d1 * d6;
	if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { float w = d2 / (d2 - d6); return a + w * ac; }
	float va = d3 * d6 - d5 * d4;
	if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { float w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); return b + w * (c - b); }
	float denom = 1.f / (va + vb + vc);
	float v = vb * denom, w = vc * denom;
	return a + ab * v + ac * w;
}
MI_DEV bool sphereTriangle(V3 center, float radius, V3 a, V3 b, V3 c, TerrainContact& out) // heightmap_collision.cpp:43-80
{
	V3 closestPoint = closestPointTriangle(center, a, b, c);
	V3 n = closestPoint - center;
	float sqDistance = sqlen(n);
	if (!(sqDistance <= radius * radius)) return false;
	float distance;
	if (sqDistance == 0.f) { V3 triNormal = cross(b - a, c - a); n = -triNormal; distance = 0.f; }
	else { distance = sqrtf(sqDistance); n = n * (1.f / distance); }
	out.point = closestPoint; out.normal = n; out.depth = radius - distance;
	return true;
}
MI_DEV void terrainIncidentEdge(V3 r, V3 normal, V3& outA, V3& outB) // heightmap_collision.cpp:6-41
{
	V3 p = v3(fabsf(normal.x), fabsf(normal.y), fabsf(normal.z));
	outA = r;
	if (p.x > p.y) outB = (p.y > p.z) ? v3(r.x, r.y, -r.z) : v3(r.x, -r.y, r.z);
	else outB = (p.x > p.z) ? v3(r.x, r.y, -r.z) : v3(-r.x, r.y, r.z);
	V3 s = v3(normal.x < 0.f ? -1.f : 1.f, normal.y < 0.f ? -1.f : 1.f, normal.z < 0.f ? -1.f : 1.f);
	outA = outA * s; outB = outB * s;
}
__device__ __noinline__ bool boxTriangle(V3 center, V3 radius, V3 a, V3 b, V3 c, TerrainContact& out) // heightmap_collision.cpp:82-423
{
	a = a - center; b = b - center; c = c - center;
	V3 f0 = b - a, f1 = c - b, f2 = a - c;
	float minPenetration = MI_FLT_MAX;
	V3 minNormal = v3s(0.f);
	u32 category = 0;
#define MI_TRI_AXIS(P0, P1, R, NX, NY, NZ, CAT) { float p0 = (P0), p1 = (P1); float r = (R); \
		float penetration = r - fmaxf(-fmaxf(p0, p1), fminf(p0, p1)); if (penetration < 0.f) return false; \
		V3 normal = v3(NX, NY, NZ); float l = length(normal); penetration *= 1.f / l; \
		if (penetration < minPenetration) { minPenetration = penetration; minNormal = normal * (1.f / l); category = CAT; } }
	MI_TRI_AXIS((a.z * f0.y) - (a.y * f0.z), (c.z * f0.y) - (c.y * f0.z), radius.y * fabsf(f0.z) + radius.z * fabsf(f0.y), 0.f, -f0.z, f0.y, 0)
	MI_TRI_AXIS((a.z * f1.y) - (a.y * f1.z), (b.z * f1.y) - (b.y * f1.z), radius.y * fabsf(f1.z) + radius.z * fabsf(f1.y), 0.f, -f1.z, f1.y, 1)
	MI_TRI_AXIS((a.z * f2.y) - (a.y * f2.z), (b.z * f2.y) - (b.y * f2.z), radius.y * fabsf(f2.z) + radius.z * fabsf(f2.y), 0.f, -f2.z, f2.y, 2)
	MI_TRI_AXIS((a.x * f0.z) - (a.z * f0.x), (c.x * f0.z) - (c.z * f0.x), radius.x * fabsf(f0.z) + radius.z * fabsf(f0.x), f0.z, 0.f, -f0.x, 0)
	MI_TRI_AXIS((a.x * f1.z) - (a.z * f1.x), (b.x * f1.z) - (b.z * f1.x), radius.x * fabsf(f1.z) + radius.z * fabsf(f1.x), f1.z, 0.f, -f1.x, 1)
	MI_TRI_AXIS((a.x * f2.z) - (a.z * f2.x), (b.x * f2.z) - (b.z * f2.x), radius.x * fabsf(f2.z) + radius.z * fabsf(f2.x), f2.z, 0.f, -f2.x, 2)
	MI_TRI_AXIS((a.y * f0.x) - (a.x * f0.y), (c.y * f0.x) - (c.x * f0.y), radius.x * fabsf(f0.y) + radius.y * fabsf(f0.x), -f0.y, f0.x, 0.f, 0)
	MI_TRI_AXIS((a.y * f1.x) - (a.x * f1.y), (b.y * f1.x) - (b.x * f1.y), radius.x * fabsf(f1.y) + radius.y * fabsf(f1.x), -f1.y, f1.x, 0.f, 1)
	MI_TRI_AXIS((a.y * f2.x) - (a.x * f2.y), (b.y * f2.x) - (b.x * f2.y), radius.x * fabsf(f2.y) + radius.y * fabsf(f2.x), -f2.y, f2.x, 0.f, 2)
#undef MI_TRI_AXIS
#define MI_BOX_FACE(PEN, NX, NY, NZ) { float penetration = (PEN); if (penetration < 0.f) return false; if (penetration < minPenetration) { minPenetration = penetration; minNormal = v3(NX, NY, NZ); category = 3; } }
	MI_BOX_FACE(fmaxf(a.x, fmaxf(b.x, c.x)) + radius.x, -1.f, 0.f, 0.f)
	MI_BOX_FACE(radius.x - fminf(a.x, fminf(b.x, c.x)), 1.f, 0.f, 0.f)
	MI_BOX_FACE(fmaxf(a.y, fmaxf(b.y, c.y)) + radius.y, 0.f, -1.f, 0.f)
	MI_BOX_FACE(radius.y - fminf(a.y, fminf(b.y, c.y)), 0.f, 1.f, 0.f)
	MI_BOX_FACE(fmaxf(a.z, fmaxf(b.z, c.z)) + radius.z, 0.f, 0.f, -1.f)
	MI_BOX_FACE(radius.z - fminf(a.z, fminf(b.z, c.z)), 0.f, 0.f, 1.f)
#undef MI_BOX_FACE
	{
		V3 triNormal = normalize(cross(f0, f1));
		float triD = dot(triNormal, a);
		float r = dot(radius, v3(fabsf(triNormal.x), fabsf(triNormal.y), fabsf(triNormal.z)));
		float penetration = r - fabsf(triD);
		if (penetration < 0.f) return false;
		if (penetration < minPenetration) { minPenetration = penetration; minNormal = triNormal; category = 4; }
	}
	V3 triCenter = (a + b + c) * (1.f / 3.f);
	if (dot(minNormal, triCenter) < 0.f) minNormal = -minNormal;
	V3 point;
	if (category < 3)
	{
		V3 a0, a1;
		terrainIncidentEdge(radius, minNormal, a0, a1);
		V3 triA = (category == 0) ? a : ((category == 1) ? b : c);
		V3 triB = (category == 0) ? b : ((category == 1) ? c : a);
		V3 pa, pb;
		closestSegmentSegmentT(a0, a1, triA, triB, pa, pb);
		point = (pa + pb) * 0.5f;
	}
	else if (category == 3)
	{
		float da = dot(minNormal, a), db = dot(minNormal, b), dc = dot(minNormal, c);
		V3 p = (da < db) ? ((da < dc) ? a : c) : ((db < dc) ? b : c);
		point = p + minNormal * (minPenetration * 0.5f);
	}
	else
	{
		V3 p = v3((minNormal.x < 0.f) ? -radius.x : radius.x, (minNormal.y < 0.f) ? -radius.y : radius.y, (minNormal.z < 0.f) ? -radius.z : radius.z);
		point = p - minNormal * (minPenetration * 0.5f);
	}
	out.point = point + center; out.normal = minNormal; out.depth = minPenetration;
	return true;
}

// One collider type per kernel instantiation (TYPE = MI_SPHERE / MI_CAPSULE / MI_AABB / MI_OBB): the shape lives in plain locals p0, p1, r, q, dir —
// sphere: p0 centre, r | capsule: p0, p1 ends, r, dir | aabb: p0 centre, p1 radius | obb: q, p0 centre, p1 radius.
template <int TYPE>
MI_DEV bool terrainTriangle(V3 p0, V3 p1, float r, Q4 q, V3 dir, V3 a, V3 b, V3 c, TerrainContact& out)
{
	if (TYPE == MI_SPHERE) return sphereTriangle(p0, r, a, b, c, out);
	if (TYPE == MI_CAPSULE) // heightmap_collision.cpp:449-470
	{
		V3 triNormal = normalize(cross(b - a, c - a));
		float d = -dot(triNormal, a);
		float ndotd = dot(dir, triNormal);
		float t = -(dot(p0, triNormal) + d) / ndotd;
		V3 trace = p0 + t * dir;
		V3 closest = closestPointTriangle(trace, a, b, c);
		V3 reference = closestPointSegmentT(closest, p0, p1);
		return sphereTriangle(reference, r, a, b, c, out);
	}
	if (TYPE == MI_AABB) return boxTriangle(p0, p1, a, b, c, out);
	// obb, heightmap_collision.cpp:492-515
	a = conjugate(q) * (a - p0); b = conjugate(q) * (b - p0); c = conjugate(q) * (c - p0);
	if (!boxTriangle(v3s(0.f), p1, a, b, c, out)) return false;
	out.normal = q * out.normal; out.point = q * out.point + p0;
	return true;
}

MI_DEV float terrainHeightAt(const TerrainParams& P, const uint16_t* __restrict__ heights, const u32* __restrict__ valid, float wx, float wz) // heightmap_collider.cpp:22-38, 123-153
{
	float cx = (wx - P.minX) * P.invChunkSize, cz = (wz - P.minZ) * P.invChunkSize;
	if (cx < 0.f || cz < 0.f || cx >= P.chunksPerDim || cz >= P.chunksPerDim) return -MI_FLT_MAX;
	u32 chunkX = (u32)cx, chunkZ = (u32)cz;
	u32 chunk = chunkZ * P.chunksPerDim + chunkX;
	if (!valid[chunk]) return -MI_FLT_MAX;
	cx = fmodf(cx, 1.f) * (TERRAIN_VERTS - 1); cz = fmodf(cz, 1.f) * (TERRAIN_VERTS - 1);
	u32 x = (u32)cx, z = (u32)cz;
	float relX = cx - x, relZ = cz - z;
	const uint16_t* H = heights + (size_t)chunk * TERRAIN_VERTS * TERRAIN_VERTS;
	float a = H[TERRAIN_VERTS * z + x] * P.heightScale, b = H[TERRAIN_VERTS * (z + 1) + x] * P.heightScale;
	float c = H[TERRAIN_VERTS * z + x + 1] * P.heightScale, d = H[TERRAIN_VERTS * (z + 1) + x + 1] * P.heightScale;
	float l0 = a + relX * (c - a), l1 = b + relX * (d - b);
	return (l0 + relZ * (l1 - l0)) + P.minY;
}

// write = false: counts[i] = number of terrain contacts of collider i (0 for other types).  write = true: their manifolds at slot base + offsets[i] + k.
// Both passes run the same instantiation (a runtime flag): the contact arithmetic must be the same machine code in the pass that counts and
// the pass that writes.
template <int TYPE>
__global__ void __launch_bounds__(64) k_heightmap(const bool write, u32 nc, u32 nb, const ColliderRec* __restrict__ colWorld, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
	const uint8_t* __restrict__ simMask, TerrainParams P, const uint16_t* __restrict__ heights, const u32* __restrict__ valid, u32* __restrict__ counts, const u32* __restrict__ offsets,
	u32* __restrict__ counters, ManifoldRec* __restrict__ manifolds, u64* __restrict__ pairSorted, u32 slotCap)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nc) return;
	const ColliderRec C = colWorld[i];
	const u32 body = colBody(C);
	if (colType(C) != (u32)TYPE) return;                       // heightmap_collision.cpp:545-570: sphere, capsule, aabb, obb only
	if (!(body < nb && simMask[body])) { if (!write) counts[i] = 0; return; }
	u32 n = 0, base = 0, fr = 0;
	if (write)
	{
		if (!counts[i]) return;
		base = counters[CTR_NUM_VALID] + offsets[i];
		float friction = clamp01(sqrtf(colFriction(C) * P.friction));
		float restitution = clamp01(fmaxf(colRestitution(C), P.restitution));
		fr = ((u32)(friction * 0xFFFF) << 16) | (u32)(restitution * 0xFFFF);
	}
#define MI_TERRAIN_EMIT(T) { if (write) { u32 slot = base + n; if (slot < slotCap) { \
		ManifoldRec rec; float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f); \
		rec.p[0] = make_float4((T).point.x, (T).point.y, (T).point.z, (T).depth); rec.p[1] = zero4; rec.p[2] = zero4; rec.p[3] = zero4; \
		rec.nf = make_float4((T).normal.x, (T).normal.y, (T).normal.z, __uint_as_float(fr)); rec.ids = make_uint4(body, nb, 1u, slot); \
		manifolds[slot] = rec; pairSorted[slot] = ((u64)(TERRAIN_SLOT | n) << 32) | i; } else counters[CTR_TERRAIN_OVERFLOW] = 1u; } ++n; }

	V3 p0 = v3s(0.f), p1 = v3s(0.f), dir = v3s(0.f), lowest; float r = 0.f; Q4 q = q4(0.f, 0.f, 0.f, 1.f);
	const V3 down = v3(0.f, -1.f, 0.f);
	if (TYPE == MI_SPHERE) { p0 = v3(C.a.x, C.a.y, C.a.z); r = C.a.w; lowest = normalize(down) * r + p0; }            // sphere_support_fn, collision_gjk.h:6-15
	else if (TYPE == MI_CAPSULE)
	{
		p0 = v3(C.a.x, C.a.y, C.a.z); p1 = v3(C.a.w, C.b.x, C.b.y); r = C.b.z; dir = normalize(p1 - p0);
		float distA = dot(down, p0), distB = dot(down, p1);                                                             // capsule_support_fn, collision_gjk.h:17-28
		lowest = (distA > distB ? p0 : p1) + normalize(down) * r;
	}
	else if (TYPE == MI_AABB)
	{
		V3 lo = v3(C.a.x, C.a.y, C.a.z), hi = v3(C.a.w, C.b.x, C.b.y);
		p0 = (lo + hi) * 0.5f; p1 = (hi - lo) * 0.5f;
		lowest = v3(hi.x, lo.y, hi.z);                                                                                  // aabb_support_fn with dir (0, -1, 0): x and z are not negative -> max corner
	}
	else
	{
		q = q4(C.a.x, C.a.y, C.a.z, C.a.w); p0 = v3(C.b.x, C.b.y, C.b.z); p1 = v3(C.b.w, C.c.x, C.c.y);
		V3 d = conjugate(q) * down;                                                                                     // obb_support_fn, collision_gjk.h:63-75
		V3 rr = v3(d.x < 0.f ? -p1.x : p1.x, d.y < 0.f ? -p1.y : p1.y, d.z < 0.f ? -p1.z : p1.z);
		lowest = p0 + q * rr;
	}
	// iterateTrianglesInVolume, heightmap_collider.h:156-206
	const V3 corner = v3(P.minX, P.minY, P.minZ);
	V3 vmin = v3f4(aabbMin[i]) - corner, vmax = v3(aabbMax[i].x, aabbMax[i].y + 10.f, aabbMax[i].z) - corner;
	vmin.x *= P.invChunkSize; vmin.z *= P.invChunkSize; vmax.x *= P.invChunkSize; vmax.z *= P.invChunkSize;
	const int cpd = (int)P.chunksPerDim;
	const u32 minX = (u32)max((int)vmin.x, 0), minZ = (u32)max((int)vmin.z, 0);
	const u32 maxX = (u32)min(max((int)vmax.x, 0), cpd - 1), maxZ = (u32)min(max((int)vmax.z, 0), cpd - 1);
	vmin.y *= P.invAmplitudeScale; vmax.y *= P.invAmplitudeScale;
	const u32 minHeight = (u32)(uint16_t)(clamp01(vmin.y) * 65535), maxHeight = (u32)(uint16_t)(clamp01(vmax.y) * 65535);
	for (u32 z = minZ; z <= maxZ; ++z)
		for (u32 x = minX; x <= maxX; ++x)
		{
			const u32 chunk = z * P.chunksPerDim + x;
			if (!valid[chunk]) continue;
			float relMinX = fmaxf(vmin.x - x, 0.f), relMinZ = fmaxf(vmin.z - z, 0.f);
			float relMaxX = (vmax.x > (x + 1)) ? 1.f : fmodf(vmax.x, 1.f), relMaxZ = (vmax.z > (z + 1)) ? 1.f : fmodf(vmax.z, 1.f);
			u32 cMinX = (u32)(relMinX * TERRAIN_VERTS), cMinZ = (u32)(relMinZ * TERRAIN_VERTS), cMaxX = (u32)(relMaxX * TERRAIN_VERTS), cMaxZ = (u32)(relMaxZ * TERRAIN_VERTS);
			cMaxX = min(cMaxX, TERRAIN_VERTS - 2u); cMaxZ = min(cMaxZ, TERRAIN_VERTS - 2u); // the pyramid's leaves end at cell 127
			const V3 chunkMin = v3(x * P.chunkSize, 0.f, z * P.chunkSize) + corner;
			const uint16_t* H = heights + (size_t)chunk * TERRAIN_VERTS * TERRAIN_VERTS;
			for (u32 cz = cMinZ; cz <= cMaxZ; ++cz)
				for (u32 cx = cMinX; cx <= cMaxX; ++cx)
				{
					u32 ha = H[TERRAIN_VERTS * cz + cx], hb = H[TERRAIN_VERTS * (cz + 1) + cx], hc = H[TERRAIN_VERTS * cz + cx + 1], hd = H[TERRAIN_VERTS * (cz + 1) + cx + 1];
					u32 lo = min(min(ha, hb), min(hc, hd)), hi = max(max(ha, hb), max(hc, hd));
					if (hi < minHeight || lo > maxHeight) continue;
					float x0 = (float)cx * P.chunkScale, x1 = (float)(cx + 1) * P.chunkScale, z0 = (float)cz * P.chunkScale, z1 = (float)(cz + 1) * P.chunkScale;
					V3 posA = v3(x0, ha * P.heightScale, z0) + chunkMin, posB = v3(x0, hb * P.heightScale, z1) + chunkMin;
					V3 posC = v3(x1, hc * P.heightScale, z0) + chunkMin, posD = v3(x1, hd * P.heightScale, z1) + chunkMin;
					TerrainContact t0, t1;
					bool hit0 = terrainTriangle<TYPE>(p0, p1, r, q, dir, posA, posB, posC, t0);
					bool hit1 = terrainTriangle<TYPE>(p0, p1, r, q, dir, posC, posB, posD, t1);
					if (hit0) MI_TERRAIN_EMIT(t0)
					if (hit1) MI_TERRAIN_EMIT(t1)
				}
		}
	float h = terrainHeightAt(P, heights, valid, lowest.x, lowest.z); // heightmap_collision.cpp:573-580
	if (lowest.y < h) { TerrainContact t; t.normal = down; t.point = lowest; t.depth = h - lowest.y; MI_TERRAIN_EMIT(t) }
#undef MI_TERRAIN_EMIT
	if (!write) counts[i] = n;
}

__global__ void k_heightmap_finish(u32* __restrict__ counters, const u32* __restrict__ counts, const u32* __restrict__ offsets, u32 nc, u32 slotCap)
{
	u32 total = offsets[nc - 1] + counts[nc - 1];
	u32 base = counters[CTR_NUM_VALID];
	counters[CTR_TERRAIN_BASE] = base;
	counters[CTR_NUM_VALID] = min(base + total, slotCap);
}

template <int TYPE>
static void launchTerrainPass(World& w, bool write, const TerrainParams& P, u32 slotCap)
{
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_heightmap<TYPE>), dim3((w.nc + 63) / 64), dim3(64), 0, w.stream, write, w.nc, w.nb, w.colWorld.p, w.aabbMin.p, w.aabbMax.p, w.simMask.p, P, w.terrainHeights.p, w.terrainValid.p,
		w.terrainCounts.p, w.terrainOffsets.p, w.dCounters.p, w.manifolds.p, (u64*)w.pairsSorted.p, slotCap);
}
static void launchTerrainPasses(World& w, bool write, const TerrainParams& P, u32 slotCap)
{
	launchTerrainPass<MI_SPHERE>(w, write, P, slotCap); launchTerrainPass<MI_CAPSULE>(w, write, P, slotCap); launchTerrainPass<MI_AABB>(w, write, P, slotCap); launchTerrainPass<MI_OBB>(w, write, P, slotCap);
}

void launch_heightmap(World& w, u32 numPairs, u32 slotCap)
{
	if (!w.terrainChunksPerDim || !w.nc) return;
	TerrainParams P;
	P.chunksPerDim = w.terrainChunksPerDim; P.chunkSize = w.terrainChunkSize; P.invChunkSize = 1.f / w.terrainChunkSize; P.chunkScale = w.terrainChunkSize / (TERRAIN_VERTS - 1);
	P.heightScale = w.terrainAmplitude / 65535; P.invAmplitudeScale = 1.f / w.terrainAmplitude;
	P.minX = w.terrainMinCorner[0]; P.minY = w.terrainMinCorner[1]; P.minZ = w.terrainMinCorner[2]; P.friction = w.terrainMaterial[1]; P.restitution = w.terrainMaterial[0];
	w.terrainCounts.ensure((size_t)w.nc + 1, w.stream); w.terrainOffsets.ensure((size_t)w.nc + 1, w.stream);
	MI_CHECK(hipMemsetAsync(w.terrainCounts.p, 0, sizeof(u32) * ((size_t)w.nc + 1), w.stream)); // cylinders and hulls: no kernel touches their entry
	launchTerrainPasses(w, false, P, slotCap);
	size_t bytes = 0;
	MI_CHECK(rocprim::exclusive_scan(nullptr, bytes, w.terrainCounts.p, w.terrainOffsets.p, 0u, w.nc, rocprim::plus<u32>(), w.stream));
	w.tempStorage.ensure(bytes, w.stream);
	MI_CHECK(rocprim::exclusive_scan(w.tempStorage.p, bytes, w.terrainCounts.p, w.terrainOffsets.p, 0u, w.nc, rocprim::plus<u32>(), w.stream));
	launchTerrainPasses(w, true, P, slotCap);
	hipLaunchKernelGGL(k_heightmap_finish, dim3(1), dim3(1), 0, w.stream, w.dCounters.p, w.terrainCounts.p, w.terrainOffsets.p, w.nc, slotCap);
}
