// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's heightmap terrain collision:
// src/terrain/heightmap_collider.h:15-206, heightmap_collider.cpp:5-153 (chunks of 129 x 129 uint16 heights, min/max mip pyramid,
// quadtree triangle iteration) and src/physics/heightmap_collision.cpp:6-618 (sphere / capsule / AABB / OBB vs triangle, the
// "lowest point under the terrain" contact).  Contacts come out in the reference's order (depth-first over the mip pyramid); each
// carries the key (chunk z, chunk x, cell z, cell x, triangle) so that follow-mode tests can re-order them the way the device emits them.
#pragma once
#include "onarrow.h"
#include <vector>
#include <cmath>
#include <cfloat>

namespace orc
{

static const u32 TERRAIN_VERTS = 129u; // TERRAIN_LOD_0_VERTICES_PER_DIMENSION

static inline float fracf(float v) { return fmodf(v, 1.f); }                                   // core/math.h:40

static inline vec3 closestPoint_PointTriangle(vec3 p, vec3 a, vec3 b, vec3 c) // bounding_volumes.cpp:1317-1367
{
	vec3 ab = b - a, ac = c - a, ap = p - a;
	float d1 = dot(ab, ap), d2 = dot(ac, ap);
	if (d1 <= 0.f && d2 <= 0.f) return a;
	vec3 bp = p - b;
	float d3 = dot(ab, bp), d4 = dot(ac, bp);
	if (d3 >= 0.f && d4 <= d3) return b;
	float vc = d1 * d4 - d3 * d2;
	if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 / (d1 - d3); return a + v * ab; }
	vec3 cp = p - c;
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

struct terrain_contact { collision_contact contact; u32 key[5]; }; // key: chunk z, chunk x, cell z, cell x, triangle (0 / 1); the lowest-point contact has key[0] = ~0u

static inline u32 collideSphereVsTriangle(vec3 center, float radius, vec3 a, vec3 b, vec3 c, collision_contact& contact) // heightmap_collision.cpp:43-80
{
	vec3 closestPoint = closestPoint_PointTriangle(center, a, b, c);
	vec3 n = closestPoint - center;
	float sqDistance = squaredLength(n);
	if (sqDistance <= radius * radius)
	{
		float distance;
		if (sqDistance == 0.f) { vec3 triNormal = cross(b - a, c - a); n = -triNormal; distance = 0.f; }
		else { distance = sqrtf(sqDistance); n *= 1.f / distance; }
		contact.point = closestPoint; contact.normal = n; contact.penetrationDepth = radius - distance;
		return 1;
	}
	return 0;
}

static inline void terrainAABBIncidentEdge(vec3 aabbRadius, vec3 normal, vec3& outA, vec3& outB) // heightmap_collision.cpp:6-41
{
	vec3 p(fabsf(normal.x), fabsf(normal.y), fabsf(normal.z));
	outA = aabbRadius;
	if (p.x > p.y) { outB = (p.y > p.z) ? vec3(aabbRadius.x, aabbRadius.y, -aabbRadius.z) : vec3(aabbRadius.x, -aabbRadius.y, aabbRadius.z); }
	else { outB = (p.x > p.z) ? vec3(aabbRadius.x, aabbRadius.y, -aabbRadius.z) : vec3(-aabbRadius.x, aabbRadius.y, aabbRadius.z); }
	vec3 s(normal.x < 0.f ? -1.f : 1.f, normal.y < 0.f ? -1.f : 1.f, normal.z < 0.f ? -1.f : 1.f);
	outA = outA * s; outB = outB * s;
}

static inline u32 collideAABBvsTriangle(vec3 center, vec3 radius, vec3 a, vec3 b, vec3 c, collision_contact& contact) // heightmap_collision.cpp:82-423
{
	a -= center; b -= center; c -= center;
	vec3 f0 = b - a, f1 = c - b, f2 = a - c;
	float minPenetration = FLT_MAX;
	vec3 minNormal(0.f);
	u32 category = 0; // 0..2: edge axis of triangle edge 0 / 1 / 2; 3: box face; 4: triangle face
	// nine edge-cross axes: for each box axis (x, y, z) the three triangle edges.  P = the projections the reference picks for that edge
	// (edge 0: vertices a and c, edges 1 and 2: a and b); R = box radius on the axis; N = the axis.
#define ORC_TRI_AXIS(P0, P1, R, N, CAT) { float p0 = (P0), p1 = (P1); float r = (R); \
		float penetration = r - std::max(-std::max(p0, p1), std::min(p0, p1)); if (penetration < 0.f) return 0; \
		vec3 normal = N; float l = length(normal); penetration *= 1.f / l; \
		if (penetration < minPenetration) { minPenetration = penetration; minNormal = normal * (1.f / l); category = CAT; } }
	ORC_TRI_AXIS((a.z * f0.y) - (a.y * f0.z), (c.z * f0.y) - (c.y * f0.z), radius.y * fabsf(f0.z) + radius.z * fabsf(f0.y), vec3(0.f, -f0.z, f0.y), 0)
	ORC_TRI_AXIS((a.z * f1.y) - (a.y * f1.z), (b.z * f1.y) - (b.y * f1.z), radius.y * fabsf(f1.z) + radius.z * fabsf(f1.y), vec3(0.f, -f1.z, f1.y), 1)
	ORC_TRI_AXIS((a.z * f2.y) - (a.y * f2.z), (b.z * f2.y) - (b.y * f2.z), radius.y * fabsf(f2.z) + radius.z * fabsf(f2.y), vec3(0.f, -f2.z, f2.y), 2)
	ORC_TRI_AXIS((a.x * f0.z) - (a.z * f0.x), (c.x * f0.z) - (c.z * f0.x), radius.x * fabsf(f0.z) + radius.z * fabsf(f0.x), vec3(f0.z, 0.f, -f0.x), 0)
	ORC_TRI_AXIS((a.x * f1.z) - (a.z * f1.x), (b.x * f1.z) - (b.z * f1.x), radius.x * fabsf(f1.z) + radius.z * fabsf(f1.x), vec3(f1.z, 0.f, -f1.x), 1)
	ORC_TRI_AXIS((a.x * f2.z) - (a.z * f2.x), (b.x * f2.z) - (b.z * f2.x), radius.x * fabsf(f2.z) + radius.z * fabsf(f2.x), vec3(f2.z, 0.f, -f2.x), 2)
	ORC_TRI_AXIS((a.y * f0.x) - (a.x * f0.y), (c.y * f0.x) - (c.x * f0.y), radius.x * fabsf(f0.y) + radius.y * fabsf(f0.x), vec3(-f0.y, f0.x, 0.f), 0)
	ORC_TRI_AXIS((a.y * f1.x) - (a.x * f1.y), (b.y * f1.x) - (b.x * f1.y), radius.x * fabsf(f1.y) + radius.y * fabsf(f1.x), vec3(-f1.y, f1.x, 0.f), 1)
	ORC_TRI_AXIS((a.y * f2.x) - (a.x * f2.y), (b.y * f2.x) - (b.x * f2.y), radius.x * fabsf(f2.y) + radius.y * fabsf(f2.x), vec3(-f2.y, f2.x, 0.f), 2)
#undef ORC_TRI_AXIS
#define ORC_BOX_FACE(PEN, N) { float penetration = (PEN); if (penetration < 0.f) return 0; if (penetration < minPenetration) { minPenetration = penetration; minNormal = N; category = 3; } }
	ORC_BOX_FACE(std::max(a.x, std::max(b.x, c.x)) + radius.x, vec3(-1.f, 0.f, 0.f))
	ORC_BOX_FACE(radius.x - std::min(a.x, std::min(b.x, c.x)), vec3(1.f, 0.f, 0.f))
	ORC_BOX_FACE(std::max(a.y, std::max(b.y, c.y)) + radius.y, vec3(0.f, -1.f, 0.f))
	ORC_BOX_FACE(radius.y - std::min(a.y, std::min(b.y, c.y)), vec3(0.f, 1.f, 0.f))
	ORC_BOX_FACE(std::max(a.z, std::max(b.z, c.z)) + radius.z, vec3(0.f, 0.f, -1.f))
	ORC_BOX_FACE(radius.z - std::min(a.z, std::min(b.z, c.z)), vec3(0.f, 0.f, 1.f))
#undef ORC_BOX_FACE
	{
		vec3 triNormal = normalize(cross(f0, f1));
		float triD = dot(triNormal, a);
		float r = dot(radius, vec3(fabsf(triNormal.x), fabsf(triNormal.y), fabsf(triNormal.z)));
		float penetration = r - fabsf(triD);
		if (penetration < 0.f) return 0;
		if (penetration < minPenetration) { minPenetration = penetration; minNormal = triNormal; category = 4; }
	}
	vec3 triCenter = (a + b + c) * (1.f / 3.f);
	if (dot(minNormal, triCenter) < 0.f) { minNormal = -minNormal; }
	vec3 point;
	if (category < 3)
	{
		vec3 a0, a1;
		terrainAABBIncidentEdge(radius, minNormal, a0, a1);
		vec3 triA = (category == 0) ? a : (category == 1) ? b : c;
		vec3 triB = (category == 0) ? b : (category == 1) ? c : a;
		vec3 pa, pb;
		closestPoint_SegmentSegment(line_segment{ a0, a1 }, line_segment{ triA, triB }, pa, pb);
		point = (pa + pb) * 0.5f;
	}
	else if (category == 3)
	{
		float da = dot(minNormal, a), db = dot(minNormal, b), dc = dot(minNormal, c);
		vec3 p = (da < db) ? ((da < dc) ? a : c) : ((db < dc) ? b : c);
		point = p + minNormal * (minPenetration * 0.5f);
	}
	else
	{
		vec3 p((minNormal.x < 0.f) ? -radius.x : radius.x, (minNormal.y < 0.f) ? -radius.y : radius.y, (minNormal.z < 0.f) ? -radius.z : radius.z);
		point = p - minNormal * (minPenetration * 0.5f);
	}
	point += center;
	contact.point = point; contact.normal = minNormal; contact.penetrationDepth = minPenetration;
	return 1;
}

struct heightmap_min_max { u16 min, max; };
struct heightmap_chunk
{
	std::vector<u16> heights;                         // 129 x 129, empty = no heights (heightmap_collider.h:27)
	std::vector<std::vector<heightmap_min_max>> mips; // setHeights, heightmap_collider.cpp:40-121
	void setHeights(const u16* h)
	{
		heights.assign(h, h + TERRAIN_VERTS * TERRAIN_VERTS);
		u32 numSegments = TERRAIN_VERTS - 1;
		u32 numMips = 8; // log2(128) + 1
		mips.assign(numMips, {});
		mips[0].resize(numSegments * numSegments);
		for (u32 z = 0; z < numSegments; ++z)
			for (u32 x = 0; x < numSegments; ++x)
			{
				u16 a = heights[TERRAIN_VERTS * z + x], b = heights[TERRAIN_VERTS * (z + 1) + x], c = heights[TERRAIN_VERTS * z + x + 1], d = heights[TERRAIN_VERTS * (z + 1) + x + 1];
				mips[0][numSegments * z + x] = { std::min(a, std::min(b, std::min(c, d))), std::max(a, std::max(b, std::max(c, d))) };
			}
		for (u32 i = 1; i < numMips; ++i)
		{
			u32 readStride = numSegments;
			numSegments >>= 1;
			mips[i].resize(numSegments * numSegments);
			for (u32 z = 0; z < numSegments; ++z)
				for (u32 x = 0; x < numSegments; ++x)
				{
					heightmap_min_max a = mips[i - 1][readStride * (2 * z) + 2 * x], b = mips[i - 1][readStride * (2 * z + 1) + 2 * x];
					heightmap_min_max c = mips[i - 1][readStride * (2 * z) + 2 * x + 1], d = mips[i - 1][readStride * (2 * z + 1) + 2 * x + 1];
					mips[i][numSegments * z + x] = { std::min(a.min, std::min(b.min, std::min(c.min, d.min))), std::max(a.max, std::max(b.max, std::max(c.max, d.max))) };
				}
		}
	}
	// iterateTrianglesInVolume, heightmap_collider.h:36-124: func(a, b, c, cellX, cellZ, triangle)
	template <typename F> void iterateTrianglesInVolume(u32 volMinX, u32 volMinZ, u32 volMaxX, u32 volMaxZ, u32 volMinY, u32 volMaxY, float chunkScale, float heightScale, vec3 chunkMinCorner, const F& func) const
	{
		if (heights.empty()) { return; }
		struct stack_entry { u16 mipLevel, x, z; };
		std::vector<stack_entry> stack;
		stack.push_back({ (u16)(mips.size() - 1), 0, 0 });
		while (!stack.empty())
		{
			stack_entry entry = stack.back(); stack.pop_back();
			u32 minX = (u32)entry.x << entry.mipLevel, minZ = (u32)entry.z << entry.mipLevel;
			u32 maxX = (((u32)entry.x + 1) << entry.mipLevel) - 1, maxZ = (((u32)entry.z + 1) << entry.mipLevel) - 1;
			if (maxX < volMinX || minX > volMaxX) continue;
			if (maxZ < volMinZ || minZ > volMaxZ) continue;
			u32 numSegmentsPerDim = (TERRAIN_VERTS - 1) >> entry.mipLevel;
			heightmap_min_max minmax = mips[entry.mipLevel][entry.z * numSegmentsPerDim + entry.x];
			if (minmax.max < volMinY || minmax.min > volMaxY) continue;
			if (entry.mipLevel == 0)
			{
				u32 stride = TERRAIN_VERTS;
				float heightA = heights[stride * entry.z + entry.x] * heightScale, heightB = heights[stride * (entry.z + 1) + entry.x] * heightScale;
				float heightC = heights[stride * entry.z + entry.x + 1] * heightScale, heightD = heights[stride * (entry.z + 1) + entry.x + 1] * heightScale;
				float x0 = (float)(entry.x) * chunkScale, x1 = (float)(entry.x + 1) * chunkScale, z0 = (float)(entry.z) * chunkScale, z1 = (float)(entry.z + 1) * chunkScale;
				vec3 posA = vec3(x0, heightA, z0) + chunkMinCorner, posB = vec3(x0, heightB, z1) + chunkMinCorner;
				vec3 posC = vec3(x1, heightC, z0) + chunkMinCorner, posD = vec3(x1, heightD, z1) + chunkMinCorner;
				func(posA, posB, posC, (u32)entry.x, (u32)entry.z, 0u);
				func(posC, posB, posD, (u32)entry.x, (u32)entry.z, 1u);
			}
			else
			{
				stack.push_back({ (u16)(entry.mipLevel - 1), (u16)(2 * entry.x + 0), (u16)(2 * entry.z + 0) });
				stack.push_back({ (u16)(entry.mipLevel - 1), (u16)(2 * entry.x + 0), (u16)(2 * entry.z + 1) });
				stack.push_back({ (u16)(entry.mipLevel - 1), (u16)(2 * entry.x + 1), (u16)(2 * entry.z + 0) });
				stack.push_back({ (u16)(entry.mipLevel - 1), (u16)(2 * entry.x + 1), (u16)(2 * entry.z + 1) });
			}
		}
	}
	float getHeightAt(float cx, float cz, float heightScale, float heightOffset) const // heightmap_collider.cpp:123-153
	{
		if (heights.empty()) { return -FLT_MAX; }
		u32 x = (u32)cx, z = (u32)cz;
		float relX = cx - x, relZ = cz - z;
		u32 stride = TERRAIN_VERTS;
		float a = heights[stride * z + x] * heightScale, b = heights[stride * (z + 1) + x] * heightScale;
		float c = heights[stride * z + x + 1] * heightScale, d = heights[stride * (z + 1) + x + 1] * heightScale;
		return lerpf(lerpf(a, c, relX), lerpf(b, d, relX), relZ) + heightOffset;
	}
};

struct heightmap // heightmap_collider_component, heightmap_collider.h:127-152
{
	u32 chunksPerDim = 0;
	float chunkSize = 0.f, invChunkSize = 0.f, chunkScale = 0.f, heightScale = 0.f, invAmplitudeScale = 1.f;
	physics_material material{};
	vec3 minCorner = vec3(0.f);
	std::vector<heightmap_chunk> chunks;
	void create(u32 chunksPerDim_, float chunkSize_, physics_material m)
	{
		chunksPerDim = chunksPerDim_; chunkSize = chunkSize_; invChunkSize = 1.f / chunkSize_; material = m;
		chunks.assign((size_t)chunksPerDim * chunksPerDim, {});
		chunkScale = chunkSize / (TERRAIN_VERTS - 1);
	}
	void update(vec3 minCorner_, float amplitudeScale) { minCorner = minCorner_; invAmplitudeScale = 1.f / amplitudeScale; heightScale = amplitudeScale / 65535; }
	float getHeightAt(float wx, float wz) const // heightmap_collider.cpp:22-38
	{
		float cx = (wx - minCorner.x) * invChunkSize, cz = (wz - minCorner.z) * invChunkSize;
		if (cx < 0.f || cz < 0.f || cx >= chunksPerDim || cz >= chunksPerDim) { return -FLT_MAX; }
		u32 chunkX = (u32)cx, chunkZ = (u32)cz;
		cx = fracf(cx) * (TERRAIN_VERTS - 1); cz = fracf(cz) * (TERRAIN_VERTS - 1);
		return chunks[chunkZ * chunksPerDim + chunkX].getHeightAt(cx, cz, heightScale, minCorner.y);
	}
	template <typename F> void iterateTrianglesInVolume(bounding_box volume, const F& func) const // heightmap_collider.h:156-206: func(a, b, c, key[5])
	{
		volume.minCorner -= minCorner; volume.maxCorner -= minCorner;
		volume.minCorner.x *= invChunkSize; volume.minCorner.z *= invChunkSize; volume.maxCorner.x *= invChunkSize; volume.maxCorner.z *= invChunkSize;
		u32 minX = (u32)std::max((int32_t)volume.minCorner.x, 0), minZ = (u32)std::max((int32_t)volume.minCorner.z, 0);
		u32 maxX = (u32)std::min(std::max((int32_t)volume.maxCorner.x, 0), (int32_t)chunksPerDim - 1), maxZ = (u32)std::min(std::max((int32_t)volume.maxCorner.z, 0), (int32_t)chunksPerDim - 1);
		volume.minCorner.y *= invAmplitudeScale; volume.maxCorner.y *= invAmplitudeScale;
		u16 minHeight = (u16)(clamp01(volume.minCorner.y) * 65535), maxHeight = (u16)(clamp01(volume.maxCorner.y) * 65535);
		for (u32 z = minZ; z <= maxZ; ++z)
			for (u32 x = minX; x <= maxX; ++x)
			{
				float relMinX = std::max(volume.minCorner.x - x, 0.f), relMinZ = std::max(volume.minCorner.z - z, 0.f);
				float relMaxX = (volume.maxCorner.x > (x + 1)) ? 1.f : fracf(volume.maxCorner.x), relMaxZ = (volume.maxCorner.z > (z + 1)) ? 1.f : fracf(volume.maxCorner.z);
				u32 cMinX = (u32)(relMinX * TERRAIN_VERTS), cMinZ = (u32)(relMinZ * TERRAIN_VERTS), cMaxX = (u32)(relMaxX * TERRAIN_VERTS), cMaxZ = (u32)(relMaxZ * TERRAIN_VERTS);
				vec3 chunkMinCorner = vec3(x * chunkSize, 0.f, z * chunkSize) + minCorner;
				chunks[z * chunksPerDim + x].iterateTrianglesInVolume(cMinX, cMinZ, cMaxX, cMaxZ, minHeight, maxHeight, chunkScale, heightScale, chunkMinCorner,
					[&](vec3 a, vec3 b, vec3 c, u32 cellX, u32 cellZ, u32 tri) { const u32 key[5] = { z, x, cellZ, cellX, tri }; func(a, b, c, key); });
			}
	}
};

// heightmapCollision for one rigid-body collider (heightmap_collision.cpp:522-583).  Cylinder and hull colliders have no case in the
// reference's switch (their `lowestPoint` is read uninitialised there); they produce nothing here.
static inline void heightmapContacts(const heightmap& hm, const collider_union& collider, bounding_box aabb, std::vector<terrain_contact>& out)
{
	aabb.maxCorner.y += 10.f;
	vec3 lowestPoint;
	auto push = [&out](const collision_contact& c, const u32* key) { terrain_contact t; t.contact = c; for (int i = 0; i < 5; ++i) t.key[i] = key[i]; out.push_back(t); };
	const vec3 down(0.f, -1.f, 0.f);
	switch (collider.type)
	{
		case collider_type_sphere:
		{
			bounding_sphere s = collider.sphere();
			hm.iterateTrianglesInVolume(aabb, [&](vec3 a, vec3 b, vec3 c, const u32* key) { collision_contact ct{}; if (collideSphereVsTriangle(s.center, s.radius, a, b, c, ct)) push(ct, key); });
			lowestPoint = sphere_support_fn{ s }(down);
		} break;
		case collider_type_capsule:
		{
			bounding_capsule capsule = collider.capsule();
			vec3 origin = capsule.positionA, direction = normalize(capsule.positionB - capsule.positionA);
			hm.iterateTrianglesInVolume(aabb, [&](vec3 a, vec3 b, vec3 c, const u32* key)
			{
				vec3 triNormal = normalize(cross(b - a, c - a));
				float d = -dot(triNormal, a);
				float ndotd = dot(direction, triNormal);
				float t = -(dot(origin, triNormal) + d) / ndotd;
				vec3 trace = origin + t * direction;
				vec3 closest = closestPoint_PointTriangle(trace, a, b, c);
				vec3 reference = closestPoint_PointSegment(closest, line_segment{ capsule.positionA, capsule.positionB });
				collision_contact ct{}; if (collideSphereVsTriangle(reference, capsule.radius, a, b, c, ct)) push(ct, key);
			});
			lowestPoint = capsule_support_fn{ capsule }(down);
		} break;
		case collider_type_aabb:
		{
			bounding_box box = collider.aabb();
			vec3 center = box.getCenter(), radius = box.getRadius();
			hm.iterateTrianglesInVolume(aabb, [&](vec3 a, vec3 b, vec3 c, const u32* key) { collision_contact ct{}; if (collideAABBvsTriangle(center, radius, a, b, c, ct)) push(ct, key); });
			lowestPoint = aabb_support_fn{ box }(down);
		} break;
		case collider_type_obb:
		{
			bounding_oriented_box obb = collider.obb();
			hm.iterateTrianglesInVolume(aabb, [&](vec3 a, vec3 b, vec3 c, const u32* key)
			{
				a = conjugate(obb.rotation) * (a - obb.center); b = conjugate(obb.rotation) * (b - obb.center); c = conjugate(obb.rotation) * (c - obb.center);
				collision_contact ct{};
				if (collideAABBvsTriangle(vec3(0.f, 0.f, 0.f), obb.radius, a, b, c, ct)) { ct.normal = obb.rotation * ct.normal; ct.point = obb.rotation * ct.point + obb.center; push(ct, key); }
			});
			lowestPoint = obb_support_fn{ obb }(down);
		} break;
		default: return;
	}
	float heightAtLowestPoint = hm.getHeightAt(lowestPoint.x, lowestPoint.z);
	if (lowestPoint.y < heightAtLowestPoint)
	{
		collision_contact ct{}; ct.normal = down; ct.point = lowestPoint; ct.penetrationDepth = heightAtLowestPoint - lowestPoint.y;
		const u32 key[5] = { ~0u, 0, 0, 0, 0 };
		push(ct, key);
	}
}

} // namespace orc
