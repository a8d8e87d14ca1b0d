// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// Narrowphase restated from the reference's src/physics/collision_narrow.cpp (scalar intersection()
// family — the "SIMD narrowphase" is dead code there, SURVEY finding 2), collision_gjk.{h,cpp},
// collision_epa.{h,cpp}.  Hull pairs are out of scope (§8 a19).
#pragma once
#include "oshapes.h"
#include <utility>

namespace orc {

struct contact_info { vec3 point; float penetrationDepth; };

// collision_narrow.cpp:40-46
struct contact_manifold
{
	contact_info contacts[4];
	vec3 collisionNormal; // From a to b.
	u32 numContacts;
};

struct vertex_penetration_pair { vec3 vertex; float penetrationDepth; };

// collision_narrow.cpp:56-146
static inline void findStableContactManifold(vertex_penetration_pair* vertices, u32 numVertices, vec3 normal, contact_manifold& outContact)
{
	if (numVertices > 4)
	{
		vec3 searchDir = getTangent(normal);
		float bestDistance = dot(searchDir, vertices[0].vertex);
		u32 resultIndex = 0;
		for (u32 i = 1; i < numVertices; ++i)
		{
			float distance = dot(searchDir, vertices[i].vertex);
			if (distance > bestDistance) { resultIndex = i; bestDistance = distance; }
		}
		outContact.contacts[0].penetrationDepth = vertices[resultIndex].penetrationDepth;
		outContact.contacts[0].point = vertices[resultIndex].vertex;

		bestDistance = 0.f; resultIndex = 0;
		for (u32 i = 0; i < numVertices; ++i)
		{
			float sqDistance = squaredLength(vertices[i].vertex - outContact.contacts[0].point);
			if (sqDistance > bestDistance) { resultIndex = i; bestDistance = sqDistance; }
		}
		outContact.contacts[1].penetrationDepth = vertices[resultIndex].penetrationDepth;
		outContact.contacts[1].point = vertices[resultIndex].vertex;

		float bestArea = 0.f; resultIndex = 0;
		for (u32 i = 0; i < numVertices; ++i)
		{
			vec3 qa = outContact.contacts[0].point - vertices[i].vertex;
			vec3 qb = outContact.contacts[1].point - vertices[i].vertex;
			float area = 0.5f * dot(cross(qa, qb), normal);
			if (area > bestArea) { resultIndex = i; bestArea = area; }
		}
		outContact.contacts[2].penetrationDepth = vertices[resultIndex].penetrationDepth;
		outContact.contacts[2].point = vertices[resultIndex].vertex;

		bestArea = 0.f; resultIndex = 0;
		for (u32 i = 0; i < numVertices; ++i)
		{
			vec3 qa = outContact.contacts[0].point - vertices[i].vertex;
			vec3 qb = outContact.contacts[1].point - vertices[i].vertex;
			vec3 qc = outContact.contacts[2].point - vertices[i].vertex;
			float area1 = 0.5f * dot(cross(qa, qb), normal);
			float area2 = 0.5f * dot(cross(qb, qc), normal);
			float area3 = 0.5f * dot(cross(qc, qa), normal);
			float area = std::max(std::max(area1, area2), area3);
			if (area > bestArea) { resultIndex = i; bestArea = area; }
		}
		outContact.contacts[3].penetrationDepth = vertices[resultIndex].penetrationDepth;
		outContact.contacts[3].point = vertices[resultIndex].vertex;
		outContact.numContacts = 4;
	}
	else
	{
		outContact.numContacts = numVertices;
		for (u32 i = 0; i < numVertices; ++i)
		{
			outContact.contacts[i].penetrationDepth = vertices[i].penetrationDepth;
			outContact.contacts[i].point = vertices[i].vertex;
		}
	}
}

// collision_narrow.cpp:148-152
struct clipping_polygon { vertex_penetration_pair points[16]; u32 numPoints = 0; };

// collision_narrow.cpp:154-163
static inline vertex_penetration_pair clipAgainstPlane(vertex_penetration_pair a, vertex_penetration_pair b, float aDist, float bDist)
{
	aDist = fabsf(aDist); bDist = fabsf(bDist);
	float total = aDist + bDist;
	float t = aDist / total;
	return { lerp(a.vertex, b.vertex, t), lerpf(a.penetrationDepth, b.penetrationDepth, t) };
}

// collision_narrow.cpp:166-222
static inline void sutherlandHodgmanClipping(clipping_polygon& input, const vec4* clipPlanes, u32 numClipPlanes, clipping_polygon& output)
{
	clipping_polygon* in = &input;
	clipping_polygon* out = &output;
	u32 clipIndex = 0;
	for (; clipIndex < numClipPlanes; ++clipIndex)
	{
		vec4 clipPlane = clipPlanes[clipIndex];
		out->numPoints = 0;
		if (in->numPoints == 0) { break; }
		vertex_penetration_pair startPoint = in->points[in->numPoints - 1];
		for (u32 i = 0; i < in->numPoints; ++i)
		{
			vertex_penetration_pair endPoint = in->points[i];
			float startDist = signedDistanceToPlane(startPoint.vertex, clipPlane);
			float endDist = signedDistanceToPlane(endPoint.vertex, clipPlane);
			bool startInside = startDist > 0.f;
			bool endInside = endDist > 0.f;
			if (startInside && endInside) { out->points[out->numPoints++] = endPoint; }
			else if (startInside) { out->points[out->numPoints++] = clipAgainstPlane(startPoint, endPoint, startDist, endDist); }
			else if (!startInside && endInside)
			{
				out->points[out->numPoints++] = clipAgainstPlane(startPoint, endPoint, startDist, endDist);
				out->points[out->numPoints++] = endPoint;
			}
			startPoint = endPoint;
		}
		std::swap(in, out);
	}
	if (clipIndex % 2 == 0)
	{
		for (u32 i = 0; i < input.numPoints; ++i) { output.points[i] = input.points[i]; }
		output.numPoints = input.numPoints;
	}
}

// collision_narrow.cpp:225-254
static inline void getAABBClippingPlanes(vec3 aabbRadius, vec3 normal, vec3* clipPlanePoints, vec3* clipPlaneNormals)
{
	vec3 p = vabs(normal);
	u32 maxElement = (p.x > p.y) ? ((p.x > p.z) ? 0 : 2) : ((p.y > p.z) ? 1 : 2);
	u32 axis0 = (maxElement + 1) % 3;
	u32 axis1 = (maxElement + 2) % 3;
	{ vec3 n(0.f); n[axis0] = 1.f; clipPlaneNormals[0] = n; clipPlanePoints[0] = -aabbRadius; }
	{ vec3 n(0.f); n[axis1] = 1.f; clipPlaneNormals[1] = n; clipPlanePoints[1] = -aabbRadius; }
	{ vec3 n(0.f); n[axis0] = -1.f; clipPlaneNormals[2] = n; clipPlanePoints[2] = aabbRadius; }
	{ vec3 n(0.f); n[axis1] = -1.f; clipPlaneNormals[3] = n; clipPlanePoints[3] = aabbRadius; }
}

// collision_narrow.cpp:257-289
static inline void getAABBIncidentVertices(vec3 aabbRadius, vec3 normal, clipping_polygon& polygon)
{
	vec3 p = vabs(normal);
	u32 maxElement = (p.x > p.y) ? ((p.x > p.z) ? 0 : 2) : ((p.y > p.z) ? 1 : 2);
	float s = normal[maxElement] < 0.f ? 1.f : -1.f;
	u32 axis0 = (maxElement + 1) % 3;
	u32 axis1 = (maxElement + 2) % 3;
	float d = aabbRadius[maxElement] * s;
	float min0 = -aabbRadius[axis0], min1 = -aabbRadius[axis1];
	float max0 = aabbRadius[axis0], max1 = aabbRadius[axis1];
	polygon.numPoints = 4;
	polygon.points[0].vertex[maxElement] = d; polygon.points[0].vertex[axis0] = min0; polygon.points[0].vertex[axis1] = min1;
	polygon.points[1].vertex[maxElement] = d; polygon.points[1].vertex[axis0] = max0; polygon.points[1].vertex[axis1] = min1;
	polygon.points[2].vertex[maxElement] = d; polygon.points[2].vertex[axis0] = max0; polygon.points[2].vertex[axis1] = max1;
	polygon.points[3].vertex[maxElement] = d; polygon.points[3].vertex[axis0] = min0; polygon.points[3].vertex[axis1] = max1;
}

// collision_narrow.cpp:291-299
static inline vec4 getAABBReferencePlane(const bounding_box& b, vec3 normal)
{
	vec3 point((normal.x < 0.f) ? b.minCorner.x : b.maxCorner.x,
		(normal.y < 0.f) ? b.minCorner.y : b.maxCorner.y,
		(normal.z < 0.f) ? b.minCorner.z : b.maxCorner.z);
	return createPlane(point, normal);
}

// collision_narrow.cpp:301-336
static inline void getAABBIncidentEdge(vec3 aabbRadius, vec3 normal, vec3& outA, vec3& outB)
{
	vec3 p = vabs(normal);
	outA = vec3(aabbRadius.x, aabbRadius.y, aabbRadius.z);
	if (p.x > p.y)
	{
		if (p.y > p.z) { outB = vec3(aabbRadius.x, aabbRadius.y, -aabbRadius.z); }
		else { outB = vec3(aabbRadius.x, -aabbRadius.y, aabbRadius.z); }
	}
	else
	{
		if (p.x > p.z) { outB = vec3(aabbRadius.x, aabbRadius.y, -aabbRadius.z); }
		else { outB = vec3(-aabbRadius.x, aabbRadius.y, aabbRadius.z); }
	}
	float sx = normal.x < 0.f ? -1.f : 1.f;
	float sy = normal.y < 0.f ? -1.f : 1.f;
	float sz = normal.z < 0.f ? -1.f : 1.f;
	outA *= vec3(sx, sy, sz);
	outB *= vec3(sx, sy, sz);
}

// collision_narrow.cpp:339-369
static inline bool clipPointsAndBuildContact(clipping_polygon& polygon, const vec4* clipPlanes, u32 numClipPlanes, const vec4& referencePlane, contact_manifold& outContact)
{
	clipping_polygon clippedPolygon;
	sutherlandHodgmanClipping(polygon, clipPlanes, numClipPlanes, clippedPolygon);
	if (clippedPolygon.numPoints > 0)
	{
		for (u32 i = 0; i < clippedPolygon.numPoints; ++i)
		{
			if (clippedPolygon.points[i].penetrationDepth < 0.f)
			{
				clippedPolygon.points[i] = clippedPolygon.points[clippedPolygon.numPoints - 1];
				--clippedPolygon.numPoints;
				--i;
			}
			else
			{
				clippedPolygon.points[i].vertex += referencePlane.xyz() * clippedPolygon.points[i].penetrationDepth;
			}
		}
		if (clippedPolygon.numPoints > 0)
		{
			findStableContactManifold(clippedPolygon.points, clippedPolygon.numPoints, outContact.collisionNormal, outContact);
			return true;
		}
	}
	return false;
}

// ---------------------------------------------------------------------------------------------------
// GJK — collision_gjk.h:6-75 (support functors), :140-238 (driver); collision_gjk.cpp:6-212 (simplex update)
// ---------------------------------------------------------------------------------------------------
struct sphere_support_fn { bounding_sphere s; vec3 operator()(vec3 dir) const { return normalize(dir) * s.radius + s.center; } };
struct capsule_support_fn
{
	bounding_capsule c;
	vec3 operator()(vec3 dir) const
	{
		float distA = dot(dir, c.positionA);
		float distB = dot(dir, c.positionB);
		vec3 fartherPoint = distA > distB ? c.positionA : c.positionB;
		return normalize(dir) * c.radius + fartherPoint;
	}
};
struct cylinder_support_fn
{
	bounding_cylinder c;
	vec3 operator()(vec3 dir) const
	{
		float distA = dot(dir, c.positionA);
		float distB = dot(dir, c.positionB);
		vec3 fartherPoint = distA > distB ? c.positionA : c.positionB;
		vec3 n = c.positionA - c.positionB;
		vec3 projectedDir = noz(cross(cross(n, dir), n));
		return fartherPoint + projectedDir * c.radius;
	}
};
struct aabb_support_fn
{
	bounding_box b;
	vec3 operator()(vec3 dir) const
	{
		return vec3((dir.x < 0.f) ? b.minCorner.x : b.maxCorner.x,
			(dir.y < 0.f) ? b.minCorner.y : b.maxCorner.y,
			(dir.z < 0.f) ? b.minCorner.z : b.maxCorner.z);
	}
};
struct obb_support_fn
{
	bounding_oriented_box b;
	vec3 operator()(vec3 dir) const
	{
		dir = conjugate(b.rotation) * dir;
		vec3 r(dir.x < 0.f ? -b.radius.x : b.radius.x, dir.y < 0.f ? -b.radius.y : b.radius.y, dir.z < 0.f ? -b.radius.z : b.radius.z);
		return b.center + b.rotation * r;
	}
};

// collision_gjk.h:77-100
struct hull_support_fn
{
	bounding_hull h;
	vec3 operator()(vec3 dir) const
	{
		dir = conjugate(h.rotation) * dir;
		vec3 result(0.f);
		float maxDist = -FLT_MAX;
		for (const vec3& v : (*hullGeometryTable())[h.geometryIndex].vertices)
		{
			float d = dot(dir, v);
			if (d > maxDist) { maxDist = d; result = v; }
		}
		return h.position + h.rotation * result;
	}
};

struct gjk_support_point
{
	vec3 shapeAPoint, shapeBPoint, minkowski;
	gjk_support_point() {}
	gjk_support_point(vec3 a, vec3 b) { shapeAPoint = a; shapeBPoint = b; minkowski = a - b; }
};
struct gjk_simplex { gjk_support_point a, b, c, d; u32 numPoints = 0; };

template <typename A, typename B>
static inline gjk_support_point support(const A& a, const B& b, vec3 dir) { return gjk_support_point(a(dir), b(-dir)); }
static inline vec3 crossABA(vec3 a, vec3 b) { return cross(cross(a, b), a); }

enum gjk_internal_success { gjk_stop, gjk_dont_stop, gjk_unexpected_error };

// collision_gjk.cpp:6-212
static inline gjk_internal_success updateGJKSimplex(gjk_simplex& s, const gjk_support_point& a, vec3& dir)
{
	if (s.numPoints == 2)
	{
		vec3 ao = -a.minkowski;
		vec3 ab = s.b.minkowski - a.minkowski;
		vec3 ac = s.c.minkowski - a.minkowski;
		vec3 abc = cross(ab, ac);
		vec3 abp = cross(ab, abc);
		if (dot(ao, abp) > 0.f) { s.c = a; dir = crossABA(ab, ao); return gjk_dont_stop; }
		vec3 acp = cross(abc, ac);
		if (dot(ao, acp) > 0.f) { s.b = a; dir = crossABA(ac, ao); return gjk_dont_stop; }
		if (dot(ao, abc) >= 0.f) { s.d = s.b; s.b = a; s.numPoints = 3; dir = abc; return gjk_dont_stop; }
		if (dot(ao, -abc) >= 0.f) { s.d = s.c; s.c = s.b; s.b = a; s.numPoints = 3; dir = -abc; return gjk_dont_stop; }
		return gjk_unexpected_error;
	}
	if (s.numPoints == 3)
	{
		vec3 ao = -a.minkowski;
		vec3 ab = s.b.minkowski - a.minkowski;
		vec3 ac = s.c.minkowski - a.minkowski;
		vec3 ad = s.d.minkowski - a.minkowski;
		vec3 bcd = cross(s.c.minkowski - s.b.minkowski, s.d.minkowski - s.b.minkowski);
		if (dot(bcd, dir) > 0.00001f || dot(bcd, s.b.minkowski) < -0.00001f) { return gjk_unexpected_error; }
		vec3 abc = cross(ac, ab);
		vec3 abd = cross(ab, ad);
		vec3 adc = cross(ad, ac);
		i32 flags = 0;
		const i32 overABCFlag = 1, overABDFlag = 2, overADCFlag = 4;
		flags |= (dot(abc, ao) > 0.f) ? overABCFlag : 0;
		flags |= (dot(abd, ao) > 0.f) ? overABDFlag : 0;
		flags |= (dot(adc, ao) > 0.f) ? overADCFlag : 0;
		if (flags == (overABCFlag | overABDFlag | overADCFlag)) { return gjk_unexpected_error; }
		if (flags == 0) { return gjk_stop; }
		if (flags == overABCFlag)
		{
		overABC1:
			if (dot(cross(abc, ab), ao) > 0.f) { s.c = a; s.numPoints = 2; dir = crossABA(ab, ao); return gjk_dont_stop; }
		overABC2:
			if (dot(cross(ac, abc), ao) > 0.f) { s.b = a; s.numPoints = 2; dir = crossABA(ac, ao); return gjk_dont_stop; }
			s.d = a; dir = abc; return gjk_dont_stop;
		}
		if (flags == overABDFlag)
		{
		overABD1:
			if (dot(cross(abd, ad), ao) > 0.f) { s.b = s.d; s.c = a; s.numPoints = 2; dir = crossABA(ad, ao); return gjk_dont_stop; }
		overABD2:
			if (dot(cross(ab, abd), ao) > 0.f) { s.c = a; s.numPoints = 2; dir = crossABA(ab, ao); return gjk_dont_stop; }
			s.c = a; dir = abd; return gjk_dont_stop;
		}
		if (flags == overADCFlag)
		{
		overADC1:
			if (dot(cross(adc, ac), ao) > 0.f) { s.b = a; s.numPoints = 2; dir = crossABA(ac, ao); return gjk_dont_stop; }
		overADC2:
			if (dot(cross(ad, adc), ao) > 0.f) { s.b = a; s.c = s.d; s.numPoints = 2; dir = crossABA(ad, ao); return gjk_dont_stop; }
			s.b = a; dir = adc; return gjk_dont_stop;
		}
		if (flags == (overABCFlag | overABDFlag)) { if (dot(cross(abc, ab), ao) > 0.f) { goto overABD1; } goto overABC2; }
		if (flags == (overABDFlag | overADCFlag)) { if (dot(cross(abd, ad), ao) > 0.f) { goto overADC1; } goto overABD2; }
		if (flags == (overADCFlag | overABCFlag)) { if (dot(cross(adc, ac), ao) > 0.f) { goto overABC1; } goto overADC2; }
		return gjk_unexpected_error;
	}
	return gjk_unexpected_error;
}

// Iteration cap: the reference's while(true) (collision_gjk.h:210) has no cap; we bound it so a device port
// can drain.  GJK_MAX_ITERATIONS is never reached on the test scenes (asserted by tests via orc_gjk_max_iters()).
static const u32 GJK_MAX_ITERATIONS = 64;
extern u32 g_gjkMaxItersSeen;

// collision_gjk.h:183-238
template <typename A, typename B>
static inline bool gjkIntersectionTest(const A& shapeA, const B& shapeB, gjk_simplex& outSimplex)
{
	vec3 dir(1.f, 0.1f, -0.2f);
	outSimplex.c = support(shapeA, shapeB, dir);
	if (dot(outSimplex.c.minkowski, dir) < 0.f) { return false; }
	dir = -outSimplex.c.minkowski;
	outSimplex.b = support(shapeA, shapeB, dir);
	if (dot(outSimplex.b.minkowski, dir) < 0.f) { return false; }
	dir = crossABA(outSimplex.c.minkowski - outSimplex.b.minkowski, -outSimplex.b.minkowski);
	outSimplex.numPoints = 2;
	for (u32 it = 0; it < GJK_MAX_ITERATIONS; ++it)
	{
		if (it + 1 > g_gjkMaxItersSeen) { g_gjkMaxItersSeen = it + 1; }
		if (squaredLength(dir) < 0.0001f) { return false; }
		gjk_support_point a = support(shapeA, shapeB, dir);
		if (dot(a.minkowski, dir) < 0.f) { return false; }
		gjk_internal_success success = updateGJKSimplex(outSimplex, a, dir);
		if (success == gjk_stop)
		{
			outSimplex.a = a;
			outSimplex.numPoints = 4;
			return true;
		}
		else if (success == gjk_unexpected_error) { return false; }
	}
	return false;
}

// ---------------------------------------------------------------------------------------------------
// EPA — collision_epa.h:6-168, collision_epa.cpp:5-239.
// The reference sizes its arrays at 1024 (72 KB of stack) but caps the loop at 20 iterations, so at most
// 24 points exist.  We keep the algorithm and the out-of-memory exits and size the arrays EPA_MAX_*;
// g_epaMax* record the high-water marks so tests can assert the caps are never the binding limit.
// ---------------------------------------------------------------------------------------------------
// Same caps as the device kernels (k_narrow.hip).  Measured high-water marks with the reference's 1024-entry arrays on the golden
// poses and the config-3/4 scenes: 98 triangles, 100 edges, 8 border edges, 16 GJK iterations — these caps never bind there.
static const u32 EPA_MAX_POINTS = 24, EPA_MAX_TRIANGLES = 128, EPA_MAX_EDGES = 160, EPA_MAX_BORDER = 32;
extern u32 g_epaMaxTriangles, g_epaMaxEdges, g_epaMaxBorder;

struct epa_triangle { u16 a, b, c, edgeOppositeA, edgeOppositeB, edgeOppositeC; vec3 normal; float distanceToOrigin; };
struct epa_edge { u16 a, b; u16 triangleA, triangleB; };
struct epa_triangle_info { vec3 normal; float distanceToOrigin; };
struct epa_result { vec3 point, normal; float penetrationDepth; };
enum epa_status { epa_none, epa_success, epa_out_of_memory, epa_max_num_iterations_reached };

struct epa_simplex
{
	gjk_support_point points[EPA_MAX_POINTS];
	epa_triangle triangles[EPA_MAX_TRIANGLES];
	epa_edge edges[EPA_MAX_EDGES];
	u8 active[EPA_MAX_TRIANGLES];
	u16 numTriangles, numPoints, numEdges;

	static epa_triangle_info getTriangleInfo(const gjk_support_point& a, const gjk_support_point& b, const gjk_support_point& c)
	{
		epa_triangle_info r;
		r.normal = normalize(cross(b.minkowski - a.minkowski, c.minkowski - a.minkowski));
		r.distanceToOrigin = dot(r.normal, a.minkowski);
		return r;
	}
	u16 pushPoint(const gjk_support_point& a) { if (numPoints >= EPA_MAX_POINTS) return UINT16_MAX; u16 i = numPoints++; points[i] = a; return i; }
	u16 pushTriangle(u16 a, u16 b, u16 c, u16 eA, u16 eB, u16 eC, epa_triangle_info info)
	{
		if (numTriangles >= EPA_MAX_TRIANGLES) return UINT16_MAX;
		u16 index = numTriangles++;
		active[index] = 1;
		epa_triangle& t = triangles[index];
		t.a = a; t.b = b; t.c = c; t.edgeOppositeA = eA; t.edgeOppositeB = eB; t.edgeOppositeC = eC;
		t.normal = info.normal; t.distanceToOrigin = info.distanceToOrigin;
		if (numTriangles > g_epaMaxTriangles) g_epaMaxTriangles = numTriangles;
		return index;
	}
	u16 pushEdge(u16 a, u16 b, u16 tA, u16 tB)
	{
		if (numEdges >= EPA_MAX_EDGES) return UINT16_MAX;
		u16 index = numEdges++;
		edges[index] = epa_edge{ a, b, tA, tB };
		if (numEdges > g_epaMaxEdges) g_epaMaxEdges = numEdges;
		return index;
	}
	// collision_epa.cpp:89-109
	u32 findTriangleClosestToOrigin()
	{
		u32 closest = (u32)-1;
		float minDistance = FLT_MAX;
		for (u32 i = 0; i < numTriangles; ++i)
		{
			if (active[i] && triangles[i].distanceToOrigin < minDistance) { minDistance = triangles[i].distanceToOrigin; closest = i; }
		}
		return closest;
	}
	// collision_epa.cpp:111-239
	bool addNewPointAndUpdate(const gjk_support_point& newPoint)
	{
		u8 edgeReferences[EPA_MAX_EDGES] = { 0 };
		for (u32 i = 0; i < numTriangles; ++i)
		{
			if (active[i])
			{
				epa_triangle& tri = triangles[i];
				float d = dot(tri.normal, newPoint.minkowski - points[tri.a].minkowski);
				if (d > 0.f)
				{
					++edgeReferences[tri.edgeOppositeA];
					++edgeReferences[tri.edgeOppositeB];
					++edgeReferences[tri.edgeOppositeC];
					active[i] = 0;
				}
			}
		}
		u16 borderEdgeIndices[EPA_MAX_BORDER];
		u32 numBorderEdges = 0;
		for (u32 i = 0; i < numEdges; ++i)
		{
			if (edgeReferences[i] == 1)
			{
				if (numBorderEdges >= EPA_MAX_BORDER) { return false; }
				borderEdgeIndices[numBorderEdges++] = (u16)i;
			}
		}
		if (numBorderEdges > g_epaMaxBorder) g_epaMaxBorder = numBorderEdges;

		u16 newEdgePerPoint[EPA_MAX_POINTS];
		u16 newPointIndex = pushPoint(newPoint);
		if (newPointIndex == UINT16_MAX) { return false; }
		u16 triangleOffset = numTriangles;

		for (u32 i = 0; i < numBorderEdges; ++i)
		{
			u16 edgeIndex = borderEdgeIndices[i];
			epa_edge& edge = edges[edgeIndex];
			bool triAActive = active[edge.triangleA] != 0;
			bool triBActive = active[edge.triangleB] != 0;
			u16 pointToConnect = triBActive ? edge.a : edge.b;
			u16 triangleIndex = numTriangles;
			u16 newEdgeIndex = pushEdge(pointToConnect, newPointIndex, UINT16_MAX, numTriangles);
			if (newEdgeIndex == UINT16_MAX) { return false; }
			newEdgePerPoint[pointToConnect] = newEdgeIndex;
			u16 bIndex = pointToConnect;
			u16 cIndex = triBActive ? edge.b : edge.a;
			const gjk_support_point& b = points[bIndex];
			const gjk_support_point& c = points[cIndex];
			u16 triangleIndexTest = pushTriangle(newPointIndex, bIndex, cIndex, edgeIndex, UINT16_MAX, newEdgeIndex, getTriangleInfo(newPoint, b, c));
			if (triangleIndexTest == UINT16_MAX) { return false; }
			u16& edgeInactiveTriangle = triAActive ? edge.triangleB : edge.triangleA;
			edgeInactiveTriangle = triangleIndex;
		}
		for (u32 i = 0; i < numBorderEdges; ++i)
		{
			u16 edgeIndex = borderEdgeIndices[i];
			epa_edge& edge = edges[edgeIndex];
			bool triangleBNew = edge.triangleB >= triangleOffset;
			u16 pointToConnect = triangleBNew ? edge.a : edge.b;
			u16 otherEdgeIndex = newEdgePerPoint[pointToConnect];
			epa_edge& otherEdge = edges[otherEdgeIndex];
			u16 triangleIndex = (u16)(i + triangleOffset);
			triangles[triangleIndex].edgeOppositeB = otherEdgeIndex;
			otherEdge.triangleA = triangleIndex;
		}
		return true;
	}
};

// collision_epa.h:96-168
template <typename A, typename B>
static inline epa_status epaCollisionInfo(const gjk_simplex& gjkSimplex, const A& shapeA, const B& shapeB, epa_result& outResult, u32 maxNumIterations = 20)
{
	epa_simplex epaSimplex;
	epaSimplex.numTriangles = 0; epaSimplex.numPoints = 0; epaSimplex.numEdges = 0;
	memset(epaSimplex.active, 0, sizeof(epaSimplex.active));

	epaSimplex.pushPoint(gjkSimplex.a);
	epaSimplex.pushPoint(gjkSimplex.b);
	epaSimplex.pushPoint(gjkSimplex.c);
	epaSimplex.pushPoint(gjkSimplex.d);

	epaSimplex.pushTriangle(0, 1, 3, 4, 3, 0, epa_simplex::getTriangleInfo(gjkSimplex.a, gjkSimplex.b, gjkSimplex.d));
	epaSimplex.pushTriangle(1, 2, 3, 5, 4, 1, epa_simplex::getTriangleInfo(gjkSimplex.b, gjkSimplex.c, gjkSimplex.d));
	epaSimplex.pushTriangle(2, 0, 3, 3, 5, 2, epa_simplex::getTriangleInfo(gjkSimplex.c, gjkSimplex.a, gjkSimplex.d));
	epaSimplex.pushTriangle(0, 2, 1, 1, 0, 2, epa_simplex::getTriangleInfo(gjkSimplex.a, gjkSimplex.c, gjkSimplex.b));

	epaSimplex.pushEdge(0, 1, 0, 3);
	epaSimplex.pushEdge(1, 2, 1, 3);
	epaSimplex.pushEdge(2, 0, 2, 3);
	epaSimplex.pushEdge(0, 3, 2, 0);
	epaSimplex.pushEdge(1, 3, 0, 1);
	epaSimplex.pushEdge(2, 3, 1, 2);

	u32 closestIndex = 0;
	epa_status returnCode = epa_max_num_iterations_reached;
	for (u32 iteration = 0; iteration < maxNumIterations; ++iteration)
	{
		closestIndex = epaSimplex.findTriangleClosestToOrigin();
		epa_triangle& tri = epaSimplex.triangles[closestIndex];
		gjk_support_point a = support(shapeA, shapeB, tri.normal);
		float d = dot(a.minkowski, tri.normal);
		if (d - tri.distanceToOrigin < 0.01f) { returnCode = epa_success; break; }
		if (!epaSimplex.addNewPointAndUpdate(a)) { returnCode = epa_out_of_memory; break; }
	}

	epa_triangle& tri = epaSimplex.triangles[closestIndex];
	gjk_support_point& a = epaSimplex.points[tri.a];
	gjk_support_point& b = epaSimplex.points[tri.b];
	gjk_support_point& c = epaSimplex.points[tri.c];
	vec3 bary = getBarycentricCoordinates(a.minkowski, b.minkowski, c.minkowski, tri.normal * tri.distanceToOrigin);
	vec3 pointA = bary.x * a.shapeAPoint + bary.y * b.shapeAPoint + bary.z * c.shapeAPoint;
	vec3 pointB = bary.x * a.shapeBPoint + bary.y * b.shapeBPoint + bary.z * c.shapeBPoint;
	outResult.point = 0.5f * (pointA + pointB);
	outResult.normal = tri.normal;
	outResult.penetrationDepth = tri.distanceToOrigin;
	return returnCode;
}

// ---------------------------------------------------------------------------------------------------
// intersection() family — collision_narrow.cpp:374-1527
// ---------------------------------------------------------------------------------------------------
// :374-400
static inline bool intersection(const bounding_sphere& s1, const bounding_sphere& s2, contact_manifold& outContact)
{
	vec3 n = s2.center - s1.center;
	float radiusSum = s2.radius + s1.radius;
	float sqDistance = squaredLength(n);
	if (sqDistance <= radiusSum * radiusSum)
	{
		float distance;
		if (sqDistance == 0.f) { distance = 0.f; outContact.collisionNormal = vec3(0.f, 1.f, 0.f); }
		else { distance = sqrtf(sqDistance); outContact.collisionNormal = n / distance; }
		outContact.numContacts = 1;
		outContact.contacts[0].penetrationDepth = radiusSum - distance;
		outContact.contacts[0].point = 0.5f * (s1.center + s1.radius * outContact.collisionNormal + s2.center - s2.radius * outContact.collisionNormal);
		return true;
	}
	return false;
}
// :402-406
static inline bool intersection(const bounding_sphere& s, const bounding_capsule& c, contact_manifold& outContact)
{
	vec3 closestPoint = closestPoint_PointSegment(s.center, line_segment{ c.positionA, c.positionB });
	return intersection(s, bounding_sphere{ closestPoint, c.radius }, outContact);
}
// :408-449.  NB the reference compares sqDistance <= s.radius * s.radius here (the latent bug SURVEY lists is in
// bounding_volumes.cpp:723's boolean test, not in this contact generator).  Reference line 445 scales the
// un-normalised `normal`; kept verbatim.
static inline bool intersection(const bounding_sphere& s, const bounding_cylinder& c, contact_manifold& outContact)
{
	vec3 ab = c.positionB - c.positionA;
	float t = dot(s.center - c.positionA, ab) / squaredLength(ab);
	if (t >= 0.f && t <= 1.f)
	{
		return intersection(s, bounding_sphere{ lerp(c.positionA, c.positionB, t), c.radius }, outContact);
	}
	vec3 p = (t <= 0.f) ? c.positionA : c.positionB;
	vec3 up = (t <= 0.f) ? -ab : ab;
	vec3 projectedDirToCenter = normalize(cross(cross(up, s.center - p), up));
	vec3 endA = p + projectedDirToCenter * c.radius;
	vec3 endB = p - projectedDirToCenter * c.radius;
	vec3 closestToSphere = closestPoint_PointSegment(s.center, line_segment{ endA, endB });
	vec3 normal = closestToSphere - s.center;
	float sqDistance = squaredLength(normal);
	if (sqDistance <= s.radius * s.radius)
	{
		float distance;
		if (sqDistance == 0.f) { distance = 0.f; outContact.collisionNormal = -normalize(up); }
		else { distance = sqrtf(sqDistance); outContact.collisionNormal = normal / distance; }
		outContact.numContacts = 1;
		outContact.contacts[0].penetrationDepth = s.radius - distance;
		outContact.contacts[0].point = closestToSphere + 0.5f * outContact.contacts[0].penetrationDepth * normal;
		return true;
	}
	return false;
}
// :451-478
static inline bool intersection(const bounding_sphere& s, const bounding_box& a, contact_manifold& outContact)
{
	vec3 p = closestPoint_PointAABB(s.center, a);
	vec3 n = p - s.center;
	float sqDistance = squaredLength(n);
	if (sqDistance <= s.radius * s.radius)
	{
		float dist = 0.f;
		if (sqDistance > 0.f) { dist = sqrtf(sqDistance); n /= dist; }
		else { n = vec3(0.f, 1.f, 0.f); }
		outContact.numContacts = 1;
		outContact.collisionNormal = n;
		outContact.contacts[0].penetrationDepth = s.radius - dist;
		outContact.contacts[0].point = 0.5f * (p + s.center + n * s.radius);
		return true;
	}
	return false;
}
// :480-494
static inline bool intersection(const bounding_sphere& s, const bounding_oriented_box& o, contact_manifold& outContact)
{
	bounding_box aabb = bounding_box::fromCenterRadius(o.center, o.radius);
	bounding_sphere s_ = { conjugate(o.rotation) * (s.center - o.center) + o.center, s.radius };
	if (intersection(s_, aabb, outContact))
	{
		outContact.collisionNormal = o.rotation * outContact.collisionNormal;
		outContact.contacts[0].point = o.rotation * (outContact.contacts[0].point - o.center) + o.center;
		return true;
	}
	return false;
}

// Shared by :523-612 (capsule,capsule), :614-703 (capsule,cylinder), :821-951 (cylinder,cylinder)
template <typename B_t, typename F1, typename F2, typename F3>
static inline bool capsuleLikeVsTube(const bounding_capsule& a, const B_t& b, contact_manifold& outContact, F1 endCapA, F2 endCapB, F3 general)
{
	vec3 aDir = a.positionB - a.positionA;
	vec3 bDir = normalize(b.positionB - b.positionA);
	float aDirLength = length(aDir);
	aDir *= 1.f / aDirLength;
	float parallel = dot(aDir, bDir);
	if (fabsf(parallel) > 0.99f)
	{
		vec3 pAa = a.positionA, pAb = a.positionB, pBa = b.positionA, pBb = b.positionB;
		if (parallel < 0.f) { std::swap(pBa, pBb); }
		vec3 referencePoint = a.positionA;
		float a0 = 0.f, a1 = aDirLength;
		float b0 = dot(aDir, pBa - referencePoint);
		float b1 = dot(aDir, pBb - referencePoint);
		float left = std::max(a0, b0);
		float right = std::min(a1, b1);
		if (right < left)
		{
			if (a0 > b1) { return endCapA(pAa, pBb); }
			else { return endCapB(pAb, pBa); }
		}
		vec3 contactA0 = referencePoint + left * aDir;
		vec3 contactA1 = referencePoint + right * aDir;
		vec3 contactB0 = closestPoint_PointSegment(contactA0, line_segment{ pBa, pBb });
		vec3 contactB1 = contactB0 + (right - left) * aDir;
		vec3 normal = contactB0 - contactA0;
		float d = length(normal);
		if (d < EPSILON) { d = 0.f; normal = vec3(0.f, 1.f, 0.f); }
		else { normal /= d; }
		float radiusSum = a.radius + b.radius;
		float penetration = radiusSum - d;
		if (penetration < 0.f) { return false; }
		outContact.collisionNormal = normal;
		outContact.numContacts = 2;
		outContact.contacts[0].penetrationDepth = penetration;
		outContact.contacts[0].point = (contactA0 + contactB0) * 0.5f;
		outContact.contacts[1].penetrationDepth = penetration;
		outContact.contacts[1].point = (contactA1 + contactB1) * 0.5f;
		return true;
	}
	return general();
}

// :523-612
static inline bool intersection(const bounding_capsule& a, const bounding_capsule& b, contact_manifold& outContact)
{
	return capsuleLikeVsTube(a, b, outContact,
		[&](vec3 pA, vec3 pB) { return intersection(bounding_sphere{ pA, a.radius }, bounding_sphere{ pB, b.radius }, outContact); },
		[&](vec3 pA, vec3 pB) { return intersection(bounding_sphere{ pA, a.radius }, bounding_sphere{ pB, b.radius }, outContact); },
		[&]() {
			vec3 c1, c2;
			closestPoint_SegmentSegment(line_segment{ a.positionA, a.positionB }, line_segment{ b.positionA, b.positionB }, c1, c2);
			return intersection(bounding_sphere{ c1, a.radius }, bounding_sphere{ c2, b.radius }, outContact);
		});
}
// :614-703
static inline bool intersection(const bounding_capsule& a, const bounding_cylinder& b, contact_manifold& outContact)
{
	return capsuleLikeVsTube(a, b, outContact,
		[&](vec3 pA, vec3) { return intersection(bounding_sphere{ pA, a.radius }, b, outContact); },
		[&](vec3 pA, vec3) { return intersection(bounding_sphere{ pA, a.radius }, b, outContact); },
		[&]() {
			vec3 c1, c2;
			closestPoint_SegmentSegment(line_segment{ a.positionA, a.positionB }, line_segment{ b.positionA, b.positionB }, c1, c2);
			return intersection(bounding_sphere{ c1, a.radius }, b, outContact);
		});
}

// Shared tail of :705-769 (capsule,aabb) and :953-1022 (cylinder,aabb): EPA result, then segment clipping if the
// normal is a box face and the tube is parallel to it.
template <typename Tube, typename Support>
static inline bool tubeVsAABB(const Tube& c, const Support& tubeSupport, const bounding_box& a, contact_manifold& outContact)
{
	aabb_support_fn boxSupport{ a };
	gjk_simplex gjkSimplex;
	if (!gjkIntersectionTest(tubeSupport, boxSupport, gjkSimplex)) { return false; }
	epa_result epa;
	epaCollisionInfo(gjkSimplex, tubeSupport, boxSupport, epa); // status ignored (:716-721)
	vec3 normal = epa.normal;
	outContact.collisionNormal = normal;
	outContact.numContacts = 1;
	outContact.contacts[0].penetrationDepth = epa.penetrationDepth;
	outContact.contacts[0].point = epa.point;
	if (fabsf(normal.x) > 0.99f || fabsf(normal.y) > 0.99f || fabsf(normal.z) > 0.99f)
	{
		vec3 axis = normalize(c.positionB - c.positionA);
		if (fabsf(dot(normal, axis)) < 0.01f)
		{
			vec3 clipPlanePoints[4], clipPlaneNormals[4];
			vec4 clipPlanes[4];
			vec3 aabbNormal = -normal;
			vec4 referencePlane = getAABBReferencePlane(a, aabbNormal);
			clipping_polygon polygon;
			polygon.numPoints = 2;
			vec3 pa = c.positionA + normal * c.radius;
			vec3 pb = c.positionB + normal * c.radius;
			polygon.points[0] = { pa, -signedDistanceToPlane(pa, referencePlane) };
			polygon.points[1] = { pb, -signedDistanceToPlane(pb, referencePlane) };
			vec3 aCenter = a.getCenter();
			getAABBClippingPlanes(a.getRadius(), aabbNormal, clipPlanePoints, clipPlaneNormals);
			for (u32 i = 0; i < 4; ++i)
			{
				clipPlanePoints[i] = clipPlanePoints[i] + aCenter;
				clipPlanes[i] = createPlane(clipPlanePoints[i], clipPlaneNormals[i]);
			}
			clipPointsAndBuildContact(polygon, clipPlanes, 4, referencePlane, outContact);
		}
	}
	return true;
}
// :705-769
static inline bool intersection(const bounding_capsule& c, const bounding_box& a, contact_manifold& outContact)
{
	return tubeVsAABB(c, capsule_support_fn{ c }, a, outContact);
}
// :953-1022
static inline bool intersection(const bounding_cylinder& c, const bounding_box& a, contact_manifold& outContact)
{
	return tubeVsAABB(c, cylinder_support_fn{ c }, a, outContact);
}
// :771-790 / :1024-1043
template <typename Tube>
static inline bool tubeVsOBB(const Tube& c, const bounding_oriented_box& o, contact_manifold& outContact)
{
	bounding_box aabb = bounding_box::fromCenterRadius(o.center, o.radius);
	Tube c_ = { conjugate(o.rotation) * (c.positionA - o.center) + o.center, conjugate(o.rotation) * (c.positionB - o.center) + o.center, c.radius };
	if (intersection(c_, aabb, outContact))
	{
		outContact.collisionNormal = o.rotation * outContact.collisionNormal;
		for (u32 i = 0; i < outContact.numContacts; ++i)
		{
			outContact.contacts[i].point = o.rotation * (outContact.contacts[i].point - o.center) + o.center;
		}
		return true;
	}
	return false;
}
static inline bool intersection(const bounding_capsule& c, const bounding_oriented_box& o, contact_manifold& outContact) { return tubeVsOBB(c, o, outContact); }
static inline bool intersection(const bounding_cylinder& c, const bounding_oriented_box& o, contact_manifold& outContact) { return tubeVsOBB(c, o, outContact); }

// :821-951.  Cap-to-cap branch adds a scalar to a vec3 (:891,:897 — latent reference quirk);
// vec3 + float broadcasts in the reference's math (vec3(float) ctor), kept verbatim.
static inline bool intersection(const bounding_cylinder& a, const bounding_cylinder& b, contact_manifold& outContact)
{
	vec3 aDir = a.positionB - a.positionA;
	vec3 bDir = normalize(b.positionB - b.positionA);
	float aDirLength = length(aDir);
	aDir *= 1.f / aDirLength;
	float parallel = dot(aDir, bDir);
	if (fabsf(parallel) > 0.99f)
	{
		vec3 pBa = b.positionA, pBb = b.positionB;
		if (parallel < 0.f) { std::swap(pBa, pBb); }
		vec3 referencePoint = a.positionA;
		float a0 = 0.f, a1 = aDirLength;
		float b0 = dot(aDir, pBa - referencePoint);
		float b1 = dot(aDir, pBb - referencePoint);
		float left = std::max(a0, b0);
		float right = std::min(a1, b1);
		if (right < left) { return false; }
		vec3 contactA0 = referencePoint + left * aDir;
		vec3 contactA1 = referencePoint + right * aDir;
		vec3 contactB0 = closestPoint_PointSegment(contactA0, line_segment{ pBa, pBb });
		vec3 contactB1 = contactB0 + (right - left) * aDir;
		vec3 normal = contactB0 - contactA0;
		float d = length(normal);
		float radiusSum = a.radius + b.radius;
		float penetration = radiusSum - d;
		if (penetration < 0.f) { return false; }
		float capPenetration = right - left;
		if (capPenetration < penetration)
		{
			outContact.numContacts = 1;
			outContact.contacts[0].penetrationDepth = capPenetration;
			if (b0 > a0) { outContact.collisionNormal = aDir; outContact.contacts[0].point = a.positionB - vec3(capPenetration * 0.5f); }
			else { outContact.collisionNormal = -aDir; outContact.contacts[0].point = a.positionA + vec3(capPenetration * 0.5f); }
		}
		else
		{
			if (d < EPSILON) { d = 0.f; normal = vec3(0.f, 1.f, 0.f); }
			else { normal /= d; }
			outContact.collisionNormal = normal;
			outContact.numContacts = 2;
			outContact.contacts[0].penetrationDepth = penetration;
			outContact.contacts[0].point = (contactA0 + contactB0) * 0.5f;
			outContact.contacts[1].penetrationDepth = penetration;
			outContact.contacts[1].point = (contactA1 + contactB1) * 0.5f;
		}
		return true;
	}
	else
	{
		cylinder_support_fn sa{ a }, sb{ b };
		gjk_simplex gjkSimplex;
		if (!gjkIntersectionTest(sa, sb, gjkSimplex)) { return false; }
		epa_result epa;
		epaCollisionInfo(gjkSimplex, sa, sb, epa);
		outContact.collisionNormal = epa.normal;
		outContact.numContacts = 1;
		outContact.contacts[0].penetrationDepth = epa.penetrationDepth;
		outContact.contacts[0].point = epa.point;
		return true;
	}
}

// :1074-1140
static inline bool intersection(const bounding_box& a, const bounding_box& b, contact_manifold& outContact)
{
	vec3 centerA = a.getCenter(), centerB = b.getCenter();
	vec3 radiusA = a.getRadius(), radiusB = b.getRadius();
	vec3 d = centerB - centerA;
	vec3 p = (radiusB + radiusA) - vabs(d);
	if (p.x < 0.f || p.y < 0.f || p.z < 0.f) { return false; }
	u32 minElement = (p.x < p.y) ? ((p.x < p.z) ? 0 : 2) : ((p.y < p.z) ? 1 : 2);
	float s = d[minElement] < 0.f ? -1.f : 1.f;
	float penetration = p[minElement] * s;
	vec3 normal(0.f);
	normal[minElement] = s;
	outContact.collisionNormal = normal;
	outContact.numContacts = 4;
	u32 axis0 = (minElement + 1) % 3;
	u32 axis1 = (minElement + 2) % 3;
	float min0 = std::max(a.minCorner[axis0], b.minCorner[axis0]);
	float min1 = std::max(a.minCorner[axis1], b.minCorner[axis1]);
	float max0 = std::min(a.maxCorner[axis0], b.maxCorner[axis0]);
	float max1 = std::min(a.maxCorner[axis1], b.maxCorner[axis1]);
	float depth = centerA[minElement] + radiusA[minElement] - penetration * 0.5f;
	const float c0[4] = { min0, min0, max0, max0 };
	const float c1[4] = { min1, max1, min1, max1 };
	for (u32 i = 0; i < 4; ++i)
	{
		outContact.contacts[i].penetrationDepth = penetration;
		outContact.contacts[i].point = vec3(0.f);
		outContact.contacts[i].point[axis0] = c0[i];
		outContact.contacts[i].point[axis1] = c1[i];
		outContact.contacts[i].point[minElement] = depth;
	}
	return true;
}

// :1179-1527
static inline bool intersection(const bounding_oriented_box& a, const bounding_oriented_box& b, contact_manifold& outContact)
{
	vec3 axesA[3] = { a.rotation * vec3(1.f, 0.f, 0.f), a.rotation * vec3(0.f, 1.f, 0.f), a.rotation * vec3(0.f, 0.f, 1.f) };
	vec3 axesB[3] = { b.rotation * vec3(1.f, 0.f, 0.f), b.rotation * vec3(0.f, 1.f, 0.f), b.rotation * vec3(0.f, 0.f, 1.f) };

	mat3 r;
	r.m00 = dot(axesA[0], axesB[0]); r.m10 = dot(axesA[1], axesB[0]); r.m20 = dot(axesA[2], axesB[0]);
	r.m01 = dot(axesA[0], axesB[1]); r.m11 = dot(axesA[1], axesB[1]); r.m21 = dot(axesA[2], axesB[1]);
	r.m02 = dot(axesA[0], axesB[2]); r.m12 = dot(axesA[1], axesB[2]); r.m22 = dot(axesA[2], axesB[2]);

	vec3 tw = b.center - a.center;
	vec3 t = conjugate(a.rotation) * tw;

	bool parallel = false;
	mat3 absR;
	for (u32 i = 0; i < 9; ++i)
	{
		absR.m()[i] = fabsf(r.m()[i]) + EPSILON;
		if (absR.m()[i] >= 0.99f) { parallel = true; }
	}

	float ra, rb;
	float minPenetration = FLT_MAX;
	vec3 normal;
	bool bFace = false;

	for (u32 i = 0; i < 3; ++i)
	{
		ra = a.radius[i];
		rb = dot(row(absR, i), b.radius);
		float d = t[i];
		float penetration = ra + rb - fabsf(d);
		if (penetration < 0.f) { return false; }
		if (penetration < minPenetration) { minPenetration = penetration; normal = vec3(0.f); normal[i] = 1.f; }
	}
	for (u32 i = 0; i < 3; ++i)
	{
		ra = dot(col(absR, i), a.radius);
		rb = b.radius[i];
		float d = dot(col(r, i), t);
		float penetration = ra + rb - fabsf(d);
		if (penetration < 0.f) { return false; }
		if (penetration < minPenetration) { minPenetration = penetration; normal = vec3(0.f); normal[i] = 1.f; bFace = true; }
	}

	bool edgeCollision = false;
	vec3 edgeNormal;

	if (!parallel)
	{
		float penetration; vec3 n; float l;
#define ORC_EDGE_TEST(RA, RB, DIST, NX, NY, NZ) \
		ra = RA; rb = RB; \
		penetration = ra + rb - fabsf(DIST); \
		if (penetration < 0.f) { return false; } \
		n = vec3(NX, NY, NZ); \
		l = 1.f / length(n); \
		penetration *= l; \
		if (penetration < minPenetration) { minPenetration = penetration; edgeNormal = n * l; edgeCollision = true; }

		ORC_EDGE_TEST(a.radius.y * absR.m20 + a.radius.z * absR.m10, b.radius.y * absR.m02 + b.radius.z * absR.m01, t.z * r.m10 - t.y * r.m20, 0.f, -r.m20, r.m10) // a.x x b.x
		ORC_EDGE_TEST(a.radius.y * absR.m21 + a.radius.z * absR.m11, b.radius.x * absR.m02 + b.radius.z * absR.m00, t.z * r.m11 - t.y * r.m21, 0.f, -r.m21, r.m11) // a.x x b.y
		ORC_EDGE_TEST(a.radius.y * absR.m22 + a.radius.z * absR.m12, b.radius.x * absR.m01 + b.radius.y * absR.m00, t.z * r.m12 - t.y * r.m22, 0.f, -r.m22, r.m12) // a.x x b.z
		ORC_EDGE_TEST(a.radius.x * absR.m20 + a.radius.z * absR.m00, b.radius.y * absR.m12 + b.radius.z * absR.m11, t.x * r.m20 - t.z * r.m00, r.m20, 0.f, -r.m00) // a.y x b.x
		ORC_EDGE_TEST(a.radius.x * absR.m21 + a.radius.z * absR.m01, b.radius.x * absR.m12 + b.radius.z * absR.m10, t.x * r.m21 - t.z * r.m01, r.m21, 0.f, -r.m01) // a.y x b.y
		ORC_EDGE_TEST(a.radius.x * absR.m22 + a.radius.z * absR.m02, b.radius.x * absR.m11 + b.radius.y * absR.m10, t.x * r.m22 - t.z * r.m02, r.m22, 0.f, -r.m02) // a.y x b.z
		ORC_EDGE_TEST(a.radius.x * absR.m10 + a.radius.y * absR.m00, b.radius.y * absR.m22 + b.radius.z * absR.m21, t.y * r.m00 - t.x * r.m10, -r.m10, r.m00, 0.f) // a.z x b.x
		ORC_EDGE_TEST(a.radius.x * absR.m11 + a.radius.y * absR.m01, b.radius.x * absR.m22 + b.radius.z * absR.m20, t.y * r.m01 - t.x * r.m11, -r.m11, r.m01, 0.f) // a.z x b.y
		ORC_EDGE_TEST(a.radius.x * absR.m12 + a.radius.y * absR.m02, b.radius.x * absR.m21 + b.radius.y * absR.m20, t.y * r.m02 - t.x * r.m12, -r.m12, r.m02, 0.f) // a.z x b.z
#undef ORC_EDGE_TEST
	}

	bool faceCollision = !edgeCollision;
	if (faceCollision) { if (bFace) { normal = r * normal; } }
	else { normal = edgeNormal; }
	normal = a.rotation * normal;
	if (dot(normal, tw) < 0.f) { normal = -normal; }
	outContact.collisionNormal = normal;

	if (faceCollision)
	{
		vec3 clipPlanePoints[4], clipPlaneNormals[4];
		clipping_polygon polygon;
		vec4 plane;
		if (!bFace)
		{
			getAABBClippingPlanes(a.radius, conjugate(a.rotation) * normal, clipPlanePoints, clipPlaneNormals);
			getAABBIncidentVertices(b.radius, conjugate(b.rotation) * normal, polygon);
			for (u32 i = 0; i < 4; ++i)
			{
				clipPlanePoints[i] = a.rotation * clipPlanePoints[i] + a.center;
				clipPlaneNormals[i] = a.rotation * clipPlaneNormals[i];
				polygon.points[i].vertex = b.rotation * polygon.points[i].vertex + b.center;
			}
			obb_support_fn support{ a };
			vec3 referencePlanePoint = support(normal);
			plane = createPlane(referencePlanePoint, normal);
		}
		else
		{
			getAABBClippingPlanes(b.radius, conjugate(b.rotation) * -normal, clipPlanePoints, clipPlaneNormals);
			getAABBIncidentVertices(a.radius, conjugate(a.rotation) * -normal, polygon);
			for (u32 i = 0; i < 4; ++i)
			{
				clipPlanePoints[i] = b.rotation * clipPlanePoints[i] + b.center;
				clipPlaneNormals[i] = b.rotation * clipPlaneNormals[i];
				polygon.points[i].vertex = a.rotation * polygon.points[i].vertex + a.center;
			}
			obb_support_fn support{ b };
			vec3 referencePlanePoint = support(-normal);
			plane = createPlane(referencePlanePoint, -normal);
		}
		vec4 clipPlanes[4];
		for (u32 i = 0; i < 4; ++i)
		{
			clipPlanes[i] = createPlane(clipPlanePoints[i], clipPlaneNormals[i]);
			polygon.points[i].penetrationDepth = -signedDistanceToPlane(polygon.points[i].vertex, plane);
		}
		if (!clipPointsAndBuildContact(polygon, clipPlanes, 4, plane, outContact)) { return false; }
	}
	else
	{
		vec3 a0, a1, b0, b1;
		getAABBIncidentEdge(a.radius, conjugate(a.rotation) * normal, a0, a1);
		getAABBIncidentEdge(b.radius, conjugate(b.rotation) * -normal, b0, b1);
		a0 = a.rotation * a0 + a.center; a1 = a.rotation * a1 + a.center;
		b0 = b.rotation * b0 + b.center; b1 = b.rotation * b1 + b.center;
		vec3 pa, pb;
		float sqDistance = closestPoint_SegmentSegment(line_segment{ a0, a1 }, line_segment{ b0, b1 }, pa, pb);
		outContact.numContacts = 1;
		outContact.contacts[0].penetrationDepth = sqrtf(sqDistance);
		outContact.contacts[0].point = (pa + pb) * 0.5f;
	}
	return true;
}
// :1142-1148
static inline bool intersection(const bounding_box& a, const bounding_oriented_box& b, contact_manifold& outContact)
{
	return intersection(bounding_oriented_box{ quat(0.f, 0.f, 0.f, 1.f), a.getCenter(), a.getRadius() }, b, outContact);
}

// Every (*, hull) pair: GJK + EPA, single contact, EPA status ignored — collision_narrow.cpp:496-520 (sphere), 792-818 (capsule),
// 1045-1071 (cylinder), 1150-1176 (aabb), 1529-1556 (obb), 1559-1584 (hull).
template <typename Support>
static inline bool supportVsHull(const Support& a, const bounding_hull& h, contact_manifold& outContact)
{
	hull_support_fn hullSupport{ h };
	gjk_simplex gjkSimplex;
	if (!gjkIntersectionTest(a, hullSupport, gjkSimplex)) { return false; }
	epa_result epa;
	epaCollisionInfo(gjkSimplex, a, hullSupport, epa);
	outContact.collisionNormal = epa.normal;
	outContact.numContacts = 1;
	outContact.contacts[0].penetrationDepth = epa.penetrationDepth;
	outContact.contacts[0].point = epa.point;
	return true;
}

// Dispatch on (typeA <= typeB), the 21 pairs of collision_narrow.cpp:2473-2570.
static inline bool intersectColliders(const collider_union& A, const collider_union& B, contact_manifold& m)
{
	switch (A.type)
	{
		case collider_type_sphere:
			switch (B.type)
			{
				case collider_type_sphere: return intersection(A.sphere(), B.sphere(), m);
				case collider_type_capsule: return intersection(A.sphere(), B.capsule(), m);
				case collider_type_cylinder: return intersection(A.sphere(), B.cylinder(), m);
				case collider_type_aabb: return intersection(A.sphere(), B.aabb(), m);
				case collider_type_obb: return intersection(A.sphere(), B.obb(), m);
				case collider_type_hull: return supportVsHull(sphere_support_fn{ A.sphere() }, B.hull(), m);
			}
			break;
		case collider_type_capsule:
			switch (B.type)
			{
				case collider_type_capsule: return intersection(A.capsule(), B.capsule(), m);
				case collider_type_cylinder: return intersection(A.capsule(), B.cylinder(), m);
				case collider_type_aabb: return intersection(A.capsule(), B.aabb(), m);
				case collider_type_obb: return intersection(A.capsule(), B.obb(), m);
				case collider_type_hull: return supportVsHull(capsule_support_fn{ A.capsule() }, B.hull(), m);
			}
			break;
		case collider_type_cylinder:
			switch (B.type)
			{
				case collider_type_cylinder: return intersection(A.cylinder(), B.cylinder(), m);
				case collider_type_aabb: return intersection(A.cylinder(), B.aabb(), m);
				case collider_type_obb: return intersection(A.cylinder(), B.obb(), m);
				case collider_type_hull: return supportVsHull(cylinder_support_fn{ A.cylinder() }, B.hull(), m);
			}
			break;
		case collider_type_aabb:
			switch (B.type)
			{
				case collider_type_aabb: return intersection(A.aabb(), B.aabb(), m);
				case collider_type_obb: return intersection(A.aabb(), B.obb(), m);
				case collider_type_hull: return supportVsHull(aabb_support_fn{ A.aabb() }, B.hull(), m);
			}
			break;
		case collider_type_obb:
			if (B.type == collider_type_obb) return intersection(A.obb(), B.obb(), m);
			if (B.type == collider_type_hull) return supportVsHull(obb_support_fn{ A.obb() }, B.hull(), m);
			break;
		case collider_type_hull:
			if (B.type == collider_type_hull) return supportVsHull(hull_support_fn{ A.hull() }, B.hull(), m);
			break;
	}
	return false;
}

// ---------------------------------------------------------------------------------------------------
// Boolean overlap tests for the non-collision interactions (force fields, triggers): overlapCheck,
// collision_narrow.cpp:1593-1689, over bounding_volumes.h:301-363 and bounding_volumes.cpp:704-835, 1079-1244.
// ---------------------------------------------------------------------------------------------------
static inline bool sphereVsSphere(const bounding_sphere& a, const bounding_sphere& b) // bounding_volumes.h:301-307
{
	vec3 d = a.center - b.center;
	float dist2 = dot(d, d);
	float radiusSum = a.radius + b.radius;
	return dist2 <= radiusSum * radiusSum;
}
static inline bool sphereVsCapsule(const bounding_sphere& s, const bounding_capsule& c) // :314-318
{
	vec3 closestPoint = closestPoint_PointSegment(s.center, line_segment{ c.positionA, c.positionB });
	return sphereVsSphere(s, bounding_sphere{ closestPoint, c.radius });
}
static inline bool sphereVsCylinder(const bounding_sphere& s, const bounding_cylinder& c) // bounding_volumes.cpp:704-724
{
	vec3 ab = c.positionB - c.positionA;
	float t = dot(s.center - c.positionA, ab) / squaredLength(ab);
	if (t >= 0.f && t <= 1.f) { return sphereVsSphere(s, bounding_sphere{ lerp(c.positionA, c.positionB, t), c.radius }); }
	vec3 p = (t <= 0.f) ? c.positionA : c.positionB;
	vec3 up = (t <= 0.f) ? -ab : ab;
	vec3 projectedDirToCenter = normalize(cross(cross(up, s.center - p), up));
	vec3 endA = p + projectedDirToCenter * c.radius;
	vec3 endB = p - projectedDirToCenter * c.radius;
	vec3 closestToSphere = closestPoint_PointSegment(s.center, line_segment{ endA, endB });
	float sqDistance = squaredLength(closestToSphere - s.center);
	return sqDistance <= s.radius; // sic: the squared distance against the plain radius (:723)
}
static inline bool sphereVsAABB(const bounding_sphere& s, const bounding_box& a) // bounding_volumes.h:320-326
{
	vec3 p = closestPoint_PointAABB(s.center, a);
	vec3 n = p - s.center;
	float sqDistance = squaredLength(n);
	return sqDistance <= s.radius * s.radius;
}
static inline bool sphereVsOBB(const bounding_sphere& s, const bounding_oriented_box& o) // :328-336
{
	bounding_box aabb = bounding_box::fromCenterRadius(o.center, o.radius);
	bounding_sphere s_ = { conjugate(o.rotation) * (s.center - o.center) + o.center, s.radius };
	return sphereVsAABB(s_, aabb);
}
static inline bool capsuleVsCapsule(const bounding_capsule& a, const bounding_capsule& b) // :338-343
{
	vec3 closestPoint1, closestPoint2;
	closestPoint_SegmentSegment(line_segment{ a.positionA, a.positionB }, line_segment{ b.positionA, b.positionB }, closestPoint1, closestPoint2);
	return sphereVsSphere(bounding_sphere{ closestPoint1, a.radius }, bounding_sphere{ closestPoint2, b.radius });
}
static inline bool capsuleVsCylinder(const bounding_capsule& a, const bounding_cylinder& b) // :345-350
{
	vec3 closestPoint1, closestPoint2;
	closestPoint_SegmentSegment(line_segment{ a.positionA, a.positionB }, line_segment{ b.positionA, b.positionB }, closestPoint1, closestPoint2);
	return sphereVsCylinder(bounding_sphere{ closestPoint1, a.radius }, b);
}
template <typename A, typename B> static inline bool gjkOverlap(const A& a, const B& b) { gjk_simplex simplex; return gjkIntersectionTest(a, b, simplex); }
static inline bool capsuleVsAABB(const bounding_capsule& c, const bounding_box& b) { return gjkOverlap(capsule_support_fn{ c }, aabb_support_fn{ b }); } // bounding_volumes.cpp:742-749
static inline bool capsuleVsOBB(const bounding_capsule& c, const bounding_oriented_box& o) // :751-760
{
	bounding_box aabb = bounding_box::fromCenterRadius(o.center, o.radius);
	bounding_capsule c_ = { conjugate(o.rotation) * (c.positionA - o.center) + o.center, conjugate(o.rotation) * (c.positionB - o.center) + o.center, c.radius };
	return capsuleVsAABB(c_, aabb);
}
static inline bool cylinderVsCylinder(const bounding_cylinder& a, const bounding_cylinder& b) { return gjkOverlap(cylinder_support_fn{ a }, cylinder_support_fn{ b }); } // :790-797
static inline bool cylinderVsAABB(const bounding_cylinder& c, const bounding_box& b) { return gjkOverlap(cylinder_support_fn{ c }, aabb_support_fn{ b }); } // :799-806
static inline bool cylinderVsOBB(const bounding_cylinder& c, const bounding_oriented_box& o) // :808-817
{
	bounding_box aabb = bounding_box::fromCenterRadius(o.center, o.radius);
	bounding_cylinder c_ = { conjugate(o.rotation) * (c.positionA - o.center) + o.center, conjugate(o.rotation) * (c.positionB - o.center) + o.center, c.radius };
	return cylinderVsAABB(c_, aabb);
}
// aabbVsAABB (bounding_volumes.h:352-358): oshapes.h
static inline bool obbVsOBB(const bounding_oriented_box& a, const bounding_oriented_box& b) // bounding_volumes.cpp:1079-1199
{
	float ra, rb, penetration;
	vec3 axesA[3] = { a.rotation * vec3(1.f, 0.f, 0.f), a.rotation * vec3(0.f, 1.f, 0.f), a.rotation * vec3(0.f, 0.f, 1.f) };
	vec3 axesB[3] = { b.rotation * vec3(1.f, 0.f, 0.f), b.rotation * vec3(0.f, 1.f, 0.f), b.rotation * vec3(0.f, 0.f, 1.f) };
	float r[3][3], absR[3][3]; // r[row][column] = dot(axesA[row], axesB[column])
	for (u32 i = 0; i < 3; ++i) for (u32 j = 0; j < 3; ++j) { r[i][j] = dot(axesA[i], axesB[j]); absR[i][j] = fabsf(r[i][j]) + EPSILON; }
	vec3 tw = b.center - a.center;
	vec3 t = conjugate(a.rotation) * tw;
	const float tA[3] = { t.x, t.y, t.z }, rA[3] = { a.radius.x, a.radius.y, a.radius.z }, rB[3] = { b.radius.x, b.radius.y, b.radius.z };
	for (u32 i = 0; i < 3; ++i) // a's faces
	{
		ra = rA[i];
		rb = dot(vec3(absR[i][0], absR[i][1], absR[i][2]), b.radius);
		penetration = ra + rb - fabsf(tA[i]);
		if (penetration < 0.f) { return false; }
	}
	for (u32 i = 0; i < 3; ++i) // b's faces
	{
		ra = dot(vec3(absR[0][i], absR[1][i], absR[2][i]), a.radius);
		rb = rB[i];
		float d = dot(vec3(r[0][i], r[1][i], r[2][i]), t);
		penetration = ra + rb - fabsf(d);
		if (penetration < 0.f) { return false; }
	}
#define ORC_EDGE(RA, RB, D) ra = (RA); rb = (RB); penetration = ra + rb - fabsf(D); if (penetration < 0.f) { return false; }
	ORC_EDGE(rA[1] * absR[2][0] + rA[2] * absR[1][0], rB[1] * absR[0][2] + rB[2] * absR[0][1], t.z * r[1][0] - t.y * r[2][0]) // a.x x b.x
	ORC_EDGE(rA[1] * absR[2][1] + rA[2] * absR[1][1], rB[0] * absR[0][2] + rB[2] * absR[0][0], t.z * r[1][1] - t.y * r[2][1]) // a.x x b.y
	ORC_EDGE(rA[1] * absR[2][2] + rA[2] * absR[1][2], rB[0] * absR[0][1] + rB[1] * absR[0][0], t.z * r[1][2] - t.y * r[2][2]) // a.x x b.z
	ORC_EDGE(rA[0] * absR[2][0] + rA[2] * absR[0][0], rB[1] * absR[1][2] + rB[2] * absR[1][1], t.x * r[2][0] - t.z * r[0][0]) // a.y x b.x
	ORC_EDGE(rA[0] * absR[2][1] + rA[2] * absR[0][1], rB[0] * absR[1][2] + rB[2] * absR[1][0], t.x * r[2][1] - t.z * r[0][1]) // a.y x b.y
	ORC_EDGE(rA[0] * absR[2][2] + rA[2] * absR[0][2], rB[0] * absR[1][1] + rB[1] * absR[1][0], t.x * r[2][2] - t.z * r[0][2]) // a.y x b.z
	ORC_EDGE(rA[0] * absR[1][0] + rA[1] * absR[0][0], rB[1] * absR[2][2] + rB[2] * absR[2][1], t.y * r[0][0] - t.x * r[1][0]) // a.z x b.x
	ORC_EDGE(rA[0] * absR[1][1] + rA[1] * absR[0][1], rB[0] * absR[2][2] + rB[2] * absR[2][0], t.y * r[0][1] - t.x * r[1][1]) // a.z x b.y
	ORC_EDGE(rA[0] * absR[1][2] + rA[1] * absR[0][2], rB[0] * absR[2][1] + rB[1] * absR[2][0], t.y * r[0][2] - t.x * r[1][2]) // a.z x b.z
#undef ORC_EDGE
	return true;
}
static inline bool aabbVsOBB(const bounding_box& a, const bounding_oriented_box& o) // bounding_volumes.h:360-363
{
	return obbVsOBB(bounding_oriented_box{ quat(0.f, 0.f, 0.f, 1.f), a.getCenter(), a.getRadius() }, o);
}
template <typename A> static inline bool supportOverlapsHull(const A& a, const bounding_hull& h) { return gjkOverlap(a, hull_support_fn{ h }); } // bounding_volumes.cpp:726-733, 762-769, 819-835, 1201-1244

// overlapCheck (collision_narrow.cpp:1593-1689) on (typeA <= typeB)
static inline bool overlapColliders(const collider_union& A, const collider_union& B)
{
	switch (A.type)
	{
		case collider_type_sphere:
			switch (B.type)
			{
				case collider_type_sphere: return sphereVsSphere(A.sphere(), B.sphere());
				case collider_type_capsule: return sphereVsCapsule(A.sphere(), B.capsule());
				case collider_type_cylinder: return sphereVsCylinder(A.sphere(), B.cylinder());
				case collider_type_aabb: return sphereVsAABB(A.sphere(), B.aabb());
				case collider_type_obb: return sphereVsOBB(A.sphere(), B.obb());
				case collider_type_hull: return supportOverlapsHull(sphere_support_fn{ A.sphere() }, B.hull());
			}
			break;
		case collider_type_capsule:
			switch (B.type)
			{
				case collider_type_capsule: return capsuleVsCapsule(A.capsule(), B.capsule());
				case collider_type_cylinder: return capsuleVsCylinder(A.capsule(), B.cylinder());
				case collider_type_aabb: return capsuleVsAABB(A.capsule(), B.aabb());
				case collider_type_obb: return capsuleVsOBB(A.capsule(), B.obb());
				case collider_type_hull: return supportOverlapsHull(capsule_support_fn{ A.capsule() }, B.hull());
			}
			break;
		case collider_type_cylinder:
			switch (B.type)
			{
				case collider_type_cylinder: return cylinderVsCylinder(A.cylinder(), B.cylinder());
				case collider_type_aabb: return cylinderVsAABB(A.cylinder(), B.aabb());
				case collider_type_obb: return cylinderVsOBB(A.cylinder(), B.obb());
				case collider_type_hull: return supportOverlapsHull(cylinder_support_fn{ A.cylinder() }, B.hull());
			}
			break;
		case collider_type_aabb:
			switch (B.type)
			{
				case collider_type_aabb: return aabbVsAABB(A.aabb(), B.aabb());
				case collider_type_obb: return aabbVsOBB(A.aabb(), B.obb());
				case collider_type_hull: return supportOverlapsHull(aabb_support_fn{ A.aabb() }, B.hull());
			}
			break;
		case collider_type_obb:
			if (B.type == collider_type_obb) return obbVsOBB(A.obb(), B.obb());
			if (B.type == collider_type_hull) return supportOverlapsHull(obb_support_fn{ A.obb() }, B.hull());
			break;
		case collider_type_hull:
			if (B.type == collider_type_hull) return supportOverlapsHull(hull_support_fn{ A.hull() }, B.hull());
			break;
	}
	return false;
}

} // namespace orc
