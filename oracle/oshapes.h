// ORACLE — TEST INFRASTRUCTURE ONLY (see omath.h header).
// Shapes, AABB transforms and closest-point routines restated from the reference's
// src/physics/bounding_volumes.{h,cpp}; collider record from src/physics/physics.h:40-171.
#pragma once
#include <vector>
#include "omath.h"

namespace orc {

struct line_segment { vec3 a, b; };
struct bounding_sphere { vec3 center; float radius; };
struct bounding_capsule { vec3 positionA, positionB; float radius; };
struct bounding_cylinder { vec3 positionA, positionB; float radius; };
struct bounding_oriented_box;

// bounding_volumes.h:104-133, bounding_volumes.cpp:25-149
struct bounding_box
{
	vec3 minCorner, maxCorner;

	void grow(vec3 o) { minCorner = vmin(minCorner, o); maxCorner = vmax(maxCorner, o); }
	vec3 getCenter() const { return (minCorner + maxCorner) * 0.5f; }
	vec3 getRadius() const { return (maxCorner - minCorner) * 0.5f; }
	static bounding_box negativeInfinity() { return bounding_box{ vec3(FLT_MAX), vec3(-FLT_MAX) }; }
	static bounding_box fromCenterRadius(vec3 c, vec3 r) { return bounding_box{ c - r, c + r }; }
	static bounding_box fromCenterRadius(vec3 c, float r) { return fromCenterRadius(c, vec3(r)); }
	static bounding_box fromMinMax(vec3 a, vec3 b) { return bounding_box{ a, b }; }

	// bounding_volumes.cpp:58-70 (8-corner grow, this corner order)
	bounding_box transformToAABB(quat rotation, vec3 translation) const
	{
		bounding_box result = negativeInfinity();
		result.grow(rotation * minCorner + translation);
		result.grow(rotation * vec3(maxCorner.x, minCorner.y, minCorner.z) + translation);
		result.grow(rotation * vec3(minCorner.x, maxCorner.y, minCorner.z) + translation);
		result.grow(rotation * vec3(maxCorner.x, maxCorner.y, minCorner.z) + translation);
		result.grow(rotation * vec3(minCorner.x, minCorner.y, maxCorner.z) + translation);
		result.grow(rotation * vec3(maxCorner.x, minCorner.y, maxCorner.z) + translation);
		result.grow(rotation * vec3(minCorner.x, maxCorner.y, maxCorner.z) + translation);
		result.grow(rotation * maxCorner + translation);
		return result;
	}
	inline bounding_oriented_box transformToOBB(quat rotation, vec3 translation) const;
	float volume() const { vec3 d = maxCorner - minCorner; return d.x * d.y * d.z; }
};

// bounding_volumes.h:135-151, bounding_volumes.cpp:127-142
struct bounding_oriented_box
{
	quat rotation; vec3 center; vec3 radius;

	bounding_box transformToAABB(quat rot, vec3 translation) const
	{
		bounding_box bb = { -radius, radius };
		return bb.transformToAABB(rot * this->rotation, rot * center + translation);
	}
	bounding_oriented_box transformToOBB(quat rot, vec3 translation) const
	{
		return bounding_oriented_box{ rot * this->rotation, rot * center + translation, radius };
	}
	float volume() const { vec3 d = radius * 2.f; return d.x * d.y * d.z; }
};

// bounding_volumes.cpp:72-79
inline bounding_oriented_box bounding_box::transformToOBB(quat rotation, vec3 translation) const
{
	bounding_oriented_box obb;
	obb.center = rotation * getCenter() + translation;
	obb.radius = getRadius();
	obb.rotation = rotation;
	return obb;
}

// bounding_volumes.h:27-72 (volumes; M_PI is a float-suffixed macro in core/math.h)
static inline float sphereVolume(float radius)
{
	float sqRadius = radius * radius;
	float sqRadiusPI = M_PI_F * sqRadius;
	return 4.f / 3.f * sqRadiusPI * radius;
}
static inline float capsuleVolume(const bounding_capsule& c)
{
	float sqRadius = c.radius * c.radius;
	float sqRadiusPI = M_PI_F * sqRadius;
	float sphereV = 4.f / 3.f * sqRadiusPI * c.radius;
	float height = length(c.positionA - c.positionB);
	return sphereV + sqRadiusPI * height;
}
static inline float cylinderVolume(const bounding_cylinder& c)
{
	float sqRadiusPI = M_PI_F * c.radius * c.radius;
	float height = length(c.positionA - c.positionB);
	return sqRadiusPI * height;
}

// bounding_volumes.h:166-178
static inline vec4 createPlane(vec3 point, vec3 normal) { float d = -dot(normal, point); return vec4(normal, d); }
// bounding_volumes.h:296-299: dot(vec4(p,1), plane) — vec4 dot is addElements of the 4 products (x*x+y*y)+(z*z+w*w)
// in the SSE hadd form; we use the plain left-to-right sum (fp order difference ≤ 1 ulp, below stated tolerances).
static inline float signedDistanceToPlane(vec3 p, vec4 plane) { return p.x * plane.x + p.y * plane.y + p.z * plane.z + plane.w; }

// bounding_volumes.h:352-358
static inline bool aabbVsAABB(const bounding_box& a, const bounding_box& b)
{
	if (a.maxCorner.x < b.minCorner.x || a.minCorner.x > b.maxCorner.x) return false;
	if (a.maxCorner.y < b.minCorner.y || a.minCorner.y > b.maxCorner.y) return false;
	if (a.maxCorner.z < b.minCorner.z || a.minCorner.z > b.maxCorner.z) return false;
	return true;
}
// bounding_volumes.h:365-371
static inline vec3 closestPoint_PointSegment(vec3 q, line_segment l)
{
	vec3 ab = l.b - l.a;
	float t = dot(q - l.a, ab) / squaredLength(ab);
	t = clampf(t, 0.f, 1.f);
	return l.a + t * ab;
}
// bounding_volumes.h:373-383
static inline vec3 closestPoint_PointAABB(vec3 q, const bounding_box& aabb)
{
	vec3 result;
	for (u32 i = 0; i < 3; ++i)
	{
		float v = q[i];
		if (v < aabb.minCorner[i]) v = aabb.minCorner[i];
		if (v > aabb.maxCorner[i]) v = aabb.maxCorner[i];
		result[i] = v;
	}
	return result;
}
// bounding_volumes.cpp:1251-1315
static inline float closestPoint_SegmentSegment(const line_segment& l1, const line_segment& l2, vec3& c1, vec3& c2)
{
	float s, t;
	vec3 d1 = l1.b - l1.a;
	vec3 d2 = l2.b - l2.a;
	vec3 r = l1.a - l2.a;
	float a = dot(d1, d1);
	float e = dot(d2, d2);
	float f = dot(d2, r);
	if (a <= EPSILON && e <= EPSILON)
	{
		s = t = 0.0f;
		c1 = l1.a; c2 = l2.a;
		return dot(c1 - c2, c1 - c2);
	}
	if (a <= EPSILON)
	{
		s = 0.0f;
		t = f / e;
		t = clampf(t, 0.f, 1.f);
	}
	else
	{
		float c = dot(d1, r);
		if (e <= EPSILON)
		{
			t = 0.0f;
			s = clampf(-c / a, 0.f, 1.f);
		}
		else
		{
			float b = dot(d1, d2);
			float denom = a * e - b * b;
			if (denom != 0.f) s = clampf((b * f - c * e) / denom, 0.f, 1.f);
			else s = 0.0f;
			t = (b * s + f) / e;
			if (t < 0.f) { t = 0.f; s = clampf(-c / a, 0.f, 1.f); }
			else if (t > 1.0f) { t = 1.f; s = clampf((b - c) / a, 0.f, 1.f); }
		}
	}
	c1 = l1.a + d1 * s;
	c2 = l2.a + d2 * t;
	return squaredLength(c1 - c2);
}

// bounding_volumes.h:196-231.  Convex hull: shared geometry (vertices + triangles; edges are not used on the rigid-body path) and a
// placed instance.  The reference keeps the geometries in a global table (physics.cpp:47-84) and a pointer in the world-space
// collider; here the table belongs to the world and hulls carry the index.
struct bounding_hull_face { u32 a, b, c; };
struct bounding_hull_geometry
{
	std::vector<vec3> vertices;
	std::vector<bounding_hull_face> faces;
	bounding_box aabb;
};
struct bounding_hull { quat rotation; vec3 position; u32 geometryIndex; };
// the geometry table the narrowphase reads hull vertices from (set by the world around its narrowphase calls)
inline const std::vector<bounding_hull_geometry>*& hullGeometryTable() { static const std::vector<bounding_hull_geometry>* t = nullptr; return t; }

// physics.h:40-106.  The reference's collider_union is a 64-byte tagged union with u16 indices; we widen
// objectIndex to u32 (SURVEY finding 1).
enum collider_type : u32
{
	collider_type_sphere, collider_type_capsule, collider_type_cylinder, collider_type_aabb, collider_type_obb, collider_type_hull,
	collider_type_count,
};
enum physics_object_type : u32
{
	physics_object_type_rigid_body, physics_object_type_static_collider, physics_object_type_force_field, physics_object_type_trigger,
};
struct physics_material { float restitution, friction, density; };

struct collider_union
{
	// Shape payload: 10 floats, interpreted per type (sphere: c,r | capsule/cylinder: A,B,r | aabb: min,max | obb: q,c,r |
	// hull: q, position, geometry index as a float value).
	float shape[10];
	physics_material material;
	u32 type;
	u32 objectType;
	u32 objectIndex;

	bounding_sphere sphere() const { return bounding_sphere{ vec3(shape[0], shape[1], shape[2]), shape[3] }; }
	bounding_capsule capsule() const { return bounding_capsule{ vec3(shape[0], shape[1], shape[2]), vec3(shape[3], shape[4], shape[5]), shape[6] }; }
	bounding_cylinder cylinder() const { return bounding_cylinder{ vec3(shape[0], shape[1], shape[2]), vec3(shape[3], shape[4], shape[5]), shape[6] }; }
	bounding_box aabb() const { return bounding_box{ vec3(shape[0], shape[1], shape[2]), vec3(shape[3], shape[4], shape[5]) }; }
	bounding_hull hull() const { return bounding_hull{ quat(shape[0], shape[1], shape[2], shape[3]), vec3(shape[4], shape[5], shape[6]), (u32)shape[7] }; }
	void set(const bounding_hull& h) { shape[0] = h.rotation.x; shape[1] = h.rotation.y; shape[2] = h.rotation.z; shape[3] = h.rotation.w; shape[4] = h.position.x; shape[5] = h.position.y; shape[6] = h.position.z; shape[7] = (float)h.geometryIndex; shape[8] = shape[9] = 0.f; }
	bounding_oriented_box obb() const { return bounding_oriented_box{ quat(shape[0], shape[1], shape[2], shape[3]), vec3(shape[4], shape[5], shape[6]), vec3(shape[7], shape[8], shape[9]) }; }
	void set(const bounding_sphere& s) { shape[0] = s.center.x; shape[1] = s.center.y; shape[2] = s.center.z; shape[3] = s.radius; for (int i = 4; i < 10; ++i) shape[i] = 0.f; }
	void set(const bounding_capsule& c) { shape[0] = c.positionA.x; shape[1] = c.positionA.y; shape[2] = c.positionA.z; shape[3] = c.positionB.x; shape[4] = c.positionB.y; shape[5] = c.positionB.z; shape[6] = c.radius; shape[7] = shape[8] = shape[9] = 0.f; }
	void set(const bounding_box& b) { shape[0] = b.minCorner.x; shape[1] = b.minCorner.y; shape[2] = b.minCorner.z; shape[3] = b.maxCorner.x; shape[4] = b.maxCorner.y; shape[5] = b.maxCorner.z; shape[6] = shape[7] = shape[8] = shape[9] = 0.f; }
	void set(const bounding_oriented_box& o) { shape[0] = o.rotation.x; shape[1] = o.rotation.y; shape[2] = o.rotation.z; shape[3] = o.rotation.w; shape[4] = o.center.x; shape[5] = o.center.y; shape[6] = o.center.z; shape[7] = o.radius.x; shape[8] = o.radius.y; shape[9] = o.radius.z; }
};
static_assert(sizeof(collider_union) == 64, "collider record is 64 B like the reference's");

// physics.h:347-354 — layout {point, depth, normal, packed} is load-bearing.
struct collision_contact
{
	vec3 point;
	float penetrationDepth;
	vec3 normal;
	u32 friction_restitution;
};
static_assert(sizeof(collision_contact) == 32, "32 B contact");

// constraints.h:53-56 widened to u32
struct constraint_body_pair { u32 rbA, rbB; };
// collision_broad.h:8-13 widened to u32
struct collider_pair { u32 colliderA, colliderB; };

} // namespace orc
