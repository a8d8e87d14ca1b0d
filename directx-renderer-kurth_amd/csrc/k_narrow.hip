// Narrowphase: prune / classify / bucket (reference collision_narrow.cpp:2346-2453), then one branch-uniform kernel family per
// group of type pairs: closed forms (:374-612, :1074-1140), box-box SAT + Sutherland-Hodgman clipping (:1179-1527) and
// GJK + EPA for capsule / cylinder vs box and cylinder vs cylinder (:705-790, :821-1043, collision_gjk.{h,cpp}, collision_epa.{h,cpp}).
// One thread per candidate pair; every pair writes a 96-byte ManifoldRec (count 0 = no collision) so contact generation needs no
// atomics and keeps pair order.  Every pair with a convex hull is GJK + EPA with one contact (:496-520, 792-818, 1045-1071, 1150-1176,
// 1529-1584); the hull's support function walks its vertex list (collision_gjk.h:77-100).
#include "world.h"
#include <rocprim/rocprim.hpp>
#include "events.h"
#include <cstddef>
EventSink sinkOf(World& w);
#include <algorithm>

void prim_sort_pairs_u32_u64(World& w, const u32* kin, u32* kout, const u64* vin, u64* vout, u32 n, u32 bits);

#define KEY_INVALID 63u
#define KEY_ZONE 62u   // rigid body vs force-field / trigger collider: boolean overlap check only (collision_narrow.cpp:2378-2395)

struct Man { V3 n; float4 p[4]; u32 count; };

MI_DEV bool typeSupported(u32 t) { return t <= MI_HULL; }

// ---------------------------------------------------------------------------------------------------------------
// K5: prune + classify.  Reads the 16-B tag quarter of both colliders.
// ---------------------------------------------------------------------------------------------------------------
// Equal types: the reference's sweep reports a pair as (collider whose start endpoint comes later on the sorting axis, the one already
// active) (collision_broad.cpp:127) and the narrowphase swaps equal-type pairs (collision_narrow.cpp:2374), so A = the collider
// whose box STARTS FIRST on this step's sorting axis.  Equal starts: the reference's order is that of its endpoint array (a stable
// insertion sort carried over from earlier frames); here the lower collider index comes first.
__global__ void __launch_bounds__(256) k_classify(const u32* __restrict__ counters, u32 nb, const uint2* __restrict__ pairs, const ColliderRec* __restrict__ colWorld, const float4* __restrict__ aabbMin, u32 stepParity,
	u32* __restrict__ pairKey, u64* __restrict__ pairPacked, u32 numLaunched)
{
	u32 p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= numLaunched) return;
	if (p >= counters[CTR_NUM_PAIRS]) { pairKey[p] = KEY_INVALID; pairPacked[p] = 0ull; return; } // (the launch is sized before the host knows the pair count: the surplus sorts behind everything)
	uint2 pr = pairs[p];
	float4 da = colWorld[pr.x].d, db = colWorld[pr.y].d;
	u32 tA = __float_as_uint(da.x), tB = __float_as_uint(db.x), bA = __float_as_uint(da.y), bB = __float_as_uint(db.y);
	u32 key = KEY_INVALID;
	bool rbA = bA < nb, rbB = bB < nb;
	if ((rbA || rbB) && !(rbA && rbB && bA == bB) && typeSupported(tA) && typeSupported(tB)) // :2358-2369
	{
		bool swap = tA > tB; // :2374: A = the lower type
		if (tA == tB)
		{
			const u32 axis = counters[CTR_SAP_AXIS + stepParity];
			float4 ma = aabbMin[pr.x], mb = aabbMin[pr.y];
			float sa = axis == 0u ? ma.x : (axis == 1u ? ma.y : ma.z), sb = axis == 0u ? mb.x : (axis == 1u ? mb.y : mb.z);
			swap = sb < sa || (sb == sa && pr.y < pr.x);
		}
		if (swap) { u32 t = pr.x; pr.x = pr.y; pr.y = t; t = tA; tA = tB; tB = t; }
		u32 zone = (__float_as_uint(da.w) | __float_as_uint(db.w)) & 0xFFu; // a force-field / trigger collider is in the pair
		key = zone ? KEY_ZONE : tA * 6 + tB;
	}
	pairKey[p] = key;
	pairPacked[p] = ((u64)pr.y << 32) | pr.x;
}

__global__ void k_bucket_offsets(u32* __restrict__ counters, const u32* __restrict__ keySorted)
{
	u32 b = threadIdx.x; // bucket key 0..63
	u32 n = counters[CTR_NUM_PAIRS];
	u32 lo = 0, hi = n;
	while (lo < hi) { u32 mid = (lo + hi) >> 1; if (keySorted[mid] < b) lo = mid + 1; else hi = mid; }
	counters[CTR_BUCKET_START + b] = lo; // collision keys are <= 5*6+5 = 35; KEY_ZONE (overlap checks) and KEY_INVALID sort last: manifold slots end where KEY_ZONE starts
	if (b == KEY_ZONE) { counters[CTR_NUM_VALID] = lo; counters[CTR_EPA_COUNT] = 0; counters[CTR_EPA_COUNT_HULL] = 0; counters[CTR_NUM_ACTIVE] = 0; counters[CTR_NUM_CONTACTS] = 0; }
}

// ---------------------------------------------------------------------------------------------------------------
// Shape accessors
// ---------------------------------------------------------------------------------------------------------------
struct Sphere { V3 c; float r; };
struct Capsule { V3 a, b; float r; };
struct Box { V3 lo, hi; };
struct Obb { Q4 q; V3 c, r; };
MI_DEV Sphere asSphere(const ColliderRec& c) { Sphere s; s.c = v3(c.a.x, c.a.y, c.a.z); s.r = c.a.w; return s; }
MI_DEV Capsule asCapsule(const ColliderRec& c) { Capsule s; s.a = v3(c.a.x, c.a.y, c.a.z); s.b = v3(c.a.w, c.b.x, c.b.y); s.r = c.b.z; return s; }
MI_DEV Box asBox(const ColliderRec& c) { Box s; s.lo = v3(c.a.x, c.a.y, c.a.z); s.hi = v3(c.a.w, c.b.x, c.b.y); return s; }
MI_DEV Obb asObb(const ColliderRec& c) { Obb s; s.q = q4f4(c.a); s.c = v3(c.b.x, c.b.y, c.b.z); s.r = v3(c.b.w, c.c.x, c.c.y); return s; }
MI_DEV V3 boxCenter(const Box& b) { return (b.lo + b.hi) * 0.5f; }
MI_DEV V3 boxRadius(const Box& b) { return (b.hi - b.lo) * 0.5f; }

MI_DEV V3 closestPointSegment(V3 q, V3 a, V3 b) // bounding_volumes.h:365-371
{
	V3 ab = b - a;
	float t = dot(q - a, ab) / sqlen(ab);
	t = clampf(t, 0.f, 1.f);
	return a + t * ab;
}
MI_DEV float closestSegmentSegment(V3 l1a, V3 l1b, V3 l2a, V3 l2b, V3& c1, V3& c2) // bounding_volumes.cpp:1251-1315
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
	c1 = l1a + d1 * s;
	c2 = l2a + d2 * t;
	return sqlen(c1 - c2);
}

// ---------------------------------------------------------------------------------------------------------------
// Closed-form pair kernels
// ---------------------------------------------------------------------------------------------------------------
MI_DEV bool sphereSphere(Sphere s1, Sphere s2, Man& m) // collision_narrow.cpp:374-400
{
	V3 n = s2.c - s1.c;
	float radiusSum = s2.r + s1.r;
	float sq = sqlen(n);
	if (sq <= radiusSum * radiusSum)
	{
		float distance;
		if (sq == 0.f) { distance = 0.f; m.n = v3(0.f, 1.f, 0.f); }
		else { distance = sqrtf(sq); m.n = n / distance; }
		m.count = 1;
		V3 pt = 0.5f * (s1.c + s1.r * m.n + s2.c - s2.r * m.n);
		m.p[0] = make_float4(pt.x, pt.y, pt.z, radiusSum - distance);
		return true;
	}
	return false;
}
MI_DEV bool sphereCapsule(Sphere s, Capsule c, Man& m) // :402-406
{
	Sphere s2; s2.c = closestPointSegment(s.c, c.a, c.b); s2.r = c.r;
	return sphereSphere(s, s2, m);
}
MI_DEV bool sphereBox(Sphere s, Box a, Man& m) // :451-478
{
	V3 p = v3(fminf(fmaxf(s.c.x, a.lo.x), a.hi.x), fminf(fmaxf(s.c.y, a.lo.y), a.hi.y), fminf(fmaxf(s.c.z, a.lo.z), a.hi.z)); // closestPoint_PointAABB
	V3 n = p - s.c;
	float sq = sqlen(n);
	if (sq <= s.r * s.r)
	{
		float dist = 0.f;
		if (sq > 0.f) { dist = sqrtf(sq); n = n / dist; }
		else { n = v3(0.f, 1.f, 0.f); }
		m.count = 1;
		m.n = n;
		V3 pt = 0.5f * (p + s.c + n * s.r);
		m.p[0] = make_float4(pt.x, pt.y, pt.z, s.r - dist);
		return true;
	}
	return false;
}
MI_DEV bool sphereObb(Sphere s, Obb o, Man& m) // :480-494
{
	Box aabb; aabb.lo = o.c - o.r; aabb.hi = o.c + o.r;
	Sphere s_; s_.c = conjugate(o.q) * (s.c - o.c) + o.c; s_.r = s.r;
	if (sphereBox(s_, aabb, m))
	{
		m.n = o.q * m.n;
		V3 pt = o.q * (v3f4(m.p[0]) - o.c) + o.c;
		m.p[0] = make_float4(pt.x, pt.y, pt.z, m.p[0].w);
		return true;
	}
	return false;
}
MI_DEV bool sphereCylinder(Sphere s, Capsule c, Man& m) // :408-449 (a cylinder is stored as {A, B, r} like a capsule)
{
	V3 ab = c.b - c.a;
	float t = dot(s.c - c.a, ab) / sqlen(ab);
	if (t >= 0.f && t <= 1.f)
	{
		Sphere s2; s2.c = lerp(c.a, c.b, t); s2.r = c.r;
		return sphereSphere(s, s2, m);
	}
	V3 p = (t <= 0.f) ? c.a : c.b;
	V3 up = (t <= 0.f) ? -ab : ab;
	V3 projectedDirToCenter = normalize(cross(cross(up, s.c - p), up));
	V3 endA = p + projectedDirToCenter * c.r;
	V3 endB = p - projectedDirToCenter * c.r;
	V3 closestToSphere = closestPointSegment(s.c, endA, endB);
	V3 normal = closestToSphere - s.c;
	float sq = sqlen(normal);
	if (sq <= s.r * s.r)
	{
		float distance;
		if (sq == 0.f) { distance = 0.f; m.n = -normalize(up); }
		else { distance = sqrtf(sq); m.n = normal / distance; }
		m.count = 1;
		float depth = s.r - distance;
		V3 pt = closestToSphere + 0.5f * depth * normal; // scales the un-normalised normal, as :445 does
		m.p[0] = make_float4(pt.x, pt.y, pt.z, depth);
		return true;
	}
	return false;
}
// capsule vs capsule (:523-612) and capsule vs cylinder (:614-703): identical except for what the end-cap / general cases test against
template <bool B_IS_CYLINDER>
MI_DEV bool capsuleTube(Capsule a, Capsule b, Man& m)
{
	V3 aDir = a.b - a.a;
	V3 bDir = normalize(b.b - b.a);
	float aDirLength = length(aDir);
	aDir *= 1.f / aDirLength;
	float parallel = dot(aDir, bDir);
	if (fabsf(parallel) > 0.99f)
	{
		V3 pAa = a.a, pAb = a.b, pBa = b.a, pBb = b.b;
		if (parallel < 0.f) { V3 t = pBa; pBa = pBb; pBb = t; }
		V3 ref = a.a;
		float a0 = 0.f, a1 = aDirLength;
		float b0 = dot(aDir, pBa - ref), b1 = dot(aDir, pBb - ref);
		float left = fmaxf(a0, b0), right = fminf(a1, b1);
		if (right < left)
		{
			Sphere sa, sb; sa.r = a.r; sb.r = b.r;
			if (a0 > b1) { sa.c = pAa; sb.c = pBb; } else { sa.c = pAb; sb.c = pBa; }
			if (B_IS_CYLINDER) return sphereCylinder(sa, b, m);
			return sphereSphere(sa, sb, m);
		}
		V3 contactA0 = ref + left * aDir, contactA1 = ref + right * aDir;
		V3 contactB0 = closestPointSegment(contactA0, pBa, pBb);
		V3 contactB1 = contactB0 + (right - left) * aDir;
		V3 normal = contactB0 - contactA0;
		float d = length(normal);
		if (d < MI_EPSILON) { d = 0.f; normal = v3(0.f, 1.f, 0.f); } else { normal = normal / d; }
		float penetration = (a.r + b.r) - d;
		if (penetration < 0.f) return false;
		m.n = normal; m.count = 2;
		V3 p0 = (contactA0 + contactB0) * 0.5f, p1 = (contactA1 + contactB1) * 0.5f;
		m.p[0] = make_float4(p0.x, p0.y, p0.z, penetration);
		m.p[1] = make_float4(p1.x, p1.y, p1.z, penetration);
		return true;
	}
	V3 c1, c2;
	closestSegmentSegment(a.a, a.b, b.a, b.b, c1, c2);
	Sphere sa, sb; sa.c = c1; sa.r = a.r; sb.c = c2; sb.r = b.r;
	if (B_IS_CYLINDER) return sphereCylinder(sa, b, m);
	return sphereSphere(sa, sb, m);
}
MI_DEV bool capsuleCapsule(Capsule a, Capsule b, Man& m) { return capsuleTube<false>(a, b, m); }
MI_DEV bool capsuleCylinder(Capsule a, Capsule b, Man& m) { return capsuleTube<true>(a, b, m); }
// cylinder vs cylinder, parallel branch (:821-903).  Returns 0 = no collision, 1 = manifold written, 2 = not parallel (GJK + EPA).
MI_DEV int cylinderCylinderParallel(Capsule a, Capsule b, Man& m)
{
	V3 aDir = a.b - a.a;
	V3 bDir = normalize(b.b - b.a);
	float aDirLength = length(aDir);
	aDir *= 1.f / aDirLength;
	float parallel = dot(aDir, bDir);
	if (!(fabsf(parallel) > 0.99f)) return 2;
	V3 pBa = b.a, pBb = b.b;
	if (parallel < 0.f) { V3 t = pBa; pBa = pBb; pBb = t; }
	V3 ref = a.a;
	float a0 = 0.f, a1 = aDirLength;
	float b0 = dot(aDir, pBa - ref), b1 = dot(aDir, pBb - ref);
	float left = fmaxf(a0, b0), right = fminf(a1, b1);
	if (right < left) return 0;
	V3 contactA0 = ref + left * aDir, contactA1 = ref + right * aDir;
	V3 contactB0 = closestPointSegment(contactA0, pBa, pBb);
	V3 contactB1 = contactB0 + (right - left) * aDir;
	V3 normal = contactB0 - contactA0;
	float d = length(normal);
	float penetration = (a.r + b.r) - d;
	if (penetration < 0.f) return 0;
	float capPenetration = right - left;
	if (capPenetration < penetration)
	{
		m.count = 1;
		V3 pt;
		if (b0 > a0) { m.n = aDir; pt = a.b - v3s(capPenetration * 0.5f); }   // vec3 - float broadcast, as :891/:897 do
		else { m.n = -aDir; pt = a.a + v3s(capPenetration * 0.5f); }
		m.p[0] = make_float4(pt.x, pt.y, pt.z, capPenetration);
	}
	else
	{
		if (d < MI_EPSILON) { d = 0.f; normal = v3(0.f, 1.f, 0.f); } else { normal = normal / d; }
		m.n = normal; m.count = 2;
		V3 p0 = (contactA0 + contactB0) * 0.5f, p1 = (contactA1 + contactB1) * 0.5f;
		m.p[0] = make_float4(p0.x, p0.y, p0.z, penetration);
		m.p[1] = make_float4(p1.x, p1.y, p1.z, penetration);
	}
	return 1;
}
MI_DEV bool boxBoxAxisAligned(Box a, Box b, Man& m) // :1074-1140
{
	V3 centerA = boxCenter(a), centerB = boxCenter(b), radiusA = boxRadius(a), radiusB = boxRadius(b);
	V3 d = centerB - centerA;
	V3 p = (radiusB + radiusA) - vabs(d);
	if (p.x < 0.f || p.y < 0.f || p.z < 0.f) return false;
	u32 minElement = (p.x < p.y) ? ((p.x < p.z) ? 0 : 2) : ((p.y < p.z) ? 1 : 2);
	float s = vget(d, minElement) < 0.f ? -1.f : 1.f;
	float penetration = vget(p, minElement) * s;
	V3 normal = v3s(0.f); vset(normal, minElement, s);
	m.n = normal; m.count = 4;
	u32 axis0 = (minElement + 1) % 3, axis1 = (minElement + 2) % 3;
	float min0 = fmaxf(vget(a.lo, axis0), vget(b.lo, axis0)), min1 = fmaxf(vget(a.lo, axis1), vget(b.lo, axis1));
	float max0 = fminf(vget(a.hi, axis0), vget(b.hi, axis0)), max1 = fminf(vget(a.hi, axis1), vget(b.hi, axis1));
	float depth = vget(centerA, minElement) + vget(radiusA, minElement) - penetration * 0.5f;
	for (u32 i = 0; i < 4; ++i)
	{
		V3 pt = v3s(0.f);
		vset(pt, axis0, (i < 2) ? min0 : max0);
		vset(pt, axis1, (i & 1) ? max1 : min1);
		vset(pt, minElement, depth);
		m.p[i] = make_float4(pt.x, pt.y, pt.z, penetration);
	}
	return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Manifold helpers — collision_narrow.cpp:56-369.  Polygon vertices are float4 (xyz, penetrationDepth).
// ---------------------------------------------------------------------------------------------------------------
// A clipping polygon: up to 16 points (x, y, z, penetration) — collision_narrow.cpp:148-152 — kept OUTSIDE the registers (its points are
// indexed by loop counters): vertex i lives at pts[i * stride].  The box-box kernel gives every lane a column of an LDS array
// (stride 64: a lane's accesses are LDS round trips, not scratch-memory ones through the vector memory path — its waves spent
// 61 % of their cycles waiting for those); the lone lane that clips a capsule's segment behind EPA uses its finished polytope's LDS (stride 1).
struct Poly { float4* pts; u32 stride; u32 n; };
#define PP(P_, I_) (P_).pts[(I_) * (P_).stride]
struct PolyStore { float4* a; float4* b; u32 stride; };

MI_DEV float4 clipAgainstPlane(float4 a, float4 b, float aDist, float bDist) // :154-163
{
	aDist = fabsf(aDist); bDist = fabsf(bDist);
	float t = aDist / (aDist + bDist);
	return make_float4(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w));
}
MI_DEV void sutherlandHodgman(Poly& input, const float4* clipPlanes, u32 numClipPlanes, Poly& output) // :166-222
{
	Poly* in = &input; Poly* out = &output;
	u32 clipIndex = 0;
	for (; clipIndex < numClipPlanes; ++clipIndex)
	{
		float4 plane = clipPlanes[clipIndex];
		out->n = 0;
		if (in->n == 0) break;
		float4 startPoint = PP(*in, in->n - 1);
		for (u32 i = 0; i < in->n; ++i)
		{
			float4 endPoint = PP(*in, i);
			float startDist = signedDistanceToPlane(v3f4(startPoint), plane);
			float endDist = signedDistanceToPlane(v3f4(endPoint), plane);
			bool startInside = startDist > 0.f, endInside = endDist > 0.f;
			if (startInside && endInside) { PP(*out, out->n++) = endPoint; }
			else if (startInside) { PP(*out, out->n++) = clipAgainstPlane(startPoint, endPoint, startDist, endDist); }
			else if (!startInside && endInside)
			{
				PP(*out, out->n++) = clipAgainstPlane(startPoint, endPoint, startDist, endDist);
				PP(*out, out->n++) = endPoint;
			}
			startPoint = endPoint;
		}
		Poly* tmp = in; in = out; out = tmp;
	}
	if (clipIndex % 2 == 0)
	{
		for (u32 i = 0; i < input.n; ++i) PP(output, i) = PP(input, i);
		output.n = input.n;
	}
}
MI_DEV void findStableContactManifold(const Poly& poly, V3 normal, Man& m) // :56-146
{
	const u32 nv = poly.n;
#define v(I_) PP(poly, I_)
	if (nv > 4)
	{
		V3 searchDir = getTangent(normal);
		float best = dot(searchDir, v3f4(v(0)));
		u32 ri = 0;
		for (u32 i = 1; i < nv; ++i) { float d = dot(searchDir, v3f4(v(i))); if (d > best) { ri = i; best = d; } }
		m.p[0] = v(ri);
		best = 0.f; ri = 0;
		for (u32 i = 0; i < nv; ++i) { float d = sqlen(v3f4(v(i)) - v3f4(m.p[0])); if (d > best) { ri = i; best = d; } }
		m.p[1] = v(ri);
		best = 0.f; ri = 0;
		for (u32 i = 0; i < nv; ++i)
		{
			V3 qa = v3f4(m.p[0]) - v3f4(v(i)), qb = v3f4(m.p[1]) - v3f4(v(i));
			float area = 0.5f * dot(cross(qa, qb), normal);
			if (area > best) { ri = i; best = area; }
		}
		m.p[2] = v(ri);
		best = 0.f; ri = 0;
		for (u32 i = 0; i < nv; ++i)
		{
			V3 qa = v3f4(m.p[0]) - v3f4(v(i)), qb = v3f4(m.p[1]) - v3f4(v(i)), qc = v3f4(m.p[2]) - v3f4(v(i));
			float area1 = 0.5f * dot(cross(qa, qb), normal);
			float area2 = 0.5f * dot(cross(qb, qc), normal);
			float area3 = 0.5f * dot(cross(qc, qa), normal);
			float area = fmaxf(fmaxf(area1, area2), area3);
			if (area > best) { ri = i; best = area; }
		}
		m.p[3] = v(ri);
		m.count = 4;
	}
	else
	{
		m.count = nv;
#pragma unroll
		for (u32 i = 0; i < 4; ++i) if (i < nv) m.p[i] = v(i); // (static indices: a manifold's four points stay in registers)
	}
#undef v
}
MI_DEV bool clipPointsAndBuildContact(Poly& polygon, Poly& clipped, const float4* clipPlanes, float4 referencePlane, Man& m) // :339-369 (clipped: the second polygon's storage)
{
	clipped.n = 0;
	sutherlandHodgman(polygon, clipPlanes, 4, clipped);
	if (clipped.n > 0)
	{
		for (u32 i = 0; i < clipped.n; ++i)
		{
			float4 pt = PP(clipped, i);
			if (pt.w < 0.f)
			{
				PP(clipped, i) = PP(clipped, clipped.n - 1);
				--clipped.n;
				--i;
			}
			else
			{
				float d = pt.w;
				pt.x += referencePlane.x * d; pt.y += referencePlane.y * d; pt.z += referencePlane.z * d;
				PP(clipped, i) = pt;
			}
		}
		if (clipped.n > 0) { findStableContactManifold(clipped, m.n, m); return true; }
	}
	return false;
}
MI_DEV void getAABBClippingPlanes(V3 radius, V3 normal, V3* pts, V3* nrm) // :225-254
{
	V3 p = vabs(normal);
	u32 maxElement = (p.x > p.y) ? ((p.x > p.z) ? 0 : 2) : ((p.y > p.z) ? 1 : 2);
	u32 axis0 = (maxElement + 1) % 3, axis1 = (maxElement + 2) % 3;
	V3 n;
	n = v3s(0.f); vset(n, axis0, 1.f); nrm[0] = n; pts[0] = -radius;
	n = v3s(0.f); vset(n, axis1, 1.f); nrm[1] = n; pts[1] = -radius;
	n = v3s(0.f); vset(n, axis0, -1.f); nrm[2] = n; pts[2] = radius;
	n = v3s(0.f); vset(n, axis1, -1.f); nrm[3] = n; pts[3] = radius;
}
MI_DEV void getAABBIncidentVertices(V3 radius, V3 normal, V3* verts) // :257-289
{
	V3 p = vabs(normal);
	u32 maxElement = (p.x > p.y) ? ((p.x > p.z) ? 0 : 2) : ((p.y > p.z) ? 1 : 2);
	float s = vget(normal, maxElement) < 0.f ? 1.f : -1.f;
	u32 axis0 = (maxElement + 1) % 3, axis1 = (maxElement + 2) % 3;
	float d = vget(radius, maxElement) * s;
	float min0 = -vget(radius, axis0), min1 = -vget(radius, axis1), max0 = vget(radius, axis0), max1 = vget(radius, axis1);
	for (u32 i = 0; i < 4; ++i)
	{
		V3 v = v3s(0.f);
		vset(v, maxElement, d);
		vset(v, axis0, (i == 1 || i == 2) ? max0 : min0);
		vset(v, axis1, (i >= 2) ? max1 : min1);
		verts[i] = v;
	}
}
MI_DEV void getAABBIncidentEdge(V3 radius, V3 normal, V3& outA, V3& outB) // :301-336
{
	V3 p = vabs(normal);
	outA = radius;
	if (p.x > p.y) { if (p.y > p.z) outB = v3(radius.x, radius.y, -radius.z); else outB = v3(radius.x, -radius.y, radius.z); }
	else { if (p.x > p.z) outB = v3(radius.x, radius.y, -radius.z); else outB = v3(-radius.x, radius.y, radius.z); }
	V3 sgn = v3(normal.x < 0.f ? -1.f : 1.f, normal.y < 0.f ? -1.f : 1.f, normal.z < 0.f ? -1.f : 1.f);
	outA = outA * sgn; outB = outB * sgn;
}
MI_DEV V3 obbSupport(const Obb& b, V3 dir) // collision_gjk.h:63-75
{
	dir = conjugate(b.q) * dir;
	V3 r = v3(dir.x < 0.f ? -b.r.x : b.r.x, dir.y < 0.f ? -b.r.y : b.r.y, dir.z < 0.f ? -b.r.z : b.r.z);
	return b.c + b.q * r;
}

// OBB vs OBB — collision_narrow.cpp:1179-1527
MI_DEV bool obbObb(const Obb& a, const Obb& b, Man& m, const PolyStore& store)
{
	V3 ax = a.q * v3(1.f, 0.f, 0.f), ay = a.q * v3(0.f, 1.f, 0.f), az = a.q * v3(0.f, 0.f, 1.f);
	V3 bx = b.q * v3(1.f, 0.f, 0.f), by = b.q * v3(0.f, 1.f, 0.f), bz = b.q * v3(0.f, 0.f, 1.f);
	M3 r;
	r.m00 = dot(ax, bx); r.m10 = dot(ay, bx); r.m20 = dot(az, bx);
	r.m01 = dot(ax, by); r.m11 = dot(ay, by); r.m21 = dot(az, by);
	r.m02 = dot(ax, bz); r.m12 = dot(ay, bz); r.m22 = dot(az, bz);
	V3 tw = b.c - a.c;
	V3 t = conjugate(a.q) * tw;
	M3 absR;
	absR.m00 = fabsf(r.m00) + MI_EPSILON; absR.m10 = fabsf(r.m10) + MI_EPSILON; absR.m20 = fabsf(r.m20) + MI_EPSILON;
	absR.m01 = fabsf(r.m01) + MI_EPSILON; absR.m11 = fabsf(r.m11) + MI_EPSILON; absR.m21 = fabsf(r.m21) + MI_EPSILON;
	absR.m02 = fabsf(r.m02) + MI_EPSILON; absR.m12 = fabsf(r.m12) + MI_EPSILON; absR.m22 = fabsf(r.m22) + MI_EPSILON;
	bool parallel = absR.m00 >= 0.99f || absR.m10 >= 0.99f || absR.m20 >= 0.99f || absR.m01 >= 0.99f || absR.m11 >= 0.99f || absR.m21 >= 0.99f
		|| absR.m02 >= 0.99f || absR.m12 >= 0.99f || absR.m22 >= 0.99f;

	float ra, rb;
	float minPenetration = MI_FLT_MAX;
	V3 normal = v3s(0.f);
	bool bFace = false;
	for (u32 i = 0; i < 3; ++i)
	{
		ra = vget(a.r, i);
		rb = dot(mrow(absR, i), b.r);
		float penetration = ra + rb - fabsf(vget(t, i));
		if (penetration < 0.f) return false;
		if (penetration < minPenetration) { minPenetration = penetration; normal = v3s(0.f); vset(normal, i, 1.f); }
	}
	for (u32 i = 0; i < 3; ++i)
	{
		ra = dot(mcol(absR, i), a.r);
		rb = vget(b.r, i);
		float d = dot(mcol(r, i), t);
		float penetration = ra + rb - fabsf(d);
		if (penetration < 0.f) return false;
		if (penetration < minPenetration) { minPenetration = penetration; normal = v3s(0.f); vset(normal, i, 1.f); bFace = true; }
	}
	bool edgeCollision = false;
	V3 edgeNormal = v3s(0.f);
	if (!parallel)
	{
		float penetration, l; V3 n;
#define MI_EDGE_TEST(RA, RB, DIST, NX, NY, NZ) \
		ra = RA; rb = RB; penetration = ra + rb - fabsf(DIST); \
		if (penetration < 0.f) return false; \
		n = v3(NX, NY, NZ); l = 1.f / length(n); penetration *= l; \
		if (penetration < minPenetration) { minPenetration = penetration; edgeNormal = n * l; edgeCollision = true; }
		MI_EDGE_TEST(a.r.y * absR.m20 + a.r.z * absR.m10, b.r.y * absR.m02 + b.r.z * absR.m01, t.z * r.m10 - t.y * r.m20, 0.f, -r.m20, r.m10)
		MI_EDGE_TEST(a.r.y * absR.m21 + a.r.z * absR.m11, b.r.x * absR.m02 + b.r.z * absR.m00, t.z * r.m11 - t.y * r.m21, 0.f, -r.m21, r.m11)
		MI_EDGE_TEST(a.r.y * absR.m22 + a.r.z * absR.m12, b.r.x * absR.m01 + b.r.y * absR.m00, t.z * r.m12 - t.y * r.m22, 0.f, -r.m22, r.m12)
		MI_EDGE_TEST(a.r.x * absR.m20 + a.r.z * absR.m00, b.r.y * absR.m12 + b.r.z * absR.m11, t.x * r.m20 - t.z * r.m00, r.m20, 0.f, -r.m00)
		MI_EDGE_TEST(a.r.x * absR.m21 + a.r.z * absR.m01, b.r.x * absR.m12 + b.r.z * absR.m10, t.x * r.m21 - t.z * r.m01, r.m21, 0.f, -r.m01)
		MI_EDGE_TEST(a.r.x * absR.m22 + a.r.z * absR.m02, b.r.x * absR.m11 + b.r.y * absR.m10, t.x * r.m22 - t.z * r.m02, r.m22, 0.f, -r.m02)
		MI_EDGE_TEST(a.r.x * absR.m10 + a.r.y * absR.m00, b.r.y * absR.m22 + b.r.z * absR.m21, t.y * r.m00 - t.x * r.m10, -r.m10, r.m00, 0.f)
		MI_EDGE_TEST(a.r.x * absR.m11 + a.r.y * absR.m01, b.r.x * absR.m22 + b.r.z * absR.m20, t.y * r.m01 - t.x * r.m11, -r.m11, r.m01, 0.f)
		MI_EDGE_TEST(a.r.x * absR.m12 + a.r.y * absR.m02, b.r.x * absR.m21 + b.r.y * absR.m20, t.y * r.m02 - t.x * r.m12, -r.m12, r.m02, 0.f)
#undef MI_EDGE_TEST
	}
	bool faceCollision = !edgeCollision;
	if (faceCollision) { if (bFace) normal = r * normal; }
	else normal = edgeNormal;
	normal = a.q * normal;
	if (dot(normal, tw) < 0.f) normal = -normal;
	m.n = normal;

	if (faceCollision)
	{
		V3 cpp[4], cpn[4], verts[4];
		Poly polygon; polygon.pts = store.a; polygon.stride = store.stride; polygon.n = 4;
		Poly clipped; clipped.pts = store.b; clipped.stride = store.stride;
		float4 plane;
		if (!bFace)
		{
			getAABBClippingPlanes(a.r, conjugate(a.q) * normal, cpp, cpn);
			getAABBIncidentVertices(b.r, conjugate(b.q) * normal, verts);
			for (u32 i = 0; i < 4; ++i) { cpp[i] = a.q * cpp[i] + a.c; cpn[i] = a.q * cpn[i]; verts[i] = b.q * verts[i] + b.c; }
			plane = createPlane(obbSupport(a, normal), normal);
		}
		else
		{
			getAABBClippingPlanes(b.r, conjugate(b.q) * -normal, cpp, cpn);
			getAABBIncidentVertices(a.r, conjugate(a.q) * -normal, verts);
			for (u32 i = 0; i < 4; ++i) { cpp[i] = b.q * cpp[i] + b.c; cpn[i] = b.q * cpn[i]; verts[i] = a.q * verts[i] + a.c; }
			plane = createPlane(obbSupport(b, -normal), -normal);
		}
		float4 clipPlanes[4];
		for (u32 i = 0; i < 4; ++i)
		{
			clipPlanes[i] = createPlane(cpp[i], cpn[i]);
			PP(polygon, i) = make_float4(verts[i].x, verts[i].y, verts[i].z, -signedDistanceToPlane(verts[i], plane));
		}
		if (!clipPointsAndBuildContact(polygon, clipped, clipPlanes, plane, m)) return false;
	}
	else
	{
		V3 a0, a1, b0, b1;
		getAABBIncidentEdge(a.r, conjugate(a.q) * normal, a0, a1);
		getAABBIncidentEdge(b.r, conjugate(b.q) * -normal, b0, b1);
		a0 = a.q * a0 + a.c; a1 = a.q * a1 + a.c; b0 = b.q * b0 + b.c; b1 = b.q * b1 + b.c;
		V3 pa, pb;
		float sq = closestSegmentSegment(a0, a1, b0, b1, pa, pb);
		V3 pt = (pa + pb) * 0.5f;
		m.count = 1;
		m.p[0] = make_float4(pt.x, pt.y, pt.z, sqrtf(sq));
	}
	return true;
}

// ---------------------------------------------------------------------------------------------------------------
// GJK + EPA for a tube (capsule / cylinder) vs an axis-aligned box or a cylinder — collision_gjk.h:17-61,140-238;
// collision_gjk.cpp:6-212; collision_epa.h:96-168; collision_epa.cpp:5-239.  Array sizes: the reference's 1024-entry arrays can
// hold at most 24 points after its 20 iterations; triangles/edges are sized above the measured high-water marks (98 / 100 over
// all test scenes, same caps in oracle/onarrow.h), and the reference's out-of-memory exits are kept.
// ---------------------------------------------------------------------------------------------------------------
#define EPA_MAX_POINTS 24
#define EPA_MAX_TRIANGLES 128
#define EPA_MAX_EDGES 160
#define EPA_MAX_BORDER 32
#define GJK_MAX_ITERATIONS 64

struct SupportPoint { V3 a, b, mk; };
struct GjkSimplex { SupportPoint a, b, c, d; u32 numPoints; };

MI_DEV V3 capsuleSupport(const Capsule& c, V3 dir)
{
	float distA = dot(dir, c.a), distB = dot(dir, c.b);
	V3 farther = distA > distB ? c.a : c.b;
	return normalize(dir) * c.r + farther;
}
MI_DEV V3 cylinderSupport(const Capsule& c, V3 dir) // collision_gjk.h:30-46
{
	float distA = dot(dir, c.a), distB = dot(dir, c.b);
	V3 farther = distA > distB ? c.a : c.b;
	V3 n = c.a - c.b;
	V3 projectedDir = noz(cross(cross(n, dir), n));
	return farther + projectedDir * c.r;
}
MI_DEV V3 boxSupport(const Box& b, V3 dir) { return v3((dir.x < 0.f) ? b.lo.x : b.hi.x, (dir.y < 0.f) ? b.lo.y : b.hi.y, (dir.z < 0.f) ? b.lo.z : b.hi.z); }
struct Hull { Q4 q; V3 pos; u32 first, count; };
MI_DEV V3 hullSupport(const Hull& h, const float4* __restrict__ hullVerts, V3 dir) // collision_gjk.h:77-100: first vertex with the largest dot product
{
	dir = conjugate(h.q) * dir;
	V3 result = v3s(0.f);
	float maxDist = -MI_FLT_MAX;
	for (u32 i = 0; i < h.count; ++i)
	{
		V3 v = v3f4(hullVerts[h.first + i]);
		float d = dot(dir, v);
		if (d > maxDist) { maxDist = d; result = v; }
	}
	return h.pos + h.q * result;
}
// The two convex shapes of one GJK/EPA instance.  A: capsule, cylinder, sphere (tubeA.a = centre), axis-aligned box, OBB or hull;
// B: axis-aligned box, cylinder or hull.
enum { SUP_CAPSULE = 0, SUP_CYLINDER = 1, SUP_SPHERE = 2, SUP_BOX = 3, SUP_OBB = 4, SUP_HULL = 5 };
struct SupShapes { Capsule tubeA, tubeB; Box box, boxA; Obb obbA; Hull hullA, hullB; u32 kindA, kindB; const float4* hullVerts; };
// FAMILY 0: tube (capsule / cylinder) vs box or cylinder — the pairs of the BASELINE configs; FAMILY 1: anything vs hull.  Separate
// kernel instances: with the family fixed at compile time the unused shapes drop out of the registers (one build for everything
// needed 256 VGPRs and halved the occupancy of the tube pairs).
template <int FAMILY>
MI_DEV V3 supportA(const SupShapes& sh, V3 dir)
{
	if (FAMILY == 0) return (sh.kindA == SUP_CYLINDER) ? cylinderSupport(sh.tubeA, dir) : capsuleSupport(sh.tubeA, dir);
	switch (sh.kindA)
	{
		case SUP_CAPSULE: return capsuleSupport(sh.tubeA, dir);
		case SUP_CYLINDER: return cylinderSupport(sh.tubeA, dir);
		case SUP_SPHERE: return normalize(dir) * sh.tubeA.r + sh.tubeA.a; // collision_gjk.h:6-15
		case SUP_BOX: return boxSupport(sh.boxA, dir);
		case SUP_OBB: return obbSupport(sh.obbA, dir);
		default: return hullSupport(sh.hullA, sh.hullVerts, dir);
	}
}
template <int FAMILY>
MI_DEV SupportPoint supportPair(const SupShapes& sh, V3 dir)
{
	SupportPoint s;
	s.a = supportA<FAMILY>(sh, dir);
	if (FAMILY == 0) s.b = (sh.kindB == SUP_BOX) ? boxSupport(sh.box, -dir) : cylinderSupport(sh.tubeB, -dir);
	else s.b = hullSupport(sh.hullB, sh.hullVerts, -dir);
	s.mk = s.a - s.b;
	return s;
}
MI_DEV V3 crossABA(V3 a, V3 b) { return cross(cross(a, b), a); }

// returns 0 = stop, 1 = continue, 2 = error
MI_DEV int updateGJKSimplex(GjkSimplex& s, const SupportPoint& a, V3& dir)
{
	if (s.numPoints == 2)
	{
		V3 ao = -a.mk, ab = s.b.mk - a.mk, ac = s.c.mk - a.mk;
		V3 abc = cross(ab, ac);
		V3 abp = cross(ab, abc);
		if (dot(ao, abp) > 0.f) { s.c = a; dir = crossABA(ab, ao); return 1; }
		V3 acp = cross(abc, ac);
		if (dot(ao, acp) > 0.f) { s.b = a; dir = crossABA(ac, ao); return 1; }
		if (dot(ao, abc) >= 0.f) { s.d = s.b; s.b = a; s.numPoints = 3; dir = abc; return 1; }
		if (dot(ao, -abc) >= 0.f) { s.d = s.c; s.c = s.b; s.b = a; s.numPoints = 3; dir = -abc; return 1; }
		return 2;
	}
	if (s.numPoints == 3)
	{
		V3 ao = -a.mk, ab = s.b.mk - a.mk, ac = s.c.mk - a.mk, ad = s.d.mk - a.mk;
		V3 bcd = cross(s.c.mk - s.b.mk, s.d.mk - s.b.mk);
		if (dot(bcd, dir) > 0.00001f || dot(bcd, s.b.mk) < -0.00001f) return 2;
		V3 abc = cross(ac, ab), abd = cross(ab, ad), adc = cross(ad, ac);
		const int ABC = 1, ABD = 2, ADC = 4;
		int flags = 0;
		flags |= (dot(abc, ao) > 0.f) ? ABC : 0;
		flags |= (dot(abd, ao) > 0.f) ? ABD : 0;
		flags |= (dot(adc, ao) > 0.f) ? ADC : 0;
		if (flags == (ABC | ABD | ADC)) return 2;
		if (flags == 0) return 0;
		// The reference's goto web (collision_gjk.cpp:95-205) as a small state machine: entry = (face, first|second test).
		int face, second = 0;
		if (flags == ABC) { face = ABC; }
		else if (flags == ABD) { face = ABD; }
		else if (flags == ADC) { face = ADC; }
		else if (flags == (ABC | ABD)) { if (dot(cross(abc, ab), ao) > 0.f) { face = ABD; } else { face = ABC; second = 1; } }
		else if (flags == (ABD | ADC)) { if (dot(cross(abd, ad), ao) > 0.f) { face = ADC; } else { face = ABD; second = 1; } }
		else { if (dot(cross(adc, ac), ao) > 0.f) { face = ABC; } else { face = ADC; second = 1; } }
		if (face == ABC)
		{
			if (!second && dot(cross(abc, ab), ao) > 0.f) { s.c = a; s.numPoints = 2; dir = crossABA(ab, ao); return 1; }
			if (dot(cross(ac, abc), ao) > 0.f) { s.b = a; s.numPoints = 2; dir = crossABA(ac, ao); return 1; }
			s.d = a; dir = abc; return 1;
		}
		if (face == ABD)
		{
			if (!second && dot(cross(abd, ad), ao) > 0.f) { s.b = s.d; s.c = a; s.numPoints = 2; dir = crossABA(ad, ao); return 1; }
			if (dot(cross(ab, abd), ao) > 0.f) { s.c = a; s.numPoints = 2; dir = crossABA(ab, ao); return 1; }
			s.c = a; dir = abd; return 1;
		}
		{
			if (!second && dot(cross(adc, ac), ao) > 0.f) { s.b = a; s.numPoints = 2; dir = crossABA(ac, ao); return 1; }
			if (dot(cross(ad, adc), ao) > 0.f) { s.b = a; s.c = s.d; s.numPoints = 2; dir = crossABA(ad, ao); return 1; }
			s.b = a; dir = adc; return 1;
		}
	}
	return 2;
}

template <int FAMILY>
MI_DEV bool gjkPair(const SupShapes& sh, GjkSimplex& sx)
{
	V3 dir = v3(1.f, 0.1f, -0.2f);
	sx.c = supportPair<FAMILY>(sh, dir);
	if (dot(sx.c.mk, dir) < 0.f) return false;
	dir = -sx.c.mk;
	sx.b = supportPair<FAMILY>(sh, dir);
	if (dot(sx.b.mk, dir) < 0.f) return false;
	dir = crossABA(sx.c.mk - sx.b.mk, -sx.b.mk);
	sx.numPoints = 2;
	for (u32 it = 0; it < GJK_MAX_ITERATIONS; ++it)
	{
		if (sqlen(dir) < 0.0001f) return false;
		SupportPoint a = supportPair<FAMILY>(sh, dir);
		if (dot(a.mk, dir) < 0.f) return false;
		int res = updateGJKSimplex(sx, a, dir);
		if (res == 0) { sx.a = a; sx.numPoints = 4; return true; }
		if (res == 2) return false;
	}
	return false;
}


MI_DEV V3 barycentric(V3 a, V3 b, V3 c, V3 p) // math.cpp:1390-1408
{
	V3 v0 = b - a, v1 = c - a, v2 = p - a;
	float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
	float denom = d00 * d11 - d01 * d01;
	denom = (fabsf(denom) < MI_EPSILON) ? 1.f : denom;
	float v = (d11 * d20 - d01 * d21) / denom;
	float w = (d00 * d21 - d01 * d20) / denom;
	float u = 1.0f - v - w;
	return v3(u, v, w);
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-cooperative EPA (collision_epa.h:96-168, collision_epa.cpp:5-239): ONE WAVE expands ONE polytope.  The polytope
// (points, triangles, edges) lives in LDS (7.6 KB per wave); the per-iteration scans the serial algorithm does over all
// triangles / edges are strided over the 64 lanes, the closest-triangle search is a shuffle min-reduction on (distance, index)
// — the same "first strictly smaller" winner as the serial scan — and border edges are compacted in index order with
// ballot + popcount, so indices, tie-breaks and therefore results are those of the serial reference algorithm.
// Support points are recomputed redundantly by every lane (wave-uniform values, no LDS traffic).
// ---------------------------------------------------------------------------------------------------------------
struct alignas(16) EpaWave
{
	float pa[EPA_MAX_POINTS][3], pb[EPA_MAX_POINTS][3];          // support points on A and B; minkowski = a - b (bit-identical to the stored one)
	float tnx[EPA_MAX_TRIANGLES], tny[EPA_MAX_TRIANGLES], tnz[EPA_MAX_TRIANGLES], tdist[EPA_MAX_TRIANGLES];
	// (indices are below 65535 = EPA_NONE32: 16 bits each keep a wave's polytope at 6.3 KB instead of 9.6, i.e. more resident workgroups)
	uint16_t ta[EPA_MAX_TRIANGLES], tb[EPA_MAX_TRIANGLES], tc[EPA_MAX_TRIANGLES], teA[EPA_MAX_TRIANGLES], teB[EPA_MAX_TRIANGLES], teC[EPA_MAX_TRIANGLES];
	uint8_t tactive[EPA_MAX_TRIANGLES];
	uint16_t ea[EPA_MAX_EDGES], eb[EPA_MAX_EDGES], etA[EPA_MAX_EDGES], etB[EPA_MAX_EDGES];
	u32 refs[EPA_MAX_EDGES];
	u32 border[EPA_MAX_BORDER], newEdgePerPoint[EPA_MAX_POINTS];
};
static_assert(offsetof(EpaWave, tnx) % 16 == 0 && sizeof(EpaWave) % 16 == 0 && 4 * sizeof(float) * EPA_MAX_TRIANGLES >= 2 * 16 * sizeof(float4), "the triangle arrays double as two 16-point clipping polygons (capsuleBoxFinish)");
#define EPA_NONE32 0xFFFFu
#define EPA_GROUP 32u                 // lanes per polytope (>= EPA_MAX_BORDER and EPA_MAX_POINTS: one lane per border edge / point)
#define EPA_GROUP_MASK 0xFFFFFFFFu
#define WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier()

__device__ __forceinline__ V3 epaMk(const EpaWave& e, u32 i) { return v3(e.pa[i][0], e.pa[i][1], e.pa[i][2]) - v3(e.pb[i][0], e.pb[i][1], e.pb[i][2]); }
__device__ __forceinline__ void epaSetPoint(EpaWave& e, u32 i, const SupportPoint& p)
{
	e.pa[i][0] = p.a.x; e.pa[i][1] = p.a.y; e.pa[i][2] = p.a.z; e.pb[i][0] = p.b.x; e.pb[i][1] = p.b.y; e.pb[i][2] = p.b.z;
}
__device__ __forceinline__ void epaSetTri(EpaWave& e, u32 t, u32 a, u32 b, u32 c, u32 eA, u32 eB, u32 eC, V3 mka, V3 mkb, V3 mkc)
{
	V3 n = normalize(cross(mkb - mka, mkc - mka)); // getTriangleInfo (collision_epa.cpp:5-11)
	e.tnx[t] = n.x; e.tny[t] = n.y; e.tnz[t] = n.z; e.tdist[t] = dot(n, mka);
	e.ta[t] = a; e.tb[t] = b; e.tc[t] = c; e.teA[t] = eA; e.teB[t] = eB; e.teC[t] = eC; e.tactive[t] = 1;
}

// Runs on a GROUP of EPA_GROUP lanes (two polytopes per wave: the scalar part of an iteration — support point, termination test, the
// bookkeeping — is issued once for both, and the kernel is bound by instruction issue); every lane of a group returns the same
// (point, normal, depth).  `lane` is the lane inside the group.  Counts and indices are uniform inside a group, not across the wave:
// the two groups' loops and exits diverge freely, every cross-lane operation (shuffles below EPA_GROUP, ballots cut to the group's
// bits) stays inside a group, and WAVE_SYNC orders a group's LDS traffic whatever the other group is doing.
template <int FAMILY>
__device__ void epaWave(EpaWave& e, u32 lane, u32 groupShift, const GjkSimplex& g, const SupShapes& sh, V3& outPoint, V3& outNormal, float& outDepth)
{
	u32 numTris = 4, numPoints = 4, numEdges = 6;
	if (lane == 0)
	{
		epaSetPoint(e, 0, g.a); epaSetPoint(e, 1, g.b); epaSetPoint(e, 2, g.c); epaSetPoint(e, 3, g.d);
		epaSetTri(e, 0, 0, 1, 3, 4, 3, 0, g.a.mk, g.b.mk, g.d.mk);
		epaSetTri(e, 1, 1, 2, 3, 5, 4, 1, g.b.mk, g.c.mk, g.d.mk);
		epaSetTri(e, 2, 2, 0, 3, 3, 5, 2, g.c.mk, g.a.mk, g.d.mk);
		epaSetTri(e, 3, 0, 2, 1, 1, 0, 2, g.a.mk, g.c.mk, g.b.mk);
		const u32 E[6][4] = { { 0, 1, 0, 3 }, { 1, 2, 1, 3 }, { 2, 0, 2, 3 }, { 0, 3, 2, 0 }, { 1, 3, 0, 1 }, { 2, 3, 1, 2 } };
		for (u32 i = 0; i < 6; ++i) { e.ea[i] = E[i][0]; e.eb[i] = E[i][1]; e.etA[i] = E[i][2]; e.etB[i] = E[i][3]; }
	}
	WAVE_SYNC();

	u32 closest = 0;
	for (u32 it = 0; it < 20; ++it)
	{
		// findTriangleClosestToOrigin (collision_epa.cpp:89-109): lowest index among the minimal distances
		float bd = MI_FLT_MAX; u32 bi = 0xFFFFFFFFu;
		for (u32 t = lane; t < numTris; t += EPA_GROUP) { if (e.tactive[t]) { float d = e.tdist[t]; if (d < bd) { bd = d; bi = t; } } }
		for (int o = EPA_GROUP / 2; o > 0; o >>= 1)
		{
			float od = __shfl_xor(bd, o); u32 oi = __shfl_xor(bi, o);
			if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
		}
		closest = bi;
		if (closest == 0xFFFFFFFFu) { closest = 0; break; }
		V3 tn = v3(e.tnx[closest], e.tny[closest], e.tnz[closest]);
		SupportPoint np = supportPair<FAMILY>(sh, tn);
		float dd = dot(np.mk, tn);
		if (dd - e.tdist[closest] < 0.01f) break;

		// addNewPointAndUpdate (collision_epa.cpp:111-239)
		for (u32 i = lane; i < numEdges; i += EPA_GROUP) e.refs[i] = 0;
		if (lane < EPA_MAX_POINTS) e.newEdgePerPoint[lane] = 0;
		WAVE_SYNC();
		for (u32 t = lane; t < numTris; t += EPA_GROUP)
		{
			if (e.tactive[t])
			{
				float d = dot(v3(e.tnx[t], e.tny[t], e.tnz[t]), np.mk - epaMk(e, e.ta[t]));
				if (d > 0.f) { atomicAdd(&e.refs[e.teA[t]], 1u); atomicAdd(&e.refs[e.teB[t]], 1u); atomicAdd(&e.refs[e.teC[t]], 1u); e.tactive[t] = 0; }
			}
		}
		WAVE_SYNC();
		u32 nBorder = 0;
		for (u32 base = 0; base < numEdges; base += EPA_GROUP) // border edges in edge-index order
		{
			u32 i = base + lane;
			bool flag = i < numEdges && e.refs[i] == 1;
			u32 mask = (u32)(__ballot(flag) >> groupShift) & EPA_GROUP_MASK; // (the other group's lanes, if they are in this loop at all, vote in their own bits)
			u32 pos = nBorder + __popc(mask & ((1u << lane) - 1u));
			if (flag && pos < EPA_MAX_BORDER) e.border[pos] = i;
			nBorder += __popc(mask);
		}
		if (nBorder > EPA_MAX_BORDER) break;                                     // "out of memory" exits, as the serial code's pushes would hit them
		if (numPoints >= EPA_MAX_POINTS) break;
		if (numEdges + nBorder > EPA_MAX_EDGES || numTris + nBorder > EPA_MAX_TRIANGLES) break;
		u32 newPoint = numPoints++;
		if (lane == 0) epaSetPoint(e, newPoint, np);
		WAVE_SYNC();
		u32 triOffset = numTris;
		if (lane < nBorder)
		{
			u32 ei = e.border[lane];
			u32 ea = e.ea[ei], eb = e.eb[ei], tA = e.etA[ei], tB = e.etB[ei];
			bool triAActive = e.tactive[tA] != 0, triBActive = e.tactive[tB] != 0;
			u32 ptc = triBActive ? ea : eb;
			u32 triIndex = numTris + lane, newEdge = numEdges + lane;
			e.ea[newEdge] = ptc; e.eb[newEdge] = newPoint; e.etA[newEdge] = EPA_NONE32; e.etB[newEdge] = triIndex;
			atomicMax(&e.newEdgePerPoint[ptc], newEdge);                      // serial code: last writer (= highest index) wins
			u32 bI = ptc, cI = triBActive ? eb : ea;
			epaSetTri(e, triIndex, newPoint, bI, cI, ei, EPA_NONE32, newEdge, np.mk, epaMk(e, bI), epaMk(e, cI));
			if (triAActive) e.etB[ei] = triIndex; else e.etA[ei] = triIndex;
		}
		numEdges += nBorder; numTris += nBorder;
		WAVE_SYNC();
		if (lane < nBorder) // fix up the indices left open above
		{
			u32 ei = e.border[lane];
			bool triBNew = e.etB[ei] >= triOffset;
			u32 ptc = triBNew ? e.ea[ei] : e.eb[ei];
			u32 other = e.newEdgePerPoint[ptc];
			u32 triIndex = lane + triOffset;
			e.teB[triIndex] = other;
			e.etA[other] = triIndex;
		}
		WAVE_SYNC();
	}
	V3 tn = v3(e.tnx[closest], e.tny[closest], e.tnz[closest]);
	float tdist = e.tdist[closest];
	u32 ia = e.ta[closest], ib = e.tb[closest], ic = e.tc[closest];
	V3 bary = barycentric(epaMk(e, ia), epaMk(e, ib), epaMk(e, ic), tn * tdist);
	V3 pointA = bary.x * v3(e.pa[ia][0], e.pa[ia][1], e.pa[ia][2]) + bary.y * v3(e.pa[ib][0], e.pa[ib][1], e.pa[ib][2]) + bary.z * v3(e.pa[ic][0], e.pa[ic][1], e.pa[ic][2]);
	V3 pointB = bary.x * v3(e.pb[ia][0], e.pb[ia][1], e.pb[ia][2]) + bary.y * v3(e.pb[ib][0], e.pb[ib][1], e.pb[ib][2]) + bary.z * v3(e.pb[ic][0], e.pb[ic][1], e.pb[ic][2]);
	outPoint = 0.5f * (pointA + pointB);
	outNormal = tn;
	outDepth = tdist;
	WAVE_SYNC(); // the next instance reuses this LDS block
}

// Everything of intersection(capsule, aabb) after EPA — collision_narrow.cpp:723-768
MI_DEV void capsuleBoxFinish(V3 point, V3 normal, float depth, const Capsule& c, const Box& a, Man& m, const PolyStore& store)
{
	m.n = normal; m.count = 1;
	m.p[0] = make_float4(point.x, point.y, point.z, depth);
	if (fabsf(normal.x) > 0.99f || fabsf(normal.y) > 0.99f || fabsf(normal.z) > 0.99f)
	{
		V3 axis = normalize(c.b - c.a);
		if (fabsf(dot(normal, axis)) < 0.01f)
		{
			V3 cpp[4], cpn[4];
			float4 clipPlanes[4];
			V3 aabbNormal = -normal;
			V3 refPoint = v3((aabbNormal.x < 0.f) ? a.lo.x : a.hi.x, (aabbNormal.y < 0.f) ? a.lo.y : a.hi.y, (aabbNormal.z < 0.f) ? a.lo.z : a.hi.z); // getAABBReferencePlane :291-299
			float4 referencePlane = createPlane(refPoint, aabbNormal);
			Poly polygon; polygon.pts = store.a; polygon.stride = store.stride; polygon.n = 2;
			Poly clipped; clipped.pts = store.b; clipped.stride = store.stride;
			V3 pa = c.a + normal * c.r, pb = c.b + normal * c.r;
			PP(polygon, 0) = make_float4(pa.x, pa.y, pa.z, -signedDistanceToPlane(pa, referencePlane));
			PP(polygon, 1) = make_float4(pb.x, pb.y, pb.z, -signedDistanceToPlane(pb, referencePlane));
			V3 aCenter = boxCenter(a);
			getAABBClippingPlanes(boxRadius(a), aabbNormal, cpp, cpn);
			for (u32 i = 0; i < 4; ++i) clipPlanes[i] = createPlane(cpp[i] + aCenter, cpn[i]);
			clipPointsAndBuildContact(polygon, clipped, clipPlanes, referencePlane, m);
		}
	}
}
// Keys (typeA * 6 + typeB) that go through GJK + EPA.
MI_DEV bool gjkKey(u32 key) { return key == 5 || key == 9 || key == 10 || key == 11 || key == 14 || key == 15 || key == 16 || key == 17 || key == 23 || key == 29 || key == 35; }
MI_DEV Hull asHull(const ColliderRec& c, const float4* __restrict__ hullInfo)
{
	Hull h; h.q = q4f4(c.a); h.pos = v3(c.b.x, c.b.y, c.b.z);
	u32 g = (u32)c.b.w;
	h.first = __float_as_uint(hullInfo[2 * g].w); h.count = __float_as_uint(hullInfo[2 * g + 1].w);
	return h;
}
// Operands of one GJK/EPA instance by bucket key: 9 capsule-aabb, 10 capsule-obb, 14 cylinder-cylinder, 15 cylinder-aabb,
// 16 cylinder-obb (the obb variants work in the box's frame, :771-790, :1024-1043), and x-hull for x = 5 sphere, 11 capsule,
// 17 cylinder, 23 aabb, 29 obb, 35 hull.
template <int FAMILY>
MI_DEV void gjkOperands(u32 key, const ColliderRec& A, const ColliderRec& B, const float4* __restrict__ hullInfo, const float4* __restrict__ hullVerts, SupShapes& sh, Obb& o)
{
	sh.hullVerts = hullVerts;
	sh.tubeA = asCapsule(A); sh.tubeB = sh.tubeA;
	sh.box.lo = v3s(0.f); sh.box.hi = v3s(0.f); sh.boxA = sh.box;
	sh.obbA.q = q4(0.f, 0.f, 0.f, 1.f); sh.obbA.c = v3s(0.f); sh.obbA.r = v3s(0.f);
	sh.hullA.q = sh.obbA.q; sh.hullA.pos = v3s(0.f); sh.hullA.first = 0; sh.hullA.count = 0; sh.hullB = sh.hullA;
	sh.kindA = (key >= 14) ? SUP_CYLINDER : SUP_CAPSULE; sh.kindB = SUP_BOX;
	if (FAMILY == 1) // x vs hull
	{
		sh.kindB = SUP_HULL; sh.hullB = asHull(B, hullInfo);
		switch (key / 6)
		{
			case 0: { Sphere s = asSphere(A); sh.kindA = SUP_SPHERE; sh.tubeA.a = s.c; sh.tubeA.b = s.c; sh.tubeA.r = s.r; } break;
			case 1: sh.kindA = SUP_CAPSULE; break;
			case 2: sh.kindA = SUP_CYLINDER; break;
			case 3: sh.kindA = SUP_BOX; sh.boxA = asBox(A); break;
			case 4: sh.kindA = SUP_OBB; sh.obbA = asObb(A); break;
			default: sh.kindA = SUP_HULL; sh.hullA = asHull(A, hullInfo); break;
		}
		return;
	}
	if (key == 14) { sh.kindB = SUP_CYLINDER; sh.tubeB = asCapsule(B); return; }
	if (key == 9 || key == 15) { sh.box = asBox(B); return; }
	o = asObb(B);
	sh.box.lo = o.c - o.r; sh.box.hi = o.c + o.r;
	Capsule c_; c_.a = conjugate(o.q) * (sh.tubeA.a - o.c) + o.c; c_.b = conjugate(o.q) * (sh.tubeA.b - o.c) + o.c; c_.r = sh.tubeA.r;
	sh.tubeA = c_;
}

// ---------------------------------------------------------------------------------------------------------------
// K6/K7: pair kernels + contact emit (writeScalarContact, collision_narrow.cpp:2221-2253)
// ---------------------------------------------------------------------------------------------------------------
MI_DEV void writeManifold(ManifoldRec* __restrict__ out, u32 slot, const Man& m, bool hit, const ColliderRec& A, const ColliderRec& B, u32 pairIndexLo)
{
	ManifoldRec r;
	u32 count = hit ? m.count : 0;
	float friction = clamp01(sqrtf(colFriction(A) * colFriction(B)));
	float restitution = clamp01(fmaxf(colRestitution(A), colRestitution(B)));
	u32 fr = ((u32)(friction * 0xFFFF) << 16) | (u32)(restitution * 0xFFFF);
	for (u32 k = 0; k < 4; ++k) r.p[k] = (k < count) ? m.p[k] : make_float4(0.f, 0.f, 0.f, 0.f);
	r.nf = make_float4(m.n.x, m.n.y, m.n.z, __uint_as_float(fr));
	r.ids = make_uint4(colBody(A), colBody(B), count, pairIndexLo);
	out[slot] = r;
}

enum { GROUP_CLOSED = 0, GROUP_BOX = 1 };

// GROUP_CLOSED scans all valid slots and skips foreign buckets (its buckets are scattered over the key space);
// GROUP_BOX covers the contiguous slot range of keys 22..28 (aabb-obb, obb-obb; key 23, aabb-hull, is skipped: GJK + EPA).
template <int GROUP>
__global__ void __launch_bounds__(GROUP == GROUP_CLOSED ? 256 : 64) k_narrow(const u32* __restrict__ counters, const u32* __restrict__ keySorted, const u64* __restrict__ pairSorted,
	const ColliderRec* __restrict__ colWorld, ManifoldRec* __restrict__ manifolds)
{
	// the box kernel's clipping polygons: two per lane, 16 points each, point i of lane l at [i * 64 + l] (conflict-free 16-byte accesses)
	__shared__ float4 sPoly[GROUP == GROUP_BOX ? 2 * 16 * 64 : 1];
	const PolyStore store = { sPoly + (GROUP == GROUP_BOX ? threadIdx.x : 0u), sPoly + (GROUP == GROUP_BOX ? 16u * 64u + threadIdx.x : 0u), 64u };
	u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (GROUP == GROUP_BOX) { slot += counters[CTR_BUCKET_START + 22]; if (slot >= counters[CTR_BUCKET_START + 29]) return; }
	if (slot >= counters[CTR_NUM_VALID]) return;
	u32 key = keySorted[slot];
	bool mine;
	if (GROUP == GROUP_CLOSED) mine = (key == 0 || key == 1 || key == 2 || key == 3 || key == 4 || key == 7 || key == 8 || key == 21);
	else mine = (key == 22 || key == 28);
	if (!mine) return;
	u64 packed = pairSorted[slot];
	u32 ia = (u32)packed, ib = (u32)(packed >> 32);
	ColliderRec A = colWorld[ia], B = colWorld[ib];
	Man m; m.count = 0; m.n = v3(0.f, 1.f, 0.f);
	bool hit = false;
	if (GROUP == GROUP_CLOSED)
	{
		switch (key)
		{
			case 0: hit = sphereSphere(asSphere(A), asSphere(B), m); break;
			case 1: hit = sphereCapsule(asSphere(A), asCapsule(B), m); break;
			case 2: hit = sphereCylinder(asSphere(A), asCapsule(B), m); break;
			case 3: hit = sphereBox(asSphere(A), asBox(B), m); break;
			case 4: hit = sphereObb(asSphere(A), asObb(B), m); break;
			case 7: hit = capsuleCapsule(asCapsule(A), asCapsule(B), m); break;
			case 8: hit = capsuleCylinder(asCapsule(A), asCapsule(B), m); break;
			case 21: hit = boxBoxAxisAligned(asBox(A), asBox(B), m); break;
		}
	}
	else
	{
		Obb oa;
		if (key == 22) { Box b = asBox(A); oa.q = q4(0.f, 0.f, 0.f, 1.f); oa.c = boxCenter(b); oa.r = boxRadius(b); } // aabb -> obb (:1142-1148)
		else oa = asObb(A);
		hit = obbObb(oa, asObb(B), m, store);
	}
	writeManifold(manifolds, slot, m, hit, A, B, slot);
}

// Tube vs box / cylinder vs cylinder, phase 1: the GJK boolean test over the contiguous slot range of keys 9..16 (9, 10, 14, 15, 16
// occur; 11 is a hull pair, 12 and 13 cannot occur with typeA <= typeB).  Parallel cylinder pairs finish here in closed form.  Misses write
// their empty manifold; hits append (slot, simplex) to the EPA work list so that phase 2 runs with dense waves — the
// expanding polytope needs ~3.7 KB of private scratch per lane and 20 serial iterations, which must not idle behind misses.
template <int FAMILY>
__global__ void __launch_bounds__(256) k_gjk(u32* __restrict__ counters, const u32* __restrict__ keySorted, const u64* __restrict__ pairSorted,
	const ColliderRec* __restrict__ colWorld, ManifoldRec* __restrict__ manifolds, u32* __restrict__ epaList, float4* __restrict__ gjkSimplex,
	const float4* __restrict__ hullInfo, const float4* __restrict__ hullVerts, u32 listCap)
{
	// family 0: the contiguous slot range of keys 9..16 (11 = capsule-hull is skipped); family 1: keys 5..35 with key % 6 == 5
	u32 slot = counters[CTR_BUCKET_START + (FAMILY == 0 ? 9 : 5)] + blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= counters[CTR_BUCKET_START + (FAMILY == 0 ? 17 : 36)]) return;
	u32 key = keySorted[slot];
	if (FAMILY == 0 ? (key % 6 == 5) : (key % 6 != 5)) return;
	u64 packed = pairSorted[slot];
	ColliderRec A = colWorld[(u32)packed], B = colWorld[(u32)(packed >> 32)];
	SupShapes sh; Obb o;
	gjkOperands<FAMILY>(key, A, B, hullInfo, hullVerts, sh, o);
	if (key == 14)
	{
		Man m; m.count = 0; m.n = v3(0.f, 1.f, 0.f);
		int r = cylinderCylinderParallel(sh.tubeA, sh.tubeB, m);
		if (r != 2) { writeManifold(manifolds, slot, m, r == 1, A, B, slot); return; }
	}
	GjkSimplex sx;
	if (!gjkPair<FAMILY>(sh, sx))
	{
		Man m; m.count = 0; m.n = v3(0.f, 1.f, 0.f);
		writeManifold(manifolds, slot, m, false, A, B, slot);
		return;
	}
	u32 j = atomicAdd(&counters[FAMILY == 0 ? CTR_EPA_COUNT : CTR_EPA_COUNT_HULL], 1u); // order of the work list does not affect any result
	if (FAMILY == 1) j = listCap - 1u - j; // the hull pairs fill the list from its end
	epaList[j] = slot;
	float4* S = gjkSimplex + (size_t)j * 9;
	// 4 support points x 9 floats, packed as 9 float4 (written out: an array of pointers to the four points goes through scratch memory)
	const SupportPoint &p0 = sx.a, &p1 = sx.b, &p2 = sx.c, &p3 = sx.d;
	S[0] = make_float4(p0.a.x, p0.a.y, p0.a.z, p0.b.x); S[1] = make_float4(p0.b.y, p0.b.z, p0.mk.x, p0.mk.y); S[2] = make_float4(p0.mk.z, p1.a.x, p1.a.y, p1.a.z);
	S[3] = make_float4(p1.b.x, p1.b.y, p1.b.z, p1.mk.x); S[4] = make_float4(p1.mk.y, p1.mk.z, p2.a.x, p2.a.y); S[5] = make_float4(p2.a.z, p2.b.x, p2.b.y, p2.b.z);
	S[6] = make_float4(p2.mk.x, p2.mk.y, p2.mk.z, p3.a.x); S[7] = make_float4(p3.a.y, p3.a.z, p3.b.x, p3.b.y); S[8] = make_float4(p3.b.z, p3.mk.x, p3.mk.y, p3.mk.z);
}

// Phase 2: EPA (EPA_GROUP lanes per hit = two hits per wave, polytopes in LDS) + face clipping (the group's first lane) for every GJK
// hit.  Groups stride over the work list.
#define EPA_WAVES_PER_BLOCK 4
#define EPA_PER_WAVE (64 / EPA_GROUP)
template <int FAMILY>
__global__ void __launch_bounds__(64 * EPA_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(FAMILY == 0 ? 3 : 2, FAMILY == 0 ? 3 : 3))) k_epa(const u32* __restrict__ counters, const u32* __restrict__ keySorted, const u64* __restrict__ pairSorted,
	const ColliderRec* __restrict__ colWorld, ManifoldRec* __restrict__ manifolds, const u32* __restrict__ epaList, const float4* __restrict__ gjkSimplex,
	const float4* __restrict__ hullInfo, const float4* __restrict__ hullVerts, u32 listCap)
{
	__shared__ EpaWave shared[EPA_WAVES_PER_BLOCK * EPA_PER_WAVE];
	const u32 wave = threadIdx.x >> 6, group = (threadIdx.x & 63u) / EPA_GROUP, lane = threadIdx.x & (EPA_GROUP - 1u);
	EpaWave& e = shared[wave * EPA_PER_WAVE + group];
	u32 numHits = counters[FAMILY == 0 ? CTR_EPA_COUNT : CTR_EPA_COUNT_HULL];
	u32 stride = gridDim.x * EPA_WAVES_PER_BLOCK * EPA_PER_WAVE;
	for (u32 jj = (blockIdx.x * EPA_WAVES_PER_BLOCK + wave) * EPA_PER_WAVE + group; jj < numHits; jj += stride)
	{
		u32 j = (FAMILY == 0) ? jj : listCap - 1u - jj;
		u32 slot = epaList[j];
		u32 key = keySorted[slot];
		u64 packed = pairSorted[slot];
		ColliderRec A = colWorld[(u32)packed], B = colWorld[(u32)(packed >> 32)];
		SupShapes sh; Obb o;
		gjkOperands<FAMILY>(key, A, B, hullInfo, hullVerts, sh, o);
		const float4* S = gjkSimplex + (size_t)j * 9;
		GjkSimplex sx; sx.numPoints = 4;
		{
			const float4 s0 = S[0], s1 = S[1], s2 = S[2], s3 = S[3], s4 = S[4], s5 = S[5], s6 = S[6], s7 = S[7], s8 = S[8];
			sx.a.a = v3(s0.x, s0.y, s0.z); sx.a.b = v3(s0.w, s1.x, s1.y); sx.a.mk = v3(s1.z, s1.w, s2.x);
			sx.b.a = v3(s2.y, s2.z, s2.w); sx.b.b = v3(s3.x, s3.y, s3.z); sx.b.mk = v3(s3.w, s4.x, s4.y);
			sx.c.a = v3(s4.z, s4.w, s5.x); sx.c.b = v3(s5.y, s5.z, s5.w); sx.c.mk = v3(s6.x, s6.y, s6.z);
			sx.d.a = v3(s6.w, s7.x, s7.y); sx.d.b = v3(s7.z, s7.w, s8.x); sx.d.mk = v3(s8.y, s8.z, s8.w);
		}
		V3 point, normal; float depth;
		epaWave<FAMILY>(e, lane, group * EPA_GROUP, sx, sh, point, normal, depth);
		if (lane == 0)
		{
			Man m; m.count = 0; m.n = v3(0.f, 1.f, 0.f);
			if (FAMILY == 1 || key == 14) { m.n = normal; m.count = 1; m.p[0] = make_float4(point.x, point.y, point.z, depth); } // :905-950; hull pairs
			else
			{
				const PolyStore store = { (float4*)e.tnx, (float4*)e.tnx + 16, 1u }; // the finished polytope's triangle arrays (2 KB, 16-byte aligned) hold the two 16-point clipping polygons: LDS, not scratch
				capsuleBoxFinish(point, normal, depth, sh.tubeA, sh.box, m, store);
			}
			if (key == 10 || key == 16) // back to world space (:779-787, :1032-1040)
			{
				m.n = o.q * m.n;
#pragma unroll
				for (u32 i = 0; i < 4; ++i)
				{
					if (i >= m.count) break;
					V3 pt = o.q * (v3f4(m.p[i]) - o.c) + o.c;
					m.p[i] = make_float4(pt.x, pt.y, pt.z, m.p[i].w);
				}
			}
			writeManifold(manifolds, slot, m, true, A, B, slot);
		}
	}
}


// ---------------------------------------------------------------------------------------------------------------
// Non-collision interactions: rigid body vs force-field / trigger collider — overlapCheck, collision_narrow.cpp:1593-1689, over the
// boolean tests of bounding_volumes.h:301-363 and bounding_volumes.cpp:704-835, 1079-1244.  One lane per pair of the KEY_ZONE bucket.
// A hit sets the body's bit of the field (k_apply_fields adds the forces in ascending field id) or enters the
// (trigger, body) pair into this step's overlap set (enter event if the previous step's set does not hold it).
// ---------------------------------------------------------------------------------------------------------------
MI_DEV bool ovSphereSphere(V3 ca, float ra, V3 cb, float rb) { V3 d = ca - cb; float dist2 = dot(d, d); float radiusSum = ra + rb; return dist2 <= radiusSum * radiusSum; }
MI_DEV bool ovSphereCylinder(V3 sc, float sr, Capsule c) // bounding_volumes.cpp:704-724
{
	V3 ab = c.b - c.a;
	float t = dot(sc - c.a, ab) / sqlen(ab);
	if (t >= 0.f && t <= 1.f) return ovSphereSphere(sc, sr, lerp(c.a, c.b, t), c.r);
	V3 p = (t <= 0.f) ? c.a : c.b;
	V3 up = (t <= 0.f) ? -ab : ab;
	V3 projectedDirToCenter = normalize(cross(cross(up, sc - p), up));
	V3 endA = p + projectedDirToCenter * c.r;
	V3 endB = p - projectedDirToCenter * c.r;
	V3 closestToSphere = closestPointSegment(sc, endA, endB);
	float sqDistance = sqlen(closestToSphere - sc);
	return sqDistance <= sr; // sic (:723)
}
MI_DEV bool ovSphereBox(V3 sc, float sr, Box a) // bounding_volumes.h:320-326
{
	V3 p = v3(fminf(fmaxf(sc.x, a.lo.x), a.hi.x), fminf(fmaxf(sc.y, a.lo.y), a.hi.y), fminf(fmaxf(sc.z, a.lo.z), a.hi.z));
	V3 n = p - sc;
	return sqlen(n) <= sr * sr;
}
MI_DEV bool ovObbObb(const Obb& a, const Obb& b) // bounding_volumes.cpp:1079-1199
{
	V3 ax = a.q * v3(1.f, 0.f, 0.f), ay = a.q * v3(0.f, 1.f, 0.f), az = a.q * v3(0.f, 0.f, 1.f);
	V3 bx = b.q * v3(1.f, 0.f, 0.f), by = b.q * v3(0.f, 1.f, 0.f), bz = b.q * v3(0.f, 0.f, 1.f);
	M3 r;
	r.m00 = dot(ax, bx); r.m10 = dot(ay, bx); r.m20 = dot(az, bx);
	r.m01 = dot(ax, by); r.m11 = dot(ay, by); r.m21 = dot(az, by);
	r.m02 = dot(ax, bz); r.m12 = dot(ay, bz); r.m22 = dot(az, bz);
	V3 tw = b.c - a.c;
	V3 t = conjugate(a.q) * tw;
	M3 absR;
	absR.m00 = fabsf(r.m00) + MI_EPSILON; absR.m10 = fabsf(r.m10) + MI_EPSILON; absR.m20 = fabsf(r.m20) + MI_EPSILON;
	absR.m01 = fabsf(r.m01) + MI_EPSILON; absR.m11 = fabsf(r.m11) + MI_EPSILON; absR.m21 = fabsf(r.m21) + MI_EPSILON;
	absR.m02 = fabsf(r.m02) + MI_EPSILON; absR.m12 = fabsf(r.m12) + MI_EPSILON; absR.m22 = fabsf(r.m22) + MI_EPSILON;
	float ra, rb;
	for (u32 i = 0; i < 3; ++i) { ra = vget(a.r, i); rb = dot(mrow(absR, i), b.r); if (ra + rb - fabsf(vget(t, i)) < 0.f) return false; }
	for (u32 i = 0; i < 3; ++i) { ra = dot(mcol(absR, i), a.r); rb = vget(b.r, i); float d = dot(mcol(r, i), t); if (ra + rb - fabsf(d) < 0.f) return false; }
#define MI_OV_EDGE(RA, RB, DIST) ra = RA; rb = RB; if (ra + rb - fabsf(DIST) < 0.f) return false;
	MI_OV_EDGE(a.r.y * absR.m20 + a.r.z * absR.m10, b.r.y * absR.m02 + b.r.z * absR.m01, t.z * r.m10 - t.y * r.m20)
	MI_OV_EDGE(a.r.y * absR.m21 + a.r.z * absR.m11, b.r.x * absR.m02 + b.r.z * absR.m00, t.z * r.m11 - t.y * r.m21)
	MI_OV_EDGE(a.r.y * absR.m22 + a.r.z * absR.m12, b.r.x * absR.m01 + b.r.y * absR.m00, t.z * r.m12 - t.y * r.m22)
	MI_OV_EDGE(a.r.x * absR.m20 + a.r.z * absR.m00, b.r.y * absR.m12 + b.r.z * absR.m11, t.x * r.m20 - t.z * r.m00)
	MI_OV_EDGE(a.r.x * absR.m21 + a.r.z * absR.m01, b.r.x * absR.m12 + b.r.z * absR.m10, t.x * r.m21 - t.z * r.m01)
	MI_OV_EDGE(a.r.x * absR.m22 + a.r.z * absR.m02, b.r.x * absR.m11 + b.r.y * absR.m10, t.x * r.m22 - t.z * r.m02)
	MI_OV_EDGE(a.r.x * absR.m10 + a.r.y * absR.m00, b.r.y * absR.m22 + b.r.z * absR.m21, t.y * r.m00 - t.x * r.m10)
	MI_OV_EDGE(a.r.x * absR.m11 + a.r.y * absR.m01, b.r.x * absR.m22 + b.r.z * absR.m20, t.y * r.m01 - t.x * r.m11)
	MI_OV_EDGE(a.r.x * absR.m12 + a.r.y * absR.m02, b.r.x * absR.m21 + b.r.y * absR.m20, t.y * r.m02 - t.x * r.m12)
#undef MI_OV_EDGE
	return true;
}
MI_DEV bool overlapColliders(u32 key, const ColliderRec& A, const ColliderRec& B, const float4* __restrict__ hullInfo, const float4* __restrict__ hullVerts)
{
	switch (key)
	{
		case 0: { Sphere a = asSphere(A), b = asSphere(B); return ovSphereSphere(a.c, a.r, b.c, b.r); }
		case 1: { Sphere s = asSphere(A); Capsule c = asCapsule(B); return ovSphereSphere(s.c, s.r, closestPointSegment(s.c, c.a, c.b), c.r); }
		case 2: { Sphere s = asSphere(A); return ovSphereCylinder(s.c, s.r, asCapsule(B)); }
		case 3: { Sphere s = asSphere(A); return ovSphereBox(s.c, s.r, asBox(B)); }
		case 4: { Sphere s = asSphere(A); Obb o = asObb(B); Box b; b.lo = o.c - o.r; b.hi = o.c + o.r; return ovSphereBox(conjugate(o.q) * (s.c - o.c) + o.c, s.r, b); }
		case 7: { Capsule a = asCapsule(A), b = asCapsule(B); V3 c1, c2; closestSegmentSegment(a.a, a.b, b.a, b.b, c1, c2); return ovSphereSphere(c1, a.r, c2, b.r); }
		case 8: { Capsule a = asCapsule(A), b = asCapsule(B); V3 c1, c2; closestSegmentSegment(a.a, a.b, b.a, b.b, c1, c2); return ovSphereCylinder(c1, a.r, b); }
		case 21: { Box a = asBox(A), b = asBox(B); return !(a.hi.x < b.lo.x || a.lo.x > b.hi.x || a.hi.y < b.lo.y || a.lo.y > b.hi.y || a.hi.z < b.lo.z || a.lo.z > b.hi.z); }
		case 22: { Box b = asBox(A); Obb oa; oa.q = q4(0.f, 0.f, 0.f, 1.f); oa.c = boxCenter(b); oa.r = boxRadius(b); return ovObbObb(oa, asObb(B)); }
		case 28: return ovObbObb(asObb(A), asObb(B));
		default: break;
	}
	SupShapes sh; Obb o; GjkSimplex sx;
	if (key % 6 == 5) { gjkOperands<1>(key, A, B, hullInfo, hullVerts, sh, o); return gjkPair<1>(sh, sx); } // x vs hull
	if (key == 9 || key == 10 || key == 14 || key == 15 || key == 16) { gjkOperands<0>(key, A, B, hullInfo, hullVerts, sh, o); return gjkPair<0>(sh, sx); } // no closed form for parallel cylinders here (:790-797)
	return false;
}

__global__ void __launch_bounds__(64) k_zone_overlap(u32* __restrict__ counters, const u64* __restrict__ pairSorted, const ColliderRec* __restrict__ colWorld,
	const float4* __restrict__ hullInfo, const float4* __restrict__ hullVerts, u32 nb, u32* __restrict__ fieldMask, u32 fieldWords, PairSetView triggers, EventSink sink)
{
	u32 slot = counters[CTR_BUCKET_START + KEY_ZONE] + blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= counters[CTR_BUCKET_START + KEY_INVALID]) return;
	u64 packed = pairSorted[slot];
	ColliderRec A = colWorld[(u32)packed], B = colWorld[(u32)(packed >> 32)];
	if (!overlapColliders(colType(A) * 6 + colType(B), A, B, hullInfo, hullVerts)) return;
	bool zoneIsA = (__float_as_uint(A.d.w) & 0xFFu) != 0;
	u32 flags = __float_as_uint(zoneIsA ? A.d.w : B.d.w);
	u32 body = colBody(zoneIsA ? B : A), zoneType = flags & 0xFFu, zoneIndex = flags >> 8;
	if (body >= nb) return;
	if (zoneType == 2u) { if (fieldWords) atomicOr(&fieldMask[(size_t)body * fieldWords + (zoneIndex >> 5)], 1u << (zoneIndex & 31u)); }
	else if (triggers.cur)
	{
		u64 key = ((u64)zoneIndex << 32) | body;
		if (pairSetInsert(triggers.cur, triggers.mask, triggers.shift, key, counters) && !pairSetContains(triggers.prev, triggers.mask, triggers.shift, key) && eventIsMine(sink, body, body, nb))
			eventWritePlain(sink, EVENT_TRIGGER_ENTER, zoneIndex, body, 0xFFFFFFFFu, body);
	}
}

void launch_narrowphase(World& w, u32 numPairs)
{
	if (!numPairs)
	{
		if (w.terrainChunksPerDim) hipLaunchKernelGGL(k_bucket_offsets, dim3(1), dim3(64), 0, w.stream, w.dCounters.p, w.pairKeySorted.p); // no pairs: all buckets empty, the terrain contacts start at slot 0
		return;
	}
	dim3 grid((numPairs + 255) / 256), block(256);
	hipLaunchKernelGGL(k_classify, grid, block, 0, w.stream, w.dCounters.p, w.nb, w.pairs.p, w.colWorld.p, w.aabbMin.p, w.stats.numInternalSteps & 1u, w.pairKey.p, (u64*)w.pairsSorted.p + numPairs, numPairs);
	// sort (bucket key, packed pair): unsorted packed pairs live in the upper half of pairsSorted, sorted ones in the lower half
	csort_pairs_u64(w, w.pairKey.p, w.pairKeySorted.p, (const u64*)w.pairsSorted.p + numPairs, (u64*)w.pairsSorted.p, numPairs, 64);
	hipLaunchKernelGGL(k_bucket_offsets, dim3(1), dim3(64), 0, w.stream, w.dCounters.p, w.pairKeySorted.p);
	const u64* sortedPairs = (const u64*)w.pairsSorted.p;
	u32 listCap = (u32)w.pairCap;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gjk<0>), grid, block, 0, w.stream, w.dCounters.p, w.pairKeySorted.p, sortedPairs, w.colWorld.p, w.manifolds.p, w.epaList.p, w.gjkSimplex.p, w.hullInfo.p, w.hullVerts.p, listCap);
	if (!w.hulls.empty())
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gjk<1>), grid, block, 0, w.stream, w.dCounters.p, w.pairKeySorted.p, sortedPairs, w.colWorld.p, w.manifolds.p, w.epaList.p, w.gjkSimplex.p, w.hullInfo.p, w.hullVerts.p, listCap);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_narrow<GROUP_CLOSED>), grid, block, 0, w.stream, w.dCounters.p, w.pairKeySorted.p, sortedPairs, w.colWorld.p, w.manifolds.p);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_narrow<GROUP_BOX>), dim3((numPairs + 63) / 64), dim3(64), 0, w.stream, w.dCounters.p, w.pairKeySorted.p, sortedPairs, w.colWorld.p, w.manifolds.p);
	// The waves stride over the hit list: launch exactly as many workgroups as are resident at once (registers allow 3 per CU, the 38.5 KB
	// of LDS would allow 4), or the last quarter of a 4-per-CU grid runs as a second, mostly empty round.
	static int epaPerCU[2] = { 0, 0 }; static int numCUs = 0;
	if (!numCUs)
	{
		hipDeviceProp_t prop; MI_CHECK(hipGetDeviceProperties(&prop, w.device)); numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		MI_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&epaPerCU[0], k_epa<0>, 64 * EPA_WAVES_PER_BLOCK, 0));
		MI_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&epaPerCU[1], k_epa<1>, 64 * EPA_WAVES_PER_BLOCK, 0));
		if (const char* e = getenv("MI_EPA_BLOCKS_PER_CU")) epaPerCU[0] = epaPerCU[1] = std::max(1, atoi(e));
		for (int& v : epaPerCU) if (v < 1) v = 3;
	}
	u32 epaBlocks = std::min<u32>((numPairs + EPA_WAVES_PER_BLOCK - 1) / EPA_WAVES_PER_BLOCK, (u32)(numCUs * epaPerCU[0]));
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_epa<0>), dim3(epaBlocks), dim3(64 * EPA_WAVES_PER_BLOCK), 0, w.stream, w.dCounters.p, w.pairKeySorted.p, sortedPairs, w.colWorld.p, w.manifolds.p, w.epaList.p, w.gjkSimplex.p, w.hullInfo.p, w.hullVerts.p, listCap);
	if (!w.hulls.empty())
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_epa<1>), dim3(std::min<u32>((numPairs + EPA_WAVES_PER_BLOCK - 1) / EPA_WAVES_PER_BLOCK, (u32)(numCUs * epaPerCU[1]))), dim3(64 * EPA_WAVES_PER_BLOCK), 0, w.stream, w.dCounters.p, w.pairKeySorted.p, sortedPairs, w.colWorld.p, w.manifolds.p, w.epaList.p, w.gjkSimplex.p, w.hullInfo.p, w.hullVerts.p, listCap);
}

// Force fields and triggers: boolean overlap tests on the pairs behind the collision buckets (collision_narrow.cpp:2573-2593).  Not
// part of launch_narrowphase: it enters pairs into the trigger set and raises events, so it must run exactly once per step.
void launch_zone_overlap(World& w, u32 numPairs)
{
	if (!numPairs || (w.fields.empty() && w.triggers.empty())) return;
	const u64* sortedPairs = (const u64*)w.pairsSorted.p;
	PairSetView tv = { w.triggers.empty() ? nullptr : w.triggerSet[w.triggerCur].p, w.triggers.empty() ? nullptr : w.triggerSet[w.triggerCur ^ 1].p, w.triggerSetSize - 1, 64u - (u32)__builtin_ctz(w.triggerSetSize ? w.triggerSetSize : 2u) };
	EventSink sink = sinkOf(w);
	hipLaunchKernelGGL(k_zone_overlap, dim3((numPairs + 63) / 64), dim3(64), 0, w.stream, w.dCounters.p, sortedPairs, w.colWorld.p, w.hullInfo.p, w.hullVerts.p, w.nb,
		w.fieldMask.p, w.fieldWords, tv, sink);
}
