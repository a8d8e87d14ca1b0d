// Force fields, trigger events, collision begin/end events (row N2 of SURVEY §8f) — reference physics.cpp:759-787, 952-1178.
// None of this is launched for a world without force fields / triggers / enabled collision events.
#include "events.h"

// ---------------------------------------------------------------------------------------------------------------
// Force fields -> force accumulators, before the force integration (physics.cpp:963-967 localized, :1273 global).
// A body's localized forces are added in ascending field id (deterministic; the reference adds them in the order of its pair list),
// then the sum of the global fields.  Consumes and clears the per-body field bits set by k_zone_overlap.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_apply_fields(u32 nb, float4* __restrict__ force, u32* __restrict__ fieldMask, u32 fieldWords, const float4* __restrict__ fieldForce,
	float gx, float gy, float gz, u32 anyGlobal, const uint8_t* __restrict__ simMask)
{
	u32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nb) return;
	float4 F = force[2 * i];
	bool changed = false;
	for (u32 wd = 0; wd < fieldWords; ++wd)
	{
		u32 bits = fieldMask[(size_t)i * fieldWords + wd];
		if (!bits) continue;
		fieldMask[(size_t)i * fieldWords + wd] = 0u;
		while (bits)
		{
			u32 f = wd * 32u + (u32)__builtin_ctz(bits);
			bits &= bits - 1u;
			float4 ff = fieldForce[f];
			F.x += ff.x; F.y += ff.y; F.z += ff.z;
			changed = true;
		}
	}
	if (!simMask[i]) return;
	if (anyGlobal) { F.x += gx; F.y += gy; F.z += gz; changed = true; }
	if (changed) force[2 * i] = F;
}

void launch_apply_fields(World& w)
{
	if (w.fields.empty() || !w.nb) return;
	w.uploadFields();
	if (!w.fieldWords && !w.anyGlobalForce) return;
	hipLaunchKernelGGL(k_apply_fields, dim3((w.nb + 255) / 256), dim3(256), 0, w.stream, w.nb, w.force.p, w.fieldMask.p, w.fieldWords, w.fieldForce.p,
		w.globalForce[0], w.globalForce[1], w.globalForce[2], w.anyGlobalForce ? 1u : 0u, w.simMask.p);
}

// ---------------------------------------------------------------------------------------------------------------
// Leave / end events: every key of the previous step's set that this step's set does not hold; the slot is emptied for the
// table's next turn as "this step".
// ---------------------------------------------------------------------------------------------------------------
template <bool COLLISIONS>
__global__ void __launch_bounds__(256) k_scan_previous_set(PairSetView set, EventSink sink, const ColliderRec* __restrict__ colWorld, u32 nb, u32 emit)
{
	u32 s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s > set.mask) return;
	u64 key = set.prev[s];
	if (key == PAIRSET_EMPTY) return;
	set.prev[s] = PAIRSET_EMPTY;
	if (!emit || pairSetContains(set.cur, set.mask, set.shift, key)) return;
	u32 a = (u32)(key >> 32), b = (u32)key;
	if (COLLISIONS)
	{
		u32 bodyA = colBody(colWorld[a]), bodyB = colBody(colWorld[b]);
		if (eventIsMine(sink, bodyA, bodyB, nb)) eventWritePlain(sink, EVENT_COLLISION_END, a, b, bodyA < nb ? bodyA : 0xFFFFFFFFu, bodyB < nb ? bodyB : 0xFFFFFFFFu);
	}
	else if (eventIsMine(sink, b, b, nb)) eventWritePlain(sink, EVENT_TRIGGER_LEAVE, a, b, 0xFFFFFFFFu, b);
}

static PairSetView viewOf(DevBuf<u64>* tables, u32 size, u32 cur)
{
	PairSetView v; v.cur = tables[cur].p; v.prev = tables[cur ^ 1].p; v.mask = size - 1; v.shift = 64u - (u32)__builtin_ctz(size);
	return v;
}
EventSink sinkOf(World& w) { EventSink s = { (EventRec*)w.eventRing.p, w.dCounters.p, w.eventCap, w.stats.numInternalSteps, w.slabSize > 1 ? w.slabCode.p : nullptr }; return s; }

void launch_trigger_events(World& w)
{
	if (w.triggers.empty() || !w.triggerSetSize) return;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_previous_set<false>), dim3((w.triggerSetSize + 255) / 256), dim3(256), 0, w.stream, viewOf(w.triggerSet, w.triggerSetSize, w.triggerCur), sinkOf(w), w.colWorld.p, w.nb, 1u);
	w.triggerCur ^= 1u;
}

// ---------------------------------------------------------------------------------------------------------------
// Begin events: one lane per manifold slot; a slot with contacts enters its collider pair into this step's set, a pair the previous
// step's set does not hold raises the event — mean contact point and normal, relative velocity of the two bodies at that point from
// the velocities after the force integration (physics.cpp:1075-1115).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_collision_begin(u32* __restrict__ counters, const u64* __restrict__ pairSorted, const ManifoldRec* __restrict__ manifolds,
	const float4* __restrict__ vel, const float4* __restrict__ cog, u32 nb, PairSetView set, EventSink sink, u32 emit)
{
	u32 slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= counters[CTR_BUCKET_START + 62]) return; // pair manifolds only: terrain contacts raise no events (physics.cpp:1049: colliderB < numColliders)
	uint4 ids = manifolds[slot].ids;
	u32 count = ids.z;
	if (!count) return;
	u64 packed = pairSorted[slot];
	u32 a = (u32)packed, b = (u32)(packed >> 32);
	u64 key = ((u64)a << 32) | b;
	if (!pairSetInsert(set.cur, set.mask, set.shift, key, counters)) return;
	if (!emit || pairSetContains(set.prev, set.mask, set.shift, key)) return;
	if (!eventIsMine(sink, ids.x, ids.y, nb)) return;
	EventRec* e = eventAppend(sink);
	if (!e) return;
	ManifoldRec m = manifolds[slot];
	float norm = 1.f / (float)count;
	V3 point = v3s(0.f), normal = v3s(0.f), n = v3(m.nf.x, m.nf.y, m.nf.z);
	for (u32 k = 0; k < count; ++k) { point += v3f4(m.p[k]); normal += n; }
	point = point * norm; normal = normal * norm;
	u32 bodyA = ids.x, bodyB = ids.y; // nb = the static dummy: zero velocity at the origin (physics.cpp:1279)
	V3 velA = v3f4(vel[2 * bodyA]) + cross(v3f4(vel[2 * bodyA + 1]), point - v3f4(cog[bodyA]));
	V3 velB = v3f4(vel[2 * bodyB]) + cross(v3f4(vel[2 * bodyB + 1]), point - v3f4(cog[bodyB]));
	V3 rel = velB - velA;
	EventRec r; r.kind = EVENT_COLLISION_BEGIN; r.step = sink.step; r.a = a; r.b = b; r.bodyA = bodyA < nb ? bodyA : 0xFFFFFFFFu; r.bodyB = bodyB < nb ? bodyB : 0xFFFFFFFFu;
	r.position[0] = point.x; r.position[1] = point.y; r.position[2] = point.z;
	r.normal[0] = normal.x; r.normal[1] = normal.y; r.normal[2] = normal.z;
	r.relativeVelocity[0] = rel.x; r.relativeVelocity[1] = rel.y; r.relativeVelocity[2] = rel.z;
	*e = r;
}

void launch_collision_events(World& w, u32 numPairs)
{
	if (!(w.collisionBeginEvents || w.collisionEndEvents) || !w.collisionSetSize) return;
	PairSetView v = viewOf(w.collisionSet, w.collisionSetSize, w.collisionCur);
	if (numPairs)
		hipLaunchKernelGGL(k_collision_begin, dim3((numPairs + 255) / 256), dim3(256), 0, w.stream, w.dCounters.p, (const u64*)w.pairsSorted.p, w.manifolds.p, w.vel.p, w.cog.p, w.nb, v, sinkOf(w), w.collisionBeginEvents ? 1u : 0u);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_previous_set<true>), dim3((w.collisionSetSize + 255) / 256), dim3(256), 0, w.stream, v, sinkOf(w), w.colWorld.p, w.nb, w.collisionEndEvents ? 1u : 0u);
	w.collisionCur ^= 1u;
}
