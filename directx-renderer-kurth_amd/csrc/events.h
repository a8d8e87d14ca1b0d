// Device-side event machinery (row N2 of SURVEY §8f): trigger enter/leave and collision begin/end events — reference physics.cpp:952-1178.
// The reference sorts this frame's overlap list and merges it against the previous frame's on the CPU.  Here both frames live in
// HBM as open-addressing hash sets of 64-bit pair keys: a pair that is inserted now and absent from the previous set raises an
// enter/begin event, a scan of the previous set raises leave/end events for keys absent now (and clears the table for its next
// use).  Events are appended to a device ring; the host sorts each step's events by pair key when it drains them, which is the
// order the reference's merge loop calls back in.
#pragma once
#include "world.h"

struct EventRec { u32 kind, step, a, b, bodyA, bodyB; float position[3], normal[3], relativeVelocity[3]; }; // = mi_event (include/mi_physics.h)
enum { EVENT_TRIGGER_ENTER = 0, EVENT_TRIGGER_LEAVE = 1, EVENT_COLLISION_BEGIN = 2, EVENT_COLLISION_END = 3 };
#define PAIRSET_EMPTY 0xFFFFFFFFFFFFFFFFull
#define PAIRSET_MAX_PROBES 2048u

struct PairSetView { u64* cur; u64* prev; u32 mask, shift; };
struct EventSink { EventRec* events; u32* counters; u32 capacity, step; const uint8_t* slabCode; };
// Slab worlds (one rank per GPU): a pair near a cut is simulated — and its overlap sets are kept — on both ranks.  Exactly one of
// them REPORTS its events: the rank that owns the pair's dynamic body with the lower index (for a trigger: the body).  That rank
// always simulates both bodies while they touch (the partner is its ghost); the sets themselves are not filtered, so a pair that stays
// in contact while its reporter changes hands raises nothing.  physics.cpp:1037-1178: one begin / end per pair.
MI_DEV bool eventIsMine(const EventSink& sink, u32 bodyA, u32 bodyB, u32 nb)
{
	if (!sink.slabCode) return true;
	u32 reporter = (bodyA < nb && bodyB < nb) ? min(bodyA, bodyB) : (bodyA < nb ? bodyA : bodyB);
	return reporter >= nb || sink.slabCode[reporter] == MI_SLAB_OWNED;
}

MI_DEV u32 pairSetHash(u64 key, u32 shift) { return (u32)((key * 0x9E3779B97F4A7C15ull) >> shift); }
// Returns true if this call inserted the key (exactly one caller per distinct key does).  A full table raises CTR_EVENT_OVERFLOW bit 1.
MI_DEV bool pairSetInsert(u64* __restrict__ table, u32 mask, u32 shift, u64 key, u32* __restrict__ counters)
{
	u32 h = pairSetHash(key, shift);
	for (u32 probe = 0; probe < PAIRSET_MAX_PROBES; ++probe)
	{
		u64 old = atomicCAS((unsigned long long*)&table[h], (unsigned long long)PAIRSET_EMPTY, (unsigned long long)key);
		if (old == PAIRSET_EMPTY) return true;
		if (old == key) return false;
		h = (h + 1) & mask;
	}
	atomicOr(&counters[CTR_EVENT_OVERFLOW], 2u);
	return false;
}
MI_DEV bool pairSetContains(const u64* __restrict__ table, u32 mask, u32 shift, u64 key)
{
	u32 h = pairSetHash(key, shift);
	for (u32 probe = 0; probe < PAIRSET_MAX_PROBES; ++probe)
	{
		u64 v = table[h];
		if (v == key) return true;
		if (v == PAIRSET_EMPTY) return false;
		h = (h + 1) & mask;
	}
	return false;
}
MI_DEV EventRec* eventAppend(const EventSink& sink)
{
	u32 j = atomicAdd(&sink.counters[CTR_EVENT_COUNT], 1u);
	if (j >= sink.capacity) { atomicOr(&sink.counters[CTR_EVENT_OVERFLOW], 1u); return nullptr; }
	return sink.events + j;
}
MI_DEV void eventWritePlain(const EventSink& sink, u32 kind, u32 a, u32 b, u32 bodyA, u32 bodyB)
{
	EventRec* e = eventAppend(sink);
	if (!e) return;
	EventRec r; r.kind = kind; r.step = sink.step; r.a = a; r.b = b; r.bodyA = bodyA; r.bodyB = bodyB;
	for (int k = 0; k < 3; ++k) { r.position[k] = 0.f; r.normal[k] = 0.f; r.relativeVelocity[k] = 0.f; }
	*e = r;
}
