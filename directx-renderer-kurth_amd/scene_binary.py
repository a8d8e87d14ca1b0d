"""Scene exchange with the reference engine's BINARY entity streams, physics subset.

The reference serialises ONE entity to memory as the sequence of its components in the fixed order of `serialized_components`
(serialization_binary.cpp:105-133): per component a `bool` (1 byte: present?) followed, when present, by the component's payload
(:420-443; `serializeEntityToMemory` / `deserializeEntityFromMemory`, :452-465).  Payloads are raw `memcpy`s of the C++ structs
(:136-148) unless a specialisation says otherwise.  The ones on the rigid-body path, with the byte layouts that follow from the
struct definitions under the MSVC x64 ABI (`quat` holds an `__m128`, core/math.h:292-304, so it and everything containing it is
16-byte aligned; nothing else on this path is aligned beyond 4 except the hull's pointer union):

    tag_component                 char name[16]                                                   (scene/components.h:6-17)
    transform_component           trs = quat rotation @0, vec3 position @16, vec3 scale @28; 48 B  (core/math.h:494-499)
    position / position_rotation / position_scale components: 12 / 32 (vec3 @0, quat @16) / 24 B  (components.h:44-75)
    dynamic_transform_component   no payload                                                      (serialization_binary.cpp:152-153)
    rigid_body_component          112 B: localCOG @0, invMass @12, invInertia @16 (mat3, column-major: core/math.h:4, 389-396),
                                  gravityFactor @52, linearDamping @56, angularDamping @60, linearVelocity @64, angularVelocity @76,
                                  forceAccumulator @88, torqueAccumulator @100                    (physics/rigid_body.h:31-45)
    force_field_component         vec3 force                                                      (physics/physics.h:182-185)
    cloth_component               width, height (f32), gridSizeX, gridSizeY (u32), totalMass, stiffness, damping, gravityFactor  (:173-199)
    cloth_render_component        no payload                                                      (:201-202)
    physics_reference_component   u32 numColliders, numColliders x collider_union (80 B: shape union @0 (48 B), physics_material
                                  {i32 type, restitution, friction, density} @48, u8 type @64, u8 objectType @65, u16 objectIndex @66),
                                  u32 numConstraints, per constraint {i32 constraint_type, u32 entityA, u32 entityB, the constraint's
                                  POD}                                                            (:204-274; physics/physics.h:84-106)
    heightmap_collider_component  u32 chunksPerDim, f32 chunkSize, physics_material (16 B)        (:378-394; written for a scene with
                                  a heightmap, skipped on read: the heights come from the terrain generator, not from the stream)

Everything else in the group (mesh, lights, terrain, grass, placement, water) is written as "absent"; a stream that holds one of
them cannot be read here (their payloads are renderer structs) and raises.  The layouts are DERIVED from the reference's headers —
the reference cannot be compiled here, so no stream written by the engine itself was available to check them against; the tests
pin the round trip and the documented offsets.

A constraint is listed in the stream of BOTH entities it joins (each walks its own constraint edges, :213-231); reading a set of
entities adds it once.  Unlike the YAML layout the binary one carries velocities and accumulators (raw rigid_body_component), so a
running world can be written and resumed from it.  Triggers are not in `serialized_components` and are not written (nor by the reference).
"""
import struct

import numpy as np

from . import scenes

CONSTRAINT_BYTES = [28, 24, 40, 104, 120, 72]       # distance, ball, fixed, hinge, cone_twist, slider (constraints.h:73-520)
CONSTRAINT_NAMES = ["distance", "ball", "fixed", "hinge", "cone_twist", "slider"]
COLLIDER_UNION_BYTES = 80
# order of serialized_components (serialization_binary.cpp:105-133)
GROUP = ["tag", "transform", "position", "position_rotation", "position_scale", "dynamic_transform", "mesh", "point_light", "spot_light",
         "rigid_body", "force_field", "cloth", "cloth_render", "physics_reference", "terrain", "heightmap_collider", "grass", "proc_placement", "water"]
_NO_PAYLOAD = {"dynamic_transform", "cloth_render", "proc_placement"}
_UNSUPPORTED = {"mesh", "point_light", "spot_light", "terrain", "grass", "water"}
OBJECT_RIGID_BODY, OBJECT_STATIC, OBJECT_FORCE_FIELD = 0, 1, 2                 # physics_object_type, physics.h:49-57


def _f(*v):
    return struct.pack("<%df" % len(v), *[float(np.float32(x)) for x in v])


def _collider_union(ctype, shape, material, object_type=OBJECT_RIGID_BODY, object_index=0):
    s = [float(x) for x in shape]
    if ctype == scenes.SPHERE:
        body = _f(*s[0:4])
    elif ctype in (scenes.CAPSULE, scenes.CYLINDER, scenes.AABB):
        body = _f(*s[0:(7 if ctype != scenes.AABB else 6)])
    elif ctype == scenes.OBB:
        body = _f(*s[0:10])                                                     # quat rotation @0, center @16, radius @28
    elif ctype == scenes.HULL:
        body = _f(*s[0:7]) + b"\0" * 4 + struct.pack("<I", int(s[7])) + b"\0" * 4  # quat @0, position @16, geometryIndex @32 (8-byte union)
    else:
        raise ValueError("unknown collider type %r" % (ctype,))
    out = body.ljust(48, b"\0")
    out += struct.pack("<i3f", -1, *[float(np.float32(x)) for x in material])   # physics_material_type_none: only used for sound selection
    out += struct.pack("<BBH", ctype, object_type, object_index & 0xFFFF)
    return out.ljust(COLLIDER_UNION_BYTES, b"\0")


def _collider_from_union(raw):
    ctype, = struct.unpack_from("<B", raw, 64)
    _, restitution, friction, density = struct.unpack_from("<i3f", raw, 48)
    if ctype == scenes.SPHERE:
        shape = struct.unpack_from("<4f", raw, 0)
    elif ctype in (scenes.CAPSULE, scenes.CYLINDER):
        shape = struct.unpack_from("<7f", raw, 0)
    elif ctype == scenes.AABB:
        shape = struct.unpack_from("<6f", raw, 0)
    elif ctype == scenes.OBB:
        shape = struct.unpack_from("<10f", raw, 0)
    elif ctype == scenes.HULL:
        shape = struct.unpack_from("<7f", raw, 0) + (float(struct.unpack_from("<I", raw, 32)[0]),)
    else:
        raise ValueError("collider_union with type %d" % ctype)
    return ctype, tuple(float(x) for x in shape), (restitution, friction, density)


def write_entity(components):
    """One entity's stream from {component name: payload bytes (b'' for the payload-free ones)}: serializeEntityToMemory, :452-457."""
    out = bytearray()
    for name in GROUP:
        if name in components:
            if name in _UNSUPPORTED:
                raise ValueError("component %s is outside the physics subset" % name)
            out += b"\x01" + (b"" if name in _NO_PAYLOAD else bytes(components[name]))
        else:
            out += b"\x00"
    return bytes(out)


def read_entity(stream):
    """{component name: payload bytes} of one entity's stream: deserializeEntityFromMemory, :459-465 (the whole stream must be consumed)."""
    fixed = {"tag": 16, "transform": 48, "position": 12, "position_rotation": 32, "position_scale": 24, "rigid_body": 112, "force_field": 12, "cloth": 32,
             "heightmap_collider": 24}
    off, out = 0, {}
    for name in GROUP:
        if off >= len(stream):
            raise ValueError("entity stream ends inside the component group (at %s)" % name)
        present = stream[off]; off += 1
        if not present:
            continue
        if name in _UNSUPPORTED:
            raise ValueError("entity stream holds a %s component: outside the physics subset" % name)
        if name in _NO_PAYLOAD:
            out[name] = b""
            continue
        if name == "physics_reference":
            start = off
            n, = struct.unpack_from("<I", stream, off); off += 4 + n * COLLIDER_UNION_BYTES
            m, = struct.unpack_from("<I", stream, off); off += 4
            for _ in range(m):
                t, = struct.unpack_from("<i", stream, off)
                if not 0 <= t < len(CONSTRAINT_BYTES):
                    raise ValueError("constraint of type %d in an entity stream" % t)
                off += 12 + CONSTRAINT_BYTES[t]
            size = off - start; off = start
        else:
            size = fixed[name]
        if off + size > len(stream):
            raise ValueError("entity stream ends inside its %s component" % name)
        out[name] = bytes(stream[off:off + size]); off += size
    if off != len(stream):
        raise ValueError("%d bytes left over behind the component group" % (len(stream) - off))
    return out


def _physics_reference(colliders, constraints):
    out = struct.pack("<I", len(colliders)) + b"".join(colliders) + struct.pack("<I", len(constraints))
    for t, a, b, pod in constraints:
        if len(pod) != CONSTRAINT_BYTES[t]:
            raise ValueError("constraint of type %s with %d bytes of data" % (CONSTRAINT_NAMES[t], len(pod)))
        out += struct.pack("<iII", t, a, b) + bytes(pod)
    return out


def dump_entities(scene, transforms=None, velocities=None, mass_properties=None, constraint_pods=None, constraints=None):
    """[(entity id, stream), ...] of `scene`: body i is entity i, the static colliders and the force fields follow.
    transforms ([n, 7]) / velocities ([n, 6]) / mass_properties ([n, 13] as world.mass_properties() returns them: localCOG, invMass,
    invInertia in the reference's memory order) fill the raw components.  Constraints: either constraint_pods = {kind: [bytes, ...]}
    in add order next to scene.joints, or constraints = [(type index, body a, body b, pod bytes), ...].
    The stream keeps no global order of the constraints: reading gives them entity by entity, each at its first listing (a list that
    is already in that order reads back unchanged)."""
    n = len(scene.bodies)
    per_body_cols = [[] for _ in range(n)]
    statics = []
    for body, ctype, shape, mat, pos, rot in scene.colliders:
        if body == scenes.STATIC:
            statics.append((ctype, shape, mat, pos, rot))
        else:
            per_body_cols[body].append(_collider_union(ctype, shape, mat, OBJECT_RIGID_BODY, body))
    if constraints is None:
        constraints, counters = [], {}
        for j in scene.joints if constraint_pods else ():
            kind = j[0][:-6] if j[0].endswith("_local") else j[0]
            k = counters.get(kind, 0); counters[kind] = k + 1
            constraints.append((CONSTRAINT_NAMES.index(kind), j[1], j[2], bytes(constraint_pods[kind][k])))
    per_body_cons = [[] for _ in range(n)]
    for t, a, b, pod in constraints:
        rec = (t, a, b, bytes(pod))
        per_body_cons[a].append(rec)
        if b != a:
            per_body_cons[b].append(rec)
    out = []
    for i, (pos, rot, kin, g, ld, ad) in enumerate(scene.bodies):
        if transforms is not None:
            pos, rot = transforms[i][0:3], transforms[i][3:7]
        mp = np.asarray(mass_properties[i], np.float32) if mass_properties is not None else np.zeros(13, np.float32)
        v = np.asarray(velocities[i], np.float32) if velocities is not None else np.zeros(6, np.float32)
        inv_mass = 0.0 if kin else float(mp[3]) if mass_properties is not None else 1.0
        rb = _f(*mp[0:3]) + _f(inv_mass) + _f(*mp[4:13]) + _f(g, ld, ad) + _f(*v[0:3]) + _f(*v[3:6]) + _f(0, 0, 0, 0, 0, 0)
        comps = {"tag": ("body_%d" % i).encode().ljust(16, b"\0")[:15] + b"\0", "transform": (_f(*rot) + _f(*pos) + _f(1, 1, 1)).ljust(48, b"\0"),
                 "dynamic_transform": b"", "rigid_body": rb}
        if per_body_cols[i] or per_body_cons[i]:
            comps["physics_reference"] = _physics_reference(per_body_cols[i], per_body_cons[i])
        out.append((i, write_entity(comps)))
    next_id = n
    for k, (ctype, shape, mat, pos, rot) in enumerate(statics):
        comps = {"tag": ("static_%d" % k).encode().ljust(16, b"\0")[:15] + b"\0", "transform": (_f(*rot) + _f(*pos) + _f(1, 1, 1)).ljust(48, b"\0"),
                 "physics_reference": _physics_reference([_collider_union(ctype, shape, mat, OBJECT_STATIC, 0)], [])}
        out.append((next_id, write_entity(comps))); next_id += 1
    for k, (force, pos, rot, cols) in enumerate(scene.fields):
        comps = {"tag": ("field_%d" % k).encode().ljust(16, b"\0")[:15] + b"\0", "force_field": _f(*force)}
        if pos is not None:
            comps["transform"] = (_f(*(rot or (0, 0, 0, 1))) + _f(*pos) + _f(1, 1, 1)).ljust(48, b"\0")
        if cols:
            comps["physics_reference"] = _physics_reference([_collider_union(ct, sh, (0.0, 0.0, 0.0), OBJECT_FORCE_FIELD, k) for ct, sh in cols], [])
        out.append((next_id, write_entity(comps))); next_id += 1
    if scene.heightmap is not None:
        cpd, size, mat = scene.heightmap[0:3]
        out.append((next_id, write_entity({"tag": b"terrain".ljust(16, b"\0"), "heightmap_collider": struct.pack("<Ifi3f", cpd, size, -1, *[float(np.float32(x)) for x in mat])}))); next_id += 1
    for width, height, gx, gy, mass, stiffness, damping, gravity, pos, rot in scene.cloths:
        comps = {"transform": (_f(*rot) + _f(*pos) + _f(1, 1, 1)).ljust(48, b"\0"),
                 "cloth": struct.pack("<2f2I4f", width, height, gx, gy, mass, stiffness, damping, gravity), "cloth_render": b""}
        out.append((next_id, write_entity(comps))); next_id += 1
    return out


def load_entities(entities, name="scene", dt=1.0 / 120.0):
    """(scene, constraints, velocities) from [(entity id, stream), ...]: constraints = [(type index, body a, body b, pod bytes), ...]
    for world.add_constraint, each once; velocities [n, 6] of the bodies in entity order (what rigid_body_component held)."""
    s = scenes.Scene(name, dt)
    body_of = {}
    parsed = [(eid, read_entity(stream)) for eid, stream in entities]
    for eid, c in parsed:
        if "rigid_body" in c:
            body_of[eid] = len(body_of)
    vel = np.zeros((len(body_of), 6), np.float32)
    seen, constraints = set(), []
    for eid, c in parsed:
        pos, rot = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0)
        if "transform" in c:
            rot = struct.unpack_from("<4f", c["transform"], 0); pos = struct.unpack_from("<3f", c["transform"], 16)
        elif "position_rotation" in c:
            pos = struct.unpack_from("<3f", c["position_rotation"], 0); rot = struct.unpack_from("<4f", c["position_rotation"], 16)
        elif "position" in c or "position_scale" in c:
            pos = struct.unpack_from("<3f", c.get("position", c.get("position_scale")), 0)
        cols, cons = [], []
        if "physics_reference" in c:
            raw = c["physics_reference"]
            n, = struct.unpack_from("<I", raw, 0)
            cols = [_collider_from_union(raw[4 + k * COLLIDER_UNION_BYTES: 4 + (k + 1) * COLLIDER_UNION_BYTES]) for k in range(n)]
            off = 4 + n * COLLIDER_UNION_BYTES
            m, = struct.unpack_from("<I", raw, off); off += 4
            for _ in range(m):
                t, a, b = struct.unpack_from("<iII", raw, off); off += 12
                cons.append((t, a, b, raw[off:off + CONSTRAINT_BYTES[t]])); off += CONSTRAINT_BYTES[t]
        if "rigid_body" in c:
            rb = c["rigid_body"]
            inv_mass, = struct.unpack_from("<f", rb, 12)
            g, ld, ad = struct.unpack_from("<3f", rb, 52)
            b = s.add_body(pos, rot, kinematic=inv_mass == 0.0, gravity_factor=g, linear_damping=ld, angular_damping=ad)
            assert b == body_of[eid]
            vel[b] = struct.unpack_from("<6f", rb, 64)
            for ctype, shape, mat in cols:
                s.add_collider(b, ctype, shape, mat)
        elif "force_field" in c:
            s.add_force_field(struct.unpack_from("<3f", c["force_field"], 0), pos if cols else None, rot if cols else None, [(ct, sh) for ct, sh, _ in cols])
        elif "cloth" in c:
            width, height, gx, gy, mass, stiffness, damping, gravity = struct.unpack_from("<2f2I4f", c["cloth"], 0)
            s.add_cloth(width, height, gx, gy, mass, pos, rot, stiffness, damping, gravity)
        else:
            for ctype, shape, mat in cols:
                s.add_collider(scenes.STATIC, ctype, shape, mat, pos, rot)
        for t, a, b, pod in cons:
            if eid != a and eid != b:
                raise ValueError("entity %d lists a constraint between %d and %d (serialization_binary.cpp:254)" % (eid, a, b))
            key = (t, a, b, pod)
            if key not in seen:
                seen.add(key)
                constraints.append(key)
    constraints = [(t, body_of[a], body_of[b], pod) for t, a, b, pod in constraints]
    return s, constraints, vel


def pack(entities):
    """A container of ours for a list of entity streams (the reference keeps them one by one in memory): u32 count, then per entity
    u32 id, u64 size, the stream."""
    out = struct.pack("<I", len(entities))
    for eid, stream in entities:
        out += struct.pack("<IQ", eid, len(stream)) + stream
    return out


def unpack(blob):
    n, = struct.unpack_from("<I", blob, 0)
    off, out = 4, []
    for _ in range(n):
        eid, size = struct.unpack_from("<IQ", blob, off); off += 12
        if off + size > len(blob):
            raise ValueError("truncated entity container")
        out.append((eid, bytes(blob[off:off + size]))); off += size
    if off != len(blob):
        raise ValueError("bytes left over behind the last entity")
    return out
