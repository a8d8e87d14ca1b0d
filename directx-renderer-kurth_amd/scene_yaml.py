"""Scene exchange with the reference engine's YAML scene files (.sc), physics subset.

The reference writes a scene as `Scene: <name>` + `Entities:` — a sequence of maps, one per entity, with its components under fixed
keys (serialization_yaml.cpp:386-431 writes, :455-520 reads).  The ones on the rigid-body path:

    Tag: <name>                                                        tag_component
    Transform: {Position: [x, y, z], Rotation: [x, y, z, w], Scale: [x, y, z]}   transform_component (components.h:22-33; vectors are flow sequences, core/yaml.h:27-66)
    Dynamic: true                                                      dynamic_transform_component (bodies that move)
    Rigid body: {Local COG, Inv mass, Inv inertia (9 floats, row-major m00 m01 m02 m10 ..., core/yaml.h:94-110), Gravity factor,
                 Linear damping, Angular damping}                      rigid_body_component (serialization_yaml.cpp:72-99)
    Force field: {Force: [x, y, z]}                                    force_field_component (:101-121)
    Colliders: [{Type: Sphere|Capsule|Cylinder|AABB|OBB, <shape keys>, Restitution, Friction, Density}, ...]   collider_component (:124-230)

`dump_scene` writes a Scene description (directx-renderer-kurth_amd/scenes.py) in that layout, `load_scene` reads such a file
back into a Scene that `instantiate()`s into a device world or an oracle world.  Velocities are not part of the format (the
reference does not write them either).  Differences, on purpose:
  * the reference's writer has no case for cylinder colliders (their shape fields are lost, :131-160) and refuses hulls on read
    (:233-236); cylinders are written here with the capsule's keys (`Position A`, `Position B`, `Radius`), hulls are rejected;
  * constraints are a TODO in the reference's YAML (:428-433; only its binary format holds them, serialization_binary.cpp:225-260):
    they are written under a top-level `Constraints` key as {Type, A, B (entity tags), Data (the reference's POD as float/uint
    words)} — an extension a reference reader ignores.
"""
import struct

import numpy as np
import yaml

from . import scenes

TYPE_NAMES = ["Sphere", "Capsule", "Cylinder", "AABB", "OBB", "Hull"]          # colliderTypeNames, physics.h:72-80
CONSTRAINT_NAMES = ["distance", "ball", "fixed", "hinge", "cone_twist", "slider"]
CONSTRAINT_BYTES = [28, 24, 40, 104, 120, 72]


class _Flow(list):
    """A sequence emitted in flow style ([a, b, c]) like the reference's vectors."""


def _flow_representer(dumper, data):
    return dumper.represent_sequence("tag:yaml.org,2002:seq", data, flow_style=True)


class _Dumper(yaml.SafeDumper):
    pass


_Dumper.add_representer(_Flow, _flow_representer)


def _vec(v):
    return _Flow(float(np.float32(x)) for x in v)


def _collider_node(ctype, shape, material):
    n = {"Type": TYPE_NAMES[ctype]}
    s = [float(np.float32(x)) for x in shape]
    if ctype == scenes.SPHERE:
        n["Center"] = _vec(s[0:3]); n["Radius"] = s[3]
    elif ctype in (scenes.CAPSULE, scenes.CYLINDER):
        n["Position A"] = _vec(s[0:3]); n["Position B"] = _vec(s[3:6]); n["Radius"] = s[6]
    elif ctype == scenes.AABB:
        n["Min corner"] = _vec(s[0:3]); n["Max corner"] = _vec(s[3:6])
    elif ctype == scenes.OBB:
        n["Center"] = _vec(s[4:7]); n["Radius"] = _vec(s[7:10]); n["Rotation"] = _vec(s[0:4])
    else:
        raise ValueError("hull colliders have no representation in the reference's scene files (serialization_yaml.cpp:233-236)")
    n["Restitution"], n["Friction"], n["Density"] = (float(np.float32(x)) for x in material)
    return n


def _collider_from_node(n):
    ctype = TYPE_NAMES.index(n["Type"])
    mat = (float(n["Restitution"]), float(n["Friction"]), float(n["Density"]))
    if ctype == scenes.SPHERE:
        shape = list(n["Center"]) + [n["Radius"]]
    elif ctype in (scenes.CAPSULE, scenes.CYLINDER):
        shape = list(n["Position A"]) + list(n["Position B"]) + [n["Radius"]]
    elif ctype == scenes.AABB:
        shape = list(n["Min corner"]) + list(n["Max corner"])
    elif ctype == scenes.OBB:
        shape = list(n["Rotation"]) + list(n["Center"]) + list(n["Radius"])
    else:
        raise ValueError("hull colliders cannot be read from a scene file (the reference refuses them too)")
    return ctype, tuple(float(x) for x in shape), mat


def dump_scene(scene, name=None, transforms=None, mass_properties=None, constraint_pods=None):
    """YAML text of `scene`.  transforms ([n, 7], e.g. world.transforms(1)) replace the description's start poses;
    mass_properties ([n, 13]: localCOG 3, invMass, invInertia 9 as world.mass_properties() returns them) fill the `Rigid body`
    node (zeros are written otherwise: the reference recomputes them from the colliders on load, scene.h:60-63);
    constraint_pods: {kind: [bytes, ...]} in add order (world.constraint_get) for the `Constraints` extension."""
    by_body = {}
    statics = []
    for body, ctype, shape, mat, pos, rot in scene.colliders:
        if body == scenes.STATIC:
            statics.append((ctype, shape, mat, pos, rot))
        else:
            by_body.setdefault(body, []).append(_collider_node(ctype, shape, mat))
    entities = []
    for i, (pos, rot, kin, g, ld, ad) in enumerate(scene.bodies):
        if transforms is not None:
            pos, rot = transforms[i][0:3], transforms[i][3:7]
        e = {"Tag": "body_%d" % i, "Transform": {"Position": _vec(pos), "Rotation": _vec(rot), "Scale": _vec((1, 1, 1))}, "Dynamic": True}
        mp = mass_properties[i] if mass_properties is not None else np.zeros(13, np.float32)
        inv_mass = 0.0 if kin else float(mp[3]) if mass_properties is not None else 1.0
        e["Rigid body"] = {"Local COG": _vec(mp[0:3]), "Inv mass": float(np.float32(inv_mass)), "Inv inertia": _vec(np.asarray(mp[4:13]).reshape(3, 3).T.reshape(-1)),
                           "Gravity factor": float(g), "Linear damping": float(ld), "Angular damping": float(ad)}
        if i in by_body:
            e["Colliders"] = by_body[i]
        entities.append(e)
    for k, (ctype, shape, mat, pos, rot) in enumerate(statics):
        entities.append({"Tag": "static_%d" % k, "Transform": {"Position": _vec(pos), "Rotation": _vec(rot), "Scale": _vec((1, 1, 1))},
                         "Colliders": [_collider_node(ctype, shape, mat)]})
    doc = {"Scene": name or scene.name, "Entities": entities}
    if constraint_pods:
        cons = []
        counters = {}
        for j in scene.joints:
            kind = j[0][:-6] if j[0].endswith("_local") else j[0]
            k = counters.get(kind, 0); counters[kind] = k + 1
            pod = bytes(constraint_pods[kind][k])
            words = [int(w) for w in struct.unpack("<%dI" % (len(pod) // 4), pod)]
            cons.append({"Type": kind, "A": "body_%d" % j[1], "B": "body_%d" % j[2], "Data": _Flow(words)})
        doc["Constraints"] = cons
    return yaml.dump(doc, Dumper=_Dumper, sort_keys=False, default_flow_style=False)


def load_scene(text):
    """A Scene from YAML text in the reference's layout (physics keys only; everything else in the file is ignored).
    Returns (scene, constraints) where constraints = [(type index, body a, body b, pod bytes), ...] for world.add_constraint."""
    doc = yaml.load(text, Loader=yaml.SafeLoader)
    if not isinstance(doc, dict) or "Scene" not in doc:
        raise ValueError("not a scene file (no 'Scene' key: serialization_yaml.cpp:464-468)")
    s = scenes.Scene(str(doc["Scene"]))
    tags = {}
    for e in doc.get("Entities") or []:
        tr = e.get("Transform") or {}
        pos = tuple(float(x) for x in tr.get("Position", e.get("Position", {}).get("Position", (0, 0, 0)) if isinstance(e.get("Position"), dict) else (0, 0, 0)))
        rot = tuple(float(x) for x in tr.get("Rotation", (0, 0, 0, 1)))
        cols = [_collider_from_node(c) for c in (e.get("Colliders") or [])]
        rb = e.get("Rigid body")
        if rb is not None:
            b = s.add_body(pos, rot, kinematic=float(rb.get("Inv mass", 1.0)) == 0.0, gravity_factor=float(rb.get("Gravity factor", 1.0)),
                           linear_damping=float(rb.get("Linear damping", 0.4)), angular_damping=float(rb.get("Angular damping", 0.4)))
            tags[e.get("Tag", "body_%d" % b)] = b
            for ctype, shape, mat in cols:
                s.add_collider(b, ctype, shape, mat)
        else:
            for ctype, shape, mat in cols:
                s.add_collider(scenes.STATIC, ctype, shape, mat, pos, rot)
        if "Force field" in e:
            s.add_force_field(tuple(float(x) for x in e["Force field"]["Force"]), pos if cols else None, rot if cols else None, [(c[0], c[1]) for c in cols] if rb is None else ())
    constraints = []
    for c in doc.get("Constraints") or []:
        t = CONSTRAINT_NAMES.index(c["Type"])
        pod = struct.pack("<%dI" % len(c["Data"]), *[int(w) for w in c["Data"]])
        if len(pod) != CONSTRAINT_BYTES[t]:
            raise ValueError("constraint of type %s with %d bytes of data" % (c["Type"], len(pod)))
        constraints.append((t, tags[c["A"]], tags[c["B"]], pod))
    return s, constraints
