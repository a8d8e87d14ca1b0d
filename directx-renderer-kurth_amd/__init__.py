"""MI355X-native rigid-body physics stepper — Python host mirror of the C-ABI in include/mi_physics.h.

`World` exposes the reference engine's physics call shapes (study-game-engines/directx-renderer-kurth, src/physics/physics.h:
collider_component::as*, rigid_body_component(kinematic, gravityFactor, linearDamping, angularDamping),
add*ConstraintFromGlobalPoints, getConstraint, physicsStep(scene, arena, timer, settings, dt)) over libmi_physics.so,
the same way the reference's own Python side binds its Physics-Lib DLL with ctypes (learning/loco_env.py:7-45).
There is no CPU fallback: without the compiled HIP library or without a GPU, constructing a World raises.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmi_physics.so")

STATIC = 0xFFFFFFFF
SPHERE, CAPSULE, CYLINDER, AABB, OBB, HULL = range(6)
DISTANCE, BALL, FIXED, HINGE, CONE_TWIST, SLIDER = range(6)
CONSTRAINT_POD_BYTES = (28, 24, 40, 104, 120, 72)

COLLIDER_DTYPE = np.dtype([("shape", "<f4", 10), ("restitution", "<f4"), ("friction", "<f4"), ("density", "<f4"),
                           ("type", "<u4"), ("objectType", "<u4"), ("objectIndex", "<u4")])
CONTACT_DTYPE = np.dtype([("point", "<f4", 3), ("depth", "<f4"), ("normal", "<f4", 3), ("friction_restitution", "<u4")])


class Material(C.Structure):
    _fields_ = [("restitution", C.c_float), ("friction", C.c_float), ("density", C.c_float)]


class Settings(C.Structure):
    """physics_settings (reference physics.h:382-397) minus the std::function callbacks."""
    _fields_ = [("fixedFrameRate", C.c_uint32), ("frameRate", C.c_uint32), ("maxPhysicsIterationsPerFrame", C.c_uint32),
                ("numRigidSolverIterations", C.c_uint32), ("numClothVelocityIterations", C.c_uint32),
                ("numClothPositionIterations", C.c_uint32), ("numClothDriftIterations", C.c_uint32),
                ("simdBroadPhase", C.c_uint32), ("simdNarrowPhase", C.c_uint32), ("simdConstraintSolver", C.c_uint32)]

    def __init__(self, **kw):
        super().__init__(1, 120, 4, 30, 0, 1, 0, 1, 1, 1)
        for k, v in kw.items():
            setattr(self, k, v)


class WorldDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("reserveBodies", C.c_uint32), ("reserveColliders", C.c_uint32), ("reservePairs", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("numRigidBodies", C.c_uint32), ("numColliders", C.c_uint32), ("numBroadphaseOverlaps", C.c_uint32), ("numCollisions", C.c_uint32),
                ("numContacts", C.c_uint32), ("numColors", C.c_uint32), ("numJoints", C.c_uint32), ("numInternalSteps", C.c_uint32),
                ("numGraphBuilds", C.c_uint32), ("coloringRounds", C.c_uint32), ("flowProbes", C.c_uint32), ("numFlowRecoveries", C.c_uint32),
                ("msCollidersBroad", C.c_float), ("msNarrow", C.c_float), ("msSolverSetup", C.c_float), ("msSolve", C.c_float),
                ("msIntegrate", C.c_float), ("msTotal", C.c_float),
                ("avgContacts", C.c_float), ("avgCollisions", C.c_float), ("avgColors", C.c_float), ("avgBroadphaseOverlaps", C.c_float), ("avgFlowProbes", C.c_float),
                ("avgSteps", C.c_uint32),
                ("clusterTasks", C.c_uint32 * 5), ("clusterManifolds", C.c_uint32 * 5), ("clusterSharedBodies", C.c_uint32), ("clusterParts", C.c_uint32), ("numNarrowphaseRedone", C.c_uint32)]

    def asdict(self):
        return {n: (list(getattr(self, n)) if n in ("clusterTasks", "clusterManifolds") else getattr(self, n)) for n, _ in self._fields_}


# mi_event (include/mi_physics.h): trigger_event / collision_begin_event / collision_end_event as one record
EVENT_DTYPE = np.dtype([("kind", "<u4"), ("step", "<u4"), ("a", "<u4"), ("b", "<u4"), ("bodyA", "<u4"), ("bodyB", "<u4"),
                        ("position", "<f4", 3), ("normal", "<f4", 3), ("relativeVelocity", "<f4", 3)])
TRIGGER_ENTER, TRIGGER_LEAVE, COLLISION_BEGIN, COLLISION_END = range(4)

EXPORTED_SYMBOLS = [
    "mi_world_create", "mi_world_destroy", "mi_last_error", "mi_snapshot_size", "mi_snapshot_save", "mi_world_restore", "mi_add_body", "mi_add_hull_geometry", "mi_add_collider", "mi_add_static_collider",
    "mi_add_distance_constraint_local", "mi_add_distance_constraint_global", "mi_add_ball_constraint_local", "mi_add_ball_constraint_global",
    "mi_add_fixed_constraint_global", "mi_add_hinge_constraint_global", "mi_add_cone_twist_constraint_global", "mi_add_slider_constraint_global", "mi_add_constraint",
    "mi_constraint_get", "mi_constraint_set", "mi_delete_constraint", "mi_delete_all_constraints", "mi_delete_all_constraints_from_body", "mi_delete_body",
    "mi_add_force_field", "mi_set_force_field", "mi_add_trigger", "mi_add_force_field_collider", "mi_add_trigger_collider", "mi_set_force_field_transform", "mi_set_trigger_transform", "mi_enable_collision_events", "mi_drain_events",
    "mi_set_heightmap", "mi_heightmap_set_chunk", "mi_heightmap_update", "mi_heightmap_height_at",
    "mi_add_cloth", "mi_cloth_set_fixed_vertices", "mi_cloth_set_properties", "mi_set_cloth_iterations", "mi_num_cloths", "mi_cloth_num_particles", "mi_cloth_read",
    "mi_test_physics_interaction", "mi_apply_force_torque", "mi_set_velocity",
    "mi_set_transform", "mi_write_transforms", "mi_write_velocities", "mi_step", "mi_step_internal", "mi_synchronize", "mi_read_transforms", "mi_read_velocities", "mi_read_mass_properties",
    "mi_get_stats", "mi_enable_validation", "mi_enable_stage_timing", "mi_num_bodies", "mi_num_colliders", "mi_device_pointers", "mi_state_to_device_buffers", "mi_state_from_device_buffers", "mi_slab_configure", "mi_slab_message_bytes", "mi_slab_pack", "mi_slab_unpack", "mi_slab_read_codes", "mi_debug_num_pairs", "mi_debug_read_pairs", "mi_debug_sorting_axis",
    "mi_debug_read_world_colliders", "mi_debug_num_manifold_slots", "mi_debug_read_manifolds", "mi_debug_num_colors", "mi_debug_read_schedule",
    "mi_debug_read_joint_order", "mi_debug_read_body_state", "mi_debug_flow_trace",
    "mi_debug_set_replay", "mi_debug_num_replay_batches", "mi_debug_read_replay_batches",
]


def build(force=False):
    """Compile libmi_physics.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j6"], stdout=subprocess.DEVNULL)
    build_locomotion()
    return _LIB_PATH


LOCOMOTION_LIB_PATH = os.path.join(_HERE, "libmi_locomotion.so")
LOCOMOTION_SYMBOLS = ["getPhysicsStateSize", "getPhysicsActionSize", "getPhysicsRanges", "resetPhysics", "updatePhysics", "setPhysicsSeed"]


def build_locomotion():
    """Host-side C++ over the C-ABI: the reference's ragdoll RL environment (learned_locomotion.cpp:395-489) as libmi_locomotion.so."""
    src = os.path.join(_HERE, "host", "locomotion_env.cpp")
    if os.path.exists(LOCOMOTION_LIB_PATH) and os.path.getmtime(LOCOMOTION_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(_LIB_PATH)):
        return LOCOMOTION_LIB_PATH
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(os.path.dirname(_HERE), "include"), src,
                           "-L" + _HERE, "-lmi_physics", "-Wl,-rpath,$ORIGIN", "-o", LOCOMOTION_LIB_PATH])
    return LOCOMOTION_LIB_PATH


_lib = None


def load_library():
    """dlopen the C-ABI library.  Raises if it has not been built — there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("libmi_physics.so is missing: run __graft_entry__.build() (hipcc) first; there is no CPU fallback")
        lib = C.CDLL(_LIB_PATH)
        lib.mi_world_create.restype = C.c_void_p
        lib.mi_world_restore.restype = C.c_void_p
        lib.mi_snapshot_size.restype = C.c_uint64
        lib.mi_slab_message_bytes.restype = C.c_uint64
        lib.mi_last_error.restype = C.c_char_p
        lib.mi_heightmap_height_at.restype = C.c_float
        for name in EXPORTED_SYMBOLS:
            fn = getattr(lib, name)
            if name.startswith("mi_add_") or name in ("mi_num_bodies", "mi_num_colliders", "mi_debug_num_pairs", "mi_debug_num_manifold_slots", "mi_debug_num_colors", "mi_drain_events", "mi_num_cloths", "mi_cloth_num_particles"):
                fn.restype = C.c_uint32
        _lib = lib
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class PhysicsError(RuntimeError):
    pass


class World:
    """One physics world on one MI355X.  Method names follow the reference's free functions / component factories."""

    def __init__(self, device=-1, reserve_bodies=0, reserve_colliders=0, reserve_pairs=0):
        self.lib = load_library()
        desc = WorldDesc(device, reserve_bodies, reserve_colliders, reserve_pairs)
        h = self.lib.mi_world_create(C.byref(desc))
        if not h:
            raise PhysicsError("mi_world_create failed: %s" % (self.lib.mi_last_error(None) or b"").decode())
        self.w = C.c_void_p(h)
        self.timer = C.c_float(0.0)

    def snapshot(self):
        """Self-contained binary image of the world (checkpoint); World.restore(blob) continues it bit-identically."""
        n = int(self.lib.mi_snapshot_size(self.w))
        buf = np.zeros(n, np.uint8)
        self._check(self.lib.mi_snapshot_save(self.w, _p(buf), C.c_uint64(n)))
        return buf.tobytes()

    @classmethod
    def restore(cls, blob, device=-1):
        self = cls.__new__(cls)
        self.lib = load_library()
        desc = WorldDesc(device, 0, 0, 0)
        buf = np.frombuffer(blob, np.uint8)
        h = self.lib.mi_world_restore(C.byref(desc), _p(buf), C.c_uint64(len(buf)))
        if not h:
            raise PhysicsError("mi_world_restore failed: %s" % (self.lib.mi_last_error(None) or b"").decode())
        self.w = C.c_void_p(h)
        self.timer = C.c_float(0.0)
        return self

    def close(self):
        if getattr(self, "w", None):
            self.lib.mi_world_destroy(self.w)
            self.w = None

    def __del__(self):
        self.close()

    def _check(self, code):
        if code:
            raise PhysicsError("mi_physics error %d: %s" % (code, (self.lib.mi_last_error(self.w) or b"").decode()))

    def _id(self, v):
        if v == 0xFFFFFFFF:
            raise PhysicsError((self.lib.mi_last_error(self.w) or b"invalid id").decode())
        return v

    # ---- add API ------------------------------------------------------------------------------------------
    def add_body(self, pos, rot=(0, 0, 0, 1), kinematic=False, gravity_factor=1.0, linear_damping=0.4, angular_damping=0.4):
        return self._id(self.lib.mi_add_body(self.w, int(kinematic), C.c_float(gravity_factor), C.c_float(linear_damping), C.c_float(angular_damping), _f(pos), _f(rot)))

    def add_hull_geometry(self, vertices, triangles):
        """bounding_hull_geometry::fromMesh (reference bounding_volumes.cpp:1394-1452): convex vertex set + outward-facing triangles."""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3); t = np.ascontiguousarray(triangles, np.uint32).reshape(-1, 3)
        return self._id(self.lib.mi_add_hull_geometry(self.w, _f(v), C.c_uint32(len(v)), t.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(len(t))))

    def add_collider(self, body, ctype, shape, material):
        s = np.zeros(10, np.float32); s[:len(shape)] = shape
        m = Material(*material)
        return self._id(self.lib.mi_add_collider(self.w, C.c_uint32(body), C.c_uint32(ctype), _f(s), C.byref(m)))

    def add_static_collider(self, ctype, shape, material, pos=(0, 0, 0), rot=(0, 0, 0, 1)):
        s = np.zeros(10, np.float32); s[:len(shape)] = shape
        m = Material(*material)
        return self._id(self.lib.mi_add_static_collider(self.w, C.c_uint32(ctype), _f(s), C.byref(m), _f(pos), _f(rot)))

    # ---- force fields, triggers, events (physics.h:182-203, 356-380; physics.cpp:759-787, 952-1178) ----
    def add_force_field(self, force, pos=None, rot=None):
        """force_field_component: without colliders it acts on every body, with colliders on the bodies overlapping them."""
        return self._id(self.lib.mi_add_force_field(self.w, _f(force), _f(pos) if pos is not None else None, _f(rot) if rot is not None else None))

    def set_force_field(self, field, force):
        self._check(self.lib.mi_set_force_field(self.w, C.c_uint32(field), _f(force)))

    def add_trigger(self, pos=None, rot=None):
        """trigger_component: enter / leave events for rigid bodies overlapping its colliders come out of drain_events()."""
        return self._id(self.lib.mi_add_trigger(self.w, _f(pos) if pos is not None else None, _f(rot) if rot is not None else None))

    def add_force_field_collider(self, field, ctype, shape):
        s = np.zeros(10, np.float32); s[:len(shape)] = shape
        return self._id(self.lib.mi_add_force_field_collider(self.w, C.c_uint32(field), C.c_uint32(ctype), _f(s)))

    def add_trigger_collider(self, trigger, ctype, shape):
        s = np.zeros(10, np.float32); s[:len(shape)] = shape
        return self._id(self.lib.mi_add_trigger_collider(self.w, C.c_uint32(trigger), C.c_uint32(ctype), _f(s)))

    def set_force_field_transform(self, field, pos, rot=(0, 0, 0, 1)):
        self._check(self.lib.mi_set_force_field_transform(self.w, C.c_uint32(field), _f(pos), _f(rot)))

    def set_trigger_transform(self, trigger, pos, rot=(0, 0, 0, 1)):
        self._check(self.lib.mi_set_trigger_transform(self.w, C.c_uint32(trigger), _f(pos), _f(rot)))

    def enable_collision_events(self, begin=True, end=True):
        """physics_settings::collisionBeginCallback / collisionEndCallback set; enable before the first step."""
        self._check(self.lib.mi_enable_collision_events(self.w, int(begin), int(end)))

    def drain_events(self, capacity=1 << 16):
        """Pending trigger / collision events in the reference's callback order, as a structured array (EVENT_DTYPE)."""
        chunks = []
        while True:
            out = np.zeros(capacity, EVENT_DTYPE)
            n = self.lib.mi_drain_events(self.w, _p(out), C.c_uint32(capacity))
            chunks.append(out[:n])
            if n < capacity:
                break
        return np.concatenate(chunks) if len(chunks) > 1 else chunks[0]

    # ---- heightmap terrain (heightmap_collider.h:127-152) ----
    def set_heightmap(self, chunks_per_dim, chunk_size, material, min_corner, amplitude_scale):
        m = Material(*material)
        self._check(self.lib.mi_set_heightmap(self.w, C.c_uint32(chunks_per_dim), C.c_float(chunk_size), C.byref(m), _f(min_corner), C.c_float(amplitude_scale)))

    def heightmap_set_chunk(self, x, z, heights):
        h = np.ascontiguousarray(heights, np.uint16).reshape(129, 129)
        self._check(self.lib.mi_heightmap_set_chunk(self.w, C.c_uint32(x), C.c_uint32(z), _p(h)))

    def heightmap_update(self, min_corner, amplitude_scale):
        self._check(self.lib.mi_heightmap_update(self.w, _f(min_corner), C.c_float(amplitude_scale)))

    def heightmap_height_at(self, x, z):
        return float(self.lib.mi_heightmap_height_at(self.w, C.c_float(x), C.c_float(z)))

    # ---- cloth (cloth.h:5-60; stepped after the rigid bodies, physics.cpp:1354-1358) ----
    def add_cloth(self, width, height, grid_x, grid_y, total_mass, stiffness=0.5, damping=0.3, gravity_factor=1.0):
        return self._id(self.lib.mi_add_cloth(self.w, C.c_float(width), C.c_float(height), C.c_uint32(grid_x), C.c_uint32(grid_y), C.c_float(total_mass),
                                              C.c_float(stiffness), C.c_float(damping), C.c_float(gravity_factor)))

    def cloth_set_fixed_vertices(self, cloth, pos, rot=(0, 0, 0, 1), move_rigid=False):
        self._check(self.lib.mi_cloth_set_fixed_vertices(self.w, C.c_uint32(cloth), _f(pos), _f(rot), int(move_rigid)))

    def cloth_set_properties(self, cloth, total_mass, stiffness, damping, gravity_factor):
        self._check(self.lib.mi_cloth_set_properties(self.w, C.c_uint32(cloth), C.c_float(total_mass), C.c_float(stiffness), C.c_float(damping), C.c_float(gravity_factor)))

    def set_cloth_iterations(self, velocity=0, position=1, drift=0):
        self._check(self.lib.mi_set_cloth_iterations(self.w, C.c_uint32(velocity), C.c_uint32(position), C.c_uint32(drift)))

    def set_cloth_colour_order(self, on=True):
        """The device always solves cloth constraints in colour order; present so that a scene instantiates into either world."""

    def cloth_state(self, cloth):
        n = self.lib.mi_cloth_num_particles(self.w, C.c_uint32(cloth))
        p = np.zeros((n, 3), np.float32); v = np.zeros((n, 3), np.float32)
        self._check(self.lib.mi_cloth_read(self.w, C.c_uint32(cloth), _p(p), _p(v)))
        return p, v

    def add_constraint(self, ctype, a, b, pod):
        """addConstraint(a, b, const T&) (reference physics.h:239-244): the constraint as its POD bytes (layout of constraint_get)."""
        buf = np.ascontiguousarray(np.frombuffer(bytes(pod), np.uint8))
        assert len(buf) == (28, 24, 40, 104, 120, 72)[ctype]
        cid = self.lib.mi_add_constraint(self.w, C.c_uint32(ctype), C.c_uint32(a), C.c_uint32(b), _p(buf))
        if cid == 0xFFFFFFFF:
            self._check(1)
        return cid

    def add_distance_constraint_local(self, a, b, la, lb, distance):
        return self._id(self.lib.mi_add_distance_constraint_local(self.w, a, b, _f(la), _f(lb), C.c_float(distance)))

    def add_distance_constraint_global(self, a, b, ga, gb):
        return self._id(self.lib.mi_add_distance_constraint_global(self.w, a, b, _f(ga), _f(gb)))

    def add_ball_constraint_local(self, a, b, la, lb):
        return self._id(self.lib.mi_add_ball_constraint_local(self.w, a, b, _f(la), _f(lb)))

    def add_ball_constraint_global(self, a, b, g):
        return self._id(self.lib.mi_add_ball_constraint_global(self.w, a, b, _f(g)))

    def add_fixed_constraint_global(self, a, b, g):
        return self._id(self.lib.mi_add_fixed_constraint_global(self.w, a, b, _f(g)))

    def add_hinge_constraint_global(self, a, b, anchor, axis, min_limit=1.0, max_limit=-1.0):
        return self._id(self.lib.mi_add_hinge_constraint_global(self.w, a, b, _f(anchor), _f(axis), C.c_float(min_limit), C.c_float(max_limit)))

    def add_cone_twist_constraint_global(self, a, b, anchor, axis, swing_limit, twist_limit):
        return self._id(self.lib.mi_add_cone_twist_constraint_global(self.w, a, b, _f(anchor), _f(axis), C.c_float(swing_limit), C.c_float(twist_limit)))

    def add_slider_constraint_global(self, a, b, anchor, axis, min_limit=1.0, max_limit=-1.0):
        return self._id(self.lib.mi_add_slider_constraint_global(self.w, a, b, _f(anchor), _f(axis), C.c_float(min_limit), C.c_float(max_limit)))

    def constraint_get(self, ctype, cid, nbytes=None):
        buf = np.zeros(nbytes or CONSTRAINT_POD_BYTES[ctype], np.uint8)
        self._check(self.lib.mi_constraint_get(self.w, ctype, cid, _p(buf)))
        return buf

    def constraint_set(self, ctype, cid, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        self._check(self.lib.mi_constraint_set(self.w, ctype, cid, _p(buf)))

    def delete_constraint(self, ctype, cid):
        self._check(self.lib.mi_delete_constraint(self.w, ctype, cid))

    def delete_all_constraints_from_body(self, body):
        self._check(self.lib.mi_delete_all_constraints_from_body(self.w, C.c_uint32(body)))

    def delete_body(self, body):
        self._check(self.lib.mi_delete_body(self.w, C.c_uint32(body)))

    def test_physics_interaction(self, origin, direction, strength=1000.0):
        """testPhysicsInteraction(scene, ray, strength) — reference physics.h:404.  Returns the pushed body's index or None."""
        r = self.lib.mi_test_physics_interaction(self.w, _f(origin), _f(direction), C.c_float(strength))
        return r - 1 if r > 0 else None

    def apply_force_torque(self, body, force, torque=(0, 0, 0)):
        self._check(self.lib.mi_apply_force_torque(self.w, body, _f(force), _f(torque)))

    def set_velocity(self, body, lin, ang=(0, 0, 0)):
        self._check(self.lib.mi_set_velocity(self.w, body, _f(lin), _f(ang)))

    def write_state(self, transforms, velocities):
        """Bulk overwrite of physics_transform1 ([n,7]) and velocities ([n,6])."""
        t = np.ascontiguousarray(transforms, np.float32); v = np.ascontiguousarray(velocities, np.float32)
        self._check(self.lib.mi_write_transforms(self.w, _p(t), C.c_uint32(len(t))))
        self._check(self.lib.mi_write_velocities(self.w, _p(v), C.c_uint32(len(v))))

    # ---- stepping -------------------------------------------------------------------------------------------
    def step(self, dt, settings=None):
        """physicsStep(scene, arena, timer, settings, dt) — reference physics.h:405."""
        settings = settings or Settings()
        self._check(self.lib.mi_step(self.w, C.byref(self.timer), C.byref(settings), C.c_float(dt)))

    def step_internal(self, dt, iterations=30):
        """One physicsStepInternal (reference physics.cpp:1180-1362) at exactly dt."""
        self._check(self.lib.mi_step_internal(self.w, C.c_float(dt), C.c_uint32(iterations)))

    def synchronize(self):
        self._check(self.lib.mi_synchronize(self.w))

    def enable_stage_timing(self, on=True):
        self._check(self.lib.mi_enable_stage_timing(self.w, int(on)))

    # ---- results --------------------------------------------------------------------------------------------
    @property
    def num_bodies(self):
        return self.lib.mi_num_bodies(self.w)

    @property
    def num_colliders(self):
        return self.lib.mi_num_colliders(self.w)

    def transforms(self, which=1):
        out = np.zeros((self.num_bodies, 7), np.float32)
        self._check(self.lib.mi_read_transforms(self.w, C.c_uint32(which), _p(out), C.c_uint32(len(out))))
        return out

    def velocities(self):
        out = np.zeros((self.num_bodies, 6), np.float32)
        self._check(self.lib.mi_read_velocities(self.w, _p(out), C.c_uint32(len(out))))
        return out

    def mass_properties(self):
        out = np.zeros((self.num_bodies, 13), np.float32)
        self._check(self.lib.mi_read_mass_properties(self.w, _p(out), C.c_uint32(len(out))))
        return out

    def enable_validation(self, on=True):
        """Debug guard: NaN / Inf scan after every stage (the reference's VALIDATE macros, physics.cpp:807-926); a step after one that
        produced a non-finite value fails."""
        self._check(self.lib.mi_enable_validation(self.w, int(on)))

    # ---- spatial slab halo (device side) -------------------------------------------------------------------
    def slab_configure(self, rank, size, axis, lo, hi, margin):
        self._check(self.lib.mi_slab_configure(self.w, C.c_uint32(rank), C.c_uint32(size), C.c_uint32(axis), C.c_float(lo), C.c_float(hi), C.c_float(margin)))

    def slab_message_bytes(self, capacity):
        return int(self.lib.mi_slab_message_bytes(C.c_uint32(capacity)))

    def slab_pack(self, left_ptr, right_ptr, capacity):
        """left_ptr / right_ptr: device addresses (int) of message buffers, or 0 for a missing neighbour."""
        self._check(self.lib.mi_slab_pack(self.w, C.c_void_p(left_ptr or None), C.c_void_p(right_ptr or None), C.c_uint32(capacity)))

    def slab_unpack(self, left_ptr, right_ptr, capacity):
        self._check(self.lib.mi_slab_unpack(self.w, C.c_void_p(left_ptr or None), C.c_void_p(right_ptr or None), C.c_uint32(capacity)))

    def slab_codes(self):
        out = np.zeros(self.num_bodies, np.uint8)
        self._check(self.lib.mi_slab_read_codes(self.w, _p(out), C.c_uint32(len(out))))
        return out

    def stats(self):
        s = Stats()
        self._check(self.lib.mi_get_stats(self.w, C.byref(s)))
        return s.asdict()

    def device_pointers(self):
        pose, vel, stream = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.mi_device_pointers(self.w, C.byref(pose), C.byref(vel), C.byref(stream)))
        return pose.value, vel.value, stream.value

    def state_to_device_buffers(self, pose_ptr, vel_ptr):
        self._check(self.lib.mi_state_to_device_buffers(self.w, C.c_void_p(pose_ptr), C.c_void_p(vel_ptr)))

    def state_from_device_buffers(self, pose_ptr, vel_ptr, mask_ptr):
        self._check(self.lib.mi_state_from_device_buffers(self.w, C.c_void_p(pose_ptr), C.c_void_p(vel_ptr), C.c_void_p(mask_ptr)))

    # ---- inspection of the last internal step (parity tests) -------------------------------------------------------
    def pairs(self):
        out = np.zeros((self.lib.mi_debug_num_pairs(self.w), 2), np.uint32)
        if len(out):
            self._check(self.lib.mi_debug_read_pairs(self.w, _p(out)))
        return out

    def sorting_axis(self):
        """(axis the last step oriented its equal-type pairs by, axis of the next step): the reference's sap_context::sortingAxis."""
        out = np.zeros(2, np.uint32)
        self._check(self.lib.mi_debug_sorting_axis(self.w, _p(out)))
        return int(out[0]), int(out[1])

    def world_colliders(self):
        n = self.num_colliders
        cols = np.zeros(n, COLLIDER_DTYPE); aabbs = np.zeros((n, 6), np.float32)
        self._check(self.lib.mi_debug_read_world_colliders(self.w, _p(cols), _p(aabbs)))
        return cols, aabbs

    def manifolds(self):
        """(ordered collider pairs [n,2], counts [n], contacts [n,4] CONTACT_DTYPE, body pairs [n,2]) per candidate pair slot."""
        n = self.lib.mi_debug_num_manifold_slots(self.w)
        pairs = np.zeros((n, 2), np.uint32); counts = np.zeros(n, np.uint32); contacts = np.zeros((n, 4), CONTACT_DTYPE); bp = np.zeros((n, 2), np.uint32)
        if n:
            self._check(self.lib.mi_debug_read_manifolds(self.w, _p(pairs), _p(counts), _p(contacts), _p(bp)))
        return pairs, counts, contacts, bp

    def schedule(self):
        """(manifold slots in Gauss-Seidel execution order, colour start offsets [66])."""
        cs = np.zeros(66, np.uint32)
        n = self.lib.mi_debug_num_manifold_slots(self.w)
        slots = np.zeros(max(n, 1), np.uint32)
        self._check(self.lib.mi_debug_read_schedule(self.w, _p(slots), _p(cs)))
        return slots[:int(cs[65])], cs

    def joint_order(self, ctype, n):
        out = np.zeros(max(n, 1), np.uint32)
        self._check(self.lib.mi_debug_read_joint_order(self.w, ctype, _p(out)))
        return out[:n]

    def flow_trace(self, enable=True, num_slots=0):
        out = np.zeros((max(1, num_slots), 32), np.uint64)
        self._check(self.lib.mi_debug_flow_trace(self.w, int(enable), _p(out) if num_slots else None, C.c_uint32(num_slots)))
        return out

    def set_replay(self, on=True):
        """Solve contacts in the REFERENCE's own order from now on (its greedy 8-wide batch schedule, batch after batch): parity facility."""
        self._check(self.lib.mi_debug_set_replay(self.w, int(on)))

    def replay_batches(self):
        """The last step's batches [numBatches, 8]: schedule position | contact << 28, 0xFFFFFFFF = empty lane."""
        self.lib.mi_debug_num_replay_batches.restype = C.c_uint32
        n = int(self.lib.mi_debug_num_replay_batches(self.w))
        out = np.zeros((max(n, 1), 8), np.uint32)
        self._check(self.lib.mi_debug_read_replay_batches(self.w, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out[:n]

    def body_state(self):
        n = self.num_bodies + 1
        cog = np.zeros((n, 4), np.float32); inv = np.zeros((n, 12), np.float32)
        self._check(self.lib.mi_debug_read_body_state(self.w, _p(cog), _p(inv), C.c_uint32(n)))
        return cog, inv
