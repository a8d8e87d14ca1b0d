"""Synthetic scene generators for the BASELINE.json configs.

A `Scene` is a backend-neutral description (bodies, colliders, joints) that `instantiate()` replays through the
add API of any world object exposing the reference's call shapes (add_body / add_collider / add_*_constraint_global):
the HIP world (`World` in this package) in production and bench, the CPU oracle in tests.

Randomness is xorshift64 exactly as the reference's `random_number_generator` (src/core/random.h:14-60), seeds as
listed in BASELINE.md.  The humanoid ragdoll geometry restates src/physics/ragdoll.cpp:10-123 (input data only).
"""
import math
import numpy as np

SPHERE, CAPSULE, CYLINDER, AABB, OBB, HULL = 0, 1, 2, 3, 4, 5
STATIC = 0xFFFFFFFF
_M64 = (1 << 64) - 1


class XorShift64:
    """src/core/random.h:14-60."""

    def __init__(self, seed):
        self.state = seed & _M64

    def u64(self):
        x = self.state
        x ^= (x << 13) & _M64
        x ^= x >> 7
        x ^= (x << 17) & _M64
        self.state = x
        return x

    def f01(self):
        return np.float32(self.u64() & 0xFFFFFFFF) / np.float32(0xFFFFFFFF)

    def between(self, lo, hi):
        return float(np.float32(lo) + self.f01() * (np.float32(hi) - np.float32(lo)))

    def unit_quat(self):
        # uniform-ish random rotation: normalised 4-vector in [-1,1]^4, rejecting the corners
        while True:
            q = np.array([self.between(-1, 1) for _ in range(4)], np.float64)
            n = float(np.dot(q, q))
            if 1e-4 < n <= 1.0:
                q = (q / math.sqrt(n)).astype(np.float32)
                return q / np.float32(np.sqrt(np.dot(q, q)))


class Scene:
    def __init__(self, name, dt=1.0 / 120.0):
        self.name = name
        self.dt = dt
        self.bodies = []      # (pos3, rot4, kinematic, gravityFactor, linDamp, angDamp)
        self.colliders = []   # (body or STATIC, type, shape[<=10], material3, static_pos3, static_rot4)
        self.joints = []      # ("hinge"|"cone_twist"|..., a, b, args...)
        self.hulls = []       # convex hull geometries: (vertices [n,3], triangles [m,3]); HULL colliders refer to them by index
        self.fields = []      # force fields: (force3, pos3 or None, rot4 or None, [(type, shape), ...]); no colliders = global
        self.triggers = []    # triggers: (pos3 or None, rot4 or None, [(type, shape), ...])
        self.collision_events = False
        self.heightmap = None  # (chunksPerDim, chunkSize, material3, minCorner3, amplitudeScale, {(x, z): uint16[129, 129]})
        self.cloths = []      # (width, height, gridX, gridY, totalMass, stiffness, damping, gravityFactor, pos3, rot4): hung from its locked upper row at (pos, rot)
        self.cloth_iterations = (0, 1, 0)
        self.joint_edits = []  # (kind, index within kind, [(byte offset, "f4"|"u4", value), ...]): getConstraint(...).field = value after creation

    def add_body(self, pos, rot=(0, 0, 0, 1), kinematic=False, gravity_factor=1.0, linear_damping=0.4, angular_damping=0.4):
        self.bodies.append((tuple(float(x) for x in pos), tuple(float(x) for x in rot), kinematic, gravity_factor, linear_damping, angular_damping))
        return len(self.bodies) - 1

    def add_collider(self, body, ctype, shape, material, pos=(0, 0, 0), rot=(0, 0, 0, 1)):
        self.colliders.append((body, ctype, tuple(float(x) for x in shape), tuple(material), tuple(pos), tuple(rot)))
        return len(self.colliders) - 1

    def add_hull_geometry(self, vertices, triangles):
        self.hulls.append((np.asarray(vertices, np.float32).reshape(-1, 3), np.asarray(triangles, np.uint32).reshape(-1, 3)))
        return len(self.hulls) - 1

    def add_force_field(self, force, pos=None, rot=None, colliders=()):
        self.fields.append((tuple(force), pos, rot, list(colliders)))
        return len(self.fields) - 1

    def add_trigger(self, pos=None, rot=None, colliders=()):
        self.triggers.append((pos, rot, list(colliders)))
        return len(self.triggers) - 1

    def add_cloth(self, width, height, grid_x, grid_y, total_mass, pos, rot=(0, 0, 0, 1), stiffness=0.5, damping=0.3, gravity_factor=1.0):
        self.cloths.append((width, height, grid_x, grid_y, total_mass, stiffness, damping, gravity_factor, tuple(pos), tuple(rot)))
        return len(self.cloths) - 1

    def add_joint(self, kind, a, b, *args):
        self.joints.append((kind, a, b) + args)

    @property
    def num_bodies(self):
        return len(self.bodies)

    def instantiate(self, world):
        for v, t in self.hulls:
            world.add_hull_geometry(v, t)
        for pos, rot, kin, g, ld, ad in self.bodies:
            world.add_body(pos, rot, kinematic=kin, gravity_factor=g, linear_damping=ld, angular_damping=ad)
        for body, ctype, shape, mat, pos, rot in self.colliders:
            if body == STATIC:
                world.add_static_collider(ctype, shape, mat, pos, rot)
            else:
                world.add_collider(body, ctype, shape, mat)
        for j in self.joints:
            kind, a, b, args = j[0], j[1], j[2], j[3:]
            if kind.endswith("_local"):   # add*ConstraintFromLocalPoints (physics.h:209-235)
                getattr(world, "add_%s_constraint_local" % kind[:-6])(a, b, *args)
            else:
                getattr(world, "add_%s_constraint_global" % kind)(a, b, *args)
        kinds = {"distance": (0, 28), "ball": (1, 24), "fixed": (2, 40), "hinge": (3, 104), "cone_twist": (4, 120), "slider": (5, 72)}
        for kind, index, edits in self.joint_edits:
            ctype, nbytes = kinds[kind]
            pod = world.constraint_get(ctype, index, nbytes)
            for offset, dtype, value in edits:
                pod[offset:offset + 4].view(np.float32 if dtype == "f4" else np.uint32)[0] = value
            world.constraint_set(ctype, index, pod)
        for force, pos, rot, cols in self.fields:
            f = world.add_force_field(force, pos, rot)
            for ctype, shape in cols:
                world.add_force_field_collider(f, ctype, shape)
        for pos, rot, cols in self.triggers:
            t = world.add_trigger(pos, rot)
            for ctype, shape in cols:
                world.add_trigger_collider(t, ctype, shape)
        if self.collision_events:
            world.enable_collision_events(True, True)
        if self.heightmap is not None:
            cpd, size, mat, corner, amplitude, chunks = self.heightmap
            world.set_heightmap(cpd, size, mat, corner, amplitude)
            for (x, z), h in sorted(chunks.items()):
                world.heightmap_set_chunk(x, z, h)
        for width, height, gx, gy, mass, stiffness, damping, gravity, pos, rot in self.cloths:
            c = world.add_cloth(width, height, gx, gy, mass, stiffness, damping, gravity)
            world.cloth_set_fixed_vertices(c, pos, rot, True)
        if self.cloths:
            world.set_cloth_iterations(*self.cloth_iterations)
            world.set_cloth_colour_order(True)
        return world


DEFAULT_MATERIAL = (0.1, 0.5, 1.0)  # restitution, friction, density (reference application.cpp:176)


def _ground(scene, half_xz, material=DEFAULT_MATERIAL):
    # the reference's "platform": static AABB centre (0,-4,0) (application.cpp:209-212), widened to the scene
    scene.add_collider(STATIC, AABB, (-half_xz, -8.0, -half_xz, half_xz, 0.0, half_xz), material)


def c1_boxes(n=64, seed=15681923):
    """C1: n OBBs, half-extents (0.5,0.5,0.5)-(1,1,2), random orientation, dropped from y in [2,20] over 8x8 m."""
    rng = XorShift64(seed)
    s = Scene("c1_boxes_%d" % n)
    _ground(s, 30.0)
    for _ in range(n):
        he = (rng.between(0.5, 1.0), rng.between(0.5, 1.0), rng.between(0.5, 2.0))
        pos = (rng.between(-4, 4), rng.between(2, 20), rng.between(-4, 4))
        rot = rng.unit_quat()
        b = s.add_body(pos, rot)
        s.add_collider(b, OBB, (0, 0, 0, 1, 0, 0, 0) + he, DEFAULT_MATERIAL)
    return s


def c2_spheres(nx=22, ny=21, nz=22, seed=519431, radius=0.5, jitter=1e-3):
    """C2: nx*ny*nz unit-density spheres r=0.5 in a lattice with +-1 mm jitter (tie-free SAP), resting on the ground."""
    rng = XorShift64(seed)
    s = Scene("c2_spheres_%d" % (nx * ny * nz))
    _ground(s, max(nx, nz) * radius * 2 + 20.0)
    d = 2.0 * radius
    for iy in range(ny):
        for iz in range(nz):
            for ix in range(nx):
                pos = ((ix - 0.5 * (nx - 1)) * d + rng.between(-jitter, jitter),
                       radius + iy * d + rng.between(0, jitter),
                       (iz - 0.5 * (nz - 1)) * d + rng.between(-jitter, jitter))
                b = s.add_body(pos)
                s.add_collider(b, SPHERE, (0, 0, 0, radius), DEFAULT_MATERIAL)
    return s


def c3_mixed(n=100000, seed=14878213, area=200.0, column_height=50, pitch=1.4, layer=1.3):
    """C3/C5: n bodies, 1:1:1 sphere r in [0.3,0.6] / capsule (half-height 0.5, r 0.25) / OBB he in [0.3,0.8],
    random orientations, poured as `column_height`-high columns (column pitch `pitch`, vertical spacing `layer`: a dense block
    that is in contact from the first step and collapses outwards) on an area x area m ground."""
    rng = XorShift64(seed)
    s = Scene("c3_mixed_%d" % n)
    _ground(s, area * 0.5 + 10.0)
    ncol = (n + column_height - 1) // column_height
    side = int(math.ceil(math.sqrt(ncol)))
    i = 0
    for c in range(ncol):
        cx = (c % side - 0.5 * (side - 1)) * pitch
        cz = (c // side - 0.5 * (side - 1)) * pitch
        for k in range(column_height):
            if i >= n:
                break
            pos = (cx + rng.between(-0.1, 0.1), 1.0 + layer * k + rng.between(0, 0.05), cz + rng.between(-0.1, 0.1))
            rot = rng.unit_quat()
            b = s.add_body(pos, rot)
            kind = i % 3
            if kind == 0:
                s.add_collider(b, SPHERE, (0, 0, 0, rng.between(0.3, 0.6)), DEFAULT_MATERIAL)
            elif kind == 1:
                s.add_collider(b, CAPSULE, (0, -0.5, 0, 0, 0.5, 0, 0.25), DEFAULT_MATERIAL)
            else:
                he = (rng.between(0.3, 0.8), rng.between(0.3, 0.8), rng.between(0.3, 0.8))
                s.add_collider(b, OBB, (0, 0, 0, 1, 0, 0, 0) + he, DEFAULT_MATERIAL)
            i += 1
    return s


def all_shapes(n=600, seed=90210377, column_height=6, pitch=1.6, layer=1.5):
    """Every in-scope collider type (sphere, capsule, cylinder, AABB, OBB) in one pile: exercises all 15 type pairs incl. the
    cylinder kernels (collision_narrow.cpp:408-449, 614-703, 821-1043).  Every fourth body starts axis-aligned so that the parallel
    branches (abs(cos) > 0.99) and the face-clipping of tubes lying on the ground are hit."""
    rng = XorShift64(seed)
    s = Scene("all_shapes_%d" % n)
    _ground(s, 30.0)
    ncol = (n + column_height - 1) // column_height
    side = int(math.ceil(math.sqrt(ncol)))
    for i in range(n):
        c, k = divmod(i, column_height)
        cx = (c % side - 0.5 * (side - 1)) * pitch
        cz = (c // side - 0.5 * (side - 1)) * pitch
        pos = (cx + rng.between(-0.1, 0.1), 1.0 + layer * k + rng.between(0, 0.05), cz + rng.between(-0.1, 0.1))
        rot = rng.unit_quat()
        if i % 4 == 0:
            rot = (0.0, 0.0, 0.0, 1.0)
        b = s.add_body(pos, rot)
        kind = i % 5
        if kind == 0:
            s.add_collider(b, SPHERE, (0, 0, 0, rng.between(0.3, 0.6)), DEFAULT_MATERIAL)
        elif kind == 1:
            s.add_collider(b, CAPSULE, (-0.5, 0, 0, 0.5, 0, 0, 0.25), DEFAULT_MATERIAL)
        elif kind == 2:
            s.add_collider(b, CYLINDER, (-0.5, 0, 0, 0.5, 0, 0, rng.between(0.25, 0.5)), DEFAULT_MATERIAL)
        elif kind == 3:
            he = (rng.between(0.3, 0.7), rng.between(0.3, 0.7), rng.between(0.3, 0.7))
            s.add_collider(b, AABB, (-he[0], -he[1], -he[2]) + he, DEFAULT_MATERIAL)
        else:
            he = (rng.between(0.3, 0.8), rng.between(0.3, 0.8), rng.between(0.3, 0.8))
            s.add_collider(b, OBB, (0, 0, 0, 1, 0, 0, 0) + he, DEFAULT_MATERIAL)
    return s


# ---- convex hull geometries (outward-facing triangles) -------------------------------------------------------
def hull_box(hx, hy, hz):
    v = [(-hx, -hy, -hz), (hx, -hy, -hz), (hx, hy, -hz), (-hx, hy, -hz), (-hx, -hy, hz), (hx, -hy, hz), (hx, hy, hz), (-hx, hy, hz)]
    t = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (2, 3, 7), (2, 7, 6), (1, 2, 6), (1, 6, 5), (0, 4, 7), (0, 7, 3)]
    return v, t


def hull_octahedron(r):
    v = [(r, 0, 0), (-r, 0, 0), (0, r, 0), (0, -r, 0), (0, 0, r), (0, 0, -r)]
    t = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    return v, t


def hull_icosahedron(r):
    p = (1.0 + math.sqrt(5.0)) / 2.0
    s = r / math.sqrt(1.0 + p * p)
    v = [(-1, p, 0), (1, p, 0), (-1, -p, 0), (1, -p, 0), (0, -1, p), (0, 1, p), (0, -1, -p), (0, 1, -p), (p, 0, -1), (p, 0, 1), (-p, 0, -1), (-p, 0, 1)]
    v = [(a * s, b * s, c * s) for a, b, c in v]
    t = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    return v, t


def all_shapes_hull(n=490, seed=55120931, column_height=7, pitch=1.7, layer=1.6):
    """all_shapes plus convex hulls (box, octahedron, icosahedron geometries): every one of the 21 collider type pairs of
    collision_narrow.cpp:2473-2570 occurs."""
    rng = XorShift64(seed)
    s = Scene("all_shapes_hull_%d" % n)
    _ground(s, 30.0)
    geoms = [s.add_hull_geometry(*hull_box(0.5, 0.35, 0.45)), s.add_hull_geometry(*hull_octahedron(0.6)), s.add_hull_geometry(*hull_icosahedron(0.55))]
    ncol = (n + column_height - 1) // column_height
    side = int(math.ceil(math.sqrt(ncol)))
    for i in range(n):
        c, k = divmod(i, column_height)
        cx = (c % side - 0.5 * (side - 1)) * pitch
        cz = (c // side - 0.5 * (side - 1)) * pitch
        pos = (cx + rng.between(-0.1, 0.1), 1.0 + layer * k + rng.between(0, 0.05), cz + rng.between(-0.1, 0.1))
        rot = rng.unit_quat()
        if i % 5 == 0:
            rot = (0.0, 0.0, 0.0, 1.0)
        b = s.add_body(pos, rot)
        kind = i % 6
        if kind == 0:
            s.add_collider(b, SPHERE, (0, 0, 0, rng.between(0.3, 0.6)), DEFAULT_MATERIAL)
        elif kind == 1:
            s.add_collider(b, CAPSULE, (-0.5, 0, 0, 0.5, 0, 0, 0.25), DEFAULT_MATERIAL)
        elif kind == 2:
            s.add_collider(b, CYLINDER, (-0.5, 0, 0, 0.5, 0, 0, rng.between(0.25, 0.5)), DEFAULT_MATERIAL)
        elif kind == 3:
            he = (rng.between(0.3, 0.7), rng.between(0.3, 0.7), rng.between(0.3, 0.7))
            s.add_collider(b, AABB, (-he[0], -he[1], -he[2]) + he, DEFAULT_MATERIAL)
        elif kind == 4:
            he = (rng.between(0.3, 0.8), rng.between(0.3, 0.8), rng.between(0.3, 0.8))
            s.add_collider(b, OBB, (0, 0, 0, 1, 0, 0, 0) + he, DEFAULT_MATERIAL)
        else:
            s.add_collider(b, HULL, (0, 0, 0, 1, 0, 0, 0, float(geoms[(i // 6) % 3])), DEFAULT_MATERIAL)
    return s


# ---- humanoid ragdoll (reference src/physics/ragdoll.cpp:10-123) ------------------------------------------
def zones(n=420, seed=7741093):
    """all_shapes_hull under force fields and triggers: an updraft column (sphere), a rotated side-wind slab (OBB through the field
    entity's rotation), a capsule- and a cylinder-shaped field, a hull-shaped trigger, a two-collider trigger (AABB + sphere that
    overlap: one event per body), two global fields (wind + a little extra gravity) and collision begin / end events.  Every boolean
    overlap test of the 21 type pairs is reached (rigid-body shapes x zone shapes)."""
    s = all_shapes_hull(n=n, seed=seed)
    s.name = "zones_%d" % n
    oct_geo = s.add_hull_geometry(*hull_octahedron(2.5))
    q = _quat_axis_angle((0.0, 0.0, 1.0), 0.5)
    s.add_force_field((0.0, 60.0, 0.0), pos=(0.0, 0.0, 0.0), colliders=[(SPHERE, (1.0, 2.5, 1.0, 2.5))])
    s.add_force_field((25.0, 0.0, 0.0), pos=(-2.0, 1.0, 0.0), rot=q, colliders=[(OBB, (0.0, 0.0, 0.38268343, 0.92387953, 0.0, 1.0, 0.0, 3.0, 0.6, 3.0))])
    s.add_force_field((0.0, 0.0, 3.0))                       # global wind
    s.add_force_field((0.0, 0.0, -40.0), pos=(3.0, 0.0, 3.0), colliders=[(CAPSULE, (0.0, 0.5, 0.0, 0.0, 3.5, 0.0, 1.2))])
    s.add_force_field((-30.0, 10.0, 0.0), pos=(-3.0, 0.0, -3.0), colliders=[(CYLINDER, (0.0, 0.2, 0.0, 1.0, 3.0, 0.5, 1.4)), (AABB, (-1.0, 0.0, -1.0, 1.0, 1.0, 1.0))])
    s.add_force_field((0.0, -0.5, 0.0), rot=q)               # global, rotated by its entity
    s.add_trigger(pos=(2.0, 2.0, -2.0), rot=q, colliders=[(HULL, (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, float(oct_geo)))])
    s.add_trigger(colliders=[(AABB, (-6.0, 0.0, -1.0, 6.0, 1.5, 1.0)), (SPHERE, (0.0, 1.0, 0.0, 1.5))])
    s.add_trigger(pos=(0.0, 4.0, 0.0), colliders=[(CAPSULE, (-3.0, 0.0, 0.0, 3.0, 0.0, 0.0, 0.8)), (CYLINDER, (0.0, 0.0, -3.0, 0.0, 0.0, 3.0, 0.8)), (OBB, (0.0, 0.38268343, 0.0, 0.92387953, 0.0, 2.0, 0.0, 2.0, 0.3, 2.0))])
    s.collision_events = True
    return s


def _quat_axis_angle(axis, angle):
    h = np.float32(angle) * np.float32(0.5)
    s = np.float32(math.sin(h))
    return (float(axis[0] * s), float(axis[1] * s), float(axis[2] * s), float(np.float32(math.cos(h))))


def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


def _qrot(q, v):
    p = (v[0], v[1], v[2], 0.0)
    c = (-q[0], -q[1], -q[2], q[3])
    r = _qmul(_qmul(q, p), c)
    return (r[0], r[1], r[2])


def terrain_heights(chunks_per_dim, seed=4242, roughness=1.0):
    """Deterministic rolling terrain: a few sine waves + per-vertex hash noise, continuous across chunk borders, as uint16 heights per chunk."""
    n = chunks_per_dim * 128 + 1
    j, i = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64))   # i = z row, j = x column
    h = 0.5 + 0.22 * np.sin(j * 0.045 + 0.3) * np.cos(i * 0.037) + 0.12 * np.sin((i + j) * 0.11 + 1.0) + 0.05 * np.sin(i * 0.31) * np.sin(j * 0.29)
    k = (i.astype(np.uint64) * np.uint64(73856093)) ^ (j.astype(np.uint64) * np.uint64(19349663)) ^ np.uint64(seed)
    k = (k * np.uint64(2654435761)) % np.uint64(1 << 32)
    h = h + roughness * 0.004 * (k.astype(np.float64) / float(1 << 32) - 0.5)
    q = np.clip(np.round(h * 65535.0), 0, 65535).astype(np.uint16)
    return {(x, z): q[z * 128:z * 128 + 129, x * 128:x * 128 + 129].copy() for z in range(chunks_per_dim) for x in range(chunks_per_dim)}


def terrain(n=300, seed=31337, chunks_per_dim=2, chunk_size=24.0, amplitude=6.0):
    """Spheres, capsules, AABB-born and OBB boxes (and a few cylinders / hulls, which the terrain ignores like the reference's) dropped on a
    rolling heightmap of 2 x 2 chunks (one chunk left without heights: a hole), no ground plane: every supported collider type against
    terrain triangles, body-body contacts on top."""
    rng = XorShift64(seed)
    s = Scene("terrain_%d" % n, dt=1.0 / 120.0)
    span = chunks_per_dim * chunk_size
    chunks = terrain_heights(chunks_per_dim)
    if chunks_per_dim > 1:
        del chunks[(chunks_per_dim - 1, chunks_per_dim - 1)]
    s.heightmap = (chunks_per_dim, chunk_size, (0.1, 0.8, 1.0), (-span * 0.5, -2.0, -span * 0.5), amplitude, chunks)
    oct_geo = s.add_hull_geometry(*hull_octahedron(0.6))
    for i in range(n):
        x, z = rng.between(-span * 0.42, span * 0.2), rng.between(-span * 0.42, span * 0.2)
        y = rng.between(3.0, 9.0)
        kind = i % 10
        rot = rng.unit_quat() if kind != 4 else (0.0, 0.0, 0.0, 1.0)
        b = s.add_body((x, y, z), rot, angular_damping=0.4 if kind != 4 else 1.0e6)
        if kind in (0, 1, 2):
            s.add_collider(b, SPHERE, (0.0, 0.0, 0.0, rng.between(0.3, 0.7)), DEFAULT_MATERIAL)
        elif kind in (3, 5):
            s.add_collider(b, CAPSULE, (0.0, -0.5, 0.0, 0.0, 0.5, 0.0, rng.between(0.2, 0.4)), DEFAULT_MATERIAL)
        elif kind == 4:   # never rotates: stays an AABB in world space
            s.add_collider(b, AABB, (-0.5, -0.4, -0.6, 0.5, 0.4, 0.6), DEFAULT_MATERIAL)
        elif kind in (6, 7):
            s.add_collider(b, OBB, (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, rng.between(0.3, 0.8), rng.between(0.3, 0.8), rng.between(0.3, 0.8)), DEFAULT_MATERIAL)
        elif kind == 8:
            s.add_collider(b, CYLINDER, (0.0, -0.4, 0.0, 0.0, 0.4, 0.0, 0.4), DEFAULT_MATERIAL)
        else:
            s.add_collider(b, HULL, (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, float(oct_geo)), DEFAULT_MATERIAL)
    return s


def cloths(n=3):
    """Cloth banners (cloth.cpp) above a small pile of boxes, under a global wind field: a 20 x 20 and a 33 x 17 cloth with all three
    iteration kinds on (they fit the LDS variant of the kernel), and one 60 x 50 cloth that does not (global-memory variant)."""
    s = c1_boxes(16)
    s.name = "cloths_%d" % n
    s.add_force_field((3.0, 0.5, 1.5))
    q = _quat_axis_angle((0.0, 1.0, 0.0), 0.6)
    s.add_cloth(10.0, 10.0, 20, 20, 8.0, (0.0, 14.0, 0.0))
    if n > 1:
        s.add_cloth(6.0, 4.0, 33, 17, 3.0, (8.0, 9.0, -3.0), q, stiffness=0.8, damping=0.1)
    if n > 2:
        s.add_cloth(12.0, 9.0, 60, 50, 20.0, (-9.0, 16.0, 4.0), q, stiffness=0.3, gravity_factor=0.7)
    s.cloth_iterations = (2, 3, 1)
    return s


def _rotate_from_to(frm, to):
    """rotateFromTo (core/math.cpp): shortest rotation taking `frm` to `to`."""
    f = np.asarray(frm, np.float64); f = f / np.linalg.norm(f)
    t = np.asarray(to, np.float64); t = t / np.linalg.norm(t)
    d = float(f @ t)
    if d >= 1.0:
        return (0.0, 0.0, 0.0, 1.0)
    if d < 1e-6 - 1.0:
        axis = np.cross((1.0, 0.0, 0.0), f)
        if not axis.any():
            axis = np.cross((0.0, 1.0, 0.0), f)
        axis = axis / np.linalg.norm(axis)
        return _quat_axis_angle(tuple(axis), math.pi)
    sq = math.sqrt((1.0 + d) * 2.0)
    c = np.cross(f, t) / sq
    q = np.array([c[0], c[1], c[2], sq * 0.5]); q = q / np.linalg.norm(q)
    return tuple(float(x) for x in q)


# hinge_constraint field offsets (constraints.h:229-257)
HINGE_MAX_MOTOR_TORQUE, HINGE_MOTOR_TYPE, HINGE_MOTOR_VELOCITY = 56, 60, 64
VELOCITY_MOTOR, POSITION_MOTOR = 0, 1

VEHICLE_PARTS = ["motor", "motorGear", "driveAxis", "frontAxis", "steeringWheel", "steeringAxis", "leftWheelSuspension", "rightWheelSuspension",
                 "leftFrontWheel", "rightFrontWheel", "leftWheelArm", "rightWheelArm", "differentialSunGear", "differentialSpiderGear",
                 "leftRearWheel", "rightRearWheel"]


def add_vehicle(s, position=(0.0, 1.5, 0.0), yaw=0.0, motor_velocity=0.0):
    """vehicle::initialize (vehicle.cpp:283-485): 16 bodies — a motor block, gears whose colliders are their capsule teeth (the gear disks
    are render-only), a rack-and-pinion steering, a differential, four cylinder wheels — held together by 11 hinges (the motor one driven,
    the steering wheel position-controlled), 1 fixed, 1 slider and 4 ball joints.  Rods and wheel suspensions carry no colliders (they
    penetrate the wheels).  Everything is laid out around the origin, joints are made from global points there, then every part is moved
    by (yaw about y, position) like the reference does at the end.  Returns {part name: body id} and {"motor" / "steering": hinge index}."""
    Y, X, Z = (0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)
    ident = (0.0, 0.0, 0.0, 1.0)
    d2r = math.pi / 180.0
    density = 2000.0
    place_q = _quat_axis_angle(Y, yaw)

    def add(a, b):
        return tuple(x + y for x, y in zip(a, b))

    def sub(a, b):
        return tuple(x - y for x, y in zip(a, b))

    def placed_point(pt):
        return add(_qrot(place_q, pt), position)

    bodies = {}
    hinge_base = sum(1 for j in s.joints if j[0] == "hinge")
    hinges = []

    def make_body(name, pos, rot):
        bodies[name] = s.add_body(placed_point(pos), _qmul(place_q, rot))
        return bodies[name]

    def joint(kind, a, b, anchor, *rest):
        args = [placed_point(anchor)]
        if kind in ("hinge", "slider"):
            args.append(_qrot(place_q, rest[0]))
            args.extend(rest[1:])
        s.add_joint(kind, bodies[a], bodies[b], *args)
        if kind == "hinge":
            hinges.append((a, b))
            return hinge_base + len(hinges) - 1
        return None

    def gear(height, radius, teeth, tooth_length, tooth_width, friction, dens):
        return ("gear", height, radius, teeth, tooth_length, tooth_width, friction, dens)

    def attach(body, attachment, rod_length, sign):
        rod_offset = rod_length * sign
        if attachment[0] == "gear":
            _, height, radius, teeth, tlen, twid, friction, dens = attachment
            for i in range(teeth):
                lr = _quat_axis_angle(Y, i * 2.0 * math.pi / teeth)
                center = add(_qrot(lr, (radius + tlen * 0.5, 0.0, 0.0)), (0.0, rod_offset, 0.0))
                half = _qrot(lr, (tlen * 0.5, 0.0, 0.0))
                s.add_collider(body, CAPSULE, sub(center, half) + add(center, half) + (twid * 0.5,), (0.2, friction, dens))
        elif attachment[0] == "wheel":
            _, height, radius, friction, dens = attachment
            s.add_collider(body, CYLINDER, (0.0, rod_offset - height * 0.5, 0.0, 0.0, rod_offset + height * 0.5, 0.0, radius), (0.2, friction, dens))

    def create_axis(name, pos, rot, desc, first=None, second=None):   # createAxis, vehicle.cpp:147-175
        b = make_body(name, pos, rot)
        attach(b, desc, 0.0, 1.0)
        if first:
            attach(b, first[0], first[1], 1.0)
        if second:
            attach(b, second[0], second[1], -1.0)
        return b

    def create_rod(name, frm, to):                                     # createRod, vehicle.cpp:257-281: no collider
        axis = sub(to, frm)
        return make_body(name, tuple((a + b) * 0.5 for a, b in zip(frm, to)), _rotate_from_to(Y, axis))

    motor_gear = gear(0.1, 0.2, 8, 0.07, 0.1, 0.0, density)
    steering_wheel_desc = gear(0.1, 0.4, 0, 0.07, 0.1, 0.0, density)
    wheel = ("wheel", 0.3, 0.7, 1.0, 50.0)
    motor_gear_y, gear_offset = 0.25, 0.26

    m = make_body("motor", (0.0, 0.0, 0.0), ident)
    s.add_collider(m, AABB, (-0.6, -0.1, -1.0, 0.6, 0.1, 1.0), (0.2, 0.0, density))

    create_axis("motorGear", (0.0, motor_gear_y, 0.0), ident, motor_gear)
    motor_hinge = joint("hinge", "motor", "motorGear", (0.0, motor_gear_y, 0.0), Y, 1.0, -1.0)

    drive_axis_length = 4.5
    create_axis("driveAxis", (0.0, motor_gear_y + gear_offset, gear_offset), _quat_axis_angle((-1.0, 0.0, 0.0), 90 * d2r), motor_gear,
                None, (motor_gear, drive_axis_length * 0.57 - 1.1))
    joint("hinge", "motor", "driveAxis", (0.0, motor_gear_y + gear_offset, gear_offset), Z, 1.0, -1.0)

    axis_length, suspension_length = 1.5, 0.4
    front_axis_offset_z = -drive_axis_length * 0.5 + gear_offset * 2.0
    front_axis_pos = (0.0, motor_gear_y + gear_offset, front_axis_offset_z)
    create_rod("frontAxis", add(front_axis_pos, (axis_length, 0.0, 0.0)), sub(front_axis_pos, (axis_length, 0.0, 0.0)))
    joint("fixed", "motor", "frontAxis", front_axis_pos)

    steering_wheel_rot = _quat_axis_angle((-1.0, 0.0, 0.0), -80 * d2r)
    steering_wheel_pos = (0.0, 1.12, 0.81)
    create_axis("steeringWheel", steering_wheel_pos, steering_wheel_rot, steering_wheel_desc, None, (motor_gear, 2.0))
    steering_hinge = joint("hinge", "motor", "steeringWheel", steering_wheel_pos, _qrot(steering_wheel_rot, (0.0, -1.0, 0.0)), 1.0, -1.0)

    steering_axis_pos = (0.0, motor_gear_y + gear_offset + 0.06, front_axis_offset_z + 0.49)
    steering_axis_length = axis_length * 1.05
    sa = make_body("steeringAxis", steering_axis_pos, steering_wheel_rot)   # createGearAxis, vehicle.cpp:177-222: a rack of 8 capsule teeth
    tooth_length, tooth_width = 0.07, 0.1
    stride = (steering_axis_length - tooth_width) / 7.0
    for i in range(8):
        x = -0.5 * steering_axis_length + 0.5 * tooth_width + i * stride
        s.add_collider(sa, CAPSULE, (x, tooth_width * 0.5 + tooth_length * 0.5, 0.0, x, tooth_width * 0.5 - tooth_length * 0.5, 0.0, tooth_width * 0.5), (0.2, 0.0, density))
    joint("slider", "motor", "steeringAxis", steering_axis_pos, X, -4.0, 4.0)
    left_rack_end = sub(steering_axis_pos, (steering_axis_length * 0.5, 0.0, 0.0))
    right_rack_end = add(steering_axis_pos, (steering_axis_length * 0.5, 0.0, 0.0))

    left_susp_pos = sub(front_axis_pos, (axis_length, 0.0, 0.0))
    left_susp_attach = add(left_susp_pos, (0.0, 0.0, suspension_length))
    make_body("leftWheelSuspension", left_susp_pos, ident)              # createWheelSuspension: no collider
    joint("hinge", "motor", "leftWheelSuspension", left_susp_pos, Y, -45 * d2r, 45 * d2r)
    right_susp_pos = add(front_axis_pos, (axis_length, 0.0, 0.0))
    right_susp_attach = add(right_susp_pos, (0.0, 0.0, suspension_length))
    make_body("rightWheelSuspension", right_susp_pos, ident)
    joint("hinge", "motor", "rightWheelSuspension", right_susp_pos, Y, -45 * d2r, 45 * d2r)

    wheel_rot = _quat_axis_angle(Z, 90 * d2r)
    left_front_pos = sub(left_susp_pos, (suspension_length * 0.5, 0.0, 0.0))
    lf = make_body("leftFrontWheel", left_front_pos, wheel_rot)          # createWheel, vehicle.cpp:224-247
    attach(lf, wheel, 0.0, 1.0)
    right_front_pos = add(right_susp_pos, (suspension_length * 0.5, 0.0, 0.0))
    rf = make_body("rightFrontWheel", right_front_pos, wheel_rot)
    attach(rf, wheel, 0.0, 1.0)
    joint("hinge", "leftFrontWheel", "leftWheelSuspension", left_front_pos, X, 1.0, -1.0)
    joint("hinge", "rightFrontWheel", "rightWheelSuspension", right_front_pos, X, 1.0, -1.0)

    create_rod("leftWheelArm", left_rack_end, left_susp_attach)
    create_rod("rightWheelArm", right_rack_end, right_susp_attach)
    joint("ball", "leftWheelSuspension", "leftWheelArm", left_susp_attach)
    joint("ball", "steeringAxis", "leftWheelArm", left_rack_end)
    joint("ball", "rightWheelSuspension", "rightWheelArm", right_susp_attach)
    joint("ball", "steeringAxis", "rightWheelArm", right_rack_end)

    rear_gear = gear(0.1, 0.5, 17, 0.07, 0.1, 0.0, density)
    rear_z, rear_x = drive_axis_length * 0.505, -gear_offset
    sun_pos = (rear_x, motor_gear_y + gear_offset, rear_z)
    create_axis("differentialSunGear", sun_pos, _quat_axis_angle((0.0, 0.0, -1.0), 90 * d2r), rear_gear)
    joint("hinge", "motor", "differentialSunGear", sun_pos, X, 1.0, -1.0)

    spider_pos = (0.11, motor_gear_y + gear_offset * 2.0, rear_z)
    create_axis("differentialSpiderGear", spider_pos, ident, motor_gear, (("none",), 0.2))
    joint("hinge", "differentialSunGear", "differentialSpiderGear", spider_pos, Y, 1.0, -1.0)

    left_rear_pos = add(spider_pos, (-gear_offset, -gear_offset, 0.0))
    right_rear_pos = add(spider_pos, (gear_offset, -gear_offset, 0.0))
    rear_rot = _quat_axis_angle((0.0, 0.0, -1.0), 90 * d2r)
    create_axis("leftRearWheel", left_rear_pos, rear_rot, motor_gear, None, (wheel, axis_length + spider_pos[0]))
    create_axis("rightRearWheel", right_rear_pos, rear_rot, motor_gear, (wheel, axis_length - spider_pos[0]), None)
    joint("hinge", "motor", "leftRearWheel", left_rear_pos, X, 1.0, -1.0)
    joint("hinge", "motor", "rightRearWheel", right_rear_pos, X, 1.0, -1.0)

    # getConstraint(scene, handle).field = ... (vehicle.cpp:371-373, 399-402)
    s.joint_edits.append(("hinge", motor_hinge, [(HINGE_MAX_MOTOR_TORQUE, "f4", 500.0), (HINGE_MOTOR_VELOCITY, "f4", motor_velocity)]))
    s.joint_edits.append(("hinge", steering_hinge, [(HINGE_MOTOR_TYPE, "u4", POSITION_MOTOR), (HINGE_MAX_MOTOR_TORQUE, "f4", 1000.0), (HINGE_MOTOR_VELOCITY, "f4", 0.0)]))
    return bodies, {"motor": motor_hinge, "steering": steering_hinge}


def vehicles(n=1, motor_velocity=3.0, pitch=8.0):
    """n gear-driven vehicles (vehicle.cpp) on the ground, motors running: compound capsule-tooth gears meshing through contacts, cylinder
    wheels on an AABB floor, every joint type but distance and cone-twist."""
    s = Scene("vehicles_%d" % n, dt=1.0 / 120.0)
    side = int(math.ceil(math.sqrt(n)))
    _ground(s, side * pitch * 0.5 + 20.0, material=(0.1, 1.0, 1.0))
    for i in range(n):
        x = (i % side - 0.5 * (side - 1)) * pitch
        z = (i // side - 0.5 * (side - 1)) * pitch
        add_vehicle(s, (x, 1.1, z), yaw=0.3 * i, motor_velocity=motor_velocity)
    return s


def add_ragdoll(s, hip, yaw=0.0):
    """Appends one 14-body humanoid (17 colliders, 7 cone-twist + 6 hinge).  Returns the body ids."""
    sc = 0.42
    mat = (0.2, 1.0, 985.0)
    d2r = math.pi / 180.0
    Z = (0.0, 0.0, 1.0)
    ident = (0.0, 0.0, 0.0, 1.0)
    parts = {  # name: (position * scale, rotation)                                    ragdoll.cpp:21-34
        "torso": ((0.0, 0.0, 0.0), ident), "head": ((0.0, 1.45, 0.0), ident),
        "lua": ((-0.6, 0.75, 0.0), _quat_axis_angle(Z, -30 * d2r)), "lla": ((-0.884, 0.044, -0.043), _quat_axis_angle(Z, -20 * d2r)),
        "rua": ((0.6, 0.75, 0.0), _quat_axis_angle(Z, 30 * d2r)), "rla": ((0.884, 0.044, -0.043), _quat_axis_angle(Z, 20 * d2r)),
        "lul": ((-0.371, -0.812, 0.0), _quat_axis_angle(Z, -10 * d2r)), "lll": ((-0.452, -1.955, 0.0), _quat_axis_angle(Z, -3.5 * d2r)),
        "lf": ((-0.498, -2.585, -0.18), ident), "lt": ((-0.498, -2.585, -0.637), ident),
        "rul": ((0.371, -0.812, 0.0), _quat_axis_angle(Z, 10 * d2r)), "rll": ((0.452, -1.955, 0.0), _quat_axis_angle(Z, 3.5 * d2r)),
        "rf": ((0.498, -2.585, -0.18), ident), "rt": ((0.498, -2.585, -0.637), ident),
    }
    order = ["torso", "head", "lua", "lla", "rua", "rla", "lul", "lll", "lf", "lt", "rul", "rll", "rf", "rt"]
    T = {k: (tuple(sc * x for x in p), q) for k, (p, q) in parts.items()}
    yawq = _quat_axis_angle((0.0, 1.0, 0.0), yaw)

    def world_tf(name):  # ragdoll.cpp:125-133: rotate about the hip, then translate
        p, q = T[name]
        wp = _qrot(yawq, p)
        return (wp[0] + hip[0], wp[1] + hip[1], wp[2] + hip[2]), _qmul(yawq, q)

    ids = {}
    for name in order:
        pos, rot = world_tf(name)
        ids[name] = s.add_body(pos, rot)

    def cap(name, a, b, r):
        s.add_collider(ids[name], CAPSULE, tuple(sc * x for x in a) + tuple(sc * x for x in b) + (sc * r,), mat)

    cap("torso", (-0.2, 0, 0), (0.2, 0, 0), 0.25); cap("torso", (-0.16, 0.32, 0), (0.16, 0.32, 0), 0.2)       # ragdoll.cpp:36-42
    cap("torso", (-0.14, 0.62, 0), (0.14, 0.62, 0), 0.22); cap("torso", (-0.14, 0.92, 0), (0.14, 0.92, 0), 0.2)
    cap("head", (0, -0.075, 0), (0, 0.075, 0), 0.25)
    for n in ("lua", "lla", "rua", "rla"):
        cap(n, (0, -0.2, 0), (0, 0.2, 0), 0.15)
    cap("lul", (0, -0.3, 0), (0, 0.3, 0), 0.25); cap("lll", (0, -0.3, 0), (0, 0.3, 0), 0.18)
    foot = tuple(sc * x for x in (0.1587, 0.1, 0.3424))
    s.add_collider(ids["lf"], AABB, tuple(-x for x in foot) + foot, mat)
    cap("lt", (-0.0587, 0, 0), (0.0587, 0, 0), 0.1)
    cap("rul", (0, -0.3, 0), (0, 0.3, 0), 0.25); cap("rll", (0, -0.3, 0), (0, 0.3, 0), 0.18)
    s.add_collider(ids["rf"], AABB, tuple(-x for x in foot) + foot, mat)
    cap("rt", (-0.0587, 0, 0), (0.0587, 0, 0), 0.1)

    def tp(name, local):  # transformPosition(partTransform, scale*local) in the ragdoll's frame, then placed
        p, q = T[name]
        r = _qrot(q, tuple(sc * x for x in local))
        lp = (r[0] + p[0], r[1] + p[1], r[2] + p[2])
        wp = _qrot(yawq, lp)
        return (wp[0] + hip[0], wp[1] + hip[1], wp[2] + hip[2])

    def td(name, d):
        return _qrot(yawq, _qrot(T[name][1], d))

    def wd(d):
        return _qrot(yawq, d)

    n2 = 1.0 / math.sqrt(2.0)
    # NB: the reference creates joints before applying initialRotation; with local anchors that is equivalent to
    # creating them in the rotated pose, which is what we do (global anchors/axes rotated by yaw).          ragdoll.cpp:108-123
    s.add_joint("cone_twist", ids["torso"], ids["head"], tp("torso", (0, 1.2, 0)), wd((0, 1, 0)), 50 * d2r, 90 * d2r)
    s.add_joint("cone_twist", ids["torso"], ids["lua"], tp("torso", (-0.4, 1.0, 0)), wd((-1, 0, 0)), 130 * d2r, 90 * d2r)
    s.add_joint("hinge", ids["lua"], ids["lla"], tp("lua", (0, -0.42, 0)), wd((n2, 0, n2)), -5 * d2r, 85 * d2r)
    s.add_joint("cone_twist", ids["torso"], ids["rua"], tp("torso", (0.4, 1.0, 0)), wd((1, 0, 0)), 130 * d2r, 90 * d2r)
    s.add_joint("hinge", ids["rua"], ids["rla"], tp("rua", (0, -0.42, 0)), wd((n2, 0, -n2)), -5 * d2r, 85 * d2r)
    s.add_joint("cone_twist", ids["torso"], ids["lul"], tp("torso", (-0.3, -0.25, 0)), td("lul", (0, -1, 0)), -1.0, 30 * d2r)
    s.add_joint("hinge", ids["lul"], ids["lll"], tp("lul", (0, -0.6, 0)), wd((1, 0, 0)), -90 * d2r, 5 * d2r)
    s.add_joint("cone_twist", ids["lll"], ids["lf"], tp("lll", (0, -0.52, 0)), td("lll", (0, -1, 0)), 75 * d2r, 20 * d2r)
    s.add_joint("hinge", ids["lf"], ids["lt"], tp("lf", (0, 0, -0.36)), wd((1, 0, 0)), -45 * d2r, 45 * d2r)
    s.add_joint("cone_twist", ids["torso"], ids["rul"], tp("torso", (0.3, -0.25, 0)), td("rul", (0, -1, 0)), -1.0, 30 * d2r)
    s.add_joint("hinge", ids["rul"], ids["rll"], tp("rul", (0, -0.6, 0)), wd((1, 0, 0)), -90 * d2r, 5 * d2r)
    s.add_joint("cone_twist", ids["rll"], ids["rf"], tp("rll", (0, -0.52, 0)), td("rll", (0, -1, 0)), 75 * d2r, 20 * d2r)
    s.add_joint("hinge", ids["rf"], ids["rt"], tp("rf", (0, 0, -0.36)), wd((1, 0, 0)), -45 * d2r, 45 * d2r)
    return [ids[n] for n in order]


def c4_ragdolls(n=256, pitch=3.0, hip_y=1.25):
    """C4: n humanoid ragdolls on a sqrt(n) x sqrt(n) grid, hips at y=1.25 (learned_locomotion.cpp:446), 60 Hz."""
    s = Scene("c4_ragdolls_%d" % n, dt=1.0 / 60.0)
    side = int(math.ceil(math.sqrt(n)))
    _ground(s, side * pitch * 0.5 + 20.0, material=(0.1, 1.0, 1.0))
    for i in range(n):
        x = (i % side - 0.5 * (side - 1)) * pitch
        z = (i // side - 0.5 * (side - 1)) * pitch
        add_ragdoll(s, (x, hip_y, z))
    return s


def c4_heap(side=8, layers=2, pitch=0.8, hip_y=1.25, layer_height=2.0):
    """Ragdolls packed at `pitch` (closer than their arm span) in `layers` layers and dropped into a heap: islands of different cluster
    tasks touch, so contacts between jointed bodies of different tasks are cut into the later phases and jointed bodies are handed
    between phases.  Alternate ragdolls are turned by 90 degrees so that the spawn is not one block of overlapping arms."""
    s = Scene("c4_heap_%dx%dx%d" % (side, side, layers), dt=1.0 / 60.0)
    _ground(s, side * pitch * 0.5 + 20.0, material=(0.1, 1.0, 1.0))
    for l in range(layers):
        for i in range(side * side):
            x = (i % side - 0.5 * (side - 1)) * pitch
            z = (i // side - 0.5 * (side - 1)) * pitch
            add_ragdoll(s, (x + 0.13 * l, hip_y + layer_height * l, z - 0.07 * l), yaw=0.5 * math.pi * ((i % side + i // side + l) % 2))
    return s


def joints_mix():
    """Every joint kind the BASELINE configs leave out, on the ground so that contacts and joints meet in one solve: two hanging
    chains held by DISTANCE joints (one added from global points, one from local points, each from a kinematic anchor), a chain of
    BALL joints added from local points, and a three-link CONE-TWIST chain whose swing and twist motors run (velocity motor on one
    joint, angle motor on the other: constraints.cpp:1880-1960), all above a pile of loose bodies they fall into."""
    s = Scene("joints_mix", dt=1.0 / 120.0)
    _ground(s, 30.0)
    mat = (0.1, 0.6, 2.0)

    def link(pos, kinematic=False, radius=0.25):
        b = s.add_body(pos, kinematic=kinematic)
        s.add_collider(b, SPHERE, (0.0, 0.0, 0.0, radius), mat)
        return b

    # distance chain from global points
    prev = link((-3.0, 6.0, 0.0), kinematic=True)
    for i in range(6):
        cur = link((-3.0 + 0.7 * (i + 1), 6.0, 0.0))
        s.add_joint("distance", prev, cur, (-3.0 + 0.7 * i, 6.0, 0.0), (-3.0 + 0.7 * (i + 1), 6.0, 0.0))
        prev = cur
    # distance chain from local points (anchors 0.1 off the centres, rest length given)
    prev = link((3.0, 6.0, 1.0), kinematic=True)
    for i in range(6):
        cur = link((3.0, 6.0, 1.0 + 0.8 * (i + 1)))
        s.add_joint("distance_local", prev, cur, (0.0, 0.0, 0.1), (0.0, 0.0, -0.1), 0.6)
        prev = cur
    # ball chain from local points
    prev = link((0.0, 7.0, -3.0), kinematic=True)
    for i in range(5):
        cur = link((0.0, 7.0 - 0.6 * (i + 1), -3.0))
        s.add_joint("ball_local", prev, cur, (0.0, -0.3, 0.0), (0.0, 0.3, 0.0))
        prev = cur
    # cone-twist chain with motors
    base = s.add_body((0.0, 3.0, 3.0), kinematic=True)
    s.add_collider(base, OBB, (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.3, 0.3, 0.3), mat)
    prev = base
    for i in range(3):
        cur = s.add_body((0.9 * (i + 1), 3.0, 3.0))
        s.add_collider(cur, CAPSULE, (-0.25, 0.0, 0.0, 0.25, 0.0, 0.0, 0.15), mat)
        s.add_joint("cone_twist", prev, cur, (0.9 * i + 0.45, 3.0, 3.0), (1.0, 0.0, 0.0), 0.9, 0.7)
        prev = cur
    # joint 0: swing velocity motor about the tangent axis; joint 1: twist angle motor; joint 2: both (mi_cone_twist_constraint byte offsets)
    s.joint_edits.append(("cone_twist", 0, [(92, "u4", 0), (96, "f4", 1.5), (100, "f4", 40.0), (104, "f4", 0.3)]))
    s.joint_edits.append(("cone_twist", 1, [(108, "u4", 1), (112, "f4", 0.5), (116, "f4", 30.0)]))
    s.joint_edits.append(("cone_twist", 2, [(92, "u4", 1), (96, "f4", 0.4), (100, "f4", 25.0), (104, "f4", 1.2), (108, "u4", 0), (112, "f4", -1.0), (116, "f4", 20.0)]))
    # loose bodies underneath
    rng = XorShift64(77120451)
    for i in range(60):
        pos = (rng.between(-4.0, 4.0), rng.between(0.4, 2.5), rng.between(-4.0, 4.0))
        b = s.add_body(pos, rng.unit_quat())
        if i % 2:
            s.add_collider(b, OBB, (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, rng.between(0.2, 0.4), rng.between(0.2, 0.4), rng.between(0.2, 0.4)), mat)
        else:
            s.add_collider(b, SPHERE, (0.0, 0.0, 0.0, rng.between(0.2, 0.4)), mat)
    return s


def rest_stacks(stacks=16, height=5):
    """Things that come to REST and stay there: stacks of axis-aligned boxes (one box wide, `height` high, dropped from 1 mm above
    each other) and single spheres and capsules lying on the ground.  Unlike a pile or a lattice of spheres, which topple and roll
    (two Gauss-Seidel orders put their bodies metres apart), a resting stack has one equilibrium: the invariants of SURVEY section
    8(c) — resting height within 1 mm, penetration within 1 mm — are meaningful here."""
    s = Scene("rest_stacks_%d" % stacks, dt=1.0 / 120.0)
    _ground(s, 40.0)
    side = int(np.ceil(np.sqrt(stacks)))
    for k in range(stacks):
        x, z = 3.0 * (k % side) - 1.5 * side, 3.0 * (k // side) - 1.5 * side
        for level in range(height):
            b = s.add_body((x, 0.5 + 0.001 + level * 1.001, z))
            s.add_collider(b, OBB, (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.5, 0.5, 0.5), DEFAULT_MATERIAL)
        b = s.add_body((x + 1.5, 0.301, z))
        s.add_collider(b, SPHERE, (0.0, 0.0, 0.0, 0.3), DEFAULT_MATERIAL)
        b = s.add_body((x, 0.251, z + 1.5))
        s.add_collider(b, CAPSULE, (-0.4, 0.0, 0.0, 0.4, 0.0, 0.0, 0.25), DEFAULT_MATERIAL)
    return s


def by_name(name):
    if name == "rest_stacks":
        return rest_stacks()
    if name == "joints_mix":
        return joints_mix()
    if name == "c1":
        return c1_boxes()
    if name == "c2":
        return c2_spheres()
    if name == "c2_small":
        return c2_spheres(10, 10, 10)
    if name == "c3":
        return c3_mixed()
    if name == "c3_small":
        return c3_mixed(3000, area=40.0, column_height=30)
    if name == "c4":
        return c4_ragdolls()
    if name == "c4_small":
        return c4_ragdolls(4)
    if name == "c4_heap":
        return c4_heap()
    if name == "c5":
        return c3_mixed(1000000, area=700.0)
    if name == "c3_mid":
        return c3_mixed(20000)
    if name == "shapes":
        return all_shapes()
    if name == "shapes_hull":
        return all_shapes_hull()
    if name == "zones":
        return zones()
    if name == "cloths":
        return cloths()
    if name == "terrain":
        return terrain()
    if name == "vehicle":
        return vehicles(1)
    if name == "vehicles":
        return vehicles(16)
    raise KeyError(name)
