"""URDF -> flat structure-of-arrays robot model.

Host-side replacement for what the reference obtains from
``p.loadURDF(path, basePosition=..., baseOrientation=..., flags=URDF_USE_SELF_COLLISION
+ URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS)``
(/root/reference/humanoid_gym/envs/dancer_env.py:58-60) followed by the
``getNumJoints`` / ``getJointInfo`` sweep that builds ``joint2Index`` and
``link2Index`` (dancer_env.py:73-79).

Numbering: joints are indexed depth-first from the root link, visiting each
link's child joints in declaration order -- the order PyBullet reports
[upstream recall, see DESIGN.md "parity unpinned"].  Fixed joints keep their
index in the name maps but are welded into their carrier body for dynamics.

Body frames: every body gets an internal frame at its centre of mass with axes
along the principal axes of inertia (a Jacobi sweep over the 3x3 tensor, only
when the tensor has off-diagonal terms), which is what Bullet's URDF importer
does before it builds a btMultiBody.

Collision: Bullet turns a collision mesh into the convex hull of its vertices, and a
hull meets the ground plane at its support vertices.  Here a mesh is reduced to (up to)
8 of its own vertices -- the support vertices of the hull in the eight diagonal
directions of its link-frame bounding box, i.e. the mesh vertices nearest the box
corners (ground contact; points on the hull, never outside it) -- plus up to 3
inscribed spheres along the longest box axis (self contact).  When only bounds are
known (``mesh_aabbs``) the box corners themselves are used.  ``<box>``, ``<sphere>``
and ``<cylinder>`` are handled analytically.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .model import FlatModel, SPHERE_GROUND, SPHERE_SELF

BIG = 1.0e30


# ----------------------------------------------------------------------------
# small rotation helpers (float64)
# ----------------------------------------------------------------------------
def rpy_to_mat(rpy: Sequence[float]) -> np.ndarray:
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]])


def quat_to_mat(q: Sequence[float]) -> np.ndarray:
    x, y, z, w = q
    n = math.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def mat_to_quat(m: np.ndarray) -> np.ndarray:
    """Rotation matrix -> xyzw quaternion with w >= 0."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    q = np.array(q)
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def principal_axes(I: np.ndarray, threshold: float = 1e-6, max_steps: int = 30) -> Tuple[np.ndarray, np.ndarray]:
    """Cyclic Jacobi diagonalisation of a symmetric 3x3 tensor.

    Returns (moments[3], V) with V's columns the principal axes, I = V diag V^T.
    Each step annihilates the largest off-diagonal element; the sweep stops
    once that element is below ``threshold`` times the trace of absolute
    diagonal terms.  The eigenvalues stay in the slot they converge in (no
    sorting), so an already diagonal tensor is returned untouched.
    """
    A = np.array(I, dtype=np.float64)
    V = np.eye(3)
    eps = np.finfo(np.float64).eps
    step = max_steps
    while step > 0:
        p, q, r = 0, 1, 2
        big = abs(A[0, 1])
        if abs(A[0, 2]) > big:
            q, r, big = 2, 1, abs(A[0, 2])
        if abs(A[1, 2]) > big:
            p, q, r, big = 1, 2, 0, abs(A[1, 2])
        t = threshold * (abs(A[0, 0]) + abs(A[1, 1]) + abs(A[2, 2]))
        if big <= t:
            if big <= eps * t:
                break
            step = 1                      # one last clean-up rotation
        apq = A[p, q]
        theta = (A[q, q] - A[p, p]) / (2.0 * apq)
        th2 = theta * theta
        if th2 * th2 < 10.0 / eps:
            tt = 1.0 / (theta + math.copysign(math.sqrt(1.0 + th2), theta if theta != 0 else 1.0))
            c = 1.0 / math.sqrt(1.0 + tt * tt)
            s = c * tt
        else:                             # nearly diagonal already
            tt = 1.0 / (theta * (2.0 + 0.5 / th2))
            c = 1.0 - 0.5 * tt * tt
            s = c * tt
        A[p, q] = A[q, p] = 0.0
        A[p, p] -= tt * apq
        A[q, q] += tt * apq
        arp, arq = A[r, p], A[r, q]
        A[r, p] = A[p, r] = c * arp - s * arq
        A[r, q] = A[q, r] = c * arq + s * arp
        vp, vq = V[:, p].copy(), V[:, q].copy()
        V[:, p] = c * vp - s * vq
        V[:, q] = c * vq + s * vp
        step -= 1
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return np.array([A[0, 0], A[1, 1], A[2, 2]]), V


# ----------------------------------------------------------------------------
# geometry helpers
# ----------------------------------------------------------------------------
def stl_vertices(path: str, scale: Sequence[float] = (1, 1, 1)) -> Optional[np.ndarray]:
    """Vertices [k, 3] of a binary (or ASCII) STL; None if the file is absent."""
    if not os.path.isfile(path):
        return None
    with open(path, "rb") as f:
        data = f.read()
    verts = None
    if len(data) >= 84:
        (ntri,) = struct.unpack_from("<I", data, 80)
        if 84 + 50 * ntri == len(data):
            rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                                count=ntri, offset=84)
            verts = rec["v"].reshape(-1, 3).astype(np.float64)
    if verts is None:                     # ASCII fallback
        pts = [ln.split()[1:4] for ln in data.decode("ascii", "ignore").splitlines()
               if ln.strip().startswith("vertex")]
        if not pts:
            return None
        verts = np.array(pts, dtype=np.float64)
    return verts * np.asarray(scale, dtype=np.float64)


def stl_aabb(path: str, scale: Sequence[float] = (1, 1, 1)) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """Axis-aligned bounds of a binary (or ASCII) STL; None if the file is absent."""
    verts = stl_vertices(path, scale)
    return None if verts is None else (verts.min(0), verts.max(0))


GROUND_TOL = 1.5e-3       # support gap the ground-point selection aims at [m]
BODY_MAX_PRIMITIVES = 64  # all primitives of one body: one 64-lane trip of the kernel's contact detection
GROUND_MAX_POINTS = 56    # per collision element (one body's ground points must fit one 64-lane pass of the kernel)


def _fixed_directions(n: int, seed: int) -> np.ndarray:
    d = np.random.default_rng(seed).normal(size=(n, 3))
    return d / np.linalg.norm(d, axis=1)[:, None]


def hull_support_points(verts: np.ndarray, tol: float = GROUND_TOL, max_points: int = GROUND_MAX_POINTS) -> List[np.ndarray]:
    """Vertices of the convex hull of ``verts`` that stand in for the hull against a PLANE: a plane with normal u touches
    a hull at its support vertex along -u, so what matters is that for every direction some kept vertex comes within
    ``tol`` of the hull's support.  First the support vertices along the 26 directions of the bounding box (faces, edges,
    corners; scaled by the inverse half extents so that a slab still yields its corners), then greedily the support vertex
    of the worst of 4000 fixed directions until the gap is <= tol (or ``max_points`` are kept).  The support vertex of a
    hull is a vertex of the point set, so no hull needs to be constructed; duplicates are dropped."""
    verts = np.unique(np.asarray(verts, float), axis=0)
    lo, hi = verts.min(0), verts.max(0)
    h = np.maximum(0.5 * (hi - lo), 1e-9)
    c = 0.5 * (lo + hi)
    idx: List[int] = []
    for sx in (-1.0, 0.0, 1.0):
        for sy in (-1.0, 0.0, 1.0):
            for sz in (-1.0, 0.0, 1.0):
                if sx == sy == sz == 0.0:
                    continue
                k = int(np.argmax((verts - c) @ (np.array([sx, sy, sz]) / h)))
                if k not in idx:
                    idx.append(k)
    D = _fixed_directions(4000, 1)
    hull_sup = (verts @ D.T).max(0)
    while len(idx) < max_points:
        gap = hull_sup - (verts[idx] @ D.T).max(0)
        j = int(gap.argmax())
        if gap[j] <= tol:
            break
        k = int(np.argmax(verts @ D[j]))
        if k in idx:
            break
        idx.append(k)
    return [verts[k].copy() for k in idx]


def support_gap(verts: np.ndarray, kept, n_dirs: int = 1000, seed: int = 0) -> float:
    """Largest (hull support - best kept vertex) over ``n_dirs`` random plane normals [m]: how far a plane can sink into
    the hull before it meets a kept vertex (tests/test_loader_and_layout.py)."""
    D = _fixed_directions(n_dirs, seed)
    return float(((np.asarray(verts, float) @ D.T).max(0) - (np.asarray(kept, float) @ D.T).max(0)).max())


def fit_capsule(verts: np.ndarray):
    """Sphere-swept segment fitted to the convex hull of ``verts`` for link-link contact: axis = the principal axis of the
    hull's vertices through the hull centroid, extent = the hull's extent along it, radius = that of the capsule with the
    hull's VOLUME and that extent (a sphere of the hull's volume if the hull is not elongated).
    Returns (centre, half axis vector, radius)."""
    from scipy.spatial import ConvexHull
    verts = np.asarray(verts, float)
    hull = ConvexHull(verts)
    hv = verts[hull.vertices]
    vol = float(hull.volume)
    # volume centroid from the hull's triangles (fan of tetrahedra from an interior point)
    o = hv.mean(0)
    tri = verts[hull.simplices]
    tv = np.abs(np.einsum("ij,ij->i", tri[:, 0] - o, np.cross(tri[:, 1] - o, tri[:, 2] - o))) / 6.0
    cen = ((tri.sum(1) + o) / 4.0 * tv[:, None]).sum(0) / tv.sum()
    w, V = np.linalg.eigh(np.cov((hv - cen).T))
    ax = V[:, int(np.argmax(w))]
    t = (hv - cen) @ ax
    tmin, tmax = float(t.min()), float(t.max())
    L = tmax - tmin
    r_sphere = (3.0 * vol / (4.0 * math.pi)) ** (1.0 / 3.0)
    if L <= 2.0 * r_sphere * 1.05:                       # not elongated: a sphere of equal volume at the centroid
        return cen, np.zeros(3), r_sphere
    # pi r^2 (L - 2 r) + 4/3 pi r^3 = vol, monotone in r on (0, L / 2): bisection
    lo_r, hi_r = 0.0, 0.5 * L
    for _ in range(60):
        r = 0.5 * (lo_r + hi_r)
        if math.pi * r * r * (L - 2.0 * r) + 4.0 / 3.0 * math.pi * r ** 3 < vol:
            lo_r = r
        else:
            hi_r = r
    r = 0.5 * (lo_r + hi_r)
    mid = cen + ax * 0.5 * (tmin + tmax)
    return mid, ax * (0.5 * L - r), r


def _box_vertices(lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    return np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], float)


def _element_primitives(ground_points, hull_verts):
    """Primitives of one collision element: its ground points (radius 0) and ONE sphere-swept segment fitted to its hull
    for link-link contact.  Each entry: (position, radius, flags, half axis)."""
    out = [(np.asarray(p, float), 0.0, SPHERE_GROUND, np.zeros(3)) for p in ground_points]
    c, half, r = fit_capsule(hull_verts)
    if r > 1e-4:
        out.append((c, float(r), SPHERE_SELF, half))
    return out


def _floats(s: Optional[str], n: int, default: float = 0.0) -> List[float]:
    if s is None:
        return [default] * n
    v = [float(x) for x in s.split()]
    assert len(v) == n, s
    return v


# ----------------------------------------------------------------------------
# the loader
# ----------------------------------------------------------------------------
def load_urdf(path: str,
              base_position: Sequence[float] = (0, 0, 0),
              base_orientation: Sequence[float] = (0, 0, 0, 1),
              self_collision: bool = False,
              self_exclude: str = "all_parents",
              default_max_force: float = 100000.0,
              default_max_vel: float = BIG,
              mesh_aabbs: Optional[Dict[str, Tuple[Sequence[float], Sequence[float]]]] = None,
              q_init: Optional[Dict[str, float]] = None,
              name: Optional[str] = None) -> FlatModel:
    """Parse ``path`` and flatten it.

    ``base_position`` / ``base_orientation`` place the root LINK frame, as the
    ``basePosition`` / ``baseOrientation`` arguments of ``loadURDF`` do
    (dancer_env.py:59).  ``self_collision`` builds the link-link pair table; ``self_exclude`` says which pairs never
    collide: "all_parents" = a link and any of its ancestors (URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS, what the
    reference passes for Dancer, dancer_env.py:60), "parent" = a link and its direct parent only (plain
    URDF_USE_SELF_COLLISION, what qibullet is recalled to pass [UPSTREAM-RECALL]).  ``mesh_aabbs`` supplies link-frame bounds for mesh
    geometry whose files are not on disk.
    """
    root = ET.parse(path).getroot()
    urdf_dir = os.path.dirname(os.path.abspath(path))

    links: Dict[str, dict] = {}
    for el in root.findall("link"):
        ln = el.get("name")
        inertial = None
        ie = el.find("inertial")
        if ie is not None:
            o = ie.find("origin")
            xyz = _floats(o.get("xyz") if o is not None else None, 3)
            rpy = _floats(o.get("rpy") if o is not None else None, 3)
            mass = float(ie.find("mass").get("value"))
            t = ie.find("inertia")
            ixx, ixy, ixz, iyy, iyz, izz = (float(t.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz"))
            inertial = dict(xyz=np.array(xyz), rpy=rpy, mass=mass,
                            I=np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]))
        cols = []
        for ce in el.findall("collision"):
            o = ce.find("origin")
            xyz = np.array(_floats(o.get("xyz") if o is not None else None, 3))
            rpy = _floats(o.get("rpy") if o is not None else None, 3)
            g = ce.find("geometry")
            cols.append((xyz, rpy_to_mat(rpy), g))
        links[ln] = dict(inertial=inertial, collisions=cols)

    joints = []
    for el in root.findall("joint"):
        o = el.find("origin")
        ax = el.find("axis")
        lim = el.find("limit")
        dyn = el.find("dynamics")
        mim = el.find("mimic")
        joints.append(dict(
            mimic=(mim.get("joint"), float(mim.get("multiplier", 1.0)), float(mim.get("offset", 0.0))) if mim is not None else None,
            **dict(
            name=el.get("name"), type=el.get("type"),
            parent=el.find("parent").get("link"), child=el.find("child").get("link"),
            xyz=np.array(_floats(o.get("xyz") if o is not None else None, 3)),
            rpy=_floats(o.get("rpy") if o is not None else None, 3),
            axis=np.array(_floats(ax.get("xyz") if ax is not None else "1 0 0", 3)),
            lower=float(lim.get("lower")) if lim is not None and lim.get("lower") is not None else None,
            upper=float(lim.get("upper")) if lim is not None and lim.get("upper") is not None else None,
            effort=float(lim.get("effort")) if lim is not None and lim.get("effort") is not None else None,
            velocity=float(lim.get("velocity")) if lim is not None and lim.get("velocity") is not None else None,
            damping=float(dyn.get("damping", 0.0)) if dyn is not None else 0.0)))

    children = {j["child"] for j in joints}
    roots = [ln for ln in links if ln not in children]
    if len(roots) != 1:
        raise ValueError(f"expected exactly one root link, found {roots}")
    base_link = roots[0]

    # depth-first numbering, child joints in declaration order
    by_parent: Dict[str, List[dict]] = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)
    ordered: List[dict] = []

    def visit(ln: str) -> None:
        for j in by_parent.get(ln, []):
            ordered.append(j)
            visit(j["child"])
    visit(base_link)

    # bodies: carrier link + everything welded to it through fixed joints
    # link -> (body id, R, t) with p_carrier = R p_link + t ; body id -1 = base
    link_body: Dict[str, Tuple[int, np.ndarray, np.ndarray]] = {base_link: (-1, np.eye(3), np.zeros(3))}
    joint_dof: List[int] = []
    dof_joint: List[int] = []
    for ji, j in enumerate(ordered):
        Rj, tj = rpy_to_mat(j["rpy"]), j["xyz"]
        pb, Rp, tp = link_body[j["parent"]]
        if j["type"] == "fixed":
            link_body[j["child"]] = (pb, Rp @ Rj, Rp @ tj + tp)
            joint_dof.append(-1)
        elif j["type"] in ("revolute", "continuous"):
            k = len(dof_joint)
            dof_joint.append(ji)
            joint_dof.append(k)
            link_body[j["child"]] = (k, np.eye(3), np.zeros(3))
        else:
            raise ValueError(f"joint {j['name']}: type {j['type']!r} not supported (revolute/continuous/fixed)")
    n = len(dof_joint)

    # lumped inertial of each body in its carrier link frame
    def lump(body: int):
        m_tot, first = 0.0, np.zeros(3)
        parts = []
        for ln, (b, R, t) in link_body.items():
            ine = links[ln]["inertial"]
            if b != body or ine is None or ine["mass"] <= 0.0:
                continue
            Ri = R @ rpy_to_mat(ine["rpy"])
            c = R @ ine["xyz"] + t
            parts.append((ine["mass"], c, Ri, ine["I"]))
            m_tot += ine["mass"]
            first += ine["mass"] * c
        if m_tot <= 0.0:
            raise ValueError(f"body {body} has no mass")
        com = first / m_tot
        if len(parts) == 1:
            m, c, Ri, I = parts[0]
            offd = abs(I[0, 1]) + abs(I[0, 2]) + abs(I[1, 2])
            if offd == 0.0:
                return m_tot, com, np.array([I[0, 0], I[1, 1], I[2, 2]]), Ri
            mom, V = principal_axes(I)
            return m_tot, com, mom, Ri @ V
        Itot = np.zeros((3, 3))
        for m, c, Ri, I in parts:
            r = c - com
            Itot += Ri @ I @ Ri.T + m * (np.dot(r, r) * np.eye(3) - np.outer(r, r))
        mom, V = principal_axes(Itot)
        return m_tot, com, mom, V

    body_mass = np.zeros(n + 1)
    body_com = np.zeros((n + 1, 3))
    body_mom = np.zeros((n + 1, 3))
    body_RL = np.zeros((n + 1, 3, 3))       # internal -> carrier link frame
    for b in range(-1, n):
        body_mass[b + 1], body_com[b + 1], body_mom[b + 1], body_RL[b + 1] = lump(b)

    parent = np.zeros(n, np.int32)
    R0 = np.zeros((n, 3, 3))
    axis = np.zeros((n, 3))
    dvec = np.zeros((n, 3))
    evec = np.zeros((n, 3))
    damping = np.zeros(n)
    lower = np.zeros(n)
    upper = np.zeros(n)
    max_force = np.zeros(n)
    max_vel = np.zeros(n)
    for k, ji in enumerate(dof_joint):
        j = ordered[ji]
        pb, Rp, tp = link_body[j["parent"]]          # parent link inside its carrier
        parent[k] = pb
        Rj = Rp @ rpy_to_mat(j["rpy"])               # child link frame -> parent carrier frame
        tj = Rp @ j["xyz"] + tp                      # pivot in parent carrier frame
        RLp, RLc = body_RL[pb + 1], body_RL[k + 1]
        R0[k] = RLc.T @ Rj.T @ RLp
        a = j["axis"] / np.linalg.norm(j["axis"])
        axis[k] = RLc.T @ a
        dvec[k] = RLp.T @ (tj - body_com[pb + 1])
        evec[k] = RLc.T @ body_com[k + 1]
        damping[k] = j["damping"]
        if j["type"] == "continuous" or j["lower"] is None or j["lower"] > j["upper"]:
            # (lower > upper is the common "no limit" convention: Bullet loads such a joint and adds no limit constraint)
            lower[k], upper[k] = -math.pi, math.pi   # dancer_env.py:80-81
        else:
            lower[k], upper[k] = j["lower"], j["upper"]
        max_force[k] = j["effort"] if j["effort"] is not None else default_max_force
        max_vel[k] = j["velocity"] if j["velocity"] is not None else default_max_vel

    frame_pos = np.zeros((n + 1, 3))
    frame_quat = np.zeros((n + 1, 4))
    for b in range(n + 1):
        frame_pos[b] = -body_RL[b].T @ body_com[b]
        frame_quat[b] = mat_to_quat(body_RL[b].T)

    # collision spheres
    s_link, s_flag, s_pos, s_rad, s_half = [], [], [], [], []
    missing_meshes = []
    for ln, (b, R, t) in link_body.items():
        for xyz, Rc, g in links[ln]["collisions"]:
            shapes = []
            if g is None:
                continue
            mesh, box, sph, cyl = g.find("mesh"), g.find("box"), g.find("sphere"), g.find("cylinder")
            if mesh is not None:
                verts = None
                if mesh_aabbs and ln in mesh_aabbs:          # only bounds are known: the box stands in for the hull
                    verts = _box_vertices(np.array(mesh_aabbs[ln][0], float), np.array(mesh_aabbs[ln][1], float))
                    shapes = _element_primitives(verts, verts)
                else:
                    fn = mesh.get("filename").replace("package://", "")
                    verts = stl_vertices(os.path.join(urdf_dir, fn), _floats(mesh.get("scale"), 3, 1.0))
                    if verts is not None:
                        shapes = _element_primitives(hull_support_points(verts), verts)
                if verts is None:
                    missing_meshes.append(ln)
                    continue
            elif box is not None:
                h = 0.5 * np.array(_floats(box.get("size"), 3))
                verts = _box_vertices(-h, h)
                shapes = _element_primitives(verts, verts)
            elif sph is not None:
                shapes = [(np.zeros(3), float(sph.get("radius")), SPHERE_GROUND | SPHERE_SELF, np.zeros(3))]
            elif cyl is not None:
                # a cylinder is taken as the capsule of the same radius and overall length: one segment for link-link
                # contact, its two end spheres (and the middle one of a long cylinder) against the ground
                r, L = float(cyl.get("radius")), float(cyl.get("length"))
                half = max(0.5 * L - r, 0.0)
                if half == 0.0:
                    shapes = [(np.zeros(3), r, SPHERE_GROUND | SPHERE_SELF, np.zeros(3))]
                else:
                    zs = [-half, half] if half < 2 * r else [-half, 0.0, half]
                    shapes = [(np.array([0, 0, z]), r, SPHERE_GROUND, np.zeros(3)) for z in zs]
                    shapes.append((np.zeros(3), r, SPHERE_SELF, np.array([0.0, 0.0, half])))
            for p, r, fl, hv in shapes:
                p_car = R @ (Rc @ p + xyz) + t
                s_link.append(b)
                s_flag.append(fl)
                s_pos.append(body_RL[b + 1].T @ (p_car - body_com[b + 1]))
                s_rad.append(r)
                s_half.append(body_RL[b + 1].T @ (R @ (Rc @ hv)))

    # A body's spheres must be contiguous (the "4 deepest points per body" rule ranks inside one run, and
    # hg_create verifies it).  link_body is in link-DFS order, so a carrier link whose welded (fixed-joint)
    # child is declared after a revolute sibling subtree would otherwise emit two separate runs: group by
    # body with a stable sort, which leaves the order inside each body as declared.
    if s_link:
        order = sorted(range(len(s_link)), key=lambda k: s_link[k])
        s_link = [s_link[k] for k in order]
        s_flag = [s_flag[k] for k in order]
        s_pos = [s_pos[k] for k in order]
        s_rad = [s_rad[k] for k in order]
        s_half = [s_half[k] for k in order]

        # one body's primitives must fit one 64-lane trip of the kernel (hg_sim.hip build_dev_model): a body welded from
        # many meshes keeps the hull support points of the union of its ground points
        keep = np.ones(len(s_link), bool)
        for b in sorted(set(s_link)):
            idx = [k for k in range(len(s_link)) if s_link[k] == b]
            if len(idx) <= BODY_MAX_PRIMITIVES:
                continue
            pts = [k for k in idx if s_flag[k] == SPHERE_GROUND and s_rad[k] == 0.0]
            room = BODY_MAX_PRIMITIVES - (len(idx) - len(pts))
            if room < 8:
                raise ValueError(f"body {b}: {len(idx) - len(pts)} collision primitives besides hull points; at most {BODY_MAX_PRIMITIVES} fit")
            P = np.array([s_pos[k] for k in pts])
            chosen = hull_support_points(P, GROUND_TOL, room)
            chosen_set = {tuple(np.round(c, 12)) for c in chosen}
            for k in pts:
                keep[k] = tuple(np.round(s_pos[k], 12)) in chosen_set
        if not keep.all():
            sel = [k for k in range(len(s_link)) if keep[k]]
            s_link = [s_link[k] for k in sel]; s_flag = [s_flag[k] for k in sel]; s_pos = [s_pos[k] for k in sel]
            s_rad = [s_rad[k] for k in sel]; s_half = [s_half[k] for k in sel]

    # self-collision pairs: different bodies, neither an ancestor of the other
    # (URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS, dancer_env.py:60)
    pair_a, pair_b = [], []
    if self_exclude not in ("all_parents", "parent"):
        raise ValueError(f"self_exclude must be 'all_parents' or 'parent', not {self_exclude!r}")
    if self_collision:
        anc = [set() for _ in range(n)]
        for k in range(n):
            p = parent[k]
            while p >= 0:
                anc[k].add(int(p))
                p = parent[p]
        for a in range(len(s_link)):
            if not (s_flag[a] & SPHERE_SELF) or s_link[a] < 0:
                continue
            for b2 in range(a + 1, len(s_link)):
                if not (s_flag[b2] & SPHERE_SELF) or s_link[b2] < 0:
                    continue
                la, lb = s_link[a], s_link[b2]
                if la == lb:
                    continue
                if self_exclude == "all_parents":
                    if la in anc[lb] or lb in anc[la]:
                        continue
                elif parent[la] == lb or parent[lb] == la:
                    continue
                pair_a.append(a)
                pair_b.append(b2)

    Rwl = quat_to_mat(base_orientation)
    base_pos0 = np.asarray(base_position, float) + Rwl @ body_com[0]
    base_quat0 = mat_to_quat(Rwl @ body_RL[0])

    qi = np.zeros(n)
    if q_init:
        names = [ordered[ji]["name"] for ji in dof_joint]
        for nm, v in q_init.items():
            qi[names.index(nm)] = v

    fm = FlatModel(
        name=name or root.get("name", "robot"), base_link=base_link,
        joint_names_all=[j["name"] for j in ordered],
        link_names_all=[j["child"] for j in ordered],
        joint_types_all=[j["type"] for j in ordered],
        joint_dof=joint_dof, dof_joint=dof_joint,
        parent=parent, R0=R0, axis=axis, d=dvec, e=evec,
        mass=body_mass[1:].copy(), inertia=body_mom[1:].copy(), joint_damping=damping,
        lower=lower, upper=upper, max_force=max_force, max_vel=max_vel, q_init=qi,
        base_mass=float(body_mass[0]), base_inertia=body_mom[0].copy(),
        base_pos0=base_pos0, base_quat0=base_quat0,
        frame_pos=frame_pos, frame_quat=frame_quat,
        sphere_link=np.array(s_link, np.int32), sphere_flags=np.array(s_flag, np.int32),
        sphere_pos=np.array(s_pos, float).reshape(-1, 3), sphere_radius=np.array(s_rad, float),
        sphere_half=np.array(s_half, float).reshape(-1, 3),
        pair_a=np.array(pair_a, np.int32), pair_b=np.array(pair_b, np.int32),
        act_joint=np.arange(n, dtype=np.int32),
        meta={"missing_meshes": missing_meshes,
              # <mimic> joints: child dof name -> [parent joint name, multiplier, offset].  Bullet ignores
              # the tag; qibullet's setAngles drives such joints from their parent (the hands' fingers).
              "mimic": {j["name"]: [j["mimic"][0], j["mimic"][1], j["mimic"][2]]
                        for j in ordered if j["mimic"] is not None and j["type"] != "fixed"},
              "base_position": list(map(float, base_position)),
              "base_orientation": list(map(float, base_orientation))})
    fm.check()
    return fm
