"""Collision-geometry report for the Dancer meshes (needs the reference's STL files; build container only).

Ground: for every link, the largest distance a plane of arbitrary orientation (1000 random normals) can sink into the
convex hull before it meets one of the kept vertices -- for the 8 bounding-box diagonals (rounds 1-2), for 26 fixed
directions, and for the shipped greedy set (humanoid_gym_amd/urdf.py hull_support_points).
Link-link: the fitted sphere-swept segment against the hull -- how far hull vertices stick out of the capsule, and how
far the capsule sticks out of the hull (sampled on the capsule surface), both in mm.

usage: python tools/geometry_report.py [/root/reference] > profiles/round3_geometry_report.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import ConvexHull
from humanoid_gym_amd.urdf import stl_vertices, hull_support_points, support_gap, fit_capsule, GROUND_TOL

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
mesh_dir = os.path.join(ref, "humanoid_gym/envs/assets/dancer/meshes")


def directions_set(verts, dirs):
    c, h = 0.5 * (verts.min(0) + verts.max(0)), np.maximum(0.5 * (verts.max(0) - verts.min(0)), 1e-9)
    return np.unique([verts[int(np.argmax((verts - c) @ (np.asarray(d, float) / h)))] for d in dirs], axis=0)


def capsule_fit(verts):
    c, h, r = fit_capsule(verts)
    hull = ConvexHull(verts)
    hv = verts[hull.vertices]
    L2 = float(h @ h)
    t = np.clip((hv - c) @ h / L2, -1, 1) if L2 > 0 else np.zeros(len(hv))
    out_of_capsule = np.linalg.norm(hv - (c + t[:, None] * h), axis=1) - r          # > 0: hull vertex outside the capsule
    # capsule surface samples: points at distance r from the segment
    rng = np.random.default_rng(0)
    u = rng.normal(size=(4000, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    s = rng.uniform(-1, 1, 4000)
    if L2 > 0:
        ax = h / np.sqrt(L2)
        side = u - np.outer(u @ ax, ax); side /= np.linalg.norm(side, axis=1)[:, None]
        P = np.vstack([c + s[:, None] * h + r * side, c + h + r * u[u @ ax > 0], c - h + r * u[u @ ax < 0]])
    else:
        P = c + r * u
    A, b = hull.equations[:, :3], hull.equations[:, 3]
    out_of_hull = (P @ A.T + b).max(1)                                                # > 0: capsule point outside the hull
    return r, np.sqrt(L2), out_of_capsule.max(), out_of_hull.max()


DIAG8 = [(x, y, z) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]
DIR26 = [(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if (x, y, z) != (0, 0, 0)]
print(f"{'link':18s} {'hull':>5s} | plane gap over 1000 normals [mm]: {'8 diag':>7s} {'26 dir':>7s} {'shipped':>8s} (points) |"
      f" capsule r, half [mm] {'hull out':>9s} {'caps out':>9s}")
worst = [0.0, 0.0, 0.0]
for fn in sorted(os.listdir(mesh_dir)):
    if not fn.upper().endswith(".STL"):
        continue
    verts = stl_vertices(os.path.join(mesh_dir, fn))
    if verts is None:
        print(f"{fn[:-4]:18s} (mesh not in the tree)")
        continue
    verts = np.unique(verts, axis=0)
    g8 = support_gap(verts, directions_set(verts, DIAG8))
    g26 = support_gap(verts, directions_set(verts, DIR26))
    kept = hull_support_points(verts)
    gs = support_gap(verts, kept)
    r, hl, oc, oh = capsule_fit(verts)
    worst = [max(worst[0], g8), max(worst[1], g26), max(worst[2], gs)]
    print(f"{fn[:-4]:18s} {len(ConvexHull(verts).vertices):5d} | {1e3 * g8:39.1f} {1e3 * g26:7.1f} {1e3 * gs:8.2f} ({len(kept):3d})    |"
          f" {1e3 * r:9.1f} {1e3 * hl:5.1f} {1e3 * oc:9.1f} {1e3 * oh:9.1f}")
print(f"worst plane gap: 8 diagonals {1e3 * worst[0]:.1f} mm, 26 directions {1e3 * worst[1]:.1f} mm, shipped {1e3 * worst[2]:.2f} mm "
      f"(selection tolerance {1e3 * GROUND_TOL:.1f} mm on 4000 other directions)")
