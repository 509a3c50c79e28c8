"""Flattens the reference's Dancer URDF (+ hull support vertices and bounding boxes of the STLs) into the JSON
model shipped with the package, so that GPU boxes without /root/reference can
simulate it.  Only derived numbers are written, never the URDF text.

    python tools/flatten_dancer.py [/root/reference]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from humanoid_gym_amd.urdf import load_urdf  # noqa: E402

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
urdf = os.path.join(ref, "humanoid_gym/envs/assets/dancer/dancer_urdf_model.URDF")
# spawn pose and flags: /root/reference/humanoid_gym/envs/dancer_env.py:58-60
m = load_urdf(urdf, base_position=[-0.3, -0.3, 0.3], base_orientation=[0, 0.7071068, -0.7071068, 0],
              self_collision=True, default_max_force=100000.0, name="dancer")
m.meta["source"] = "humanoid_gym/envs/assets/dancer/dancer_urdf_model.URDF + meshes/*.STL (hull support vertices, bounding boxes)"
out = os.path.join(ROOT, "humanoid_gym_amd", "robots", "dancer.json")
with open(out, "w") as f:
    f.write(m.to_json())
print(out, m.n_links, "dofs,", len(m.sphere_link), "spheres,", len(m.pair_a), "pairs, mass", m.total_mass())
