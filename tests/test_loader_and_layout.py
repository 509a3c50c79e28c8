"""Integer half of the contract: joint/link index maps, topology, name lists (bit-exact)."""
import json
import math
import os

import numpy as np
import pytest

from humanoid_gym_amd.model import FlatModel
from humanoid_gym_amd.robots import NAO_JOINTS, PEPPER_JOINTS, load_robot
from humanoid_gym_amd.urdf import load_urdf, principal_axes, quat_to_mat, mat_to_quat, rpy_to_mat
import humanoid_gym_amd as hg

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "dancer_layout.json")))
REF_URDF = "/root/reference/humanoid_gym/envs/assets/dancer/dancer_urdf_model.URDF"


def test_dancer_maps_match_golden():
    m, cfg = load_robot("dancer")
    assert m.joint_names_all == GOLD["joint_names"]
    assert m.link_names_all == GOLD["link_names"]
    assert m.parent.tolist() == GOLD["parent_index"]
    assert m.base_link == GOLD["base_link"]
    assert m.joint2Index == {n: i for i, n in enumerate(GOLD["joint_names"])}     # dancer_env.py:77
    assert m.link2Index == {n: i for i, n in enumerate(GOLD["link_names"])}       # dancer_env.py:79
    assert m.n_links == 18 and m.n_act == 18
    assert abs(m.total_mass() - GOLD["total_mass"]) < 1e-12
    # world parameters: dancer_env.py:63-65
    assert cfg.gravity == (0.0, 0.0, -10.0) and cfg.solver_iters == 150 and abs(cfg.dt - 1 / 240) < 1e-15


def test_dancer_axes_and_reset_pose():
    m, _ = load_robot("dancer")
    # joint axes are stored in the internal (principal-axis) frame: rotate back to the URDF link frame
    for i, ax in enumerate(GOLD["axis"]):
        R_int_to_link = quat_to_mat(m.frame_quat[i + 1]).T
        a = R_int_to_link @ m.axis[i]
        g = np.array(ax) / np.linalg.norm(ax)
        assert np.abs(a - g).max() < 1e-9, i
    # reset pose: whole-body COM and lowest mesh corner computed in the survey
    n = m.n_links
    Rb, xb = quat_to_mat(m.base_quat0), m.base_pos0
    Rw, xw = [None] * n, [None] * n
    for i in range(n):
        p = m.parent[i]
        Rp, xp = (Rb, xb) if p < 0 else (Rw[p], xw[p])
        Rw[i] = Rp @ m.R0[i].T
        xw[i] = xp + Rp @ m.d[i] + Rw[i] @ m.e[i]
    com = (m.base_mass * xb + sum(m.mass[i] * xw[i] for i in range(n))) / m.total_mass()
    assert np.abs(com - np.array(GOLD["reset_com"])).max() < 1e-4
    zs = [((xb if b < 0 else xw[b]) + (Rb if b < 0 else Rw[b]) @ m.sphere_pos[s])[2] - m.sphere_radius[s]
          for s, b in enumerate(m.sphere_link) if m.sphere_flags[s] & 1]      # the points tested against the plane
    assert abs(min(zs) - GOLD["reset_lowest_z"]) < 1e-4
    # base link frame sits where loadURDF was told to put it (dancer_env.py:59)
    Rl = Rb @ quat_to_mat(m.frame_quat[0])
    pl = xb + Rb @ m.frame_pos[0]
    assert np.abs(pl - np.array([-0.3, -0.3, 0.3])).max() < 1e-12
    assert np.abs(Rl - np.array([[-1, 0, 0], [0, 0, -1], [0, -1, 0]])).max() < 1e-6


@pytest.mark.skipif(not os.path.exists(REF_URDF), reason="reference tree not present (GPU box)")
def test_shipped_dancer_json_equals_fresh_flatten():
    fresh = load_urdf(REF_URDF, base_position=[-0.3, -0.3, 0.3], base_orientation=[0, 0.7071068, -0.7071068, 0],
                      self_collision=True, name="dancer")
    shipped, _ = load_robot("dancer")
    for k in FlatModel._ARRAYS:
        np.testing.assert_array_equal(np.asarray(getattr(fresh, k)), np.asarray(getattr(shipped, k)), err_msg=k)
    assert fresh.meta["missing_meshes"] == ["body_chest"]      # .MISSING_LARGE_BLOBS


def test_self_collision_pairs_exclude_all_parents():
    m, _ = load_robot("dancer")
    anc = [set() for _ in range(m.n_links)]
    for k in range(m.n_links):
        p = m.parent[k]
        while p >= 0:
            anc[k].add(int(p)); p = m.parent[p]
    assert len(m.pair_a) > 0
    for a, b in zip(m.pair_a, m.pair_b):
        la, lb = m.sphere_link[a], m.sphere_link[b]
        assert la != lb and la not in anc[lb] and lb not in anc[la] and la >= 0 and lb >= 0


def test_softbank_like_name_orders():
    nao, cfg = load_robot("nao")
    assert NAO_JOINTS == GOLD["nao"] and PEPPER_JOINTS == GOLD["pepper"]
    assert [nao.dof_names[k] for k in nao.act_joint] == GOLD["nao"]               # nao_env.py:30-31
    pep, _ = load_robot("pepper")
    assert [pep.dof_names[k] for k in pep.act_joint] == GOLD["pepper"]            # pepper_env.py:28-39
    rom, _ = load_robot("romeo")
    assert all("Finger" not in n and "Thumb" not in n for n in rom.dof_names)     # romeo_env.py:25
    assert rom.n_act == rom.n_links == 37
    assert cfg.gravity[2] == -9.81 and cfg.solver_iters == 50
    for m in (nao, pep, rom):
        idx = np.asarray(m.act_joint)
        assert np.all(m.lower[idx] <= m.q_init[idx]) and np.all(m.q_init[idx] <= m.upper[idx])
        assert np.all(m.parent < np.arange(m.n_links))


def test_env_ids_match_reference_registration():
    sim_ids = [i for i in GOLD["env_ids"] if i != "nao-real-v0"]                  # hardware env: out of scope
    assert sorted(hg.ENTRY_POINTS) == sorted(sim_ids)


def test_json_roundtrip_and_cstruct():
    m, _ = load_robot("nao")
    m2 = FlatModel.from_json(m.to_json())
    for k in FlatModel._ARRAYS:
        np.testing.assert_array_equal(np.asarray(getattr(m, k)), np.asarray(getattr(m2, k)))
    c = m.to_c()
    assert c.n_links == 26 and c.n_act == 22 and c.parent[0] == -1


def test_rotation_helpers():
    rng = np.random.default_rng(0)
    for _ in range(20):
        R = rpy_to_mat(rng.uniform(-3, 3, 3))
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
        assert np.abs(quat_to_mat(mat_to_quat(R)) - R).max() < 1e-12
        A = rng.normal(size=(3, 3)); I = A @ A.T + np.eye(3)
        mom, V = principal_axes(I)
        assert np.abs(V @ np.diag(mom) @ V.T - I).max() < 1e-5 * np.abs(I).max()
        assert abs(np.linalg.det(V) - 1) < 1e-9
    mom, V = principal_axes(np.diag([1.0, 2.0, 3.0]))
    assert mom.tolist() == [1.0, 2.0, 3.0] and np.array_equal(V, np.eye(3))      # diagonal input untouched


def test_ragged_and_edge_inputs():
    m, _ = load_robot("dancer")
    m.parent = m.parent.copy(); m.parent[3] = 5                                   # child before parent
    with pytest.raises(ValueError):
        m.check()
    with pytest.raises(KeyError):
        load_robot("atlas")


def test_mimic_tags_are_recorded(tmp_path):
    """<mimic> joints (the fingers of the real NAO/Pepper hands) become ordinary dofs, as in Bullet, and the
    loader records (parent, multiplier, offset) so the env can drive them like qibullet's setAngles."""
    from _mimic_urdf import write
    from humanoid_gym_amd.robots import NAO_JOINTS
    m, _ = load_robot("nao", write(str(tmp_path / "nao_fingers.urdf")))
    assert m.n_links == 28 and [m.dof_names[k] for k in m.act_joint] == NAO_JOINTS
    assert m.meta["mimic"] == {"LFinger11": ["LHand", 0.9, 0.05], "RFinger11": ["RHand", 0.9, 0.05]}
    assert m.parent[m.dof_of("LFinger11")] == m.dof_of("LHand")
    plain, _ = load_robot("nao")
    assert plain.meta["mimic"] == {}


def test_hull_support_points_are_vertices_of_the_point_set():
    """Mesh collision geometry: ground points are vertices of the mesh's convex hull chosen so that a plane
    of ANY orientation meets a kept vertex within GROUND_TOL of where it would meet the hull -- members of the vertex
    set, exactly the corners for a box, and strictly inside the bounding box for a rounded body."""
    from humanoid_gym_amd.urdf import hull_support_points, support_gap, GROUND_TOL, GROUND_MAX_POINTS
    rng = np.random.default_rng(5)
    box = np.array([[x, y, z] for x in (-0.1, 0.2) for y in (-0.3, 0.05) for z in (0.0, 0.02)])
    cloud = np.vstack([box, box.mean(0) + 0.3 * (rng.random((200, 3)) - 0.5) * (box.max(0) - box.min(0))])
    pts = hull_support_points(cloud)
    assert len(pts) == 8 and {tuple(p) for p in pts} == {tuple(p) for p in box}
    assert support_gap(cloud, pts) < 1e-12
    ball = rng.normal(size=(500, 3))
    ball = 0.05 * ball / np.linalg.norm(ball, axis=1, keepdims=True) * np.array([1.0, 2.0, 0.5])
    pts = np.array(hull_support_points(ball))
    assert all(any(np.array_equal(p, v) for v in ball) for p in pts) and len(pts) <= GROUND_MAX_POINTS
    lo, hi = ball.min(0), ball.max(0)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    assert np.linalg.norm(pts[:, None] - corners[None], axis=2).min() > 0.01    # an ellipsoid has no corners
    # the greedy rule: more points, smaller gap (fresh directions, not the 4000 the selection used), monotonically
    gaps = [support_gap(ball, hull_support_points(ball, tol=0.0, max_points=k)) for k in (8, 16, 32, 56)]
    assert all(a >= b for a, b in zip(gaps, gaps[1:])) and gaps[-1] < 0.5 * gaps[0]
    octa = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    assert len(hull_support_points(octa)) == 6 and len(hull_support_points(np.vstack([octa, octa]))) == 6   # duplicates are dropped


def test_capsule_fit_and_segment_distance():
    """Link-link geometry: one sphere-swept segment per collision element, fitted to the element's convex hull (axis =
    principal axis through the volume centroid, extent = the hull's, radius = equal volume); a box 2 x 2 x 10 gives
    the capsule along z, a cube gives the sphere of its volume."""
    from humanoid_gym_amd.urdf import fit_capsule, _box_vertices
    c, h, r = fit_capsule(_box_vertices(np.array([-1.0, -1.0, -5.0]) + 3.0, np.array([1.0, 1.0, 5.0]) + 3.0))
    assert np.abs(c - 3.0).max() < 1e-9 and np.abs(np.abs(h) - [0, 0, 5.0 - r]).max() < 1e-9
    vol = math.pi * r * r * 2 * np.linalg.norm(h) + 4.0 / 3.0 * math.pi * r ** 3
    assert abs(vol - 40.0) < 1e-6 and 1.0 < r < math.sqrt(2.0)                   # between the inscribed and the circumscribed radius
    c, h, r = fit_capsule(_box_vertices(-np.ones(3), np.ones(3)))
    assert np.abs(h).max() == 0.0 and abs(4.0 / 3.0 * math.pi * r ** 3 - 8.0) < 1e-9


def test_dancer_ground_points_come_from_the_reference_meshes():
    """With the reference tree present: every ground point of the shipped Dancer model is a vertex of that link's STL
    (in the URDF link frame), and a plane of any orientation (1000 random normals per link) meets a kept vertex within
    2 mm of where it would meet the link's convex hull -- against 5.8 - 29.5 mm for the 8 bounding-box diagonals the
    model used before (the figures are printed by tools/geometry_report.py).  Always: at most 56
    points per link, contiguous per body, one link-link primitive per link."""
    import os
    from humanoid_gym_amd.urdf import stl_vertices, support_gap, GROUND_MAX_POINTS
    m, _ = load_robot("dancer")
    ground = [s for s in range(len(m.sphere_link)) if m.sphere_flags[s] & 1]
    assert all(m.sphere_radius[s] == 0.0 and np.abs(m.sphere_half[s]).max() == 0.0 for s in ground)
    per_link = {}
    for s in ground:
        per_link.setdefault(int(m.sphere_link[s]), []).append(s)
    assert len(per_link) == 18 and all(8 <= len(v) <= GROUND_MAX_POINTS for v in per_link.values())
    selfp = [s for s in range(len(m.sphere_link)) if m.sphere_flags[s] & 2]
    assert sorted(int(m.sphere_link[s]) for s in selfp) == sorted(per_link) and all(m.sphere_radius[s] > 0.005 for s in selfp)
    mesh_dir = "/root/reference/humanoid_gym/envs/assets/dancer/meshes"
    if not os.path.isdir(mesh_dir):
        pytest.skip("reference meshes not on this machine")
    worst = 0.0
    for b, idx in per_link.items():
        verts = stl_vertices(os.path.join(mesh_dir, m.link_names_all[b] + ".STL"))
        assert verts is not None, m.link_names_all[b]
        # internal (COM, principal axes) frame -> URDF link frame
        R = quat_to_mat(m.frame_quat[b + 1])
        kept = np.array([R.T @ (m.sphere_pos[s] - m.frame_pos[b + 1]) for s in idx])
        for p_link in kept:
            assert np.abs(verts - p_link).sum(1).min() < 1e-6, m.link_names_all[b]
        gap = support_gap(verts, kept, n_dirs=1000, seed=123 + b)
        assert gap <= 2e-3, (m.link_names_all[b], gap)
        worst = max(worst, gap)
    assert worst > 0.0


def test_welded_child_after_revolute_sibling_keeps_a_bodys_spheres_contiguous(tmp_path):
    """A carrier link with collision geometry, a revolute child subtree, and THEN a fixed-joint child that
    carries more collision geometry (real SoftBank URDFs are full of welded links): the carrier's spheres
    come from two links visited at different times, and must still form one run per body -- hg_create
    rejects a model otherwise ('collision spheres of one body must be contiguous')."""
    import ctypes as C
    from humanoid_gym_amd import _capi
    from humanoid_gym_amd.model import SimConfig
    from humanoid_gym_amd.urdf import load_urdf

    def link(name, box=None):
        col = f'<collision><geometry><box size="{box} {box} {box}"/></geometry></collision>' if box else ""
        return (f'<link name="{name}"><inertial><mass value="0.2"/><inertia ixx="1e-3" ixy="0" ixz="0" iyy="1e-3" '
                f'iyz="0" izz="1e-3"/></inertial>{col}</link>')

    def joint(name, kind, parent, child, xyz):
        lim = '<limit lower="-1" upper="1" effort="5" velocity="5"/>' if kind == "revolute" else ""
        return (f'<joint name="{name}" type="{kind}"><parent link="{parent}"/><child link="{child}"/>'
                f'<origin xyz="{xyz}"/><axis xyz="0 1 0"/>{lim}</joint>')

    text = "<robot name=\"welded\">" + "".join([
        link("base", 0.1), link("arm", 0.04), link("hand", 0.03), link("lid", 0.05), link("tip", 0.02),
        joint("j_arm", "revolute", "base", "arm", "0 0.1 0"),
        joint("j_hand", "revolute", "arm", "hand", "0 0.1 0"),
        joint("w_lid", "fixed", "base", "lid", "0 0 0.1"),            # welded to the base, declared after the arm subtree
        joint("w_tip", "fixed", "arm", "tip", "0 0.05 0.05"),         # welded to the arm, declared after its revolute child
    ]) + "</robot>"
    path = tmp_path / "welded.urdf"
    path.write_text(text)
    m = load_urdf(str(path), self_collision=True, name="welded")
    assert m.n_links == 2 and m.joint_names_all == ["j_arm", "j_hand", "w_tip", "w_lid"]      # DFS, declaration order
    sl = m.sphere_link.tolist()
    runs = [b for k, b in enumerate(sl) if k == 0 or sl[k - 1] != b]
    assert len(runs) == len(set(runs)) and set(sl) == {-1, 0, 1}, sl                         # one run per body
    assert sl.count(-1) > 8 and sl.count(0) > 8                                               # base+lid, arm+tip merged
    assert all(m.sphere_link[a] != m.sphere_link[b] for a, b in zip(m.pair_a, m.pair_b))
    L = _capi.load_library()
    h = C.c_void_p()
    cm, cc = m.to_c(), SimConfig(enable_self=True).to_c()
    rc = L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h))
    assert rc in (0, -2), (rc, L.hg_last_error())           # accepted: 0 on a GPU box, "no device" here -- never HG_EINVAL
    if rc == 0:
        L.hg_destroy(h)


def test_a_body_welded_from_many_boxes_keeps_at_most_64_primitives(tmp_path):
    """One body's primitives share one 64-lane trip of the kernel's contact detection (hg_create refuses more).  A carrier
    with nine welded boxes would have 80 hull points + 10 capsules: the loader keeps the hull support points of the UNION
    (what a plane can touch of the whole body), every capsule, and hg_create accepts the model."""
    import ctypes as C
    from humanoid_gym_amd import _capi
    from humanoid_gym_amd.model import SimConfig
    from humanoid_gym_amd.urdf import load_urdf, BODY_MAX_PRIMITIVES

    def link(name, box):
        return (f'<link name="{name}"><inertial><mass value="0.2"/><inertia ixx="1e-3" ixy="0" ixz="0" iyy="1e-3" iyz="0" izz="1e-3"/>'
                f'</inertial><collision><geometry><box size="{box}"/></geometry></collision></link>')

    parts = [link("base", "0.1 0.1 0.1"), link("arm", "0.02 0.02 0.2")]
    joints = ['<joint name="j" type="revolute"><parent link="base"/><child link="arm"/><origin xyz="0 0 0.2"/><axis xyz="0 1 0"/>'
              '<limit lower="-1" upper="1" effort="5" velocity="5"/></joint>']
    for k in range(9):
        parts.append(link(f"plate{k}", "0.05 0.03 0.02"))
        joints.append(f'<joint name="w{k}" type="fixed"><parent link="base"/><child link="plate{k}"/>'
                      f'<origin xyz="{0.08 * math.cos(k)} {0.08 * math.sin(k)} {0.02 * k - 0.08}" rpy="0 0 {0.7 * k}"/></joint>')
    path = tmp_path / "plates.urdf"
    path.write_text('<robot name="plates">' + "".join(parts + joints) + "</robot>")
    m = load_urdf(str(path), self_collision=True, name="plates")
    on_base = [k for k in range(len(m.sphere_link)) if m.sphere_link[k] == -1]
    assert len(on_base) <= BODY_MAX_PRIMITIVES
    assert sum(1 for k in on_base if m.sphere_flags[k] & 2) == 10                       # every capsule kept
    pts = np.array([m.sphere_pos[k] for k in on_base if m.sphere_flags[k] & 1])
    assert 26 <= len(pts) <= BODY_MAX_PRIMITIVES - 10
    L = _capi.load_library()
    h = C.c_void_p()
    cm, cc = m.to_c(), SimConfig(enable_self=True).to_c()
    rc = L.hg_create(C.byref(cm), C.byref(cc), 2, 0, C.byref(h))
    assert rc in (0, -2), (rc, L.hg_last_error())
    if rc == 0:
        L.hg_destroy(h)
    # and the check it protects from: 65 primitives on one body are refused with a message, not run
    extra = BODY_MAX_PRIMITIVES + 1 - len(on_base)
    first = on_base[0]
    crowded = load_urdf(str(path), self_collision=True, name="plates")      # ... with copies of a base point appended to the base's run
    k_end = max(on_base) + 1
    for name in ("sphere_link", "sphere_flags", "sphere_radius"):
        arr = np.asarray(getattr(crowded, name))
        setattr(crowded, name, np.concatenate([arr[:k_end], np.repeat(arr[first:first + 1], extra, 0), arr[k_end:]]))
    for name in ("sphere_pos", "sphere_half"):
        arr = np.asarray(getattr(crowded, name))
        setattr(crowded, name, np.concatenate([arr[:k_end], np.repeat(arr[first:first + 1], extra, 0), arr[k_end:]]))
    crowded.pair_a = np.where(np.asarray(crowded.pair_a) >= k_end, np.asarray(crowded.pair_a) + extra, crowded.pair_a)
    crowded.pair_b = np.where(np.asarray(crowded.pair_b) >= k_end, np.asarray(crowded.pair_b) + extra, crowded.pair_b)
    rc = L.hg_create(C.byref(crowded.to_c()), C.byref(cc), 2, 0, C.byref(h))
    assert rc == -4 and b"more than 64 collision primitives on one body" in L.hg_last_error(), (rc, L.hg_last_error())


def test_hands_variants_of_nao_and_romeo():
    """romeo_env.py:24-26 filters Finger / Thumb joints out of the CONTROLLED set: they are simulated dofs of the
    reference's robots.  hands=True / "<robot>_hands" load the stand-ins that have them; the env-facing joint lists
    are unchanged, the fingers mimic LHand / RHand and start where the Stand posture puts the hands."""
    import ctypes as C
    from humanoid_gym_amd import _capi
    nao, cfg = load_robot("nao_hands")
    plain, _ = load_robot("nao")
    assert nao.n_links == 42 and plain.n_links == 26
    assert [nao.dof_names[k] for k in nao.act_joint] == NAO_JOINTS                     # nao_env.py:30-31
    fingers = [n for n in nao.dof_names if "Finger" in n or "Thumb" in n]
    assert len(fingers) == 16 and {"LFinger11", "LFinger23", "RThumb2"} <= set(fingers)
    assert set(nao.meta["mimic"]) == set(fingers) and all(v[0] in ("LHand", "RHand") for v in nao.meta["mimic"].values())
    assert abs(nao.q_init[nao.dof_of("LFinger11")] - 0.95 * nao.q_init[nao.dof_of("LHand")]) < 1e-12
    assert nao.n_links + 3 * cfg.max_contacts == 66                                    # more rows than lanes
    rom, _ = load_robot("romeo", hands=True)
    assert rom.n_links == 61 and rom.n_act == 37
    assert not any("Finger" in rom.dof_names[k] or "Thumb" in rom.dof_names[k] for k in rom.act_joint)
    assert sum(1 for n in rom.dof_names if "Finger" in n or "Thumb" in n) == 24
    with pytest.raises(KeyError):
        load_robot("pepper", hands=True)
    # the library accepts them on the host (a kernel variant exists); on this box it then finds no device
    L = _capi.load_library()
    h = C.c_void_p()
    for m in (nao, rom):
        cm, cc = m.to_c(), cfg.to_c()
        assert L.hg_create(C.byref(cm), C.byref(cc), 4, 0, C.byref(h)) in (0, -2), L.hg_last_error()
        if h.value:
            L.hg_destroy(h)
    cc = cfg.to_c(); cc.max_contacts = 12                                              # catch-all variant: 61 dofs, 12 contact slots
    cm = rom.to_c()
    assert L.hg_create(C.byref(cm), C.byref(cc), 4, 0, C.byref(h)) in (0, -2), L.hg_last_error()
    if h.value:
        L.hg_destroy(h)
    cm.n_links = 63                                                                    # beyond HG_MAX_LINKS
    assert L.hg_create(C.byref(cm), C.byref(cc), 4, 0, C.byref(h)) == -1 and b"n_links" in L.hg_last_error()
