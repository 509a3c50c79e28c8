"""Random articulated trees as URDF text, for the generic-kernel parity tests (not a robot of the
reference: the point is to drive every kernel variant -- row counts above 32, depth 12, nested
branching -- through the same oracle comparison as the four robots)."""
import numpy as np


def random_tree_urdf(n_links, max_depth, n_chains, branch_at=(), seed=0, name="tree"):
    """n_chains chains hang off the base; `branch_at` lists (chain, depth) positions where an extra
    side chain forks off (nesting them exercises the saved-acceleration slots)."""
    rng = np.random.default_rng(seed)
    out = [f'<robot name="{name}">']

    def link(ln, half, collide=True):
        m = float(rng.uniform(0.05, 0.4))
        ixx, iyy = (float(m * rng.uniform(0.5, 2.0) * 1e-3) for _ in range(2))
        izz = float(rng.uniform(abs(ixx - iyy) + 1e-5, ixx + iyy - 1e-6))
        com = rng.uniform(-0.02, 0.02, 3)
        out.append(f'<link name="{ln}"><inertial><origin xyz="{com[0]:.4f} {com[1]:.4f} {com[2]:.4f}" '
                   f'rpy="{rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f}"/>'
                   f'<mass value="{m:.4f}"/><inertia ixx="{ixx:.3e}" ixy="0" ixz="0" iyy="{iyy:.3e}" iyz="0" izz="{izz:.3e}"/>'
                   f'</inertial>' + (f'<collision><geometry><box size="{2 * half[0]:.3f} {2 * half[1]:.3f} {2 * half[2]:.3f}"/>'
                                     f'</geometry></collision>' if collide else '') + '</link>')

    def joint(jn, parent, child):
        xyz = rng.uniform(-0.06, 0.06, 3)
        xyz[2] = -abs(xyz[2]) - 0.03
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        out.append(f'<joint name="{jn}" type="revolute"><parent link="{parent}"/><child link="{child}"/>'
                   f'<origin xyz="{xyz[0]:.4f} {xyz[1]:.4f} {xyz[2]:.4f}" rpy="{rng.uniform(-1, 1):.3f} {rng.uniform(-1, 1):.3f} 0"/>'
                   f'<axis xyz="{ax[0]:.4f} {ax[1]:.4f} {ax[2]:.4f}"/><limit lower="-2.0" upper="2.0" effort="5.0" velocity="20"/>'
                   f'<dynamics damping="{rng.uniform(0, 0.02):.4f}"/></joint>')

    link("base", (0.06, 0.05, 0.04))
    made = 0
    # plan: (parent link name, depth of the new link) work list, chains first
    chains = [["base"] for _ in range(n_chains)]
    k = 0
    forks = {}
    while made < n_links:
        progressed = False
        for c in range(len(chains)):
            if made >= n_links:
                break
            depth = len(chains[c]) - 1 + forks.get(c, 0)
            if depth >= max_depth:
                continue
            ln = f"l{made}"
            link(ln, tuple(rng.uniform(0.01, 0.03, 3)), collide=(made % 4 == 0))     # HG_MAX_SPHERES bounds the total
            joint(f"j{made}", chains[c][-1], ln)
            chains[c].append(ln)
            made += 1
            progressed = True
            for (bc, bd) in branch_at:
                if bc == c and len(chains[c]) - 1 == bd and (bc, bd) not in forks.values():
                    chains.append([ln])                       # a side chain forks off this link
                    forks[len(chains) - 1] = bd
        if not progressed:
            break
        k += 1
    out.append("</robot>")
    return "\n".join(out), made


def write_random_tree(path, **kw):
    text, made = random_tree_urdf(**kw)
    with open(path, "w") as f:
        f.write(text)
    return path, made


def degenerate_pair_urdf():
    """Two children of the base whose collision boxes are centred on their joint pivots and overlap: the inscribed
    spheres touch for ever and NO coordinate of the robot moves one centre against the other -- a contact row whose
    Jacobian vanishes identically (found by tools/fuzz_trees.py; see HG_ROW_DEN_EPS in include/hg_model.h)."""
    def link(name, m, box):
        return (f'<link name="{name}"><inertial><origin xyz="0.004 -0.003 0.01" rpy="0.1 0.2 0.3"/><mass value="{m}"/>'
                f'<inertia ixx="3e-4" ixy="0" ixz="0" iyy="4e-4" iyz="0" izz="5e-4"/></inertial>'
                f'<collision><geometry><box size="{box} {box} {box}"/></geometry></collision></link>')

    def joint(name, parent, child, xyz, axis):
        return (f'<joint name="{name}" type="revolute"><parent link="{parent}"/><child link="{child}"/><origin xyz="{xyz}" rpy="0.2 -0.1 0"/>'
                f'<axis xyz="{axis}"/><limit lower="-2.0" upper="2.0" effort="5.0" velocity="20"/></joint>')

    return "\n".join(['<robot name="degenerate_pair">', link("base", 0.5, 0.08),
                      link("left", 0.3, 0.04), link("right", 0.25, 0.04), link("left_tip", 0.1, 0.03), link("right_tip", 0.1, 0.03),
                      joint("j_left", "base", "left", "0.012 0 -0.05", "0.6 0 0.8"), joint("j_right", "base", "right", "-0.012 0 -0.05", "0 0.8 0.6"),
                      joint("j_left_tip", "left", "left_tip", "0.03 0.05 -0.06", "1 0 0"), joint("j_right_tip", "right", "right_tip", "-0.03 0.05 -0.06", "0 1 0"),
                      "</robot>"])
