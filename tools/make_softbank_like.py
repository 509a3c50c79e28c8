"""Writes the stand-in URDFs for NAO, Pepper and Romeo into humanoid_gym_amd/robots/.

The reference gets these robots from qibullet's downloaded resources
(nao_env.py:20, pepper_env.py:18, romeo_env.py:17), which are not in the
reference tree nor in this image.  These descriptions keep the joint NAMES,
the tree SHAPE and the joint LIMITS of the public SoftBank documentation as far
as remembered; masses, link lengths and collision shapes are approximate.
They are labelled "*_like" everywhere for that reason.

Frames follow the SoftBank convention: x forward, y left, z up; every joint
frame is parallel to the torso frame at the zero posture.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "humanoid_gym_amd", "robots")


def box_inertia(m, sx, sy, sz):
    return (m * (sy * sy + sz * sz) / 12, m * (sx * sx + sz * sz) / 12, m * (sx * sx + sy * sy) / 12)


class Urdf:
    def __init__(self, name):
        self.name, self.links, self.joints = name, [], []

    def link(self, name, mass, com, size, col=None, col_origin=None):
        """Box-shaped body: inertia of a solid box `size` about `com`; collision `col`
        is ("box", (sx,sy,sz)) / ("sphere", r) / ("cylinder", (r, L)) / None."""
        ixx, iyy, izz = box_inertia(mass, *size)
        s = [f'  <link name="{name}">', '    <inertial>',
             f'      <origin xyz="{com[0]:.6g} {com[1]:.6g} {com[2]:.6g}" rpy="0 0 0"/>',
             f'      <mass value="{mass:.6g}"/>',
             f'      <inertia ixx="{ixx:.6g}" ixy="0" ixz="0" iyy="{iyy:.6g}" iyz="0" izz="{izz:.6g}"/>',
             '    </inertial>']
        if col is not None:
            o = col_origin if col_origin is not None else com
            s += ['    <collision>', f'      <origin xyz="{o[0]:.6g} {o[1]:.6g} {o[2]:.6g}" rpy="0 0 0"/>', '      <geometry>']
            if col[0] == "box":
                s.append(f'        <box size="{col[1][0]:.6g} {col[1][1]:.6g} {col[1][2]:.6g}"/>')
            elif col[0] == "sphere":
                s.append(f'        <sphere radius="{col[1]:.6g}"/>')
            else:
                s.append(f'        <cylinder radius="{col[1][0]:.6g}" length="{col[1][1]:.6g}"/>')
            s += ['      </geometry>', '    </collision>']
        s.append('  </link>')
        self.links.append("\n".join(s))

    def joint(self, name, parent, child, xyz, axis, lower=None, upper=None, effort=1.5, velocity=7.0,
              damping=0.0, jtype=None):
        jtype = jtype or ("continuous" if lower is None else "revolute")
        s = [f'  <joint name="{name}" type="{jtype}">',
             f'    <origin xyz="{xyz[0]:.6g} {xyz[1]:.6g} {xyz[2]:.6g}" rpy="0 0 0"/>',
             f'    <parent link="{parent}"/>', f'    <child link="{child}"/>',
             f'    <axis xyz="{axis[0]:.9g} {axis[1]:.9g} {axis[2]:.9g}"/>']
        if jtype == "revolute":
            s.append(f'    <limit lower="{lower:.6f}" upper="{upper:.6f}" effort="{effort:.6g}" velocity="{velocity:.6g}"/>')
        elif jtype == "continuous":
            s.append(f'    <limit effort="{effort:.6g}" velocity="{velocity:.6g}"/>')
        s += [f'    <dynamics damping="{damping:.6g}" friction="0"/>', '  </joint>']
        self.joints.append("\n".join(s))

    def mimic_joint(self, name, parent, child, xyz, axis, lower, upper, mimic, multiplier=1.0, offset=0.0,
                    effort=0.3, velocity=8.3):
        """revolute joint that carries a <mimic> tag: Bullet simulates it as an ordinary dof, qibullet's
        setAngles drives it from the joint it mimics (the finger joints of the SoftBank hands)."""
        self.joint(name, parent, child, xyz, axis, lower, upper, effort, velocity)
        self.joints[-1] = self.joints[-1].replace(
            "  </joint>", f'    <mimic joint="{mimic}" multiplier="{multiplier:.6g}" offset="{offset:.6g}"/>\n  </joint>')

    def write(self, path, comment):
        with open(path, "w") as f:
            f.write('<?xml version="1.0"?>\n<!-- ' + comment + ' -->\n')
            f.write(f'<robot name="{self.name}">\n')
            # links and joints interleaved in creation order is not needed: URDF is order-free
            # except for child-joint declaration order, which the joint list preserves.
            f.write("\n".join(self.links) + "\n" + "\n".join(self.joints) + "\n</robot>\n")


X, Y, Z = (1, 0, 0), (0, 1, 0), (0, 0, 1)
S2 = 0.70710678118


def add_hand(u, side, chains, phalanx=0.014, mass=0.002):
    """Finger joints of one hand, hanging off the `<side>_gripper` link and mimicking `<Side>Hand`.
    chains: {"Finger1": 3, "Finger2": 3, "Thumb": 2} -> LFinger11..13, LFinger21..23, LThumb1..2: the joint
    names of SoftBank's hands [public documentation / upstream recall]; dimensions and masses are approximate
    (light phalanges, a few grams).  Fingers curl about y, the thumb about z; no collision geometry."""
    S, s = side, side.lower()
    for ci, (chain, count) in enumerate(chains.items()):
        parent = f"{s}_gripper"
        yoff = (ci - 0.5 * (len(chains) - 1)) * 0.012
        for k in range(1, count + 1):
            jn = f"{S}{chain}{k}"
            ln = f"{jn}_link"
            u.link(ln, mass, (0.5 * phalanx, 0, 0), (phalanx, 0.008, 0.008))
            xyz = (0.02, yoff, 0.0) if k == 1 else (phalanx, 0, 0)
            axis = Z if chain.startswith("Thumb") else Y
            u.mimic_joint(jn, parent, ln, xyz, axis, 0.0, 1.0, f"{S}Hand", multiplier=0.95, offset=0.0)
            parent = ln


NAO_HAND = {"Finger1": 3, "Finger2": 3, "Thumb": 2}                    # 8 joints per hand, 16 in all: 26 + 16 = 42 dofs
ROMEO_HAND = {"Finger1": 3, "Finger2": 3, "Finger3": 3, "Thumb": 3}    # 12 per hand, 24 in all: 37 + 24 = 61 dofs


def make_nao(hands=False):
    u = Urdf("nao_hands_like" if hands else "nao_like")
    u.link("torso", 1.05, (-0.004, 0, 0.043), (0.10, 0.13, 0.19), ("box", (0.10, 0.13, 0.19)))
    # head
    u.link("Neck", 0.064, (0, 0, -0.027), (0.03, 0.03, 0.05))
    u.link("Head", 0.605, (0, 0, 0.053), (0.11, 0.11, 0.11), ("sphere", 0.062), (0, 0, 0.055))
    u.joint("HeadYaw", "torso", "Neck", (0, 0, 0.1265), Z, -2.0857, 2.0857, 1.55, 8.27)
    u.joint("HeadPitch", "Neck", "Head", (0, 0, 0), Y, -0.6720, 0.5149, 1.33, 7.19)
    # legs (left then right), before the arms as in SoftBank's chain order
    for side, sy in (("L", 1.0), ("R", -1.0)):
        yawpitch_axis = (0, S2, -S2) if side == "L" else (0, S2, S2)
        lim = dict(
            HipYawPitch=(-1.145303, 0.740810),
            HipRoll=(-0.379472, 0.790477) if side == "L" else (-0.790477, 0.379472),
            HipPitch=(-1.535889, 0.484090),
            KneePitch=(-0.092346, 2.112528) if side == "L" else (-0.103083, 2.120198),
            AnklePitch=(-1.189516, 0.922747) if side == "L" else (-1.186448, 0.932056),
            AnkleRoll=(-0.397880, 0.769001) if side == "L" else (-0.768992, 0.397935))
        u.link(f"{side}Pelvis", 0.070, (-0.008, 0, -0.004), (0.04, 0.04, 0.04))
        u.link(f"{side}Hip", 0.130, (-0.015, 0, 0.0), (0.04, 0.04, 0.04))
        u.link(f"{side}Thigh", 0.390, (0.001, sy * 0.002, -0.054), (0.07, 0.07, 0.10), ("cylinder", (0.035, 0.10)), (0, 0, -0.05))
        u.link(f"{side}Tibia", 0.292, (0.004, sy * 0.002, -0.045), (0.06, 0.06, 0.10), ("cylinder", (0.032, 0.10)), (0, 0, -0.0515))
        u.link(f"{side}AnklePitch", 0.134, (0.0, 0, 0.007), (0.04, 0.04, 0.04))
        u.link(f"{side.lower()}_ankle", 0.162, (0.025, sy * 0.003, -0.032), (0.16, 0.09, 0.03),
               ("box", (0.16, 0.088, 0.03)), (0.02, sy * 0.003, -0.03019))
        u.joint(f"{side}HipYawPitch", "torso", f"{side}Pelvis", (0, sy * 0.050, -0.085), yawpitch_axis, *lim["HipYawPitch"], 3.35, 4.16)
        u.joint(f"{side}HipRoll", f"{side}Pelvis", f"{side}Hip", (0, 0, 0), X, *lim["HipRoll"], 3.35, 4.16)
        u.joint(f"{side}HipPitch", f"{side}Hip", f"{side}Thigh", (0, 0, 0), Y, *lim["HipPitch"], 3.02, 6.40)
        u.joint(f"{side}KneePitch", f"{side}Thigh", f"{side}Tibia", (0, 0, -0.100), Y, *lim["KneePitch"], 3.02, 6.40)
        u.joint(f"{side}AnklePitch", f"{side}Tibia", f"{side}AnklePitch", (0, 0, -0.1029), Y, *lim["AnklePitch"], 3.02, 6.40)
        u.joint(f"{side}AnkleRoll", f"{side}AnklePitch", f"{side.lower()}_ankle", (0, 0, 0), X, *lim["AnkleRoll"], 3.35, 4.16)
    for side, sy in (("L", 1.0), ("R", -1.0)):
        lim = dict(
            ShoulderPitch=(-2.0857, 2.0857),
            ShoulderRoll=(-0.3142, 1.3265) if side == "L" else (-1.3265, 0.3142),
            ElbowYaw=(-2.0857, 2.0857),
            ElbowRoll=(-1.5446, -0.0349) if side == "L" else (0.0349, 1.5446),
            WristYaw=(-1.8238, 1.8238), Hand=(0.0, 1.0))
        u.link(f"{side}Shoulder", 0.075, (-0.002, sy * -0.014, 0), (0.04, 0.04, 0.04))
        u.link(f"{side}Bicep", 0.157, (0.024, sy * 0.004, 0.003), (0.10, 0.05, 0.05), ("cylinder", (0.025, 0.09)), (0.05, 0, 0))
        u.link(f"{side}Elbow", 0.065, (-0.027, 0, 0), (0.04, 0.04, 0.04))
        u.link(f"{side}ForeArm", 0.077, (0.026, sy * 0.003, 0), (0.06, 0.04, 0.04))
        u.link(f"{side.lower()}_wrist", 0.185, (0.035, sy * 0.002, 0.003), (0.08, 0.05, 0.04), ("sphere", 0.03), (0.04, 0, 0))
        u.link(f"{side.lower()}_gripper", 0.02, (0.01, 0, 0), (0.03, 0.03, 0.02))
        u.joint(f"{side}ShoulderPitch", "torso", f"{side}Shoulder", (0, sy * 0.098, 0.100), Y, *lim["ShoulderPitch"], 1.33, 8.27)
        u.joint(f"{side}ShoulderRoll", f"{side}Shoulder", f"{side}Bicep", (0, 0, 0), Z, *lim["ShoulderRoll"], 1.77, 7.19)
        u.joint(f"{side}ElbowYaw", f"{side}Bicep", f"{side}Elbow", (0.105, sy * 0.015, 0), X, *lim["ElbowYaw"], 1.55, 8.27)
        u.joint(f"{side}ElbowRoll", f"{side}Elbow", f"{side}ForeArm", (0, 0, 0), Z, *lim["ElbowRoll"], 1.33, 7.19)
        u.joint(f"{side}WristYaw", f"{side}ForeArm", f"{side.lower()}_wrist", (0.05595, 0, 0), X, *lim["WristYaw"], 0.85, 24.6)
        u.joint(f"{side}Hand", f"{side.lower()}_wrist", f"{side.lower()}_gripper", (0.05775, 0, -0.012), X, *lim["Hand"], 0.3, 8.3)
        if hands:
            add_hand(u, side, NAO_HAND)
    u.write(os.path.join(OUT, "nao_hands_like.urdf" if hands else "nao_like.urdf"),
            "NAO-like stand-in" + (" WITH the 16 finger joints of the hands (42 dofs)" if hands else "") +
            " authored for humanoid_gym_amd; NOT SoftBank's URDF (see robots/__init__.py)")


def make_pepper():
    u = Urdf("pepper_like")
    # root = mobile base (SoftBank's Tibia link): heavy, passive, on three wheels
    u.link("Tibia", 12.0, (0.0, 0, 0.13), (0.42, 0.42, 0.26), ("cylinder", (0.10, 0.22)), (0, 0, 0.19))
    for nm, ang in (("WheelFL", 1.0472), ("WheelFR", -1.0472), ("WheelB", 3.14159265)):
        import math
        px, py = 0.176 * math.cos(ang), 0.176 * math.sin(ang)
        u.link(nm + "_link", 0.35, (0, 0, 0), (0.07, 0.07, 0.07), ("sphere", 0.07))
        u.joint(nm, "Tibia", nm + "_link", (px, py, 0.07), (-math.sin(ang), math.cos(ang), 0), None, None, 0.0, 6.0)
    u.link("Thigh", 2.4, (0, 0, 0.14), (0.14, 0.14, 0.27), ("cylinder", (0.06, 0.20)), (0, 0, 0.13))
    u.link("Hip", 1.2, (0, 0, 0.04), (0.12, 0.12, 0.10))
    u.link("torso", 4.6, (0.0, 0, 0.07), (0.20, 0.26, 0.32), ("box", (0.18, 0.24, 0.30)), (0, 0, 0.07))
    u.joint("KneePitch", "Tibia", "Thigh", (0, 0, 0.334), Y, -0.5149, 0.5149, 27.7, 2.93)
    u.joint("HipPitch", "Thigh", "Hip", (0, 0, 0.268), Y, -1.0385, 1.0385, 19.8, 2.93)
    u.joint("HipRoll", "Hip", "torso", (0, 0, 0.079), X, -0.5149, 0.5149, 10.1, 2.27)
    u.link("Neck", 0.27, (0, 0, 0.02), (0.06, 0.06, 0.06))
    u.link("Head", 1.52, (0.01, 0, 0.11), (0.19, 0.17, 0.21), ("sphere", 0.10), (0.01, 0, 0.11))
    u.joint("HeadYaw", "torso", "Neck", (-0.038, 0, 0.1699), Z, -2.0857, 2.0857, 5.4, 7.33)
    u.joint("HeadPitch", "Neck", "Head", (0, 0, 0), Y, -0.7068, 0.6371, 2.7, 9.22)
    for side, sy in (("L", 1.0), ("R", -1.0)):
        lim = dict(ShoulderPitch=(-2.0857, 2.0857),
                   ShoulderRoll=(0.0087, 1.5620) if side == "L" else (-1.5620, -0.0087),
                   ElbowYaw=(-2.0857, 2.0857),
                   ElbowRoll=(-1.5620, -0.0087) if side == "L" else (0.0087, 1.5620),
                   WristYaw=(-1.8239, 1.8239), Hand=(0.02, 0.98))
        u.link(f"{side}Shoulder", 0.31, (0, sy * -0.02, 0), (0.07, 0.07, 0.07))
        u.link(f"{side}Bicep", 0.55, (0.08, sy * 0.01, 0), (0.18, 0.07, 0.07), ("cylinder", (0.035, 0.16)), (0.09, 0, 0))
        u.link(f"{side}Elbow", 0.25, (-0.02, 0, 0), (0.06, 0.06, 0.06))
        u.link(f"{side}ForeArm", 0.33, (0.06, sy * 0.005, 0), (0.15, 0.06, 0.06))
        u.link(f"{side.lower()}_wrist", 0.28, (0.03, 0, 0), (0.08, 0.07, 0.05), ("sphere", 0.04), (0.04, 0, 0))
        u.link(f"{side.lower()}_gripper", 0.05, (0.02, 0, 0), (0.05, 0.04, 0.03))
        u.joint(f"{side}ShoulderPitch", "torso", f"{side}Shoulder", (-0.057, sy * 0.14974, 0.08682), Y, *lim["ShoulderPitch"], 5.4, 7.33)
        u.joint(f"{side}ShoulderRoll", f"{side}Shoulder", f"{side}Bicep", (0, 0, 0), Z, *lim["ShoulderRoll"], 2.7, 9.22)
        u.joint(f"{side}ElbowYaw", f"{side}Bicep", f"{side}Elbow", (0.1812, sy * 0.015, 0.00013), X, *lim["ElbowYaw"], 5.4, 7.33)
        u.joint(f"{side}ElbowRoll", f"{side}Elbow", f"{side}ForeArm", (0, 0, 0), Z, *lim["ElbowRoll"], 2.7, 9.22)
        u.joint(f"{side}WristYaw", f"{side}ForeArm", f"{side.lower()}_wrist", (0.15, 0, 0), X, *lim["WristYaw"], 0.2, 17.3)
        u.joint(f"{side}Hand", f"{side.lower()}_wrist", f"{side.lower()}_gripper", (0.0695, 0, -0.03), X, *lim["Hand"], 0.14, 12.5)
    u.write(os.path.join(OUT, "pepper_like.urdf"),
            "Pepper-like stand-in authored for humanoid_gym_amd; NOT SoftBank's URDF (see robots/__init__.py)")


def make_romeo(hands=False):
    u = Urdf("romeo_hands_like" if hands else "romeo_like")
    u.link("torso", 9.0, (0, 0, 0.12), (0.20, 0.30, 0.38), ("box", (0.18, 0.28, 0.36)), (0, 0, 0.12))
    # neck / head / eyes
    u.link("NeckYawLink", 0.3, (0, 0, 0.02), (0.05, 0.05, 0.05))
    u.link("NeckPitchLink", 0.4, (0, 0, 0.04), (0.05, 0.05, 0.09))
    u.link("HeadPitchLink", 0.2, (0, 0, 0.01), (0.04, 0.04, 0.04))
    u.link("HeadRollLink", 2.2, (0.02, 0, 0.08), (0.18, 0.16, 0.20), ("sphere", 0.10), (0.02, 0, 0.08))
    u.joint("NeckYaw", "torso", "NeckYawLink", (0, 0, 0.33), Z, -1.5708, 1.5708, 8.0, 3.6)
    u.joint("NeckPitch", "NeckYawLink", "NeckPitchLink", (0, 0, 0.02), Y, -0.3491, 0.6981, 8.0, 3.6)
    u.joint("HeadPitch", "NeckPitchLink", "HeadPitchLink", (0, 0, 0.095), Y, -0.3491, 0.3491, 4.0, 4.2)
    u.joint("HeadRoll", "HeadPitchLink", "HeadRollLink", (0, 0, 0), X, -0.3491, 0.3491, 4.0, 4.2)
    for side, sy in (("L", 1.0), ("R", -1.0)):
        u.link(f"{side}EyeYawLink", 0.01, (0, 0, 0), (0.02, 0.02, 0.02))
        u.link(f"{side}EyeLink", 0.02, (0.005, 0, 0), (0.025, 0.025, 0.025))
        u.joint(f"{side}EyeYaw", "HeadRollLink", f"{side}EyeYawLink", (0.09, sy * 0.035, 0.10), Z, -0.5236, 0.5236, 0.05, 15.0)
        u.joint(f"{side}EyePitch", f"{side}EyeYawLink", f"{side}EyeLink", (0, 0, 0), Y, -0.5236, 0.5236, 0.05, 15.0)
    # trunk and legs
    u.link("body", 5.5, (0, 0, -0.05), (0.18, 0.26, 0.16), ("box", (0.16, 0.24, 0.14)), (0, 0, -0.05))
    u.joint("TrunkYaw", "torso", "body", (0, 0, -0.10), Z, -0.7854, 0.7854, 40.0, 3.0)
    for side, sy in (("L", 1.0), ("R", -1.0)):
        u.link(f"{side}HipYawLink", 0.6, (0, 0, -0.02), (0.08, 0.08, 0.08))
        u.link(f"{side}HipRollLink", 0.8, (0, 0, 0), (0.08, 0.08, 0.08))
        u.link(f"{side}Thigh", 3.6, (0.0, sy * 0.005, -0.14), (0.12, 0.12, 0.30), ("cylinder", (0.06, 0.26)), (0, 0, -0.16))
        u.link(f"{side}Tibia", 2.4, (0.0, 0, -0.13), (0.10, 0.10, 0.28), ("cylinder", (0.05, 0.24)), (0, 0, -0.15))
        u.link(f"{side}AnklePitchLink", 0.4, (0, 0, 0), (0.06, 0.06, 0.06))
        u.link(f"{side.lower()}_ankle", 1.0, (0.03, 0, -0.05), (0.26, 0.12, 0.05), ("box", (0.26, 0.11, 0.05)), (0.03, 0, -0.045))
        hr = (-0.3491, 0.5411) if side == "L" else (-0.5411, 0.3491)
        ar = (-0.3491, 0.3491)
        u.joint(f"{side}HipYaw", "body", f"{side}HipYawLink", (0, sy * 0.096, -0.13), Z, -0.5236, 0.5236, 40.0, 3.2)
        u.joint(f"{side}HipRoll", f"{side}HipYawLink", f"{side}HipRollLink", (0, 0, -0.03), X, *hr, 60.0, 3.2)
        u.joint(f"{side}HipPitch", f"{side}HipRollLink", f"{side}Thigh", (0, 0, 0), Y, -1.7279, 0.4363, 80.0, 3.9)
        u.joint(f"{side}KneePitch", f"{side}Thigh", f"{side}Tibia", (0, 0, -0.32), Y, -0.0175, 2.0944, 80.0, 3.9)
        u.joint(f"{side}AnklePitch", f"{side}Tibia", f"{side}AnklePitchLink", (0, 0, -0.30), Y, -0.9425, 0.7854, 60.0, 3.9)
        u.joint(f"{side}AnkleRoll", f"{side}AnklePitchLink", f"{side.lower()}_ankle", (0, 0, 0), X, *ar, 40.0, 3.2)
    # arms: 7 dofs + hand
    for side, sy in (("L", 1.0), ("R", -1.0)):
        u.link(f"{side}ShoulderPitchLink", 0.7, (0, sy * 0.02, 0), (0.08, 0.08, 0.08))
        u.link(f"{side}ShoulderYawLink", 1.3, (0.08, 0, 0), (0.20, 0.08, 0.08), ("cylinder", (0.04, 0.16)), (0.10, 0, 0))
        u.link(f"{side}ElbowRollLink", 0.4, (0, 0, 0), (0.06, 0.06, 0.06))
        u.link(f"{side}ElbowYawLink", 0.9, (0.08, 0, 0), (0.18, 0.07, 0.07), ("cylinder", (0.035, 0.14)), (0.09, 0, 0))
        u.link(f"{side}WristRollLink", 0.15, (0, 0, 0), (0.04, 0.04, 0.04))
        u.link(f"{side}WristYawLink", 0.15, (0, 0, 0), (0.04, 0.04, 0.04))
        u.link(f"{side.lower()}_wrist", 0.45, (0.04, 0, 0), (0.10, 0.08, 0.04), ("sphere", 0.045), (0.05, 0, 0))
        u.link(f"{side.lower()}_gripper", 0.08, (0.02, 0, 0), (0.05, 0.04, 0.03))
        sr = (-1.5708, 1.5708)
        sy_lim = (-0.1745, 1.2217) if side == "L" else (-1.2217, 0.1745)
        er = (-2.0071, -0.0175) if side == "L" else (0.0175, 2.0071)
        u.joint(f"{side}ShoulderPitch", "torso", f"{side}ShoulderPitchLink", (0, sy * 0.19, 0.26), Y, *sr, 20.0, 4.0)
        u.joint(f"{side}ShoulderYaw", f"{side}ShoulderPitchLink", f"{side}ShoulderYawLink", (0, 0, 0), Z, *sy_lim, 20.0, 4.0)
        u.joint(f"{side}ElbowRoll", f"{side}ShoulderYawLink", f"{side}ElbowRollLink", (0.205, 0, 0), X, -2.0944, 2.0944, 10.0, 5.0)
        u.joint(f"{side}ElbowYaw", f"{side}ElbowRollLink", f"{side}ElbowYawLink", (0, 0, 0), Z, *er, 10.0, 5.0)
        u.joint(f"{side}WristRoll", f"{side}ElbowYawLink", f"{side}WristRollLink", (0.1823, 0, 0), X, -3.1241, 0.6458, 3.0, 8.0)
        u.joint(f"{side}WristYaw", f"{side}WristRollLink", f"{side}WristYawLink", (0, 0, 0), Z, -0.5236, 0.5236, 3.0, 8.0)
        u.joint(f"{side}WristPitch", f"{side}WristYawLink", f"{side.lower()}_wrist", (0, 0, 0), Y, -1.0123, 0.9948, 3.0, 8.0)
        u.joint(f"{side}Hand", f"{side.lower()}_wrist", f"{side.lower()}_gripper", (0.08, 0, -0.02), X, 0.0, 1.0, 0.5, 8.0)
        if hands:
            add_hand(u, side, ROMEO_HAND, phalanx=0.025, mass=0.006)
    u.write(os.path.join(OUT, "romeo_hands_like.urdf" if hands else "romeo_like.urdf"),
            "Romeo-like stand-in" + (" WITH the 24 Finger/Thumb joints of the hands (61 dofs)" if hands else "") +
            " authored for humanoid_gym_amd; NOT SoftBank's URDF (see robots/__init__.py)")


if __name__ == "__main__":
    make_nao()
    make_nao(hands=True)
    make_pepper()
    make_romeo()
    make_romeo(hands=True)
    print("wrote nao_like.urdf nao_hands_like.urdf pepper_like.urdf romeo_like.urdf romeo_hands_like.urdf to", OUT)
