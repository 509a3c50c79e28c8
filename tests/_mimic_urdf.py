"""Writes a copy of the stand-in NAO URDF with two finger joints that <mimic> LHand / RHand (the
shape of the hands in SoftBank's real URDFs), for the mimic-joint tests."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "humanoid_gym_amd", "robots", "nao_like.urdf")

_FINGER = """
  <link name="{side}_finger_link">
    <inertial><origin xyz="0.01 0 0"/><mass value="0.004"/>
      <inertia ixx="2e-7" ixy="0" ixz="0" iyy="2e-7" iyz="0" izz="2e-7"/></inertial>
  </link>
  <joint name="{S}Finger11" type="revolute">
    <origin xyz="0.03 0 0" rpy="0 0 0"/>
    <parent link="{side}_gripper"/>
    <child link="{side}_finger_link"/>
    <axis xyz="0 1 0"/>
    <limit lower="0" upper="1.0" effort="0.3" velocity="8.3"/>
    <mimic joint="{S}Hand" multiplier="{mult}" offset="{off}"/>
  </joint>
"""


def write(path, mult=0.9, off=0.05):
    text = open(SRC).read()
    extra = _FINGER.format(side="l", S="L", mult=mult, off=off) + _FINGER.format(side="r", S="R", mult=mult, off=off)
    text = text.replace("</robot>", extra + "</robot>")
    with open(path, "w") as f:
        f.write(text)
    return path
