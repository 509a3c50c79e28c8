"""Robot descriptions and the world parameters each reference env uses.

* ``dancer``: the reference's own URDF
  (/root/reference/humanoid_gym/envs/assets/dancer/dancer_urdf_model.URDF),
  shipped here flattened (``dancer.json``, made by tools/flatten_dancer.py)
  because the GPU boxes do not hold /root/reference.  World parameters are the
  literals of dancer_env.py:58-65.
* ``nao``, ``pepper``, ``romeo``: the reference loads SoftBank's URDFs through
  qibullet (nao_env.py:20, pepper_env.py:18, romeo_env.py:17); those files are
  neither in the reference tree nor in this image.  The ``*_like.urdf`` files
  next to this module are structure-faithful stand-ins authored for this
  project (same joint names, tree shape and joint limits as far as public
  specs go; link masses and dimensions approximate).  A user who has qibullet's
  resources passes the real file through ``urdf_path=``.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

from ..model import FlatModel, SimConfig
from ..urdf import load_urdf

_HERE = os.path.dirname(os.path.abspath(__file__))

# joint orders the envs expose (nao_env.py:30-31, pepper_env.py:28-39)
NAO_JOINTS = ['LShoulderPitch', 'LShoulderRoll', 'LElbowYaw', 'LElbowRoll', 'LWristYaw',
              'RShoulderPitch', 'RShoulderRoll', 'RElbowYaw', 'RElbowRoll', 'RWristYaw',
              'LHipYawPitch', 'LHipRoll', 'LHipPitch', 'LKneePitch', 'LAnklePitch', 'LAnkleRoll',
              'RHipYawPitch', 'RHipRoll', 'RHipPitch', 'RKneePitch', 'RAnklePitch', 'RAnkleRoll']
PEPPER_JOINTS = ['LShoulderPitch', 'LShoulderRoll', 'LElbowYaw', 'LElbowRoll', 'LWristYaw', 'LHand',
                 'RShoulderPitch', 'RShoulderRoll', 'RElbowYaw', 'RElbowRoll', 'RWristYaw', 'RHand']


def dancer_config(**kw) -> SimConfig:
    # dancer_env.py:63-65; setJointMotorControlArray does not clamp targets; loadURDF is called with
    # URDF_USE_SELF_COLLISION + URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS (dancer_env.py:60)
    c = SimConfig(gravity=(0.0, 0.0, -10.0), solver_iters=150, max_contacts=12, clamp_targets=False,
                  enable_self=True)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def softbank_config(**kw) -> SimConfig:
    # PyBullet defaults under qibullet; setAngles clamps to the joint limits; qibullet loads its robots
    # with URDF_USE_SELF_COLLISION [UPSTREAM-RECALL], so link-link contacts are on here as well
    c = SimConfig(gravity=(0.0, 0.0, -9.81), solver_iters=50, max_contacts=8, clamp_targets=True,
                  enable_self=True)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def load_robot(name: str, urdf_path: Optional[str] = None, hands: bool = False) -> Tuple[FlatModel, SimConfig]:
    """name: dancer | nao | pepper | romeo; "nao_hands" / "romeo_hands" (or hands=True) select the stand-ins
    that include the finger joints of the hands (42 / 61 dofs)."""
    name = name.lower()
    if name.endswith("_hands"):
        name, hands = name[:-len("_hands")], True
    if name == "dancer":
        if urdf_path:
            m = load_urdf(urdf_path, base_position=[-0.3, -0.3, 0.3],
                          base_orientation=[0, 0.7071068, -0.7071068, 0], self_collision=True, name="dancer")
        else:
            with open(os.path.join(_HERE, "dancer.json")) as f:
                m = FlatModel.from_json(f.read())
        return m, dancer_config()
    if name in ("nao", "pepper", "romeo"):
        from .softbank_like import load_softbank_like
        return load_softbank_like(name, urdf_path, hands=hands), softbank_config()
    raise KeyError(f"unknown robot {name!r}")
