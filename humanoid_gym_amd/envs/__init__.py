from .dancer_env import DancerEnv
from .softbank_envs import NaoEnv, PepperEnv, RomeoEnv

__all__ = ["DancerEnv", "NaoEnv", "PepperEnv", "RomeoEnv"]
