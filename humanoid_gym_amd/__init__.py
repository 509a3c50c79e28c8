"""MI355X-native batched replacement for humanoid-gym's PyBullet env-step path.

Same ids as /root/reference/humanoid_gym/__init__.py:4-25 (``nao-real-v0`` is
hardware I/O and out of scope):

    import humanoid_gym_amd as hg
    env = hg.make("nao-v0", n_envs=4096)      # or gym.make("nao-v0") when gym is installed
    obs, reward, done, info = env.step(actions)
"""
from __future__ import annotations

ENTRY_POINTS = {
    "pepper-v0": "humanoid_gym_amd.envs:PepperEnv",
    "nao-v0": "humanoid_gym_amd.envs:NaoEnv",
    "romeo-v0": "humanoid_gym_amd.envs:RomeoEnv",
    "dancer-v0": "humanoid_gym_amd.envs:DancerEnv",
}


def make(env_id: str, **kwargs):
    """``gym.make`` for the four simulated ids, with the batching keywords (n_envs, device, ...)."""
    if env_id not in ENTRY_POINTS:
        raise KeyError(f"unknown env id {env_id!r}; available: {sorted(ENTRY_POINTS)}")
    import importlib
    mod, cls = ENTRY_POINTS[env_id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)


def _register_with_gym() -> bool:
    try:
        from gym.envs.registration import register
    except Exception:
        return False
    for env_id, entry in ENTRY_POINTS.items():
        try:
            register(id=env_id, entry_point=entry)
        except Exception:           # already registered (e.g. the reference package is imported too)
            pass
    return True


GYM_REGISTERED = _register_with_gym()
