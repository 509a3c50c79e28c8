"""Shared machinery of the four simulated envs: batching, tensor <-> list conversion."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .._capi import HgSim
from ..model import FlatModel, SimConfig

try:
    import gym as _gym
    _EnvBase = _gym.Env
except Exception:                       # gym is optional (absent in this image)
    class _EnvBase:                     # type: ignore[no-redef]
        metadata: dict = {}


class BatchedHumanoidEnv(_EnvBase):
    """n_envs independent copies of one robot stepped by one HIP kernel launch.

    With ``n_envs == 1`` (the reference's only mode) observations come back as
    plain Python lists / floats / bools exactly as the reference returns them;
    with ``n_envs > 1`` they are torch tensors with a leading env axis that
    stay on the GPU.  Those tensors (observations, reward, done, collision) are
    the SAME buffers on every step -- the kernel writes them in place, nothing is
    allocated per step -- so a rollout buffer must ``clone()`` what it keeps.
    """

    def __init__(self, model: FlatModel, cfg: SimConfig, n_envs: int = 1, device=None, n_substeps: int = 1,
                 auto_reset: bool = False):
        import torch
        self._torch = torch
        self.model, self.cfg = model, cfg
        self.n_envs, self.n_substeps = int(n_envs), int(n_substeps)
        self.sim = HgSim(model, cfg, self.n_envs, device)
        self.device = self.sim.device
        self.batched = self.n_envs > 1
        # on-device episode bookkeeping for the batched reward/done pipeline
        self.auto_reset = bool(auto_reset)
        z = lambda dt: torch.zeros(self.n_envs, dtype=dt, device=self.device)   # noqa: E731
        self.episode_return, self.episode_length = z(torch.float32), z(torch.int32)
        self.last_episode_return, self.last_episode_length = z(torch.float32), z(torch.int32)

    # -- helpers -----------------------------------------------------------
    def _actions_tensor(self, actions, width: int):
        t = self._torch
        if isinstance(actions, t.Tensor):
            a = actions.to(device=self.device, dtype=t.float32)
        else:
            a = t.as_tensor(np.asarray(actions, dtype=np.float32), device=self.device)
        if a.dim() == 1:
            a = a.unsqueeze(0).expand(self.n_envs, -1)
        if tuple(a.shape) != (self.n_envs, width):
            raise ValueError(f"expected {width} action values per env (got shape {tuple(a.shape)})")
        return a.contiguous()

    def _out(self, x):
        """tensor [n_envs, ...] -> python value for the single-env mode."""
        if self.batched:
            return x
        v = x[0].tolist()
        return v

    def _finish_step(self, reward, done):
        """Batched mode: accumulate returns/lengths; envs whose `done` is set are reset in place
        (masked hg_reset, no host synchronisation) when auto_reset is on."""
        t = self._torch
        reward = t.as_tensor(reward, dtype=t.float32, device=self.device).expand(self.n_envs)
        done = t.as_tensor(done, device=self.device).expand(self.n_envs) != 0
        self.episode_return += reward
        self.episode_length += 1
        self.last_episode_return = t.where(done, self.episode_return, self.last_episode_return)
        self.last_episode_length = t.where(done, self.episode_length, self.last_episode_length)
        if self.auto_reset:
            self.sim.reset(done)
            self.episode_return = t.where(done, t.zeros_like(self.episode_return), self.episode_return)
            self.episode_length = t.where(done, t.zeros_like(self.episode_length), self.episode_length)
        return reward, done

    def reset_envs(self, env_mask=None):
        self.sim.reset(env_mask)

    def render(self, mode="human"):
        """Rasterisation is outside the accelerated path (OUT OF SCOPE)."""
        return None

    def close(self):
        self.sim.close()
