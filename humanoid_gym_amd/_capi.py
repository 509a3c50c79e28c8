"""ctypes binding of libhg_sim.so (include/hg_sim.h) and a thin tensor-level wrapper.

There is no CPU fallback: if the library is missing or no HIP device is usable
the calls raise.  Buffers are caller-owned torch tensors handed over by
``data_ptr()`` (zero copy); work is enqueued on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from .model import FlatModel, HgConfig, HgModel, SimConfig, HG_MAX_CONTACTS, HG_STATE_BASE

# HG_SIM_LIB: another build of the same library (diagnostics: tools/phase_ticks.py, compiler-flag comparisons)
_LIB_PATH = os.environ.get("HG_SIM_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libhg_sim.so")
_lib = None

SYMBOLS = ["hg_last_error", "hg_abi_version", "hg_create", "hg_destroy", "hg_reset", "hg_step",
           "hg_get_joint_states", "hg_get_link_states", "hg_get_contact_pairs", "hg_get_state",
           "hg_set_state", "hg_get_layout", "hg_stats_accumulate", "hg_stats_flush",
           "hg_set_base_velocity", "hg_get_joint_torques", "hg_stats_accumulate_noalloc"]


# the ABI version that the ctypes mirrors in model.py (HgModel, HgConfig) and the prototypes below are written for
HG_ABI_VERSION = 8


class HgError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    """dlopen libhg_sim.so and declare every prototype of include/hg_sim.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise HgError(f"{p} not found: build it with `python -m humanoid_gym_amd.build` "
                      "(there is no CPU fallback for the simulator)")
    L = C.CDLL(p)
    L.hg_abi_version.restype = C.c_int
    L.hg_abi_version.argtypes = []
    if L.hg_abi_version() != HG_ABI_VERSION:      # a stale in-tree build would read the structs at the wrong strides
        raise HgError(f"{p} has ABI version {L.hg_abi_version()}, this package needs {HG_ABI_VERSION}: rebuild it "
                      "(python -m humanoid_gym_amd.build)")
    vp, fp, u8p, i32p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers
    L.hg_last_error.restype = C.c_char_p
    L.hg_last_error.argtypes = []
    L.hg_abi_version.restype = C.c_int
    L.hg_abi_version.argtypes = []
    L.hg_create.restype = C.c_int
    L.hg_create.argtypes = [C.POINTER(HgModel), C.POINTER(HgConfig), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.hg_destroy.restype = C.c_int
    L.hg_destroy.argtypes = [C.c_void_p]
    L.hg_reset.restype = C.c_int
    L.hg_reset.argtypes = [C.c_void_p, u8p, vp]
    L.hg_step.restype = C.c_int
    L.hg_step.argtypes = [C.c_void_p, fp, C.POINTER(C.c_int32), C.c_int, C.c_int, fp, fp, fp, fp, u8p, u8p, vp]
    L.hg_get_joint_states.restype = C.c_int
    L.hg_get_joint_states.argtypes = [C.c_void_p, fp, fp, vp]
    L.hg_get_joint_torques.restype = C.c_int
    L.hg_get_joint_torques.argtypes = [C.c_void_p, fp, vp]
    L.hg_get_link_states.restype = C.c_int
    L.hg_get_link_states.argtypes = [C.c_void_p, fp, fp, fp, fp, vp]
    L.hg_get_contact_pairs.restype = C.c_int
    L.hg_get_contact_pairs.argtypes = [C.c_void_p, i32p, i32p, vp, vp]
    L.hg_get_state.restype = C.c_int
    L.hg_get_state.argtypes = [C.c_void_p, fp, vp]
    L.hg_set_state.restype = C.c_int
    L.hg_set_state.argtypes = [C.c_void_p, fp, vp]
    L.hg_get_layout.restype = C.c_int
    L.hg_get_layout.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 5
    L.hg_stats_accumulate.restype = C.c_int
    L.hg_stats_accumulate.argtypes = [C.c_void_p, fp, u8p, u8p, fp, vp]
    L.hg_stats_accumulate_noalloc.restype = C.c_int
    L.hg_stats_accumulate_noalloc.argtypes = [C.c_void_p, fp, u8p, u8p, fp, vp]
    L.hg_set_base_velocity.restype = C.c_int
    L.hg_set_base_velocity.argtypes = [C.c_void_p, fp, vp]
    L.hg_stats_flush.restype = C.c_int
    L.hg_stats_flush.argtypes = [C.c_void_p, fp, vp]
    if path is None:
        _lib = L
    return L


def _check(L, rc: int, what: str) -> None:
    if rc != 0:
        raise HgError(f"{what} failed ({rc}): {L.hg_last_error().decode()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class HgSim:
    """n_envs copies of one robot on one GPU.  All tensors are torch CUDA(HIP) tensors."""

    def __init__(self, model: FlatModel, cfg: SimConfig, n_envs: int, device=None):
        import torch
        self.torch = torch
        self.L = load_library()
        if not torch.cuda.is_available():
            raise HgError("no HIP device visible to torch: the simulator has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise HgError(f"device {self.device} is not a GPU")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        self.model, self.cfg, self.n_envs = model, cfg, int(n_envs)
        self.n = model.n_links
        self.n_act = model.n_act
        self.state_size = HG_STATE_BASE + 3 * self.n
        self._cm, self._cc = model.to_c(), cfg.to_c()
        h = C.c_void_p()
        _check(self.L, self.L.hg_create(C.byref(self._cm), C.byref(self._cc), self.n_envs, dev_index, C.byref(h)),
               "hg_create")
        self._h = h
        f32, u8 = torch.float32, torch.uint8
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.device)   # noqa: E731
        self.obs_q = z(self.n_envs, self.n_act)
        self.obs_qd = z(self.n_envs, self.n_act)
        self.base_xyyaw = z(self.n_envs, 3)
        self.reward = z(self.n_envs)
        self.done = z(self.n_envs, dt=u8)
        self.collision = z(self.n_envs, dt=u8)

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self.L.hg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, env_mask=None):
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            if env_mask.numel() != self.n_envs:
                raise HgError(f"env_mask must have {self.n_envs} entries (got {env_mask.numel()})")
        _check(self.L, self.L.hg_reset(self._h, _ptr(env_mask), self._stream()), "hg_reset")

    def step(self, actions=None, joint_sel: Optional[Sequence[int]] = None, n_substeps: int = 1):
        """actions: float32 CUDA tensor [n_envs, n_sel] (or None to keep targets)."""
        sel_p, n_sel = None, 0
        if joint_sel is not None:
            sel = np.ascontiguousarray(joint_sel, np.int32)
            sel_p, n_sel = sel.ctypes.data_as(C.POINTER(C.c_int32)), int(sel.size)
        if actions is not None:
            want = n_sel if joint_sel is not None else self.n_act
            if (actions.device != self.device or actions.dtype != self.torch.float32
                    or not actions.is_contiguous() or tuple(actions.shape) != (self.n_envs, want)):
                raise HgError(f"actions must be a contiguous float32 tensor [{self.n_envs}, {want}] on {self.device}")
        _check(self.L, self.L.hg_step(self._h, _ptr(actions), sel_p, n_sel, int(n_substeps),
                                      _ptr(self.obs_q), _ptr(self.obs_qd), _ptr(self.base_xyyaw),
                                      _ptr(self.reward), _ptr(self.done), _ptr(self.collision), self._stream()),
               "hg_step")
        return self.obs_q, self.reward, self.done, self.collision

    def joint_states(self):
        t = self.torch
        q = t.empty(self.n_envs, self.n, dtype=t.float32, device=self.device)
        qd = t.empty_like(q)
        _check(self.L, self.L.hg_get_joint_states(self._h, _ptr(q), _ptr(qd), self._stream()), "hg_get_joint_states")
        return q, qd

    def joint_torques(self):
        """[n_envs, n_links] motor torque applied in the last physics step (getJointState()[3])."""
        t = self.torch
        tau = t.empty(self.n_envs, self.n, dtype=t.float32, device=self.device)
        _check(self.L, self.L.hg_get_joint_torques(self._h, _ptr(tau), self._stream()), "hg_get_joint_torques")
        return tau

    def link_states(self):
        t = self.torch
        nb = self.n + 1
        e = lambda w: t.empty(self.n_envs, nb, w, dtype=t.float32, device=self.device)   # noqa: E731
        cp, cq, fp, fq = e(3), e(4), e(3), e(4)
        _check(self.L, self.L.hg_get_link_states(self._h, _ptr(cp), _ptr(cq), _ptr(fp), _ptr(fq), self._stream()),
               "hg_get_link_states")
        return cp, cq, fp, fq

    def contact_pairs(self):
        """(pairs [n_envs, HG_MAX_CONTACTS, 2], counts [n_envs], link mask [n_envs]) of the last physics step;
        HgError on a simulator created with store_contacts=False."""
        t = self.torch
        pairs = t.empty(self.n_envs, HG_MAX_CONTACTS, 2, dtype=t.int32, device=self.device)
        counts = t.empty(self.n_envs, dtype=t.int32, device=self.device)
        mask = t.empty(self.n_envs, dtype=t.int64, device=self.device)
        _check(self.L, self.L.hg_get_contact_pairs(self._h, _ptr(pairs), _ptr(counts), _ptr(mask), self._stream()),
               "hg_get_contact_pairs")
        return pairs, counts, mask

    def get_state(self):
        t = self.torch
        s = t.empty(self.n_envs, self.state_size, dtype=t.float32, device=self.device)
        _check(self.L, self.L.hg_get_state(self._h, _ptr(s), self._stream()), "hg_get_state")
        return s

    def set_state(self, s):
        t = self.torch
        s = s.to(device=self.device, dtype=t.float32).contiguous()
        if tuple(s.shape) != (self.n_envs, self.state_size):
            raise HgError(f"state must have shape ({self.n_envs}, {self.state_size}), got {tuple(s.shape)}")
        _check(self.L, self.L.hg_set_state(self._h, _ptr(s), self._stream()), "hg_set_state")

    def stats_accumulate(self, stats):
        """stats: float32[8] tensor on this device, accumulated in place."""
        _check(self.L, self.L.hg_stats_accumulate(self._h, _ptr(self.reward), _ptr(self.done),
                                                  _ptr(self.collision), _ptr(stats), self._stream()),
               "hg_stats_accumulate")
        return stats

    def set_base_velocity(self, cmd):
        """cmd: [n_envs, 3] (vx, vy in the base heading frame [m/s], yaw rate [rad/s]) or None to release the base."""
        if cmd is None:
            _check(self.L, self.L.hg_set_base_velocity(self._h, None, self._stream()), "hg_set_base_velocity")
            return
        t = self.torch
        cmd = cmd.to(device=self.device, dtype=t.float32).contiguous()
        if tuple(cmd.shape) != (self.n_envs, 3):
            raise HgError(f"base velocity command must have shape ({self.n_envs}, 3), got {tuple(cmd.shape)}")
        _check(self.L, self.L.hg_set_base_velocity(self._h, _ptr(cmd), self._stream()), "hg_set_base_velocity")

    def stats_flush(self, stats):
        """Adds the running per-env counters the step kernel keeps (env steps, reward, done, collision)
        to stats[0:4] and clears them.  stats: float32[8] tensor on this device."""
        _check(self.L, self.L.hg_stats_flush(self._h, _ptr(stats), self._stream()), "hg_stats_flush")
        return stats
