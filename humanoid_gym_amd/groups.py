"""Env groups on their own HIP streams: the out-of-phase steady state at small env counts.

One ``hg_step`` launch over 4096 envs is a single resident round (4 waves per SIMD) in which every
wave runs the same phase at the same time: the LDS-latency-bound tree sweeps of all waves coincide,
then their VALU-bound solver sweeps do.  Splitting the envs of a GPU into G handles, each stepped on
its own stream WITHOUT joining the streams between steps, lets group A's solver overlap group B's
tree sweeps; measured on MI355X, NAO, 4096 envs: 31 M env-steps/s with one group, 38 M with two,
41 M with four (``tools/two_stream.py``).  This is the usual double-buffered env-group pattern of RL
samplers and is exact for the reference's own drivers, none of which feeds an observation back into
the next action (examples/h5_control_nao.py:14-25, random_control_nao.py:8-10).

Keep the number of busy streams of a process within its hardware queues (``GPU_MAX_HW_QUEUES``, 4 by
default): streams beyond that share a queue and their kernels serialise -- 8 groups ran at 22 M, and 4
groups next to one more busy side stream at 23 M.  Two groups (the default) leave room for the caller's
own stream and a statistics side stream.

Envs are independent, so the result of every env is bit-identical to the single-handle
run; only the schedule differs.  The C-ABI needs nothing new: distinct ``hg_sim`` handles on distinct
streams are independent by contract (include/hg_sim.h).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

import ctypes as C

from ._capi import HgError, HgSim, _check
from .model import FlatModel, SimConfig


class SimGroups:
    """``n_envs`` envs of one robot on one GPU as ``groups`` independent ``HgSim`` handles.

    ``step`` enqueues one step of every group on that group's stream, ordered after the caller's
    current stream (so freshly written actions are seen), and returns without joining.  Outputs are
    slices of the shared tensors ``obs_q / obs_qd / base_xyyaw / reward / done / collision``
    ([n_envs, ...], env-major like ``HgSim``); call ``wait()`` before reading them on the caller's
    stream.  Group g owns envs ``bounds[g] : bounds[g + 1]``.
    """

    def __init__(self, model: FlatModel, cfg: SimConfig, n_envs: int, groups: int = 2, device=None):
        import torch
        self.torch = torch
        groups = max(1, min(int(groups), int(n_envs)))
        self.n_envs, self.groups = int(n_envs), groups
        self.bounds = [int(b) for b in np.linspace(0, n_envs, groups + 1).round()]
        self.sims = [HgSim(model, cfg, self.bounds[g + 1] - self.bounds[g], device) for g in range(groups)]
        self.device = self.sims[0].device
        self.model, self.cfg = model, cfg
        self.n, self.n_act = self.sims[0].n, self.sims[0].n_act
        self.streams = [torch.cuda.Stream(self.device) for _ in range(groups)]
        f32, u8 = torch.float32, torch.uint8
        z = lambda *s, dt=f32: torch.zeros(*s, dtype=dt, device=self.device)   # noqa: E731
        self.obs_q, self.obs_qd = z(n_envs, self.n_act), z(n_envs, self.n_act)
        self.base_xyyaw, self.reward = z(n_envs, 3), z(n_envs)
        self.done, self.collision = z(n_envs, dt=u8), z(n_envs, dt=u8)
        for g, s in enumerate(self.sims):          # every handle writes straight into its slice
            a, b = self.bounds[g], self.bounds[g + 1]
            s.obs_q, s.obs_qd, s.base_xyyaw = self.obs_q[a:b], self.obs_qd[a:b], self.base_xyyaw[a:b]
            s.reward, s.done, s.collision = self.reward[a:b], self.done[a:b], self.collision[a:b]
        # per-group C arguments, built once: a step is then G plain hg_step calls (the host has ~100 us per
        # step of all groups at 4096 NAO envs before it becomes the bottleneck)
        vp = C.c_void_p
        self._L = self.sims[0].L
        self._outs = [tuple(vp(x.data_ptr()) for x in (s.obs_q, s.obs_qd, s.base_xyyaw, s.reward, s.done, s.collision))
                      for s in self.sims]
        self._sp = [vp(st.cuda_stream) for st in self.streams]
        self._ev = torch.cuda.Event()
        self._sel_key, self._sel_arr = None, None

    def close(self):
        for s in self.sims:
            s.close()

    def _fork(self):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    def step(self, actions=None, joint_sel: Optional[Sequence[int]] = None, n_substeps: int = 1,
             order_after_caller: bool = True):
        """actions: float32 CUDA tensor [n_envs, n_sel] (contiguous) or None; not joined, see ``wait``.

        ``order_after_caller=False`` skips the event that orders the group streams after the caller's stream:
        for action rows that were complete (and synchronised) before the loop started, e.g. a resident replay
        trajectory.  The event costs one record and G stream waits per step, on the host and on the GPU."""
        t = self.torch
        sel_p, n_sel = None, 0
        if joint_sel is not None:
            key = tuple(int(j) for j in joint_sel)
            if self._sel_key != key:                 # converted once per selection, not once per step
                self._sel_arr = np.ascontiguousarray(key, np.int32)
                self._sel_key = key
            sel_p, n_sel = self._sel_arr.ctypes.data_as(C.POINTER(C.c_int32)), int(self._sel_arr.size)
        base, row_bytes = 0, 0
        if actions is not None:
            want = n_sel if joint_sel is not None else self.n_act
            if (actions.device != self.device or actions.dtype != t.float32
                    or not actions.is_contiguous() or tuple(actions.shape) != (self.n_envs, want)):
                raise HgError(f"actions must be a contiguous float32 tensor [{self.n_envs}, {want}] on {self.device}")
            base, row_bytes = actions.data_ptr(), 4 * want
            if order_after_caller:
                self._ev.record(t.cuda.current_stream(self.device))     # the actions were written on the caller's stream
        for g, s in enumerate(self.sims):
            if actions is not None and order_after_caller:
                self.streams[g].wait_event(self._ev)
                actions.record_stream(self.streams[g])   # torch's allocator must not recycle it under the group's kernel
            o = self._outs[g]
            a_p = C.c_void_p(base + self.bounds[g] * row_bytes) if actions is not None else None
            _check(self._L, self._L.hg_step(s._h, a_p, sel_p, n_sel, int(n_substeps), o[0], o[1], o[2], o[3], o[4], o[5],
                                            self._sp[g]), "hg_step")
        return self.obs_q, self.reward, self.done, self.collision

    def reset(self, env_mask=None):
        t = self.torch
        ev = self._fork()
        for g, s in enumerate(self.sims):
            a, b = self.bounds[g], self.bounds[g + 1]
            with t.cuda.stream(self.streams[g]):
                self.streams[g].wait_event(ev)
                s.reset(None if env_mask is None else env_mask[a:b])

    def wait(self):
        """Make the caller's current stream wait for everything enqueued on the group streams."""
        cur = self.torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    def get_state(self):
        self.wait()
        return self.torch.cat([s.get_state() for s in self.sims], dim=0)

    def stats_flush(self, stats):
        """Add every group's running episode statistics into ``stats`` (float32[4] on the device) and clear them."""
        t = self.torch
        for g, s in enumerate(self.sims):
            with t.cuda.stream(self.streams[g]):
                s.stats_flush(stats)          # atomic adds: groups may flush concurrently
