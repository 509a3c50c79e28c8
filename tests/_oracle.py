"""ctypes binding of the CPU oracle (oracle/libhg_oracle.so).

Test infrastructure: imported only by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg.  The product package never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from humanoid_gym_amd.model import (FlatModel, HgConfig, HgModel, SimConfig,  # noqa: E402
                                    HG_MAX_CONTACTS, HG_STATE_BASE)

_LIB = None


def build_oracle() -> str:
    so = os.path.join(ROOT, "oracle", "libhg_oracle.so")
    src = os.path.join(ROOT, "oracle", "hg_oracle.c")
    hdr = os.path.join(ROOT, "include", "hg_model.h")
    if (not os.path.exists(so)) or (os.path.exists(src) and
                                    os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build_oracle())
        dp = C.POINTER(C.c_double)
        L.hgo_reset.argtypes = [C.POINTER(HgModel), dp, C.c_int]
        L.hgo_reset.restype = None
        L.hgo_step.argtypes = [C.POINTER(HgModel), C.POINTER(HgConfig), C.c_int, dp, dp,
                               C.POINTER(C.c_int32), C.c_int, C.c_int, dp, dp, dp,
                               C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
                               C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.hgo_step.restype = C.c_int
        L.hgo_step_cmd.argtypes = L.hgo_step.argtypes + [dp]
        L.hgo_step_cmd.restype = C.c_int
        L.hgo_step_full.argtypes = L.hgo_step_cmd.argtypes + [dp]
        L.hgo_step_full.restype = C.c_int
        L.hgo_link_states.argtypes = [C.POINTER(HgModel), dp, C.c_int, dp, dp, dp, dp]
        L.hgo_link_states.restype = None
        L.hgo_forward_dynamics.argtypes = [C.POINTER(HgModel), C.POINTER(HgConfig), dp, dp, dp]
        L.hgo_forward_dynamics.restype = C.c_int
        L.hgo_apply_minv.argtypes = [C.POINTER(HgModel), C.POINTER(HgConfig), dp, dp, dp]
        L.hgo_apply_minv.restype = C.c_int
        L.hgo_energy_momentum.argtypes = [C.POINTER(HgModel), C.POINTER(HgConfig), dp, dp]
        L.hgo_energy_momentum.restype = None
        L.hgo_contact_slack.argtypes = [C.POINTER(HgModel), C.POINTER(HgConfig), dp, dp]
        L.hgo_contact_slack.restype = None
        L.hgo_seg_seg_closest.argtypes = [dp] * 6
        L.hgo_seg_seg_closest.restype = None
        L.hgo_kinematics.argtypes = [C.POINTER(HgModel), dp, dp, dp]
        L.hgo_kinematics.restype = None
        L.hgo_set_row_noise.argtypes = [C.c_double, C.c_uint64]
        L.hgo_set_row_noise.restype = None
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def counted_flops_per_env_step(model: FlatModel, cfg: SimConfig, actions, joint_sel=None, n_substeps: int = 1,
                               warmup: int = 0) -> dict:
    """Flops that the oracle's formulation executes per env step, counted by the instrumented build
    oracle/libhg_oracle_flops.so (-DHGO_COUNT_FLOPS): ONE env stepped through `actions` [T][n_sel]; the first
    `warmup` steps are not counted (e.g. the fall onto the ground).  Returns {"flops_per_env_step", "mean_contacts"}."""
    so = os.path.join(ROOT, "oracle", "libhg_oracle_flops.so")
    src = os.path.join(ROOT, "oracle", "hg_oracle.c")
    if (not os.path.exists(so)) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libhg_oracle_flops.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    L.hgo_reset.argtypes = [C.POINTER(HgModel), dp, C.c_int]
    L.hgo_reset.restype = None
    L.hgo_step.argtypes = lib().hgo_step.argtypes
    L.hgo_step.restype = C.c_int
    L.hgo_flops_read.restype = C.c_double
    L.hgo_flops_reset.restype = None
    cm, cc = model.to_c(), cfg.to_c()
    n = model.n_links
    state = np.zeros((1, HG_STATE_BASE + 3 * n))
    L.hgo_reset(C.byref(cm), _dp(state), 1)
    sel_p, n_sel = None, 0
    if joint_sel is not None:
        sel = np.ascontiguousarray(joint_sel, np.int32)
        sel_p, n_sel = sel.ctypes.data_as(C.POINTER(C.c_int32)), len(sel)
    nc = np.zeros(1, np.int32)
    acts = np.ascontiguousarray(actions, np.float64)
    contacts, counted = 0, 0
    for t in range(len(acts)):
        if t == warmup:
            L.hgo_flops_reset()
        rc = L.hgo_step(C.byref(cm), C.byref(cc), 1, _dp(state), _dp(acts[t].reshape(1, -1)), sel_p, n_sel, n_substeps,
                        None, None, None, None, None, None, nc.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc:
            raise RuntimeError(f"hgo_step failed: {rc}")
        if t >= warmup:
            contacts += int(nc[0]); counted += 1
    return {"flops_per_env_step": L.hgo_flops_read() / max(counted, 1), "mean_contacts": contacts / max(counted, 1)}


class OracleSim:
    """Batched FP64 CPU simulation with the same step semantics as the HIP library."""

    def __init__(self, model: FlatModel, cfg: SimConfig, n_envs: int = 1):
        self.model, self.cfg, self.n_envs = model, cfg, n_envs
        self.cm, self.cc = model.to_c(), cfg.to_c()
        self.n = model.n_links
        self.S = HG_STATE_BASE + 3 * self.n
        self.state = np.zeros((n_envs, self.S))
        self.reset()

    def reset(self):
        lib().hgo_reset(C.byref(self.cm), _dp(self.state), self.n_envs)

    # convenient views
    @property
    def q(self):
        return self.state[:, HG_STATE_BASE:HG_STATE_BASE + self.n]

    @property
    def qd(self):
        return self.state[:, HG_STATE_BASE + self.n:HG_STATE_BASE + 2 * self.n]

    @property
    def targets(self):
        return self.state[:, HG_STATE_BASE + 2 * self.n:]

    def step(self, actions=None, joint_sel=None, n_substeps=1, base_cmd=None):
        na = self.model.n_act
        obs_q = np.zeros((self.n_envs, na))
        obs_qd = np.zeros((self.n_envs, na))
        xyyaw = np.zeros((self.n_envs, 3))
        col = np.zeros(self.n_envs, np.uint8)
        mask = np.zeros(self.n_envs, np.uint64)
        pairs = np.zeros((self.n_envs, HG_MAX_CONTACTS, 2), np.int32)
        nc = np.zeros(self.n_envs, np.int32)
        sel_p, n_sel = None, 0
        if joint_sel is not None:
            sel = np.ascontiguousarray(joint_sel, np.int32)
            sel_p, n_sel = sel.ctypes.data_as(C.POINTER(C.c_int32)), len(sel)
        act = None
        if actions is not None:
            act = np.ascontiguousarray(actions, np.float64).reshape(self.n_envs, -1)
            assert act.shape[1] == (n_sel if joint_sel is not None else na)
        cmd = None
        if base_cmd is not None:
            cmd = np.ascontiguousarray(base_cmd, np.float64).reshape(self.n_envs, 3)
        tau = np.zeros((self.n_envs, self.n))
        rc = lib().hgo_step_full(C.byref(self.cm), C.byref(self.cc), self.n_envs, _dp(self.state), _dp(act),
                                 sel_p, n_sel, n_substeps, _dp(obs_q), _dp(obs_qd), _dp(xyyaw),
                                 col.ctypes.data_as(C.POINTER(C.c_uint8)),
                                 mask.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 pairs.ctypes.data_as(C.POINTER(C.c_int32)),
                                 nc.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cmd), _dp(tau))
        if rc:
            raise RuntimeError(f"hgo_step failed: {rc}")
        return dict(q=obs_q, qd=obs_qd, base_xyyaw=xyyaw, collision=col, link_mask=mask,
                    pairs=pairs, n_contacts=nc, motor_torque=tau)

    def noise_spread(self, state_row, action_row=None, joint_sel=None, n_substeps=1, base_cmd_row=None, trials=64,
                     eps=1e-6, seed=0):
        """How far does the oracle's OWN answer for one sample move when its constraint rows carry the rounding errors
        of a float32 implementation?  `trials` copies of the sample are stepped with independent relative noise of
        standard deviation `eps` on every entry of every row's impulse response, right-hand side and diagonal
        (hgo_set_row_noise), plus one clean copy.  Returns the largest deviations from the clean answer:
        dict(q, qd, tau_rel, base) -- joint angles [rad], joint rates [rad/s], motor torques relative to
        max(1, largest clean torque), base x / y / yaw.  A kernel-oracle difference that is not larger than a small
        multiple of this is a property of the sample (a degenerate row set), not of the kernel."""
        sim = OracleSim(self.model, self.cfg, trials + 1)
        sim.state[:] = np.asarray(state_row, np.float64)[None, :]
        act = None if action_row is None else np.repeat(np.asarray(action_row, np.float64)[None, :], trials + 1, axis=0)
        cmd = None if base_cmd_row is None else np.repeat(np.asarray(base_cmd_row, np.float64)[None, :], trials + 1, axis=0)
        clean = OracleSim(self.model, self.cfg, 1)
        clean.state[:] = sim.state[:1]
        oc = clean.step(None if act is None else act[:1], joint_sel=joint_sel, n_substeps=n_substeps,
                        base_cmd=None if cmd is None else cmd[:1])
        lib().hgo_set_row_noise(float(eps), int(seed))
        try:
            o = sim.step(act, joint_sel=joint_sel, n_substeps=n_substeps, base_cmd=cmd)
        finally:
            lib().hgo_set_row_noise(0.0, 0)
        tscale = max(1.0, float(np.abs(oc["motor_torque"][0]).max()))
        dxy = o["base_xyyaw"] - oc["base_xyyaw"][0]
        dxy[:, 2] = (dxy[:, 2] + np.pi) % (2 * np.pi) - np.pi
        return dict(q=float(np.abs(sim.q - clean.q[0]).max()), qd=float(np.abs(sim.qd - clean.qd[0]).max()),
                    tau_rel=float(np.abs(o["motor_torque"] - oc["motor_torque"][0]).max()) / tscale,
                    base=float(np.abs(dxy).max()), same_contacts=bool((o["n_contacts"] == oc["n_contacts"][0]).all()))

    def link_states(self):
        nb = self.n + 1
        cp, cq = np.zeros((self.n_envs, nb, 3)), np.zeros((self.n_envs, nb, 4))
        fp, fq = np.zeros((self.n_envs, nb, 3)), np.zeros((self.n_envs, nb, 4))
        lib().hgo_link_states(C.byref(self.cm), _dp(self.state), self.n_envs, _dp(cp), _dp(cq), _dp(fp), _dp(fq))
        return cp, cq, fp, fq

    def forward_dynamics(self, env=0, tau=None):
        acc = np.zeros(6 + self.n)
        t = None if tau is None else np.ascontiguousarray(tau, np.float64)
        st = np.ascontiguousarray(self.state[env])
        rc = lib().hgo_forward_dynamics(C.byref(self.cm), C.byref(self.cc), _dp(st), _dp(t), _dp(acc))
        assert rc == 0
        return acc

    def apply_minv(self, f, env=0):
        out = np.zeros(6 + self.n)
        st = np.ascontiguousarray(self.state[env])
        f = np.ascontiguousarray(f, np.float64)
        rc = lib().hgo_apply_minv(C.byref(self.cm), C.byref(self.cc), _dp(st), _dp(f), _dp(out))
        assert rc == 0
        return out

    def energy_momentum(self, env=0):
        out = np.zeros(8)
        st = np.ascontiguousarray(self.state[env])
        lib().hgo_energy_momentum(C.byref(self.cm), C.byref(self.cc), _dp(st), _dp(out))
        return dict(kinetic=out[0], potential=out[1], p=out[2:5].copy(), L=out[5:8].copy())

    def contact_slack(self, env=0):
        """(smallest |distance - margin| of any contact candidate, smallest 4th/5th-deepest gap of a body)
        for the CURRENT pose of one env: how far its contact set is from changing."""
        out = np.zeros(2)
        st = np.ascontiguousarray(self.state[env])
        lib().hgo_contact_slack(C.byref(self.cm), C.byref(self.cc), _dp(st), _dp(out))
        return float(out[0]), float(out[1])

    def kinematics(self, env=0):
        Rw, xw = np.zeros((self.n + 1, 3, 3)), np.zeros((self.n + 1, 3))
        st = np.ascontiguousarray(self.state[env])
        lib().hgo_kinematics(C.byref(self.cm), _dp(st), _dp(Rw), _dp(xw))
        return Rw, xw


def seg_seg_closest(ca, ha, cb, hb):
    """Closest points (on a, on b) of two sphere-swept segments (centre, half axis) as oracle/hg_oracle.c computes them."""
    a = [np.ascontiguousarray(x, np.float64) for x in (ca, ha, cb, hb)]
    c1, c2 = np.zeros(3), np.zeros(3)
    lib().hgo_seg_seg_closest(*[_dp(x) for x in a], _dp(c1), _dp(c2))
    return c1, c2
