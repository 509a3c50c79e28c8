"""hg_stats_accumulate_noalloc (include/hg_sim.h): the one entry point that allocates nothing, for callers
that cannot call hipMalloc.  Called through the C-ABI and checked against hg_stats_accumulate on the same
inputs, on two calls in a row, the second on a smaller sim: the static scratch array keeps the first call's
partials, which the second call must not pick up."""
import ctypes as C

import numpy as np
import pytest

from humanoid_gym_amd.robots import load_robot

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def test_noalloc_statistics_match_the_allocating_entry_point():
    import torch
    from humanoid_gym_amd import _capi
    from humanoid_gym_amd._capi import HgSim
    L = _capi.load_library()
    model, cfg = load_robot("dancer")
    g = torch.Generator(device="cpu").manual_seed(3)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    results = []
    for n_envs in (5000, 300):                      # second call: smaller input, fewer partial slots than the first left behind
        sim = HgSim(model, cfg, n_envs, "cuda:0")
        reward = torch.rand(n_envs, generator=g).to("cuda:0") * 3.0 - 1.0
        done = (torch.rand(n_envs, generator=g) < 0.3).to(torch.uint8).to("cuda:0")
        col = (torch.rand(n_envs, generator=g) < 0.6).to(torch.uint8).to("cuda:0")
        ref = torch.zeros(8, device="cuda:0"); ref[5] = 7.0            # entries 4..7 are not touched by either
        new = ref.clone()
        assert L.hg_stats_accumulate(sim._h, _ptr(reward), _ptr(done), _ptr(col), _ptr(ref), stream) == 0
        free0, _ = torch.cuda.mem_get_info()
        assert L.hg_stats_accumulate_noalloc(sim._h, _ptr(reward), _ptr(done), _ptr(col), _ptr(new), stream) == 0
        again = torch.zeros(8, device="cuda:0"); again[5] = 7.0
        assert L.hg_stats_accumulate_noalloc(sim._h, _ptr(reward), _ptr(done), _ptr(col), _ptr(again), stream) == 0
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0                                          # no device memory was taken
        r, n = ref.cpu().numpy(), new.cpu().numpy()
        assert n[0] == r[0] == n_envs and n[2] == r[2] == float(done.sum()) and n[3] == r[3] == float(col.sum())   # integer sums: exact
        want = float(reward.double().sum())
        assert abs(n[1] - want) <= 1e-5 * max(1.0, float(reward.abs().double().sum()))          # FP32 sum, fixed order
        assert abs(r[1] - want) <= 1e-5 * max(1.0, float(reward.abs().double().sum()))
        assert np.array_equal(n[4:], r[4:]) and n[5] == 7.0
        assert torch.equal(new, again)                                 # no atomics: bit-reproducible
        # nullable inputs, like hg_stats_accumulate
        a, b = torch.zeros(8, device="cuda:0"), torch.zeros(8, device="cuda:0")
        assert L.hg_stats_accumulate(sim._h, None, None, _ptr(col), _ptr(a), stream) == 0
        assert L.hg_stats_accumulate_noalloc(sim._h, None, None, _ptr(col), _ptr(b), stream) == 0
        assert torch.equal(a, b)
        # accumulates into the vector rather than overwriting it
        assert L.hg_stats_accumulate_noalloc(sim._h, _ptr(reward), _ptr(done), _ptr(col), _ptr(new), stream) == 0
        assert new[0].item() == 2 * n_envs
        results.append(n.copy())
        sim.close()
    assert results[0][0] == 5000 and results[1][0] == 300
    assert L.hg_stats_accumulate_noalloc(None, None, None, None, None, None) == -1
    big = HgSim(model, cfg, (1 << 20) + 1, "cuda:0")                   # beyond HG_NOALLOC_MAX_ENVS: refused, clearly
    out = torch.zeros(8, device="cuda:0")
    assert L.hg_stats_accumulate_noalloc(big._h, None, None, None, _ptr(out), stream) == -4
    assert b"HG_NOALLOC_MAX_ENVS" in L.hg_last_error()
    assert L.hg_stats_accumulate(big._h, None, None, None, _ptr(out), stream) == 0 and out[0].item() == (1 << 20) + 1
    big.close()
