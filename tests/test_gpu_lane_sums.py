"""The lane sums every reduction of the resident kernels ends in (gnnx_kernels.hpp: xor32_sum / xor16_sum) go through gfx950's lane swaps in INLINE
ASSEMBLY (v_permlane32_swap / v_permlane16_swap; the builtin's pair sum is miscompiled by ROCm 7.2) with a hand-written wait state: a compiler or
hardware change must not break them silently.  The library's self-check kernel computes both forms on random data; they must agree bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from gnn_model_explainer_amd import engine


def _run(lib, device):
    rng = np.random.default_rng(11)
    for scale in (1.0, 1e-20, 1e20):
        x = torch.tensor((rng.standard_normal(64) * scale).astype(np.float32), device=device)
        out = torch.zeros(256, dtype=torch.float32, device=device)
        engine._check(lib, lib.gnnx_debug_lane_sums(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), None))
        if device != "cpu":
            torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint32).reshape(4, 64)
        assert np.array_equal(o[0], o[1]) and np.array_equal(o[2], o[3])
        v = x.cpu().numpy()
        assert np.array_equal(out.cpu().numpy()[:64], v + v[np.arange(64) ^ 32]) and np.array_equal(out.cpu().numpy()[128:192], v + v[np.arange(64) ^ 16])


@pytest.mark.gpu
def test_lane_swap_sums_equal_the_shuffle_sums_on_gpu():
    _run(engine.get_library(), "cuda")


def test_lane_sums_on_the_emulator():
    from emu.emu_engine import emu_library
    _run(emu_library(), "cpu")
