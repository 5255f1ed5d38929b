"""TEST INFRASTRUCTURE ONLY: run the product's MaskOptimJob against libgnnx_emu.so (the product's own
HIP sources compiled for the CPU emulator) with host memory standing in for device memory."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

from gnn_model_explainer_amd import engine  # noqa: E402

_lib = None


def emu_library():
    global _lib
    if _lib is None:
        _lib = engine.bind(ctypes.CDLL(build_emu.build()))
    return _lib


def emu_job(subgraphs, state_dict, graph_mode=False, analyze=True):
    return engine.MaskOptimJob(subgraphs, state_dict, graph_mode=graph_mode, device="cpu", lib=emu_library(), analyze=analyze)
