"""Shared plumbing for kernel tests: the same test bodies run against
  * the SIMT-emulator build of the kernel sources (CPU, `-m "not gpu"`), and
  * the real gfx950 library on an MI355X (`-m gpu`).
"""
import os
import subprocess

import pytest
import torch

from osu_dreamer_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(REPO, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libod_emu.so")


_built = False


def build_emu():
    """Build (once per process, one process at a time: pytest-xdist workers share the tree) the emulator library."""
    global _built
    if not _built:
        import fcntl
        with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["bash", os.path.join(EMU_DIR, "build_emu.sh")], stdout=subprocess.DEVNULL)
        _built = True
    return EMU_SO


BACKENDS = [
    pytest.param("emu", id="emu"),
    pytest.param("hip", id="hip", marks=pytest.mark.gpu),
]


@pytest.fixture(params=BACKENDS)
def dev(request):
    """Binds the library for the backend and returns the torch device to allocate on."""
    if request.param == "emu":
        _lib.use_library(build_emu())
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("hip backend selected but no GPU is visible")
    _lib._lib = None          # force the real library (raises if it was not built)
    _lib.lib()
    return torch.device("cuda:0")


def frames(x):   # (B,C,L) -> [B*L, C]
    B, C, L = x.shape
    return x.permute(0, 2, 1).reshape(B * L, C).contiguous()


def unframes(x, B, L):   # [B*L, C] -> (B,C,L)
    return x.reshape(B, L, -1).permute(0, 2, 1).contiguous()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
