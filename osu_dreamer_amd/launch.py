"""One process per GPU: turning `--gpus N` / `trainer.devices: N` into N ranks.

The reference's knob is `trainer.devices` (osu_dreamer/models/diffusion/model.yml:11), which Lightning
turns into one process per device.  Here the same knob starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... <script> <args>`
as a CHILD process — never an exec, and before anything in the parent has touched the GPU — unless the
process already is a rank of such a job (`WORLD_SIZE` set, e.g. when the caller ran torchrun itself).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence


def parse_devices(devices) -> int:
    """`trainer.devices` as Lightning accepts it: an int, a list of device indices, or "auto" / -1 (= every visible GPU; counting
    devices does not initialise the GPU)."""
    if isinstance(devices, (list, tuple)):
        return max(1, len(devices))
    if devices in ("auto", "-1", -1):
        import torch
        return max(1, torch.cuda.device_count())
    return max(1, int(devices))


def world_from_env() -> Optional[int]:
    w = os.environ.get("WORLD_SIZE")
    return int(w) if w is not None else None


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def torchrun_command(n: int, script_args: Sequence[str], module: Optional[str] = None, port: Optional[int] = None) -> List[str]:
    """The command line that runs `script_args` (or `-m module args`) as `n` local ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    if module is not None:
        cmd += ["-m", module]
    return cmd + list(script_args)


def check_world(requested: int) -> int:
    """World size this process runs in; raises when it contradicts the requested device count."""
    world = world_from_env() or 1
    if requested != world:
        raise SystemExit(f"asked for {requested} GPU rank(s) but WORLD_SIZE={world}: launch with "
                         f"`--nproc-per-node {requested}` or drop the torchrun wrapper and let --gpus/devices spawn the ranks")
    return world


def spawn_ranks_if_needed(n: int, script_args: Sequence[str], module: Optional[str] = None, env: Optional[dict] = None) -> Optional[int]:
    """If `n > 1` and this process is not already a rank, run the job as `n` torchrun children and return the
    job's exit code (the caller should `sys.exit` with it).  Returns None when the caller should carry on itself
    (n == 1, or already inside a rank).  Must be called before the first GPU call of the process."""
    if n <= 1 or world_from_env() is not None:
        return None
    e = dict(os.environ if env is None else env)
    # dmabuf IPC: on this driver RCCL's cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument` under the legacy
    # mode.  A value already in the environment wins (so the knob can be turned without editing code); what the ranks run with is logged.
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"[osu_dreamer_amd.launch] starting {n} ranks; HSA_ENABLE_IPC_MODE_LEGACY={e['HSA_ENABLE_IPC_MODE_LEGACY']}"
          f"{' (from the environment)' if 'HSA_ENABLE_IPC_MODE_LEGACY' in (os.environ if env is None else env) else ' (default)'}",
          file=sys.stderr, flush=True)
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(torchrun_command(n, script_args, module), env=e)
