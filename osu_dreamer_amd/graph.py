"""hipGraph capture of a fixed launch sequence (the sampler's step body) through the C ABI's
od_graph_* helpers — hipStreamBeginCapture / EndCapture / GraphInstantiate / GraphLaunch.
torch supplies only the side stream the capture runs on."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class CapturedLoop:
    def __init__(self, step_fn, device):
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.stream.wait_stream(torch.cuda.current_stream(device))
        self._exec = ctypes.c_void_p()
        L = _lib.lib()
        with torch.cuda.stream(self.stream):
            L.od_graph_begin(self.stream.cuda_stream)
            try:
                step_fn()                      # recorded, not executed
            finally:
                L.od_graph_end(self.stream.cuda_stream, ctypes.byref(self._exec))

    def begin(self):
        """Order the side stream after everything already queued on the caller's stream."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def replay(self):
        _lib.lib().od_graph_launch(self._exec, self.stream.cuda_stream)

    def end(self):
        """Make the caller's stream wait for the replays."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def close(self):
        self.end()
        if self._exec:
            self.stream.synchronize()
            _lib.lib().od_graph_destroy(self._exec)
            self._exec = ctypes.c_void_p()
