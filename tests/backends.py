"""Two ways of driving the SAME C-ABI entry points with the SAME test bodies:
  EmuBackend — host-emulated build of the kernel sources (tests/emu), numpy buffers, CPU test tier;
  HipBackend — the real gfx950 library, torch CUDA tensors, `-m gpu` tier."""
import ctypes
import numpy as np

from side_inputs import SideInputs


class EmuBackend:
    name = "emu"
    stream = None

    def __init__(self, lib):
        self.lib = SideInputs(lib)      # (registered side inputs -> the per-call form of the C ABI)

    def dev(self, a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float32))

    def dev_i32(self, a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.int32))

    def zeros(self, *shape):
        return np.zeros(shape, dtype=np.float32)

    def full(self, shape, v):
        return np.full(shape, v, dtype=np.float32)

    def ptr(self, h):
        return None if h is None else h.ctypes.data_as(ctypes.c_void_p)

    def np(self, h):
        return np.array(h, dtype=np.float64)

    def raw(self, h):
        """the buffer's bytes, bit for bit"""
        return np.array(h).view(np.uint8).copy()

    def bytes_buf(self, nbytes):
        return np.zeros(max(1, (nbytes + 3) // 4), dtype=np.float32)

    def sync(self):
        pass


class HipBackend:
    name = "hip"

    def __init__(self, lib):
        import torch
        self.torch = torch
        self.lib = SideInputs(lib)      # (registered side inputs -> the per-call form of the C ABI)
        self.device = torch.device("cuda:0")

    @property
    def stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def dev(self, a):
        return self.torch.tensor(np.asarray(a, dtype=np.float32), device=self.device).contiguous()

    def dev_i32(self, a):
        return self.torch.tensor(np.asarray(a, dtype=np.int32), device=self.device).contiguous()

    def zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.float32, device=self.device)

    def full(self, shape, v):
        return self.torch.full(shape, float(v), dtype=self.torch.float32, device=self.device)

    def ptr(self, h):
        return None if h is None else ctypes.c_void_p(h.data_ptr())

    def np(self, h):
        return h.detach().cpu().numpy().astype(np.float64)

    def raw(self, h):
        """the buffer's bytes, bit for bit"""
        return h.detach().contiguous().view(self.torch.uint8).cpu().numpy().copy()

    def bytes_buf(self, nbytes):
        return self.torch.zeros(max(1, (nbytes + 3) // 4), dtype=self.torch.float32, device=self.device)

    def sync(self):
        self.torch.cuda.synchronize()
