"""Test / tool helper: the REGISTERED form of the convolution side inputs (scratch arena, per-sample max words, producer-written planes)
on top of the library's per-call form (nemar_conv2d_*_ex + nemar_conv_extras).  Rounds 2-3 exported nemar_set_scratch / nemar_absmax_hint /
nemar_planes_hint (process- / thread-wide registrations); the C ABI now only takes side inputs with the call.  Kernel tests and
measurement tools were written against the registered vocabulary — this proxy keeps their bodies unchanged:

    lib = SideInputs(_lib.load())
    lib.set_scratch(ptr, nbytes); lib.absmax_hint(tensor_ptr, words_ptr, count); lib.conv2d_fwd(...)   # -> nemar_conv2d_fwd_ex

Not part of the product (nemar_amd/ops.py passes nemar_conv_extras itself)."""
import ctypes

from nemar_amd._lib import ConvExtras


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    v = ctypes.cast(p, ctypes.c_void_p).value
    return v or 0


class SideInputs:
    def __init__(self, lib):
        self._lib = lib
        self._scratch = (0, 0)
        self._max = {}          # tensor address -> (words address, count)
        self._planes = {}       # tensor address -> planes address

    def __getattr__(self, name):
        return getattr(self._lib, name)

    # ---- the registered vocabulary ----
    def set_scratch(self, ptr, nbytes):
        self._scratch = (_addr(ptr), int(nbytes)) if nbytes else (0, 0)
        return 0

    def absmax_hint(self, tensor, words, count):
        if words is None or not count:
            self._max.pop(_addr(tensor), None)
        else:
            self._max[_addr(tensor)] = (_addr(words), int(count))
        return 0

    def planes_hint(self, tensor, planes, N, C, H, W):
        if planes is None:
            self._planes.pop(_addr(tensor), None)
        else:
            self._planes[_addr(tensor)] = _addr(planes)
        return 0

    def _extras(self, src, src2=None, planes_of=None):
        e = ConvExtras()
        e.scratch, e.scratch_bytes = (self._scratch[0] or None), self._scratch[1]
        w = self._max.get(_addr(src))
        if w:
            e.src_max_words, e.src_max_count = w
        if src2 is not None:
            w2 = self._max.get(_addr(src2))
            if w2:
                e.src2_max_words, e.src2_max_count = w2
        if planes_of is not None:
            pl = self._planes.get(_addr(planes_of))
            if pl:
                e.src_planes = pl
        return ctypes.byref(e)

    # ---- the three operators: per-call side inputs ----
    def conv2d_fwd(self, x0, *a):
        return self._lib.conv2d_fwd_ex(x0, *a, self._extras(x0, planes_of=x0))

    def conv2d_bwd_data(self, gy, *a):
        return self._lib.conv2d_bwd_data_ex(gy, *a, self._extras(gy))

    def conv2d_bwd_weight(self, x0, C0, x1, C1, gy, *a):
        return self._lib.conv2d_bwd_weight_ex(x0, C0, x1, C1, gy, *a, self._extras(x0, gy))
