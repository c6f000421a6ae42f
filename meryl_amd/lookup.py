"""ctypes/torch plumbing of the exact k-mer lookup table (include/meryl_lookup.h)."""
import ctypes

import numpy as np

from . import capi

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

U64_MAX = (1 << 64) - 1


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Lookup:
    """merylExactLookup on the device: load(db, min, max) / value(kmers) / per-window stream lookups / -existence."""

    def __init__(self, handle):
        if not handle:
            raise capi.MgcError(-1, "mgc_lookup", capi.lib().mgc_lookup_error().decode("utf-8", "replace"))
        self._h = handle
        self.info = capi.LookupInfo()
        capi.check(capi.lib().mgc_lookup_get_info(self._h, ctypes.byref(self.info)), "mgc_lookup_get_info")

    @classmethod
    def load(cls, db_path, min_value=0, max_value=U64_MAX, device=-1, host_threads=16):
        return cls(capi.lib().mgc_lookup_load(db_path.encode(), min_value, max_value, device, host_threads))

    @classmethod
    def from_device(cls, keys, counts, k, min_value=0, max_value=U64_MAX):
        torch.cuda.current_stream(keys.device).synchronize()
        return cls(capi.lib().mgc_lookup_from_device(_ptr(keys), _ptr(counts), keys.shape[0], k, min_value, max_value, -1))

    def close(self):
        if self._h:
            capi.lib().mgc_lookup_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def values(self, kmers):
        """kmers: int64[n] (or int64[n, 2] {lo, hi}) cuda tensor -> int32[n] values (0 = absent)"""
        out = torch.empty(kmers.shape[0], dtype=torch.int32, device=kmers.device)
        capi.check(capi.lib().mgc_lookup_values(self._h, _ptr(kmers), kmers.shape[0], _ptr(out), _stream()), "mgc_lookup_values")
        return out

    def stream(self, bases):
        """uint8 cuda tensor of bases -> int32[n_bases]: value of the k-mer starting at every base (0: absent / broken)"""
        out = torch.empty(bases.numel(), dtype=torch.int32, device=bases.device)
        capi.check(capi.lib().mgc_lookup_stream(self._h, _ptr(bases), bases.numel(), _ptr(out), _stream()), "mgc_lookup_stream")
        return out

    def existence(self, bases, seq_start):
        """seq_start: int64[n_seq + 1] offsets into `bases` -> (total k-mers, k-mers found) per sequence as numpy uint64"""
        ss = torch.as_tensor(np.asarray(seq_start, dtype=np.int64)).to(bases.device)
        n = ss.numel() - 1
        tot = torch.empty(max(n, 1), dtype=torch.int64, device=bases.device)
        fnd = torch.empty(max(n, 1), dtype=torch.int64, device=bases.device)
        capi.check(capi.lib().mgc_lookup_existence(self._h, _ptr(bases), bases.numel(), _ptr(ss), n, _ptr(tot), _ptr(fnd), _stream()),
                   "mgc_lookup_existence")
        return tot[:n].cpu().numpy().view(np.uint64), fnd[:n].cpu().numpy().view(np.uint64)
