"""ctypes plumbing for the database writer/reader (include/meryl_db.h)."""
import ctypes

import numpy as np

from . import capi


class DbError(RuntimeError):
    pass


def _err(what):
    s = capi.lib().mdb_last_error()
    return DbError("%s: %s" % (what, s.decode("utf-8", "replace") if s else ""))


class Writer:
    """merylFileWriter + merylBlockWriter for the count path."""

    def __init__(self, path, k, w_prefix, label_size=0, part=0, n_parts=1):
        """part / n_parts: one writer of a sharded database (finish with merge_parts once all are closed)."""
        self._h = capi.lib().mdb_writer_open_ex(path.encode(), k, w_prefix, label_size, part, n_parts)
        if not self._h:
            raise _err("mdb_writer_open")

    def add_block(self, prefix, suffix_lo, counts, suffix_hi=None, label=0):
        slo = np.ascontiguousarray(suffix_lo, dtype=np.uint64)
        cnt = np.ascontiguousarray(counts, dtype=np.uint32)
        shi = None if suffix_hi is None else np.ascontiguousarray(suffix_hi, dtype=np.uint64)
        rc = capi.lib().mdb_writer_add_block_labelled(self._h, int(prefix), slo.size, slo.ctypes.data if slo.size else None,
                                                      shi.ctypes.data if shi is not None and shi.size else None,
                                                      cnt.ctypes.data if cnt.size else None, None, int(label))
        if rc != 0:
            raise _err("mdb_writer_add_block")

    def close(self):
        if self._h:
            rc = capi.lib().mdb_writer_close(self._h)
            self._h = None
            if rc != 0:
                raise _err("mdb_writer_close")


def merge_parts(path, n_parts):
    """Stitches the part files of a sharded database into the final 64+64+1 files (one caller, after every part closed)."""
    rc = capi.lib().mdb_merge_parts(path.encode(), int(n_parts))
    if rc != 0:
        raise _err("mdb_merge_parts")


def write_database(session, path, host_threads=8):
    """Count result of a meryl_amd.count.Session -> database directory."""
    rc = capi.lib().mgc_write_database(session._h, path.encode(), host_threads)
    if rc != 0:
        raise _err("mgc_write_database (%s)" % capi.last_error(session._h))


class Reader:
    def __init__(self, path):
        self._h = capi.lib().mdb_reader_open(path.encode())
        if not self._h:
            raise _err("mdb_reader_open")
        self.info = capi.DbInfo()
        capi.lib().mdb_reader_info(self._h, ctypes.byref(self.info))

    def histogram(self):
        n = self.info.hist_len
        v = np.zeros(n, dtype=np.uint64)
        o = np.zeros(n, dtype=np.uint64)
        if n:
            capi.lib().mdb_reader_histogram(self._h, v.ctypes.data, o.ctypes.data)
        return v, o

    def read_file(self, ff, labels=False):
        lo = ctypes.c_void_p()
        hi = ctypes.c_void_p()
        cn = ctypes.c_void_p()
        lb = ctypes.c_void_p()
        n = ctypes.c_uint64(0)
        rc = capi.lib().mdb_reader_read_file_ex(self._h, ff, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(cn),
                                                ctypes.byref(lb), ctypes.byref(n))
        if rc != 0:
            raise _err("mdb_reader_read_file")
        m = n.value

        def take(p, dtype):
            if m == 0:
                out = np.zeros(0, dtype=dtype)
            else:
                out = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dtype))),
                                            shape=(m,)).copy()
            capi.lib().mdb_free(p)
            return out

        out = (take(lo, np.uint64), take(hi, np.uint64), take(cn, np.uint32), take(lb, np.uint64))
        return out if labels else out[:3]

    def read_all(self, labels=False):
        cols = [[] for _ in range(4 if labels else 3)]
        for ff in range(64):
            for c, a in zip(cols, self.read_file(ff, labels)):
                c.append(a)
        return tuple(np.concatenate(c) for c in cols)

    def file_index(self, ff):
        """(prefix, position, n_kmers) rows of file ff's index"""
        n = 1 << self.info.num_blocks_bits
        arr = (capi.IndexEntry * n)()
        if capi.lib().mdb_reader_file_index(self._h, ff, arr) != 0:
            raise _err("mdb_reader_file_index")
        return [(e.prefix, e.position, e.n_kmers) for e in arr]

    def block_header(self, ff, position):
        h = capi.BlockHeader()
        if capi.lib().mdb_reader_block_header(self._h, ff, position, ctypes.byref(h)) != 0:
            raise _err("mdb_reader_block_header")
        return h

    def close(self):
        if self._h:
            capi.lib().mdb_reader_close(self._h)
            self._h = None
