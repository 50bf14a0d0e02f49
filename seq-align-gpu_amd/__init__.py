"""ctypes binding of libswg (include/swg.h, include/swg_host.h).

This package is plumbing for tests and bench.py: the product is the C-ABI
library next to this file.  Nothing here computes an alignment; if libswg.so
is missing the import fails loudly, and `Context()` raises when there is no
GPU (the library has no CPU backend).

The directory name has a hyphen (it is the name the build contract gives), so
it is imported through `swg_loader.load()` at the repo root, as module
`seq_align_gpu_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libswg.so")
DATA_DIR = os.path.join(_HERE, "data")
CLI_PATH = os.path.join(_HERE, "bin", "smith_waterman")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libswg.so is not built: run `python seq-align-gpu_amd/build.py` "
        "(or __graft_entry__.build()); there is no Python/CPU fallback")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

SWG_OK, SWG_ERR_ARG, SWG_ERR_HIP, SWG_ERR_NOMEM = 0, -1, -2, -3
SWG_ERR_STATE, SWG_ERR_RESIDUE, SWG_ERR_IO, SWG_ERR_NODEVICE = -4, -5, -6, -7

# every symbol declared in include/swg.h and include/swg_host.h
ABI_SYMBOLS = [
    "swg_create", "swg_destroy", "swg_last_error", "swg_global_error", "swg_abi_version",
    "swg_set_option", "swg_set_scoring", "swg_set_query", "swg_db_pack", "swg_db_pack_shard", "swg_db_pack_shards", "swg_db_upload",
    "swg_db_free", "swg_db_save", "swg_db_load", "swg_db_count", "swg_db_total_count", "swg_db_residues",
    "swg_db_packed_bytes", "swg_db_order", "swg_search", "swg_search_begin", "swg_search_end", "swg_search_multi",
    "swg_fill_batches16", "swg_align_hits", "swg_align_ops_bound", "swg_hit_key",
    "swg_key_hit", "swg_topk_merge_keys",
    "swg_group_create", "swg_group_destroy", "swg_group_size", "swg_group_last_error", "swg_group_set_option",
    "swg_group_set_scoring", "swg_group_set_query", "swg_group_load", "swg_group_search",
    "swg_group_align_hits", "swg_group_align_ops_bound",
    "swg_letter_index", "swg_index_letter", "swg_scoring_init", "swg_scoring_add",
    "swg_scoring_load_matrix", "swg_query_sanitize", "swg_seqs_read", "swg_seqs_free",
    "swg_seqs_to_indices", "swg_synth_db", "swg_synth_query", "swg_synth_db_similar", "swg_synth_db_family", "swg_synth_db_shard",
    "swg_synth_free", "swg_host_threads",
]


class SwgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libswg error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("reserved", C.c_int * 7)]


class Hit(C.Structure):
    _fields_ = [("score", C.c_int32), ("index", C.c_uint32)]


class Alignment(C.Structure):
    _fields_ = [("score", C.c_int32), ("index", C.c_uint32), ("q_begin", C.c_uint32), ("q_end", C.c_uint32),
                ("d_begin", C.c_uint32), ("d_end", C.c_uint32), ("n_ops", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("cells", C.c_uint64), ("cells_padded", C.c_uint64), ("bytes_alg", C.c_uint64),
        ("n_rescored", C.c_uint64), ("fill_ms", C.c_double), ("rescore_ms", C.c_double),
        ("topk_ms", C.c_double), ("total_ms", C.c_double), ("path_bits", C.c_int32),
        ("cols_per_wave", C.c_int32), ("waves", C.c_int32), ("passes", C.c_int32),
        ("workgroups", C.c_int32), ("engine", C.c_int32), ("group_lanes", C.c_int32),
        ("streams", C.c_int32), ("long_pairs", C.c_int32), ("long_cols_per_lane", C.c_int32),
        ("long_streams", C.c_int32), ("work_queue", C.c_int32), ("classes_overlapped", C.c_int32),
        ("fill_launches", C.c_int32),
        ("cell_form", C.c_int32),
        ("split_rows", C.c_int32),
        ("fill_f16_launches", C.c_int32),
        ("fill_f16_ms", C.c_double),
        ("cells_f16", C.c_uint64),
        ("last_pass_cols", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


class Batch16(C.Structure):
    _fields_ = [("db_idx_t", C.c_void_p), ("max_len", C.c_size_t), ("vector_size", C.c_size_t),
                ("max_scores", C.c_void_p)]


class Scoring(C.Structure):
    _fields_ = [("gap_open", C.c_int), ("gap_extend", C.c_int), ("match", C.c_int),
                ("mismatch", C.c_int), ("sub", (C.c_int8 * 32) * 32), ("set", C.c_uint32 * 32)]

    def table(self):
        return np.ctypeslib.as_array(self.sub).reshape(32, 32).copy()


class Seqs(C.Structure):
    _fields_ = [("n", C.c_size_t), ("names", C.c_void_p), ("name_off", C.POINTER(C.c_uint64)),
                ("seq", C.c_void_p), ("seq_off", C.POINTER(C.c_uint64))]


def _sig(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


_vp = C.c_void_p
_sig("swg_create", C.c_int, [C.POINTER(Config), C.POINTER(_vp)])
_sig("swg_destroy", None, [_vp])
_sig("swg_last_error", C.c_char_p, [_vp])
_sig("swg_global_error", C.c_char_p, [])
_sig("swg_abi_version", C.c_int, [])
_sig("swg_set_option", C.c_int, [_vp, C.c_char_p, C.c_long])
_sig("swg_set_scoring", C.c_int, [_vp, _vp, C.c_int, C.c_int])
_sig("swg_set_query", C.c_int, [_vp, _vp, C.c_size_t])
_sig("swg_db_pack", C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(_vp)])
_sig("swg_db_pack_shard", C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(_vp)])
_sig("swg_db_pack_shards", C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.POINTER(_vp)])
_sig("swg_db_upload", C.c_int, [_vp, _vp])
_sig("swg_db_free", None, [_vp])
_sig("swg_db_save", C.c_int, [_vp, C.c_char_p])
_sig("swg_db_load", C.c_int, [C.c_char_p, C.POINTER(_vp)])
_sig("swg_db_count", C.c_size_t, [_vp])
_sig("swg_db_total_count", C.c_size_t, [_vp])
_sig("swg_db_residues", C.c_uint64, [_vp])
_sig("swg_db_packed_bytes", C.c_uint64, [_vp])
_sig("swg_db_order", C.POINTER(C.c_uint32), [_vp])
_sig("swg_search", C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)])
_sig("swg_search_begin", C.c_int, [_vp, _vp, C.c_int, C.c_size_t, C.POINTER(C.c_int)])
_sig("swg_search_end", C.c_int, [_vp, C.c_int, _vp, _vp, C.POINTER(C.c_size_t), C.POINTER(Stats)])
_sig("swg_search_multi", C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.POINTER(Stats)])
_sig("swg_fill_batches16", C.c_int, [_vp, C.POINTER(Batch16), C.c_size_t, C.POINTER(C.c_double)])
_sig("swg_align_hits", C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t])
_sig("swg_align_ops_bound", C.c_size_t, [_vp, _vp])
_sig("swg_hit_key", C.c_uint64, [C.c_int32, C.c_uint32])
_sig("swg_key_hit", None, [C.c_uint64, C.POINTER(Hit)])
_sig("swg_topk_merge_keys", C.c_size_t, [_vp, C.c_size_t, C.c_size_t, _vp])
_sig("swg_group_align_hits", C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t])
_sig("swg_group_align_ops_bound", C.c_size_t, [_vp])
_sig("swg_group_create", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)])
_sig("swg_group_destroy", None, [_vp])
_sig("swg_group_size", C.c_int, [_vp])
_sig("swg_group_last_error", C.c_char_p, [_vp])
_sig("swg_group_set_option", C.c_int, [_vp, C.c_char_p, C.c_long])
_sig("swg_group_set_scoring", C.c_int, [_vp, _vp, C.c_int, C.c_int])
_sig("swg_group_set_query", C.c_int, [_vp, _vp, C.c_size_t])
_sig("swg_group_load", C.c_int, [_vp, _vp, _vp, C.c_size_t])
_sig("swg_group_search", C.c_int, [_vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)])
_sig("swg_letter_index", C.c_int, [C.c_int])
_sig("swg_index_letter", C.c_int, [C.c_int])
_sig("swg_scoring_init", None, [C.POINTER(Scoring)])
_sig("swg_scoring_add", C.c_int, [C.POINTER(Scoring), C.c_int, C.c_int, C.c_int])
_sig("swg_scoring_load_matrix", C.c_int, [C.POINTER(Scoring), C.c_char_p, C.c_char_p, C.c_size_t])
_sig("swg_query_sanitize", None, [C.POINTER(Scoring), _vp, C.c_size_t])
_sig("swg_seqs_read", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(Seqs), C.c_char_p, C.c_size_t])
_sig("swg_seqs_free", None, [C.POINTER(Seqs)])
_sig("swg_seqs_to_indices", C.c_int, [C.POINTER(Seqs), _vp, C.c_char_p])
_sig("swg_synth_db", C.c_int, [C.c_uint64, C.c_size_t, C.c_double, C.c_double, C.c_uint32,
                               C.c_uint32, C.POINTER(_vp), C.POINTER(_vp)])
_sig("swg_synth_query", None, [C.c_uint64, C.c_size_t, _vp])
_sig("swg_synth_db_similar", C.c_int, [C.c_uint64, C.c_size_t, C.c_double, C.c_double, C.c_uint32,
                                       C.c_uint32, _vp, C.c_size_t, C.c_double, C.c_double,
                                       C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)])
_sig("swg_synth_db_family", C.c_int, [C.c_uint64, C.c_size_t, C.c_double, C.c_double, C.c_uint32,
                                      C.c_uint32, _vp, C.c_size_t, C.c_double, C.c_double, C.c_double,
                                      C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)])
_sig("swg_synth_db_shard", C.c_int, [C.c_uint64, C.c_size_t, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                     _vp, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int,
                                     C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)])
_sig("swg_synth_free", None, [_vp])
# test hook, declared in csrc/swg_host_internal.h (not part of the public ABI)
_sig("swg_debug_fail_alloc", None, [C.c_int])
_sig("swg_debug_sort_count", C.c_ulong, [])
_sig("swg_debug_engine", C.c_int, [_vp, C.c_size_t, C.c_int, C.c_int, _vp])
_sig("swg_debug_plan", C.c_int, [_vp, C.c_size_t, C.c_int, _vp])
_sig("swg_debug_split", C.c_int, [_vp, C.c_size_t, C.c_uint64, _vp])
_sig("swg_debug_list_plan", C.c_int, [C.c_size_t, C.c_uint32, C.c_int, _vp, _vp])
_sig("swg_debug_pair_tokens", C.c_int, [_vp, _vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t)])


def _check(rc, ctx=None):
    if rc != SWG_OK:
        msg = lib.swg_last_error(ctx) if ctx else lib.swg_global_error()
        raise SwgError(rc, (msg or b"").decode("utf-8", "replace"))


def _i8(a):
    a = np.ascontiguousarray(a, dtype=np.int8)
    return a, a.ctypes.data_as(_vp)


# ---------------------------------------------------------------------------
# host helpers
# ---------------------------------------------------------------------------
def letters_to_indices(s):
    """Reference letters_to_index over a string; raises on an illegal letter."""
    out = np.empty(len(s), dtype=np.int8)
    for i, ch in enumerate(s):
        v = lib.swg_letter_index(ord(ch))
        if v < 0:
            raise SwgError(SWG_ERR_RESIDUE, "illegal residue %r" % ch)
        out[i] = v
    return out


def load_scoring(name_or_path, gap_open=-2, gap_extend=-1):
    """Scoring struct from a matrix file (bundled name like 'BLOSUM62' or a path)."""
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(DATA_DIR, name_or_path + ".txt")
    sc = Scoring()
    lib.swg_scoring_init(C.byref(sc))
    err = C.create_string_buffer(512)
    rc = lib.swg_scoring_load_matrix(C.byref(sc), path.encode(), err, 512)
    if rc != SWG_OK:
        raise SwgError(rc, err.value.decode())
    sc.gap_open, sc.gap_extend = gap_open, gap_extend
    return sc


def read_seqs(path, max_records=0):
    """-> (names list, residue letters bytes, offsets uint64[n+1])."""
    s = Seqs()
    err = C.create_string_buffer(512)
    rc = lib.swg_seqs_read(path.encode(), max_records, C.byref(s), err, 512)
    if rc != SWG_OK:
        raise SwgError(rc, err.value.decode())
    try:
        n = s.n
        seq_off = np.ctypeslib.as_array(s.seq_off, shape=(n + 1,)).copy()
        name_off = np.ctypeslib.as_array(s.name_off, shape=(n + 1,)).copy()
        seq = C.string_at(s.seq, int(seq_off[n])) if n else b""
        raw = C.string_at(s.names, int(name_off[n])) if n else b""
        names = [raw[int(name_off[i]):int(name_off[i + 1]) - 1].decode("utf-8", "replace") for i in range(n)]
        idx = np.empty(len(seq), dtype=np.int8)
        bad = C.create_string_buffer(2)
        rc = lib.swg_seqs_to_indices(C.byref(s), idx.ctypes.data_as(_vp), bad)
        if rc != SWG_OK:
            raise SwgError(rc, "illegal residue %r" % bad.value)
    finally:
        lib.swg_seqs_free(C.byref(s))
    return names, seq, idx, seq_off


class _Owned:
    """A buffer malloc'ed by the library, released with swg_synth_free when the last array over it dies."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes
        self.buf = (C.c_uint8 * nbytes).from_address(ptr.value)

    def __del__(self):
        if self.ptr is not None:
            lib.swg_synth_free(self.ptr)
            self.ptr = None


def _take(ptr, nbytes, dtype):
    """numpy array over a library-owned buffer WITHOUT copying it (a 10M-sequence database is 3.8 GB);
    the buffer is freed when the array (and every view of it) is gone."""
    if not nbytes:
        lib.swg_synth_free(ptr)
        return np.zeros(0, dtype=dtype)
    owner = _Owned(ptr, nbytes)
    arr = np.frombuffer(owner.buf, dtype=dtype)     # keeps owner.buf alive ...
    _OWNERS[id(owner.buf)] = owner                  # ... and the owner lives as long as its buffer does
    import weakref
    weakref.finalize(arr, _OWNERS.pop, id(owner.buf), None)
    return arr


_OWNERS = {}


def synth_db(seed, n, median=290.0, sigma_ln=0.75, min_len=20, max_len=5000, query=None,
             fraction=0.0, subst=0.05, subst_hi=None):
    """Synthetic database -> (flat int8, offsets uint64[n+1][, n_planted]).
    subst_hi: the planted sequences are a family of relatives, each with its own substitution rate
    from [subst, subst_hi]."""
    flat, off = _vp(), _vp()
    if query is None or fraction <= 0.0:
        _check(lib.swg_synth_db(seed, n, median, sigma_ln, min_len, max_len, C.byref(flat), C.byref(off)))
        planted = None
    else:
        q, qp = _i8(query)
        npl = C.c_size_t(0)
        if subst_hi is not None:
            _check(lib.swg_synth_db_family(seed, n, median, sigma_ln, min_len, max_len, qp, len(q),
                                           fraction, subst, subst_hi, C.byref(flat), C.byref(off), C.byref(npl)))
        else:
            _check(lib.swg_synth_db_similar(seed, n, median, sigma_ln, min_len, max_len, qp, len(q),
                                            fraction, subst, C.byref(flat), C.byref(off), C.byref(npl)))
        planted = npl.value
    offsets = _take(off, (n + 1) * 8, np.uint64)
    residues = _take(flat, int(offsets[n]), np.int8)
    return (residues, offsets) if planted is None else (residues, offsets, planted)


def synth_db_shard(seed, n, shard_rank, shard_count, median=290.0, sigma_ln=0.75, min_len=20, max_len=5000,
                   query=None, fraction=0.0, subst=0.05):
    """One shard (global bins b % shard_count == shard_rank) of the database synth_db(seed, n, ...) would
    return, without generating the rest -> dict(flat, offsets[n_local+1], index[n_local] global indices,
    n_total, residues_total, planted)."""
    flat, off, idx = _vp(), _vp(), _vp()
    nl, tot, npl = C.c_size_t(0), C.c_uint64(0), C.c_size_t(0)
    if query is not None and fraction > 0.0:
        q, qp = _i8(query)
        lq = len(q)
    else:
        qp, lq, fraction = None, 0, 0.0
    _check(lib.swg_synth_db_shard(seed, n, median, sigma_ln, min_len, max_len, qp, lq, fraction, subst,
                                  shard_rank, shard_count, C.byref(flat), C.byref(off), C.byref(idx),
                                  C.byref(nl), C.byref(tot), C.byref(npl)))
    n_local = nl.value
    offsets = _take(off, (n_local + 1) * 8, np.uint64)
    index = _take(idx, n_local * 4, np.uint32)
    residues = _take(flat, int(offsets[n_local]), np.int8)
    return dict(flat=residues, offsets=offsets, index=index, n_total=n, residues_total=int(tot.value),
                planted=int(npl.value))


def debug_list_plan(lq, n_pairs_guess, n_cu=256, main=(32, 16, 4)):
    """Test hook: geometry of the int16 re-run of the pairs the f16 cells flagged (no device needed)
    -> dict(K, G, W, passes)."""
    m = np.asarray(main, dtype=np.int32)
    out = np.zeros(4, dtype=np.int32)
    _check(lib.swg_debug_list_plan(lq, n_pairs_guess, n_cu, m.ctypes.data_as(_vp), out.ctypes.data_as(_vp)))
    return {"K": int(out[0]), "G": int(out[1]), "W": int(out[2]), "passes": int(out[3])}


def synth_query(seed, lq):
    out = np.empty(lq, dtype=np.int8)
    lib.swg_synth_query(seed, lq, out.ctypes.data_as(_vp))
    return out


def hit_key(score, index):
    return int(lib.swg_hit_key(int(score), int(index)))


def key_hit(key):
    h = Hit()
    lib.swg_key_hit(int(key), C.byref(h))
    return int(h.score), int(h.index)


def topk_merge_keys(keys, k):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    out = (Hit * max(k, 1))()
    m = lib.swg_topk_merge_keys(keys.ctypes.data_as(_vp), keys.size, k, C.cast(out, _vp))
    return [(int(out[i].score), int(out[i].index)) for i in range(m)]


# ---------------------------------------------------------------------------
# database + context
# ---------------------------------------------------------------------------
class Database:
    """Host-packed database shard (swg_db); `upload(ctx)` makes it resident."""

    def __init__(self, flat=None, offsets=None, shard_rank=0, shard_count=1, path=None, index=None, n_total=None):
        """flat/offsets: the whole database, of which bins b % shard_count == shard_rank are kept; or, with
        index (global index of every sequence given) and n_total, a shard that was cut elsewhere."""
        self.handle = None
        if path is not None:            # load a packed-database file written by save()
            h = _vp()
            _check(lib.swg_db_load(path.encode(), C.byref(h)))
            self.handle = h
            return
        self._flat, fp = _i8(flat)
        self._off = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = self._off.size - 1
        h = _vp()
        if index is not None:
            ix = np.ascontiguousarray(index, dtype=np.uint32)
            assert ix.size == n
            _check(lib.swg_db_pack_shard(fp, self._off.ctypes.data_as(_vp), n, ix.ctypes.data_as(_vp),
                                         int(n_total), C.byref(h)))
        else:
            _check(lib.swg_db_pack(fp, self._off.ctypes.data_as(_vp), n, shard_rank, shard_count, C.byref(h)))
        self.handle = h
        self._flat = None  # the library copied what it needs

    @classmethod
    def pack_shards(cls, flat, offsets, shard_count):
        """All shards of one database from ONE global sort (what swg_group_load uses) -> [Database] * shard_count."""
        f, fp = _i8(flat)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        hs = (_vp * shard_count)()
        _check(lib.swg_db_pack_shards(fp, off.ctypes.data_as(_vp), off.size - 1, shard_count, hs))
        out = []
        for h in hs:
            d = cls.__new__(cls)
            d.handle = _vp(h)
            out.append(d)
        return out

    count = property(lambda self: lib.swg_db_count(self.handle))
    total_count = property(lambda self: lib.swg_db_total_count(self.handle))
    residues = property(lambda self: lib.swg_db_residues(self.handle))
    packed_bytes = property(lambda self: lib.swg_db_packed_bytes(self.handle))

    def order(self):
        if self.count == 0:                 # a shard may hold nothing (fewer bins than shards)
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(lib.swg_db_order(self.handle), shape=(self.count,)).copy()

    def debug_plan(self, lq, n_cu=256):
        """Test hook: the cost model's first choice for this database and a query of lq residues (no device needed)."""
        out = np.zeros(13, dtype=np.int32)
        _check(lib.swg_debug_plan(self.handle, lq, n_cu, out.ctypes.data_as(_vp)))
        keys = ("classes", "K", "G", "W", "passes", "workgroups", "long_pairs", "long_K", "long_G", "long_W", "long_workgroups", "est_us",
                "last_pass_cols")
        return dict(zip(keys, (int(v) for v in out)))

    def debug_engine(self, lq, n_cu=256, form=2):
        """Test hook: the cost model's estimates of both engines for this database and a query of lq residues (no device
        needed) -> dict(diag_us, systolic_us, systolic_K, systolic: the model picks the systolic engine)."""
        out = np.zeros(4, dtype=np.int32)
        _check(lib.swg_debug_engine(self.handle, lq, n_cu, form, out.ctypes.data_as(_vp)))
        return {"diag_us": int(out[0]), "systolic_us": int(out[1]), "systolic_K": int(out[2]), "systolic": bool(out[3])}

    def debug_split(self, lq, qbound):
        """Test hook: the both-forms cut (a query of lq columns that can score qbound at best) -> rows, first pair of the
        f16 part in the sorted order, residues of the f16 part (no device needed)."""
        out = np.zeros(3, dtype=np.uint64)
        _check(lib.swg_debug_split(self.handle, lq, qbound, out.ctypes.data_as(_vp)))
        return {"rows": int(out[0]), "first_pair": int(out[1]), "residues_f16": int(out[2])}

    def debug_pair_tokens(self, ctx, from_host):
        """Test hook: the pair-token image (uint32 dwords) built on the device or by the host builder."""
        n = C.c_size_t(0)
        _check(lib.swg_debug_pair_tokens(ctx.handle, self.handle, 1 if from_host else 0, None, 0, C.byref(n)), ctx.handle)
        out = np.zeros(n.value, dtype=np.uint32)
        _check(lib.swg_debug_pair_tokens(ctx.handle, self.handle, 1 if from_host else 0, out.ctypes.data_as(_vp),
                                         out.size, C.byref(n)), ctx.handle)
        return out

    def save(self, path):
        _check(lib.swg_db_save(self.handle, path.encode()))

    def upload(self, ctx):
        _check(lib.swg_db_upload(ctx.handle, self.handle), ctx.handle)
        return self

    def close(self):
        if self.handle:
            lib.swg_db_free(self.handle)
            self.handle = None

    __del__ = close


class Context:
    """swg_ctx: one GPU, one query, one scoring system."""

    def __init__(self, device=0):
        self.handle = None
        cfg = Config()
        cfg.device = device
        h = _vp()
        _check(lib.swg_create(C.byref(cfg), C.byref(h)))
        self.handle = h

    def set_option(self, key, value):
        _check(lib.swg_set_option(self.handle, key.encode(), int(value)), self.handle)

    def set_scoring(self, sub, gap_open, gap_extend):
        if isinstance(sub, Scoring):
            sub = sub.table()
        s, sp = _i8(np.asarray(sub).reshape(32, 32))
        _check(lib.swg_set_scoring(self.handle, sp, int(gap_open), int(gap_extend)), self.handle)

    def set_query(self, idx):
        q, qp = _i8(idx)
        _check(lib.swg_set_query(self.handle, qp, q.size), self.handle)

    def search(self, db, want_scores=True, k=0):
        """-> (scores int32[total] or None, hits [(score, index)], stats dict)."""
        scores = np.zeros(db.total_count, dtype=np.int32) if want_scores else None
        hits = (Hit * max(k, 1))()
        nh = C.c_size_t(0)
        st = Stats()
        rc = lib.swg_search(self.handle, db.handle, scores.ctypes.data_as(_vp) if want_scores else None,
                            C.cast(hits, _vp) if k else None, k, C.byref(nh), C.byref(st))
        _check(rc, self.handle)
        return scores, [(int(hits[i].score), int(hits[i].index)) for i in range(nh.value)], st.as_dict()

    def search_multi(self, db, queries, k=0, want_scores=True):
        """Several queries (list of index arrays) against db in one pass -> (scores int32[nq, total] or None,
        hits: list of lists, stats dict)."""
        nq = len(queries)
        qoff = np.zeros(nq + 1, dtype=np.uint64)
        qoff[1:] = np.cumsum([len(q) for q in queries])
        qflat = np.ascontiguousarray(np.concatenate(queries) if nq else np.zeros(0), dtype=np.int8)
        scores = np.zeros((nq, db.total_count), dtype=np.int32) if want_scores else None
        hits = (Hit * max(k * nq, 1))()
        nh = (C.c_size_t * max(nq, 1))()
        st = Stats()
        rc = lib.swg_search_multi(self.handle, db.handle, qflat.ctypes.data_as(_vp), qoff.ctypes.data_as(_vp), nq,
                                  scores.ctypes.data_as(_vp) if want_scores else None, C.cast(hits, _vp) if k else None, k,
                                  C.cast(nh, _vp), C.byref(st))
        _check(rc, self.handle)
        out = [[(int(hits[i * k + j].score), int(hits[i * k + j].index)) for j in range(nh[i])] for i in range(nq)]
        return scores, out, st.as_dict()

    def search_begin(self, db, k=0, want_scores=False):
        """Queue a search; returns a ticket for search_end / search_end_keys."""
        t = C.c_int(-1)
        _check(lib.swg_search_begin(self.handle, db.handle, 1 if want_scores else 0, k, C.byref(t)), self.handle)
        return (t.value, db, k, want_scores)

    def search_end(self, ticket):
        t, db, k, want_scores = ticket
        scores = np.zeros(db.total_count, dtype=np.int32) if want_scores else None
        hits = (Hit * max(k, 1))()
        nh = C.c_size_t(0)
        st = Stats()
        _check(lib.swg_search_end(self.handle, t, scores.ctypes.data_as(_vp) if want_scores else None,
                                  C.cast(hits, _vp) if k else None, C.byref(nh), C.byref(st)), self.handle)
        return scores, [(int(hits[i].score), int(hits[i].index)) for i in range(nh.value)], st.as_dict()

    def search_end_keys(self, ticket):
        """As search_keys, for a search queued with search_begin."""
        t, db, k, _ = ticket
        hits = (Hit * max(k, 1))()
        nh = C.c_size_t(0)
        st = Stats()
        _check(lib.swg_search_end(self.handle, t, None, C.cast(hits, _vp), C.byref(nh), C.byref(st)), self.handle)
        return self._hit_keys(hits, k, nh.value), st.as_dict()

    def align_hits(self, db, hits, want_ops=True, ops_stride=None):
        """Alignments of the given hits [(score, index)] -> list of dicts with score, index,
        q_begin, q_end, d_begin, d_end and (want_ops) the path as a string of M/I/D."""
        n = len(hits)
        arr = (Hit * max(n, 1))()
        for i, (sc, ix) in enumerate(hits):
            arr[i].score, arr[i].index = int(sc), int(ix)
        out = (Alignment * max(n, 1))()
        stride = int(ops_stride if ops_stride is not None else lib.swg_align_ops_bound(self.handle, db.handle))
        ops = C.create_string_buffer(max(1, n * stride)) if want_ops else None
        _check(lib.swg_align_hits(self.handle, db.handle, C.cast(arr, _vp), n, C.cast(out, _vp),
                                  C.cast(ops, _vp) if want_ops else None, stride), self.handle)
        res = []
        for i in range(n):
            a = {f: int(getattr(out[i], f)) for f, _ in Alignment._fields_ if f != "reserved"}
            if want_ops:
                a["ops"] = ops.raw[i * stride:i * stride + a["n_ops"]].decode()
            res.append(a)
        return res

    @staticmethod
    def _hit_keys(hits, k, n):
        a = np.frombuffer(hits, dtype=np.dtype([("score", "<i4"), ("index", "<u4")]), count=max(k, 1))[:k]
        keys = (a["score"].astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - a["index"].astype(np.uint64))
        keys[n:] = 0
        return keys

    def search_keys(self, db, k):
        """Top-K of this shard as uint64 sort keys (score << 32 | ~index), zero-padded to k, plus
        stats: the form the multi-GPU merge exchanges (no per-hit Python objects)."""
        hits = (Hit * max(k, 1))()
        nh = C.c_size_t(0)
        st = Stats()
        _check(lib.swg_search(self.handle, db.handle, None, C.cast(hits, _vp), k, C.byref(nh), C.byref(st)),
               self.handle)
        a = np.frombuffer(hits, dtype=np.dtype([("score", "<i4"), ("index", "<u4")]), count=max(k, 1))[:k]
        keys = (a["score"].astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - a["index"].astype(np.uint64))
        keys[nh.value:] = 0
        return keys, st.as_dict()

    def fill_batches16(self, batches):
        """batches: list of (db_idx_t int8[max_len,16], vector_size) -> list of int16[vector_size]."""
        arr = (Batch16 * len(batches))()
        keep, outs = [], []
        for i, (d, vs) in enumerate(batches):
            d = np.ascontiguousarray(d, dtype=np.int8)
            o = np.zeros(16, dtype=np.int16)
            keep.append(d)
            outs.append(o)
            arr[i].db_idx_t = d.ctypes.data
            arr[i].max_len = d.shape[0]
            arr[i].vector_size = vs
            arr[i].max_scores = o.ctypes.data
        secs = C.c_double(0)
        _check(lib.swg_fill_batches16(self.handle, arr, len(batches), C.byref(secs)), self.handle)
        return [o[:vs].copy() for o, (_, vs) in zip(outs, batches)], secs.value

    def close(self):
        if self.handle:
            lib.swg_destroy(self.handle)
            self.handle = None

    __del__ = close


class Group:
    """swg_group: several GPUs in one process, shards merged with one RCCL all-reduce."""

    def __init__(self, devices, force_collective=False):
        self.handle = None
        devs = (C.c_int * len(devices))(*devices)
        h = _vp()
        _check(lib.swg_group_create(C.cast(devs, _vp), len(devices), 1 if force_collective else 0, C.byref(h)))
        self.handle = h
        self.n = len(devices)
        self.total = 0

    def _chk(self, rc):
        if rc != SWG_OK:
            raise SwgError(rc, (lib.swg_group_last_error(self.handle) or b"").decode("utf-8", "replace"))

    def set_option(self, key, value):
        self._chk(lib.swg_group_set_option(self.handle, key.encode(), int(value)))

    def set_scoring(self, sub, gap_open, gap_extend):
        if isinstance(sub, Scoring):
            sub = sub.table()
        s, sp = _i8(np.asarray(sub).reshape(32, 32))
        self._chk(lib.swg_group_set_scoring(self.handle, sp, int(gap_open), int(gap_extend)))

    def set_query(self, idx):
        q, qp = _i8(idx)
        self._chk(lib.swg_group_set_query(self.handle, qp, q.size))

    def load(self, flat, offsets):
        f, fp = _i8(flat)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(lib.swg_group_load(self.handle, fp, o.ctypes.data_as(_vp), o.size - 1))
        self.total = o.size - 1

    def search(self, want_scores=True, k=0):
        scores = np.zeros(self.total, dtype=np.int32) if want_scores else None
        hits = (Hit * max(k, 1))()
        nh = C.c_size_t(0)
        st = (Stats * self.n)()
        self._chk(lib.swg_group_search(self.handle, scores.ctypes.data_as(_vp) if want_scores else None,
                                       C.cast(hits, _vp) if k else None, k, C.byref(nh), st))
        return scores, [(int(hits[i].score), int(hits[i].index)) for i in range(nh.value)], [s.as_dict() for s in st]

    def align_hits(self, hits):
        """Alignments of hits of a group search (see Context.align_hits)."""
        n = len(hits)
        arr = (Hit * max(n, 1))()
        for i, (sc, ix) in enumerate(hits):
            arr[i].score, arr[i].index = int(sc), int(ix)
        out = (Alignment * max(n, 1))()
        stride = int(lib.swg_group_align_ops_bound(self.handle))
        ops = C.create_string_buffer(max(1, n * stride))
        self._chk(lib.swg_group_align_hits(self.handle, C.cast(arr, _vp), n, C.cast(out, _vp), C.cast(ops, _vp), stride))
        res = []
        for i in range(n):
            a = {f: int(getattr(out[i], f)) for f, _ in Alignment._fields_ if f != "reserved"}
            a["ops"] = ops.raw[i * stride:i * stride + a["n_ops"]].decode()
            res.append(a)
        return res

    def close(self):
        if self.handle:
            lib.swg_group_destroy(self.handle)
            self.handle = None

    __del__ = close
