"""TEST INFRASTRUCTURE ONLY: ctypes wrappers of the CPU checkers.

  liboracle.so       -- this repo's scalar int32 restatement (oracle/sw_oracle.c)
  _ref/libswref.so   -- the REFERENCE's own alignment.c + alignment_scoring.c,
                        compiled where they lie (oracle/Makefile); present only
                        when built in the container that has /root/reference,
                        travels to the GPU box as a built file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product (seq-align-gpu_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libswref.so")
_vp = C.c_void_p


def build(quiet=True):
    """Compile the checkers (make): liboracle.so always, _ref/ when the reference is mounted."""
    r = subprocess.run(["make", "-C", HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout)
    if not quiet:
        print(r.stdout)


_olib = None
_rlib = None


def olib():
    global _olib
    if _olib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        _olib = C.CDLL(ORACLE_SO)
        _olib.sw_oracle_pair.restype = C.c_int32
        _olib.sw_oracle_pair.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int, C.c_int]
        _olib.sw_oracle_pair_wrap16.restype = C.c_int32
        _olib.sw_oracle_pair_wrap16.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int, C.c_int]
        _olib.sw_oracle_pair_trace.restype = C.c_int32
        _olib.sw_oracle_pair_trace.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int, C.c_int,
                                               _vp, _vp, C.c_size_t, _vp]
        _olib.sw_oracle_db.restype = None
        _olib.sw_oracle_db.argtypes = [_vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.c_int, C.c_int, _vp]
        _olib.sw_oracle_letter_index.restype = C.c_int
        _olib.sw_oracle_letter_index.argtypes = [C.c_int]
        _olib.sw_oracle_topk.restype = C.c_size_t
        _olib.sw_oracle_topk.argtypes = [_vp, C.c_size_t, C.c_size_t, _vp, _vp]
    return _olib


def have_ref():
    return os.path.exists(REF_SO)


def rlib():
    global _rlib
    if _rlib is None:
        _rlib = C.CDLL(REF_SO)
        _rlib.swref_fill_batch16.restype = None
        _rlib.swref_fill_batch16.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int, C.c_int, _vp]
        _rlib.swref_fill_batches.restype = C.c_double
        _rlib.swref_fill_batches.argtypes = [_vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.c_int,
                                             C.c_int, C.c_int, _vp]
        _rlib.swref_letters_to_index.restype = C.c_int
        _rlib.swref_letters_to_index.argtypes = [C.c_int]
        _rlib.swref_max_threads.restype = C.c_int
    return _rlib


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def pair(q, d, sub, gap_open, gap_extend):
    q, d, sub = _i8(q), _i8(d), _i8(sub).reshape(32, 32)
    return int(olib().sw_oracle_pair(q.ctypes.data, q.size, d.ctypes.data, d.size, sub.ctypes.data,
                                     gap_open, gap_extend))


def pair_wrap16(q, d, sub, gap_open, gap_extend):
    q, d, sub = _i8(q), _i8(d), _i8(sub).reshape(32, 32)
    return int(olib().sw_oracle_pair_wrap16(q.ctypes.data, q.size, d.ctypes.data, d.size,
                                            sub.ctypes.data, gap_open, gap_extend))


def pair_trace(q, d, sub, gap_open, gap_extend):
    """-> (score, (q_begin, q_end, d_begin, d_end), ops str) of one pair's alignment."""
    q, d, sub = _i8(q), _i8(d), _i8(sub).reshape(32, 32)
    coords = np.zeros(4, dtype=np.uint32)
    cap = q.size + d.size + 1
    ops = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    sc = olib().sw_oracle_pair_trace(q.ctypes.data, q.size, d.ctypes.data, d.size, sub.ctypes.data,
                                     gap_open, gap_extend, coords.ctypes.data, ops, cap, C.byref(n))
    return int(sc), tuple(int(c) for c in coords), ops.value.decode()


def path_score(q, d, sub, gap_open, gap_extend, coords, ops):
    """Score of an alignment path under the three-state model: a residue pair scores
    sub[q][d]; a gap position scores gap_extend when it continues a gap of the SAME kind, else
    gap_open + gap_extend.  Checks that the path covers exactly the coordinates it reports."""
    sub = np.asarray(sub).reshape(32, 32)
    i, j = coords[0], coords[2]
    total, prev = 0, ""
    for op in ops:
        if op == "M":
            total += int(sub[int(q[i]), int(d[j])]); i += 1; j += 1
        elif op == "I":
            total += gap_extend if prev == "I" else gap_open + gap_extend; j += 1
        elif op == "D":
            total += gap_extend if prev == "D" else gap_open + gap_extend; i += 1
        else:
            raise ValueError(op)
        prev = op
    assert (i, j) == (coords[1], coords[3]), "path does not end where it says"
    return total


def score_db(q, flat, offsets, sub, gap_open, gap_extend):
    """int32 scores of every database sequence, in database order (OpenMP)."""
    q, flat, sub = _i8(q), _i8(flat), _i8(sub).reshape(32, 32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    out = np.zeros(n, dtype=np.int32)
    olib().sw_oracle_db(q.ctypes.data, q.size, flat.ctypes.data, offsets.ctypes.data, n,
                        sub.ctypes.data, gap_open, gap_extend, out.ctypes.data)
    return out


def topk(scores, k):
    scores = np.ascontiguousarray(scores, dtype=np.int32)
    idx = np.zeros(max(k, 1), dtype=np.uint32)
    sc = np.zeros(max(k, 1), dtype=np.int32)
    m = olib().sw_oracle_topk(scores.ctypes.data, scores.size, k, idx.ctypes.data, sc.ctypes.data)
    return [(int(sc[i]), int(idx[i])) for i in range(m)]


def letter_index(ch):
    return int(olib().sw_oracle_letter_index(ord(ch)))


# ---- reference-shaped batches ------------------------------------------------
def make_batch16(seqs, pad_index=31):
    """16 (or fewer) index arrays, first longest -> [max_len][16] int8 as the reference
    packer lays it out (src/alignment_cmdline.c:444-450); missing lanes are all '*'."""
    max_len = len(seqs[0])
    out = np.full((max_len, 16), pad_index, dtype=np.int8)
    for l, s in enumerate(seqs):
        assert len(s) <= max_len
        out[:len(s), l] = s
    return out


def _aligned_copy(a, align=32):
    buf = np.zeros(a.size + align, dtype=np.int8)
    o = (-buf.ctypes.data) % align
    v = buf[o:o + a.size]
    v[:] = a.ravel()
    return buf, v


def ref_batch16(q, batch, sub, gap_open, gap_extend):
    """The reference's alignment_fill_matrices on one [max_len][16] batch -> int16[16]."""
    q, batch, sub = _i8(q), _i8(batch), _i8(sub).reshape(32, 32)
    out = np.zeros(16, dtype=np.int16)
    rlib().swref_fill_batch16(q.ctypes.data, q.size, batch.ctypes.data, batch.shape[0],
                              sub.ctypes.data, gap_open, gap_extend, out.ctypes.data)
    return out


def ref_batches(q, batches, sub, gap_open, gap_extend, threads=0):
    """Reference dispatch over many batches -> (int16[n][16], seconds inside the fill regions)."""
    q, sub = _i8(q), _i8(sub).reshape(32, 32)
    keep = [_aligned_copy(_i8(b)) for b in batches]
    ptrs = (C.c_void_p * len(batches))(*[v.ctypes.data for _, v in keep])
    lens = (C.c_size_t * len(batches))(*[b.shape[0] for b in batches])
    out = np.zeros((len(batches), 16), dtype=np.int16)
    secs = rlib().swref_fill_batches(q.ctypes.data, q.size, ptrs, lens, len(batches), sub.ctypes.data,
                                     gap_open, gap_extend, threads, out.ctypes.data)
    return out, float(secs)


def db_to_batches16(flat, offsets, pad_index=31):
    """Length-sorted database -> reference-shaped batches (N must be a multiple of 16 and the
    first record of every 16 the longest: SURVEY A.7-4/5)."""
    n = len(offsets) - 1
    assert n % 16 == 0
    batches = []
    for b in range(0, n, 16):
        seqs = [flat[int(offsets[i]):int(offsets[i + 1])] for i in range(b, b + 16)]
        batches.append(make_batch16(seqs, pad_index))
    return batches
