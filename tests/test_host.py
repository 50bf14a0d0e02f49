"""CPU: the C-ABI library loads, exports every declared symbol, and its host-only logic
(residue map, matrix reader, sequence reader, packer, synthetic data, hit keys) behaves.
No compute entry point is called: there is no GPU here and the library has no CPU backend."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_names, load_golden


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(swg_[a-z0-9_]+)\s*\(", txt)))


def test_abi_exports_every_declared_symbol(swg):
    names = _declared("swg.h") + _declared("swg_host.h")
    assert len(names) >= 30
    for n in names:
        assert hasattr(swg.lib, n), "libswg.so does not export " + n
    assert sorted(names) == sorted(swg.ABI_SYMBOLS)
    assert swg.lib.swg_abi_version() == 3


def test_no_cpu_fallback(swg):
    """Without a GPU the product must refuse, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(swg.SwgError) as e:
        swg.Context(0)
    assert e.value.code == swg.SWG_ERR_NODEVICE
    assert "no CPU backend" in str(e.value)


def test_product_does_not_touch_oracle():
    """Nothing under seq-align-gpu_amd/ may reference oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "seq-align-gpu_amd")
    for d, _, files in os.walk(pkg):
        if os.path.basename(d) in ("build", "__pycache__", "bin"):
            continue
        for f in files:
            if f.endswith((".py", ".c", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(d, f), errors="replace").read()
                assert "oracle" not in txt.replace("oracle/ ", "").lower() or f == "__init__.py" and \
                    "liboracle" not in txt, (d, f)


def test_letter_index(swg, orc):
    for c in range(256):
        assert swg.lib.swg_letter_index(c) == orc.letter_index(chr(c))
    assert swg.lib.swg_index_letter(1) == ord("A") and swg.lib.swg_index_letter(31) == ord("*")
    assert swg.lib.swg_index_letter(0) == 0 and swg.lib.swg_index_letter(27) == 0
    with pytest.raises(swg.SwgError):
        swg.letters_to_indices("AC-GT")


@pytest.mark.parametrize("name,fixture", [("BLOSUM62", "blosum62_lq367"), ("PAM250", "pam250_lq128"),
                                          ("BLOSUM45", "blosum45_lq200")])
def test_matrix_reader_matches_golden_tables(swg, name, fixture):
    sc = swg.load_scoring(name)
    assert np.array_equal(sc.table(), load_golden(fixture)["sub"])
    assert (sc.gap_open, sc.gap_extend, sc.match, sc.mismatch) == (-2, -1, 2, -2)
    a, x, j = (swg.lib.swg_letter_index(ord(c)) for c in "AXJ")
    assert (sc.set[a] >> a) & 1 and (sc.set[x] >> x) & 1 and not (sc.set[j] >> j) & 1
    # query sanitisation: undefined self-pair -> X (reference src/alignment_cmdline.c:391-396)
    q = swg.letters_to_indices("AJOUW")
    swg.lib.swg_query_sanitize(C.byref(sc), q.ctypes.data_as(C.c_void_p), q.size)
    assert list(q) == [a, x, x, x, swg.lib.swg_letter_index(ord("W"))]


def test_matrix_reader_formats_and_errors(swg, tmp_path):
    def load(text):
        p = tmp_path / "m.txt"
        p.write_text(text)
        return swg.load_scoring(str(p))
    sc = load("# c\n\n   A  C\nA  5 -3 \nC -3  7\n# tail\n")
    t = sc.table()
    assert t[1, 1] == 5 and t[1, 3] == -3 and t[3, 3] == 7
    sc = load(",A,C\nA,5,-3\nC,-3,7\n")  # single-character separator mode
    assert sc.table()[3, 1] == -3
    import gzip
    p = tmp_path / "m.txt.gz"
    with gzip.open(p, "wt") as f:
        f.write("  A C\nA 1 2\nC 3 4\n")
    assert swg.load_scoring(str(p)).table()[3, 3] == 4  # gz-transparent like the reference
    for bad in ("", "# only comment\n", "  A C\nA 1\n", "  A C\nA 1 2 3\n", "  A C\nA 1 x\n",
                "1A1C\nA151\n", "  A C\nA 1 200\n", "  A ?\nA 1 2\n"):
        with pytest.raises(swg.SwgError):
            load(bad)
    with pytest.raises(swg.SwgError):
        swg.load_scoring(str(tmp_path / "missing.txt"))


def test_sequence_reader(swg, tmp_path):
    p = tmp_path / "a.fasta"
    p.write_text(">q1 first\nACDE\nFGH\n\n>q2\nkl mn\n>empty\n>last\nWW")
    names, seq, idx, off = swg.read_seqs(str(p))
    assert names == ["q1 first", "q2", "empty", "last"]
    assert seq == b"ACDEFGHklmnWW" and list(off) == [0, 7, 11, 11, 13]
    assert list(idx[:3]) == [1, 3, 4] and idx[7] == 11
    names, seq, idx, off = swg.read_seqs(str(p), 2)
    assert names == ["q1 first", "q2"]
    fq = tmp_path / "a.fastq"
    fq.write_text("@r1\nACGT\n+\n@@II\n@r2\nTT\n+r2\nII\n")
    names, seq, _, off = swg.read_seqs(str(fq))
    assert names == ["r1", "r2"] and seq == b"ACGTTT"
    pl = tmp_path / "plain.txt"
    pl.write_text("ACD\nEFG\n")
    names, seq, _, off = swg.read_seqs(str(pl))
    assert names == ["", ""] and list(off) == [0, 3, 6]
    bad = tmp_path / "bad.fasta"
    bad.write_text(">x\nAC-D\n")
    with pytest.raises(swg.SwgError) as e:
        swg.read_seqs(str(bad))
    assert e.value.code == swg.SWG_ERR_RESIDUE
    with pytest.raises(swg.SwgError):
        swg.read_seqs(str(tmp_path / "nope.fa"))


def test_mapped_fasta_reader_equals_line_reader(swg, tmp_path):
    """Plain FASTA files of a megabyte or more are parsed from a memory map by all cores; the result
    must equal the line reader's (which a gzip copy of the same bytes goes through): multi-line
    records, CR LF line ends, blank lines, white space inside sequence lines, an empty record,
    lower case, no newline at the end."""
    import gzip
    rng = np.random.default_rng(4)
    letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdxbz*", dtype=np.uint8)
    parts = []
    for i in range(6000):
        L = int(rng.integers(0, 700)) if i % 97 else 0
        seq = letters[rng.integers(0, len(letters), size=L)].tobytes()
        width = int(rng.integers(20, 90))
        eol = b"\r\n" if i % 5 == 0 else b"\n"
        parts.append(b">rec%d some text | %d" % (i, L) + eol)
        for j in range(0, L, width):
            line = seq[j:j + width]
            if i % 11 == 0 and len(line) > 4:
                line = line[:3] + b" \t" + line[3:]
            parts.append(line + eol)
        if i % 13 == 0:
            parts.append(eol)
    blob = b"\n\n" + b"".join(parts) + b">last\nACDEFGHIKL"
    assert len(blob) > (1 << 20)
    plain, zipped = tmp_path / "big.fa", tmp_path / "big.fa.gz"
    plain.write_bytes(blob)
    with gzip.open(zipped, "wb") as f:
        f.write(blob)
    a = swg.read_seqs(str(plain))
    b = swg.read_seqs(str(zipped))
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert len(a[0]) == 6001 and a[0][-1] == "last" and a[1].endswith(b"ACDEFGHIKL")
    # odd shapes of a megabyte or more: both readers must still agree record for record
    big = 1 << 20
    odd = {
        "headers_only": b">\n" * big, "one_line": b">" + b"A" * (2 * big), "header_last": b">a\n" + b"ACDE\n" * big + b">last",
        "crlf_blank": (b">r\r\n\r\nAC DE\r\n\r\n") * (big // 8), "header_at_end": b">a\n" + b"ACDEFGHIKL\n" * (big // 8) + b">z\n",
        "leading_blank": b"\n \n>a\n" + b"ACDEFGHIKL\n" * (big // 8), "leading_empty": b"\n\r\n>a\n" + b"ACDEFGHIKL\n" * (big // 8),
        "plain_lines": b"ACDEFGHIKL\n" * (big // 8),
    }
    for name, data in odd.items():
        p, z = tmp_path / name, tmp_path / (name + ".gz")
        p.write_bytes(data)
        with gzip.open(z, "wb", compresslevel=1) as f:
            f.write(data)
        a, b = swg.read_seqs(str(p)), swg.read_seqs(str(z))
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[3], b[3]), name


def test_pack_orders_bins_and_validates(swg):
    rng = np.random.default_rng(3)
    lens = rng.integers(1, 300, size=1000)
    off = np.zeros(1001, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    flat = rng.integers(1, 26, size=int(off[-1])).astype(np.int8)
    db = swg.Database(flat, off)
    assert db.count == 1000 and db.total_count == 1000 and db.residues == int(off[-1])
    order = db.order()
    assert sorted(order) == list(range(1000))
    sl = lens[order]
    assert (np.diff(sl) <= 0).all()                      # longest first
    for L in np.unique(sl):                              # stable among equal lengths
        o = order[sl == L]
        assert (np.diff(o) > 0).all()
    # what goes to the GPU: one byte per residue, every sequence a run of whole dwords, plus 16 bytes
    # per slot (length, original index, offset) and one block offset per pair of sequences
    nb = (1000 + 127) // 128
    ns = nb * 128
    expect = sum((int(l) + 3) // 4 * 4 for l in lens) + ns * 16 + 8 + (ns // 2 + 1) * 4
    assert db.packed_bytes == expect
    # shards partition the bins round-robin
    seen = []
    for r in range(3):
        s = swg.Database(flat, off, r, 3)
        o = s.order()
        want = np.concatenate([order[b * 128:(b + 1) * 128] for b in range(r, nb, 3)])
        assert np.array_equal(o, want)
        seen += list(o)
    assert sorted(seen) == list(range(1000))
    # residue 0 / out of range / bad shard -> error codes, never a crash
    bad = flat.copy()
    bad[5] = 0
    with pytest.raises(swg.SwgError) as e:
        swg.Database(bad, off)
    assert e.value.code == swg.SWG_ERR_RESIDUE
    with pytest.raises(swg.SwgError):
        swg.Database(flat, off, 3, 3)
    empty = swg.Database(np.zeros(0, np.int8), np.zeros(1, np.uint64))
    assert empty.count == 0
    # n == 0 with NULL arrays is an empty database, not a crash (the C ABI is called directly)
    import ctypes as C
    h = C.c_void_p()
    assert swg.lib.swg_db_pack(None, None, 0, 0, 1, C.byref(h)) == 0 and swg.lib.swg_db_count(h) == 0
    swg.lib.swg_db_free(h)


def test_presharded_pack_equals_sharding_the_whole(swg, tmp_path):
    """A rank that generates only its own bins (swg_synth_db_shard) and packs them (swg_db_pack_shard) holds
    byte for byte what swg_db_pack(whole database, rank, world) keeps -- the multi-GPU bench's path."""
    n = 1000
    q = swg.synth_query(21, 60)
    for kw in (dict(), dict(query=q, fraction=0.05, subst=0.05)):
        whole = swg.synth_db(21, n, max_len=400, **kw)
        flat, off = whole[0], whole[1]
        for world in (1, 2, 3, 8, 16):                    # 8 bins: 16 ranks leave half of them empty-handed
            seen, residues = [], 0
            for r in range(world):
                sh = swg.synth_db_shard(21, n, r, world, max_len=400, **kw)
                assert sh["n_total"] == n and sh["residues_total"] == int(off[-1])
                if kw:
                    assert sh["planted"] == whole[2]
                # the shard's sequences are the whole database's, at their global indices
                for i in (0, len(sh["index"]) // 2, len(sh["index"]) - 1):
                    if len(sh["index"]):
                        g = int(sh["index"][i])
                        assert np.array_equal(sh["flat"][int(sh["offsets"][i]):int(sh["offsets"][i + 1])],
                                              flat[int(off[g]):int(off[g + 1])])
                a = swg.Database(flat, off, r, world)
                b = swg.Database(sh["flat"], sh["offsets"], index=sh["index"], n_total=n)
                assert (a.count, a.total_count, a.residues) == (b.count, b.total_count, b.residues)
                assert np.array_equal(a.order(), b.order())
                pa, pb = tmp_path / "a.swg", tmp_path / "b.swg"
                a.save(str(pa))
                b.save(str(pb))
                assert pa.read_bytes() == pb.read_bytes()
                seen += list(b.order())
                residues += b.residues
            assert sorted(seen) == list(range(n)) and residues == int(off[-1])
    with pytest.raises(swg.SwgError):                     # an index outside the database is refused
        swg.Database(flat[:int(off[2])], off[:3], index=np.array([0, n], np.uint32), n_total=n)


def test_all_shards_from_one_sort(swg, tmp_path):
    """swg_db_pack_shards (what swg_group_load -- one process, several GPUs -- packs with): shard r is byte for byte
    swg_db_pack(whole, r, count), and the whole database is sorted ONCE, not once per device."""
    flat, off = swg.synth_db(33, 1500, max_len=500)
    for count in (1, 4, 7):
        before = swg.lib.swg_debug_sort_count()
        shards = swg.Database.pack_shards(flat, off, count)
        assert swg.lib.swg_debug_sort_count() == before + 1
        assert len(shards) == count
        seen = []
        for r, b in enumerate(shards):
            a = swg.Database(flat, off, r, count)
            assert (a.count, a.total_count, a.residues) == (b.count, b.total_count, b.residues)
            pa, pb = tmp_path / "a.swg", tmp_path / "b.swg"
            a.save(str(pa))
            b.save(str(pb))
            assert pa.read_bytes() == pb.read_bytes()
            seen += list(b.order())
            a.close()
            b.close()
        assert sorted(seen) == list(range(1500))
    bad = flat.copy()
    bad[17] = 0                                           # the padding residue is refused in input, for every shard at once
    with pytest.raises(swg.SwgError) as e:
        swg.Database.pack_shards(bad, off, 4)
    assert e.value.code == swg.SWG_ERR_RESIDUE
    empty = swg.Database.pack_shards(np.zeros(0, np.int8), np.zeros(1, np.uint64), 3)
    assert [d.count for d in empty] == [0, 0, 0]


def test_no_exception_crosses_the_search_entry_points(swg):
    """include/swg.h:9-12: every entry point returns a status.  A failed host allocation inside swg_search_begin /
    swg_search_end / swg_search (the plan cache is a std::map, the fall-back key list a std::vector) must come back as
    SWG_ERR_NOMEM, not as std::terminate: the test hook makes the next visit of a site throw std::bad_alloc.
    (The site inside the fall-back key vector needs a search in flight: tests/test_gpu_parity.py.)"""
    t = C.c_int(-1)
    n = C.c_size_t(0)
    swg.lib.swg_debug_fail_alloc(1)
    assert swg.lib.swg_search_begin(None, None, 0, 0, C.byref(t)) == swg.SWG_ERR_NOMEM
    assert b"swg_search_begin" in swg.lib.swg_global_error() and b"memory" in swg.lib.swg_global_error()
    assert swg.lib.swg_search_begin(None, None, 0, 0, C.byref(t)) == swg.SWG_ERR_ARG      # one shot: the hook is spent
    swg.lib.swg_debug_fail_alloc(2)
    assert swg.lib.swg_search_end(None, 0, None, None, C.byref(n), None) == swg.SWG_ERR_NOMEM
    assert b"swg_search_end" in swg.lib.swg_global_error()
    swg.lib.swg_debug_fail_alloc(1)
    assert swg.lib.swg_search(None, None, None, None, 0, C.byref(n), None) == swg.SWG_ERR_NOMEM
    swg.lib.swg_debug_fail_alloc(0)
    assert swg.lib.swg_search(None, None, None, None, 0, C.byref(n), None) == swg.SWG_ERR_ARG


def test_packed_database_file_roundtrip(swg, tmp_path):
    flat, off = swg.synth_db(5, 700, max_len=300)
    db = swg.Database(flat, off, 1, 2)
    path = str(tmp_path / "shard.swgdb")
    db.save(path)
    back = swg.Database(path=path)
    assert (back.count, back.total_count, back.residues, back.packed_bytes) == \
           (db.count, db.total_count, db.residues, db.packed_bytes)
    assert np.array_equal(back.order(), db.order())
    raw = open(path, "rb").read()
    ns = ((db.count + 127) // 128) * 128
    hdr = 8 + 8 * 8
    import struct

    def patched(at, val, size=8):
        return raw[:at] + struct.pack("<Q" if size == 8 else "<I", val) + raw[at + size:]

    swapped = bytearray(raw)                              # two neighbouring lengths out of order
    lens_at = hdr + ns * 4
    fl = np.frombuffer(raw, dtype="<u4", count=ns, offset=lens_at)
    i = int(np.nonzero(fl[:-1] > fl[1:])[0][0]) * 4 + lens_at
    swapped[i:i + 4], swapped[i + 4:i + 8] = raw[i + 4:i + 8], raw[i:i + 4]
    codes_at = hdr + ns * 8
    cases = {
        "truncated header": raw[:50], "truncated body": raw[:-7], "bad magic": b"NOTADB00" + raw[8:],
        "n_bins huge": patched(24, 1 << 40), "n_codes huge": patched(40, 1 << 62), "n_total tiny": patched(8, 1),
        "n_local lies": patched(16, db.count - 1), "residues lie": patched(32, db.residues + 1),
        "order out of range": patched(hdr, 0x7FFFFFFF, 4), "lengths unsorted": bytes(swapped),
        "residue byte with low bits": raw[:codes_at] + b"\x0b" + raw[codes_at + 1:],
        "padding residue inside a sequence": raw[:codes_at] + b"\x00" + raw[codes_at + 1:],
        "trailing junk": raw + b"\0" * 4,
    }
    for name, bad in cases.items():
        p2 = tmp_path / "bad.swgdb"
        p2.write_bytes(bad)
        with pytest.raises(swg.SwgError) as e:
            swg.Database(path=str(p2))
        assert e.value.code == swg.SWG_ERR_IO, name
    with pytest.raises(swg.SwgError):
        swg.Database(path=str(tmp_path / "missing.swgdb"))


def test_synthetic_data_is_deterministic_and_shaped(swg):
    f1, o1 = swg.synth_db(0x5EED0002, 2000)
    f2, o2 = swg.synth_db(0x5EED0002, 2000)
    assert np.array_equal(f1, f2) and np.array_equal(o1, o2)
    f3, _ = swg.synth_db(0x5EED0003, 2000)
    assert not np.array_equal(f1[:1000], f3[:1000])
    lens = np.diff(o1.astype(np.int64))
    assert (np.diff(lens) <= 0).all() and lens.min() >= 20 and lens.max() <= 5000
    assert 250 < np.median(lens) < 340 and 330 < lens.mean() < 420
    letters = set(chr(swg.lib.swg_index_letter(int(v))) for v in np.unique(f1))
    assert letters == set("ARNDCQEGHILKMFPSTWYV")
    freq_l = (f1 == swg.lib.swg_letter_index(ord("L"))).mean()
    assert 0.085 < freq_l < 0.108
    q = swg.synth_query(0x5EED0002, 367)
    assert q.size == 367 and np.array_equal(q, swg.synth_query(0x5EED0002, 367))
    fs, os_, planted = swg.synth_db(7, 500, query=q, fraction=0.1, subst=0.05)
    lens = np.diff(os_.astype(np.int64))
    assert 20 < planted < 90 and (lens == 367).sum() >= planted
    k = int(np.nonzero(lens == 367)[0][0])
    same = (fs[int(os_[k]):int(os_[k + 1])] == q).mean()
    assert same > 0.9
    # a family of relatives: every planted sequence with its own substitution rate from the range; the same
    # sequences are planted, and a range of zero width is the near-copy form byte for byte
    ff, of, pf = swg.synth_db(7, 500, query=q, fraction=0.1, subst=0.3, subst_hi=0.7)
    assert pf == planted and np.array_equal(of, os_)
    ident = [(ff[int(of[k]):int(of[k + 1])] == q).mean() for k in np.nonzero(lens == 367)[0]]
    ident = [v for v in ident if v > 0.2]                 # (an unplanted sequence of 367 residues matches ~6 %)
    assert len(ident) == planted and 0.28 < min(ident) < 0.45 and 0.6 < max(ident) < 0.78
    f0, o0, _ = swg.synth_db(7, 500, query=q, fraction=0.1, subst=0.05, subst_hi=0.05)
    assert np.array_equal(f0, fs) and np.array_equal(o0, os_)


def test_hit_keys_and_merge(swg, orc):
    assert swg.key_hit(swg.hit_key(1234, 77)) == (1234, 77)
    assert swg.hit_key(10, 5) > swg.hit_key(10, 6) > swg.hit_key(9, 0)   # score, then lower index
    rng = np.random.default_rng(1)
    scores = rng.integers(0, 50, size=5000).astype(np.int32)
    keys = np.array([swg.hit_key(int(s), i) for i, s in enumerate(scores)], dtype=np.uint64)
    for k in (1, 10, 100, 6000):
        assert swg.topk_merge_keys(keys, k) == orc.topk(scores, k)
    # merging per-shard lists gives the global list
    parts = [swg.topk_merge_keys(keys[r::4], 100) for r in range(4)]
    shard_keys = np.array([swg.hit_key(s, i) for p in parts for (s, i) in p] + [0, 0], dtype=np.uint64)
    assert swg.topk_merge_keys(shard_keys, 100) == orc.topk(scores, 100)


def test_host_threads_respect_the_process_share(swg):
    """swg_host_threads: never more than the CPUs this process may run on, and OMP_NUM_THREADS wins."""
    import subprocess, sys
    n = swg.lib.swg_host_threads()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    code = ("import sys; sys.path.insert(0, %r); import swg_loader; "
            "print(swg_loader.load().lib.swg_host_threads())" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS="1"),
                       stdout=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "1"


def test_packed_image_is_deterministic(swg, tmp_path):
    """The residue arrays are allocated uninitialised and filled by parallel loops: two packs of the
    same input (and a pack on one thread's worth of bins) must give byte-identical files, i.e. no
    byte of the image is left to chance."""
    flat, off = swg.synth_db(99, 1000, max_len=900)
    a, b = tmp_path / "a.swg", tmp_path / "b.swg"
    junk = np.full(8 << 20, 0x5A, dtype=np.uint8)      # dirty the heap between the two packs
    swg.Database(flat, off).save(str(a))
    del junk
    swg.Database(flat, off).save(str(b))
    assert a.read_bytes() == b.read_bytes()
    db = swg.Database(path=str(a))
    assert db.count == 1000 and db.residues == len(flat)


def test_built_kernels_keep_unfused_lds_reads(swg):
    """The build disables hipcc's load/store optimizer for device code so that the profile reads stay
    ds_read_b64 (conflict-free 8-byte bank slots) instead of being fused into ds_read2_b64 (half the bytes per
    LDS cycle).  If a compiler update dropped or renamed that switch nothing else would notice: the build
    looks at the code object it produced (build.verify_isa) and records the library it checked."""
    import swg_loader
    b = swg_loader.build_module()
    b.ensure_isa_checked(verbose=False)                    # a no-op when this library was checked at build time
    stamp = open(b.ISA_STAMP).read().split("\n")
    assert stamp[0] == b._lib_digest()
    counts = eval(stamp[1], {"__builtins__": {}})
    for fam in ("swg_diag_dyn_kernel", "swg_diag_kernel", "swg_fill_kernelI8CellsI16", "swg_diag32q_kernel"):
        assert counts[fam][0] > 0 and counts[fam][1] == 0, fam
