"""The `smith_waterman` tool (plain-C host over libswg): flags and stdout of the reference tool
(SURVEY A.6).  The CPU tests cover argument handling and the no-GPU failure; the GPU test is
BASELINE config 1 -- a 128-aa query against a 1k-sequence synthetic database through the tool --
checked entry by entry the way the reference's own test/tests.py does (regex over `Entry #n:` /
`score:`), against the oracle."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "seq-align-gpu_amd", "bin", "smith_waterman")
B62 = os.path.join(ROOT, "seq-align-gpu_amd", "data", "BLOSUM62.txt")
ENTRY_RX = re.compile(r"Entry\s+#(\d+):\s*score:\s*([+-]?\d+)", re.IGNORECASE)   # reference test/tests.py:52
TIME_RX = re.compile(r"Total Time: ([0-9]*\.?[0-9]+)")                             # reference benchmarks/benchmark.py:29


def _letters(swg, idx):
    return "".join(chr(swg.lib.swg_index_letter(int(v))) for v in idx)


def _write_fasta(path, names, seqs, width=60, gz=False):
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n" % n)
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def _run(*args):
    return subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_cli_argument_handling(swg, tmp_path):
    assert os.path.exists(CLI)
    r = _run()
    assert r.returncode != 0 and "usage:" in r.stderr
    r = _run("--bogus", "1")
    assert r.returncode != 0 and "Unknown argument" in r.stderr
    r = _run("--files", "a.fa")
    assert r.returncode != 0
    q = tmp_path / "q.fa"
    q.write_text(">q\nACDE\n")
    r = _run("--files", str(q), str(q))
    assert r.returncode != 0 and "--substitution_matrix is required" in r.stderr
    r = _run("--substitution_matrix", str(tmp_path / "none.txt"), "--files", str(q), str(q))
    assert r.returncode != 0 and "substitution matrix" in r.stderr
    r = _run("--substitution_matrix", B62, "--gapopen", "x", "--files", str(q), str(q))
    assert r.returncode != 0 and "--gapopen" in r.stderr
    r = _run("--substitution_matrix", B62, "--align", "--files", str(q), str(q))
    assert r.returncode != 0 and "--topk" in r.stderr
    # the reference's --stdin and --file <f> (src/alignment_cmdline.c:219-222, 268-270) name a query and no
    # database: accepted by the parser, then refused like the reference's cmdline_new does (:303-305)
    for extra in (["--stdin"], ["--file", str(q)]):
        r = _run("--substitution_matrix", B62, *extra)
        assert r.returncode != 0 and "No input specified" in r.stderr and "Unknown argument" not in r.stderr, extra
    # ... and a later --files still wins
    r = _run("--substitution_matrix", B62, "--file", str(q), "--files", str(q), str(q))
    assert "No input specified" not in r.stderr
    # flag-only options are valid in the last position (the reference's own are: --printseq etc.)
    for last in ("--timing", "--align", "--packed", "--allqueries", "--printseq", "--printfasta", "--stdin"):
        r = _run("--substitution_matrix", B62, "--files", str(q), str(q), last)
        assert "without parameter" not in r.stderr, last
    r = _run("--substitution_matrix", B62, "--files", str(q), str(q), "--topk")
    assert r.returncode != 0 and "Unknown argument without parameter: --topk" in r.stderr


def test_cli_refuses_without_gpu(swg, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    q = tmp_path / "q.fa"
    q.write_text(">q\nACDEFGHIKL\n")
    r = _run("--substitution_matrix", B62, "--files", str(q), str(q))
    assert r.returncode != 0 and "no CPU backend" in r.stderr
    assert "Query File=%s and Database File=%s" % (q, q) in r.stdout


@pytest.mark.gpu
def test_cli_config1_against_oracle(swg, orc, tmp_path):
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(0x5EED0001, 128)
    flat, off = swg.synth_db(0x5EED0001, 1024)
    # the tool does not need a sorted database nor a multiple of 16 records: shuffle, drop a few
    rng = np.random.default_rng(1)
    keep = rng.permutation(1024)[:1003]
    seqs = [_letters(swg, flat[int(off[i]):int(off[i + 1])]) for i in keep]
    seqs[5] = seqs[5].lower()                                  # case folds (letters_to_index)
    names = ["db%d some description" % i for i in range(len(seqs))]
    qf, df = tmp_path / "query.fasta", tmp_path / "db.fasta.gz"
    _write_fasta(qf, ["query1"], [_letters(swg, q)])
    _write_fasta(df, names, seqs, gz=True)
    want = [orc.pair(q, swg.letters_to_indices(s), sc.table(), -2, -1) for s in seqs]

    r = _run("--substitution_matrix", B62, "--files", str(qf), str(df))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "Query File=%s and Database File=%s" % (qf, df)
    got = {int(m.group(1)): int(m.group(2)) for m in ENTRY_RX.finditer(r.stdout)}
    assert got == dict(enumerate(want))
    assert lines[1:4] == ["Entry #0:", "score: %d" % want[0], ""]
    assert TIME_RX.search(r.stdout) and lines[-1] == "Total Entries: %d" % len(seqs)
    assert lines[-2].startswith("Total Time: ")

    # non-default gap scores, names/sequences printed, top-K report
    r = _run("--substitution_matrix", B62, "--gapopen", "-10", "--gapextend", "-1", "--printfasta", "--printseq",
             "--topk", "5", "--files", str(qf), str(df))
    assert r.returncode == 0, r.stderr
    want2 = np.array([orc.pair(q, swg.letters_to_indices(s), sc.table(), -10, -1) for s in seqs])
    lines = r.stdout.splitlines()
    assert lines[1:8] == ["query1", _letters(swg, q), "Entry #0:", names[0], seqs[0], "score: %d" % want2[0], ""]
    i16 = lines.index("Entry #16:")
    assert lines[i16 - 2:i16] == ["query1", _letters(swg, q)]          # query lines once per 16 entries
    top = lines[lines.index("Top 5 hits (score, entry, name):") + 1:][:5]
    exp = orc.topk(want2.astype(np.int32), 5)
    assert top == ["%d\t%d\t%s" % (s, i, names[i]) for s, i in exp]

    # alignments of the reported hits: gapped query line over gapped database line
    ra = _run("--substitution_matrix", B62, "--gapopen", "-10", "--gapextend", "-1", "--topk", "5", "--align",
              "--files", str(qf), str(df))
    assert ra.returncode == 0, ra.stderr
    al = ra.stdout.splitlines()
    qs = _letters(swg, q)
    for r_, (s_, i_) in enumerate(exp):
        k = next(n for n, l in enumerate(al) if l.startswith("Alignment #%d: " % r_))
        m = re.match(r"Alignment #\d+: entry (\d+) score (-?\d+) query (\d+)\.\.(\d+) entry (\d+)\.\.(\d+)$", al[k])
        ent, scv, qb, qe, db_, de = (int(x) for x in m.groups())
        assert (ent, scv) == (i_, s_)
        top_line, bot_line = al[k + 1], al[k + 2]
        assert len(top_line) == len(bot_line) and al[k + 3] == ""
        assert top_line.replace("-", "") == qs[qb:qe] and bot_line.replace("-", "").upper() == seqs[i_].upper()[db_:de]
        want_sc, co, ops = orc.pair_trace(q, swg.letters_to_indices(seqs[i_]), sc.table(), -10, -1)
        assert (want_sc, co) == (s_, (qb, qe, db_, de))
        assert "".join("I" if a == "-" else "D" if b == "-" else "M" for a, b in zip(top_line, bot_line)) == ops

    # makedb route: write the packed database once, search it without parsing or sorting
    pk = tmp_path / "db.swg"
    r2 = _run("--substitution_matrix", B62, "--savedb", str(pk), "--files", str(qf), str(df))
    assert r2.returncode == 0 and pk.exists() and "packed database written" in r2.stderr
    r3 = _run("--substitution_matrix", B62, "--packed", "--topk", "4", "--align", "--files", str(qf), str(pk))
    assert r3.returncode == 0, r3.stderr
    l3 = r3.stdout.splitlines()
    k3 = next(n for n, l in enumerate(l3) if l.startswith("Alignment #0: "))
    best = orc.topk(np.array(want, dtype=np.int32), 1)[0][1]
    assert l3[k3 + 1] == orc.pair_trace(q, swg.letters_to_indices(seqs[best]), sc.table(), -2, -1)[2]   # no letters: the path
    assert {int(m.group(1)): int(m.group(2)) for m in ENTRY_RX.finditer(r3.stdout)} == dict(enumerate(want))
    top3 = r3.stdout.splitlines()
    top3 = top3[top3.index("Top 4 hits (score, entry, name):") + 1:][:4]
    assert [tuple(int(x) for x in t.split("\t")[:2]) for t in top3] == orc.topk(np.array(want, dtype=np.int32), 4)
    assert _run("--substitution_matrix", B62, "--packed", "--printfasta", "--files", str(qf), str(pk)).returncode != 0

    # every record of the query file against the resident database
    q2 = swg.synth_query(77, 61)
    q3 = swg.synth_query(78, 300)
    qf3 = tmp_path / "queries.fasta"
    _write_fasta(qf3, ["query1", "second", "third one"], [_letters(swg, q), _letters(swg, q2), _letters(swg, q3)])
    r4 = _run("--substitution_matrix", B62, "--allqueries", "--files", str(qf3), str(df))
    assert r4.returncode == 0, r4.stderr
    blocks = re.split(r"^Query #(\d+): (.*)$", r4.stdout, flags=re.MULTILINE)
    assert [blocks[i] for i in (1, 4, 7)] == ["0", "1", "2"] and [blocks[i] for i in (2, 5, 8)] == ["query1", "second", "third one"]
    for qq, text in ((q, blocks[3]), (q2, blocks[6]), (q3, blocks[9])):
        w = [orc.pair(qq, swg.letters_to_indices(s), sc.table(), -2, -1) for s in seqs]
        assert {int(m.group(1)): int(m.group(2)) for m in ENTRY_RX.finditer(text)} == dict(enumerate(w))
        assert "Total Entries: %d" % len(seqs) in text

    # the multi-GPU route of the tool (one device here): same stream of entries
    r1 = _run("--substitution_matrix", B62, "--gpus", "1", "--topk", "3", "--align", "--files", str(qf), str(df))
    assert r1.returncode == 0, r1.stderr
    assert sum(1 for l in r1.stdout.splitlines() if l.startswith("Alignment #")) == 3
    assert {int(m.group(1)): int(m.group(2)) for m in ENTRY_RX.finditer(r1.stdout)} == dict(enumerate(want))

    # an illegal residue: the reference's message and exit status 1
    bad = tmp_path / "bad.fasta"
    bad.write_text(">x\nAC-DE\n")
    r = _run("--substitution_matrix", B62, "--files", str(qf), str(bad))
    assert r.returncode == 1 and "Error: - is not a legal character for the substitution matrix!" in r.stdout
