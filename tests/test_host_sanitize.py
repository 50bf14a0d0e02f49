"""CPU: the plain-C host helpers (matrix reader, sequence reader, residue map, synthetic data)
compiled with AddressSanitizer + UBSan and driven over good and hostile inputs.  (GPU ASan is
not available on this pool; the device code is covered by the parity tests instead.)"""
import os
import subprocess

import pytest

from conftest import ROOT

DRIVER = r'''
#include "swg.h"
#include "swg_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
    char err[256];
    for (int i = 1; i < argc; i++) {
        swg_scoring sc; swg_scoring_init(&sc);
        int rc = swg_scoring_load_matrix(&sc, argv[i], err, sizeof err);
        swg_seqs s; char bad = 0;
        int rs = swg_seqs_read(argv[i], 0, &s, err, sizeof err);
        if (rs == SWG_OK) {
            int8_t *idx = malloc(s.n ? s.seq_off[s.n] + 1 : 1);
            int ri = swg_seqs_to_indices(&s, idx, &bad);
            if (ri == SWG_OK && s.n) swg_query_sanitize(&sc, idx, s.seq_off[1]);
            free(idx);
            swg_seqs_free(&s);
        }
        printf("%s matrix=%d seqs=%d\n", argv[i], rc, rs);
    }
    for (int c = -300; c < 300; c++) { (void)swg_letter_index(c); (void)swg_index_letter(c); }
    int8_t *flat; uint64_t *off; int8_t q[77]; size_t planted = 0;
    swg_synth_query(3, 77, q);
    if (swg_synth_db(1, 500, 290.0, 0.75, 20, 5000, &flat, &off) != SWG_OK) return 2;
    swg_synth_free(flat); swg_synth_free(off);
    if (swg_synth_db_similar(2, 300, 100.0, 0.5, 1, 400, q, 77, 0.2, 0.1, &flat, &off, &planted) != SWG_OK) return 3;
    swg_synth_free(flat); swg_synth_free(off);
    if (swg_synth_db(1, 0, 290.0, 0.75, 20, 5000, &flat, &off) != SWG_OK) return 4;
    swg_synth_free(flat); swg_synth_free(off);
    puts("done");
    return 0;
}
'''


def test_host_helpers_under_asan_ubsan(tmp_path):
    src = tmp_path / "driver.c"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    host = os.path.join(ROOT, "seq-align-gpu_amd", "host")
    cmd = ["gcc", "-std=c11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
           "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)] + \
          [os.path.join(host, f) for f in ("swg_scoring.c", "swg_seqio.c", "swg_synth.c", "swg_threads.c")] + ["-lz", "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 and "sanitize" in r.stdout:
        pytest.skip("sanitizer runtime not available: " + r.stdout[:200])
    assert r.returncode == 0, r.stdout
    files = [os.path.join(ROOT, "seq-align-gpu_amd", "data", n + ".txt") for n in ("BLOSUM62", "PAM250", "BLOSUM45")]
    hostile = {
        "empty": b"", "nul": b"\x00\x00\x00", "long_line": b"  " + b"A " * 5000 + b"\nA " + b"1 " * 5000 + b"\n",
        "no_newline": b">x\nACGT", "only_header": b">h", "fastq_trunc": b"@r\nACGT\n+", "binary": bytes(range(256)) * 8,
        "crlf": b">a\r\nAC\r\nGT\r\n", "huge_num": b"  A C\nA 99999999999999999999 1\n", "sep": b",A,C\nA,1\nC,,\n",
        "gz_garbage": b"\x1f\x8b\x08\x00garbage",
    }
    # a megabyte or more and a '>' first: these go through the memory-mapped parallel reader
    big = 1 << 20
    hostile.update({
        "big_headers_only": b">\n" * big, "big_one_line": b">" + b"A" * (2 * big), "big_no_final_newline": b">a\n" + b"ACDE\n" * big + b">last",
        "big_binary": b">x\n" + bytes(range(256)) * (big // 128), "big_crlf_blank": (b">r\r\n\r\nAC DE\r\n\r\n") * (big // 8),
        "big_gt_inside": b">a\n" + b"AC>DE\n>\n\n" * (big // 4), "big_header_at_end": b">a\n" + b"ACDEFGHIKL\n" * (big // 8) + b">z\n",
        "big_leading_blank": b"\n \n>a\n" + b"ACDEFGHIKL\n" * (big // 8), "big_not_fasta": b"ACDEFGHIKL\n" * (big // 8),
    })
    for name, data in hostile.items():
        p = tmp_path / name
        p.write_bytes(data)
        files.append(str(p))
    files.append(str(tmp_path / "does_not_exist"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe)] + files, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env,
                       timeout=120)
    assert r.returncode == 0 and "done" in r.stdout and "ERROR: AddressSanitizer" not in r.stdout \
        and "runtime error" not in r.stdout, r.stdout[-3000:]
    assert "BLOSUM62.txt matrix=0" in r.stdout
