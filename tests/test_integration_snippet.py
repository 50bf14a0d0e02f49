"""CPU, only where the reference is mounted (never on the GPU box): the replacement body INTEGRATION.md
section A gives for the reference's timed loop is compiled -- syntax and types only -- against the
reference's OWN headers and include/swg.h, so that the aligner_t / scoring_t field names and types the
snippet relies on (src/alignment.h:25-37, src/alignment_scoring.h:18-37) are checked by a compiler and
not by eye.  The snippet is taken from INTEGRATION.md itself: what is documented is what is compiled."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "alignment.h")), reason="reference sources not mounted")
def test_integration_snippet_compiles_against_the_reference_headers(tmp_path):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## A."):md.index("## B.")]
    blocks = re.findall(r"```c\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 2, "section A: the reference's loop, then its replacement"
    body = blocks[1]
    assert "swg_fill_batches16" in body and "aligners[i]->seq_b_batch_indexes" in body
    setup, loop = body.split("/* instead of the OpenMP loop: */")
    setup = setup.replace('#include "swg.h"', "")
    # the variables are declared as the reference's driver declares them (src/alignment_cmdline.c:343-420)
    tu = """
#include <stdio.h>
#include <stdlib.h>
#include "alignment.h"            /* the reference's aligner_t, scoring_t */
#include "swg.h"
void patched_region(aligner_t **aligners, size_t batch_cnt, scoring_t *scoring, int8_t *query_indexes,
                    size_t query_seq_len, double total_time)
{
    size_t i;
%s
%s
    (void)total_time;
}
""" % (setup, loop)
    src = tmp_path / "patched.c"
    src.write_text(tu)
    cc = shutil.which("gcc")
    r = subprocess.run([cc, "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-mavx2", "-fsyntax-only", "-Wall", "-Werror",
                        "-Wno-unused-variable", "-I" + REF_SRC, "-I" + os.path.join(ROOT, "include"), str(src)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    # and the field types are the ones swg_batch16 takes, not merely convertible ones
    chk = tmp_path / "types.c"
    chk.write_text("""
#include "alignment.h"
#include "swg.h"
_Static_assert(_Generic(((aligner_t *)0)->seq_b_batch_indexes, int8_t *: 1, default: 0), "db_idx_t");
_Static_assert(_Generic(((aligner_t *)0)->max_scores, int16_t *: 1, default: 0), "max_scores is int16 (score_t)");
_Static_assert(_Generic(((aligner_t *)0)->vector_size, size_t: 1, default: 0), "vector_size");
_Static_assert(_Generic(((aligner_t *)0)->score_height, size_t: 1, default: 0), "score_height");
_Static_assert(sizeof(((scoring_t *)0)->swap_scores) == 32 * 32, "swap_scores is int8[32][32]");
_Static_assert(_Generic(((swg_batch16 *)0)->max_scores, int16_t *: 1, default: 0), "swg_batch16.max_scores");
""")
    r = subprocess.run([cc, "-std=c11", "-mavx2", "-fsyntax-only", "-I" + REF_SRC, "-I" + os.path.join(ROOT, "include"), str(chk)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
