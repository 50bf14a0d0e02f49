/*
 * swg_scoring.c -- residue map, scoring table and substitution-matrix reader.
 * Host-only C; behaviour mirrors reference src/alignment_scoring.c and
 * src/alignment_scoring_load.c (file format: SURVEY A.5) but returns error
 * codes where the reference prints and exits.
 */
#include "../../include/swg.h"
#include "../../include/swg_host.h"

#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

int swg_letter_index(int c)
{
    /* reference src/alignment_scoring.c:70-81 */
    if (c >= 'a' && c <= 'z') return c - 'a' + 1;
    if (c >= 'A' && c <= 'Z') return c - 'A' + 1;
    if (c == '*') return 31;
    return -1;
}

int swg_index_letter(int idx)
{
    /* reference src/alignment_scoring.c:83-92 */
    if (idx >= 1 && idx <= 26) return 'A' + idx - 1;
    if (idx == 31) return '*';
    return 0;
}

void swg_scoring_init(swg_scoring *sc)
{
    /* reference src/tools/sw_cmdline.c:27-35; unlike the reference the table
     * starts zeroed (the reference leaves swap_scores uninitialised, A.7-1) */
    memset(sc, 0, sizeof *sc);
    sc->match = 2;
    sc->mismatch = -2;
    sc->gap_open = -2;
    sc->gap_extend = -1;
}

int swg_scoring_add(swg_scoring *sc, int a, int b, int score)
{
    /* reference src/alignment_scoring.c:60-68 */
    const int ia = swg_letter_index(a), ib = swg_letter_index(b);
    if (ia < 0 || ib < 0 || score <= -128 || score >= 128) return SWG_ERR_ARG;
    sc->sub[ia][ib] = (int8_t)score;
    sc->set[ia] |= 1u << ib;
    return SWG_OK;
}

void swg_query_sanitize(const swg_scoring *sc, int8_t *idx, size_t n)
{
    /* reference src/alignment_cmdline.c:391-396 */
    const int8_t x = (int8_t)swg_letter_index('X');
    for (size_t i = 0; i < n; i++) {
        const int a = idx[i];
        if (a < 0 || a > 31 || !((sc->set[a] >> a) & 1u)) idx[i] = x;
    }
}

/* ---- line reader over zlib (plain files pass through gzread unchanged) ---- */
typedef struct {
    gzFile f;
    char *buf;
    size_t cap, len;
} linebuf;

static int lb_next(linebuf *lb)
{
    lb->len = 0;
    for (;;) {
        if (lb->cap - lb->len < 256) {
            size_t ncap = lb->cap ? lb->cap * 2 : 1024;
            char *nb = (char *)realloc(lb->buf, ncap);
            if (!nb) return -1;
            lb->buf = nb;
            lb->cap = ncap;
        }
        if (!gzgets(lb->f, lb->buf + lb->len, (int)(lb->cap - lb->len))) {
            if (lb->len == 0) return 0;
            break;
        }
        lb->len += strlen(lb->buf + lb->len);
        if (lb->len && lb->buf[lb->len - 1] == '\n') break;
    }
    while (lb->len && (lb->buf[lb->len - 1] == '\n' || lb->buf[lb->len - 1] == '\r'))
        lb->buf[--lb->len] = '\0';
    return 1;
}

static int all_space(const char *s)
{
    for (; *s; s++)
        if (!isspace((unsigned char)*s)) return 0;
    return 1;
}

static int fail(char *err, size_t errlen, const char *path, long line, const char *msg)
{
    if (err && errlen) {
        if (line >= 0)
            snprintf(err, errlen, "substitution matrix : %s (file %s, line %ld)", msg, path, line);
        else
            snprintf(err, errlen, "substitution matrix : %s (file %s)", msg, path);
    }
    return SWG_ERR_IO;
}

int swg_scoring_load_matrix(swg_scoring *sc, const char *path, char *err, size_t errlen)
{
    if (!sc || !path) return SWG_ERR_ARG;
    linebuf lb = {0};
    lb.f = gzopen(path, "r");
    if (!lb.f) return fail(err, errlen, path, -1, "couldn't read file");
    int rc = SWG_OK;
    long line = 0;
    int got;
    /* header: first line that is neither empty, all-space nor a '#' comment
     * (src/alignment_scoring_load.c:64-80) */
    while ((got = lb_next(&lb)) > 0) {
        line++;
        if (lb.len > 0 && lb.buf[0] != '#' && !all_space(lb.buf)) break;
    }
    if (got <= 0) {
        rc = fail(err, errlen, path, -1, "empty file");
        goto done;
    }
    if (lb.len < 2) {
        rc = fail(err, errlen, path, line, "too few column headings");
        goto done;
    }
    {
        /* the header's first character is the separator (:88-93) */
        const char sep = lb.buf[0];
        if ((sep >= '0' && sep <= '9') || sep == '-') {
            rc = fail(err, errlen, path, line, "numbers (0-9) and dashes (-) do not make good separators");
            goto done;
        }
        char cols[256];
        int ncols = 0;
        if (isspace((unsigned char)sep)) {
            /* whitespace mode (:98-151) */
            for (const char *p = lb.buf; *p; p++) {
                if (isspace((unsigned char)*p)) continue;
                if (ncols >= 255) {
                    rc = fail(err, errlen, path, line, "too many column headings");
                    goto done;
                }
                cols[ncols++] = *p;
                if (p[1] && !isspace((unsigned char)p[1])) {
                    rc = fail(err, errlen, path, line, "column headings must be single characters");
                    goto done;
                }
            }
            while ((got = lb_next(&lb)) > 0) {
                line++;
                const char *p = lb.buf;
                while (*p && isspace((unsigned char)*p)) p++;
                if (*p == '\0' || lb.buf[0] == '#') continue;
                const char from = *p++;
                for (int i = 0; i < ncols; i++) {
                    if (!isspace((unsigned char)*p)) {
                        rc = fail(err, errlen, path, line,
                                  *p ? "expected whitespace between elements" : "missing number value on line");
                        goto done;
                    }
                    while (*p && isspace((unsigned char)*p)) p++;
                    char *end = NULL;
                    const long v = strtol(p, &end, 10);
                    if (end == p) {
                        rc = fail(err, errlen, path, line, "missing number value on line");
                        goto done;
                    }
                    if (swg_scoring_add(sc, from, cols[i], (int)v) != SWG_OK) {
                        rc = fail(err, errlen, path, line, "illegal residue letter or score outside -127..127");
                        goto done;
                    }
                    p = end;
                }
                if (*p != '\0' && !all_space(p)) {
                    rc = fail(err, errlen, path, line, "too many columns on row");
                    goto done;
                }
            }
        } else {
            /* single-character separator mode (:152-211): "<sep>A<sep>B..." */
            for (size_t i = 0; i < lb.len; i += 2) {
                if (lb.buf[i] != sep || lb.buf[i + 1] == '\0' || ncols >= 255) {
                    rc = fail(err, errlen, path, line, "separator missing from line");
                    goto done;
                }
                cols[ncols++] = lb.buf[i + 1];
            }
            while ((got = lb_next(&lb)) > 0) {
                line++;
                if (lb.buf[0] == '#' || all_space(lb.buf)) continue;
                const char from = lb.buf[0];
                const char *p = lb.buf + 1;
                int i = 0;
                while (*p != '\0') {
                    if (*p != sep) {
                        rc = fail(err, errlen, path, line, "separator missing from line");
                        goto done;
                    }
                    p++;
                    char *end = NULL;
                    const long v = strtol(p, &end, 10);
                    if (end == p) {
                        rc = fail(err, errlen, path, line, "missing number value on line");
                        goto done;
                    }
                    if (i >= ncols) {
                        rc = fail(err, errlen, path, line, "too many columns on row");
                        goto done;
                    }
                    if (swg_scoring_add(sc, from, cols[i++], (int)v) != SWG_OK) {
                        rc = fail(err, errlen, path, line, "illegal residue letter or score outside -127..127");
                        goto done;
                    }
                    p = end;
                }
            }
        }
        if (got < 0) rc = fail(err, errlen, path, line, "out of memory");
    }
done:
    gzclose(lb.f);
    free(lb.buf);
    return rc;
}
