/*
 * swg_seqio.c -- minimal sequence-file reader (FASTA / FASTQ / plain, gzip
 * transparent).  Stands in for the un-vendored libs/seq_file the reference
 * driver calls (seq_open/seq_read, src/alignment_cmdline.c:335-339, 370-386,
 * 422-457): name = header line without its '>' or '@', sequence = all sequence
 * lines joined, newlines dropped.  Host-only C.
 */
#include "../../include/swg.h"
#include "../../include/swg_host.h"

#include <ctype.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    char *p;
    size_t len, cap;
} cbuf;

static int cb_add(cbuf *b, const char *s, size_t n)
{
    if (b->len + n + 1 > b->cap) {
        size_t nc = b->cap ? b->cap : 4096;
        while (nc < b->len + n + 1) nc *= 2;
        char *np = (char *)realloc(b->p, nc);
        if (!np) return -1;
        b->p = np;
        b->cap = nc;
    }
    memcpy(b->p + b->len, s, n);
    b->len += n;
    b->p[b->len] = '\0';
    return 0;
}

typedef struct {
    uint64_t *p;
    size_t len, cap;
} obuf;

static int ob_add(obuf *b, uint64_t v)
{
    if (b->len == b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 1024;
        uint64_t *np = (uint64_t *)realloc(b->p, nc * sizeof(uint64_t));
        if (!np) return -1;
        b->p = np;
        b->cap = nc;
    }
    b->p[b->len++] = v;
    return 0;
}

void swg_seqs_free(swg_seqs *s)
{
    if (!s) return;
    free(s->names);
    free(s->name_off);
    free(s->seq);
    free(s->seq_off);
    memset(s, 0, sizeof *s);
}

/* ---- plain (uncompressed) FASTA files: parsed from a memory map by all cores ---------------- */
/* One pass over [lo, hi) of the map, which starts at a header line.  With out == NULL it only
 * counts (records, name bytes including the terminators, residue bytes); otherwise it writes
 * the records at the given positions.  Same rules as the line reader below: a line starting with
 * '>' opens a record, every other non-empty line is sequence, white space inside it is dropped. */
typedef struct {
    size_t n, name_bytes, seq_bytes;
} fa_count;

static fa_count fa_scan(const char *d, size_t lo, size_t hi, swg_seqs *out, size_t rec, size_t npos, size_t spos)
{
    fa_count c = {0, 0, 0};
    size_t p = lo;
    while (p < hi) {
        size_t e = p;
        while (e < hi && d[e] != '\n') e++;
        size_t len = e - p;
        while (len && (d[p + len - 1] == '\r' || d[p + len - 1] == '\n')) len--;
        if (len && d[p] == '>') {
            if (out) {
                out->name_off[rec + c.n] = npos + c.name_bytes;
                out->seq_off[rec + c.n] = spos + c.seq_bytes;
                memcpy(out->names + npos + c.name_bytes, d + p + 1, len - 1);
                out->names[npos + c.name_bytes + len - 1] = '\0';
            }
            c.name_bytes += len; /* len - 1 characters and the terminator */
            c.n++;
        } else if (len) {
            for (size_t i = 0; i < len; i++) {
                const unsigned char ch = (unsigned char)d[p + i];
                if (!isspace(ch)) {
                    if (out) out->seq[spos + c.seq_bytes] = (char)ch;
                    c.seq_bytes++;
                }
            }
        }
        p = e + 1;
    }
    return c;
}

/* 1: read; 0: not a plain FASTA file of useful size (the line reader takes it); < 0: error */
static int read_fasta_mapped(const char *path, swg_seqs *out)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return 0;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < (1 << 20)) {
        close(fd);
        return 0;
    }
    const size_t size = (size_t)sb.st_size;
    const char *d = (const char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (d == MAP_FAILED) return 0;
    size_t first = 0;
    while (first < size && (d[first] == '\n' || d[first] == '\r')) first++; /* empty lines, as the line reader skips them */
    if (first >= size || d[first] != '>' || (first > 0 && d[first - 1] != '\n')) { /* gzip, FASTQ, plain lines */
        munmap((void *)d, size);
        return 0;
    }
    int T = swg_host_threads();
    if (T > 256) T = 256;
    size_t cut[257];
    cut[0] = first;
    for (int t = 1; t < T; t++) {
        size_t p = first + (size - first) / (size_t)T * (size_t)t;
        if (p < cut[t - 1]) p = cut[t - 1];
        while (p < size && !(d[p] == '>' && d[p - 1] == '\n')) p++; /* next header line */
        cut[t] = p;
    }
    cut[T] = size;
    fa_count cnt[256];
#pragma omp parallel for schedule(static, 1) num_threads(swg_host_threads())
    for (int t = 0; t < T; t++) cnt[t] = fa_scan(d, cut[t], cut[t + 1], NULL, 0, 0, 0);
    size_t n = 0, nb = 0, sbytes = 0;
    size_t rec0[256], np0[256], sp0[256];
    for (int t = 0; t < T; t++) {
        rec0[t] = n;
        np0[t] = nb;
        sp0[t] = sbytes;
        n += cnt[t].n;
        nb += cnt[t].name_bytes;
        sbytes += cnt[t].seq_bytes;
    }
    out->names = (char *)malloc(nb + 1);
    out->seq = (char *)malloc(sbytes + 1);
    out->name_off = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    out->seq_off = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    if (!out->names || !out->seq || !out->name_off || !out->seq_off) {
        munmap((void *)d, size);
        swg_seqs_free(out);
        return SWG_ERR_NOMEM;
    }
#pragma omp parallel for schedule(static, 1) num_threads(swg_host_threads())
    for (int t = 0; t < T; t++) (void)fa_scan(d, cut[t], cut[t + 1], out, rec0[t], np0[t], sp0[t]);
    out->n = n;
    out->name_off[n] = nb;
    out->seq_off[n] = sbytes;
    out->names[nb] = '\0';
    out->seq[sbytes] = '\0';
    munmap((void *)d, size);
    return 1;
}

int swg_seqs_read(const char *path, size_t max_records, swg_seqs *out, char *err, size_t errlen)
{
    if (!path || !out) return SWG_ERR_ARG;
    memset(out, 0, sizeof *out);
    if (max_records == 0 && strcmp(path, "-") != 0) {
        const int rc = read_fasta_mapped(path, out);
        if (rc == 1) return SWG_OK;
        if (rc < 0) {
            if (err) snprintf(err, errlen, "out of memory reading %s", path);
            return rc;
        }
        memset(out, 0, sizeof *out);
    }
    gzFile f = strcmp(path, "-") == 0 ? gzdopen(0, "r") : gzopen(path, "r");
    if (!f) {
        if (err) snprintf(err, errlen, "couldn't open %s", path);
        return SWG_ERR_IO;
    }
    gzbuffer(f, 1 << 20);
    cbuf names = {0}, seq = {0};
    obuf noff = {0}, soff = {0};
    size_t cap = 1 << 16;
    char *line = (char *)malloc(cap);
    int rc = SWG_OK;
    int state = 0; /* 0 none, 1 fasta body, 2 fastq seq, 3 fastq '+', 4 fastq qual */
    size_t n = 0;
    int oom = !line;
    while (!oom) {
        /* read one full line */
        size_t len = 0;
        int eof = 0;
        for (;;) {
            if (!gzgets(f, line + len, (int)(cap - len))) {
                eof = 1;
                break;
            }
            len += strlen(line + len);
            if (len && line[len - 1] == '\n') break;
            if (cap - len < 2) {
                char *nl = (char *)realloc(line, cap * 2);
                if (!nl) {
                    oom = 1;
                    break;
                }
                line = nl;
                cap *= 2;
            }
        }
        if (oom) break;
        if (eof && len == 0) break;
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = '\0';
        if (state == 3) { /* FASTQ '+' seen: this is the quality line */
            state = 0;
            if (eof) break;
            continue;
        }
        if (len == 0) {
            if (eof) break;
            continue;
        }
        if ((line[0] == '>' && state != 2) || (line[0] == '@' && (state == 0))) {
            if (max_records && n == max_records) break;
            oom |= ob_add(&noff, names.len) || ob_add(&soff, seq.len);
            oom |= cb_add(&names, line + 1, len - 1) || cb_add(&names, "", 1);
            names.len -= 0;
            n++;
            state = line[0] == '>' ? 1 : 2;
        } else if (state == 2 && line[0] == '+') {
            state = 3;
        } else if (state == 1 || state == 2) {
            /* sequence line: drop embedded blanks */
            size_t w = 0;
            for (size_t i = 0; i < len; i++)
                if (!isspace((unsigned char)line[i])) line[w++] = line[i];
            oom |= cb_add(&seq, line, w);
        } else {
            /* plain: one sequence per line, no name */
            if (max_records && n == max_records) break;
            oom |= ob_add(&noff, names.len) || ob_add(&soff, seq.len);
            oom |= cb_add(&names, "", 1);
            size_t w = 0;
            for (size_t i = 0; i < len; i++)
                if (!isspace((unsigned char)line[i])) line[w++] = line[i];
            oom |= cb_add(&seq, line, w);
            n++;
        }
        if (eof) break;
    }
    if (!oom) oom |= ob_add(&noff, names.len) || ob_add(&soff, seq.len);
    if (!oom && !seq.p) oom |= cb_add(&seq, "", 0);
    if (!oom && !names.p) oom |= cb_add(&names, "", 0);
    gzclose(f);
    free(line);
    if (oom) {
        free(names.p);
        free(seq.p);
        free(noff.p);
        free(soff.p);
        if (err) snprintf(err, errlen, "out of memory reading %s", path);
        return SWG_ERR_NOMEM;
    }
    out->n = n;
    out->names = names.p;
    out->name_off = noff.p;
    out->seq = seq.p;
    out->seq_off = soff.p;
    return rc;
}

int swg_seqs_to_indices(const swg_seqs *s, int8_t *out, char *bad)
{
    if (!s || !out) return SWG_ERR_ARG;
    const uint64_t total = s->n ? s->seq_off[s->n] : 0;
    /* letters_to_index (reference src/alignment_scoring.c:70-81) as a table, all cores: at
     * 10M sequences this loop is 3.7e9 residues */
    int8_t lut[256];
    for (int c = 0; c < 256; c++) lut[c] = (int8_t)swg_letter_index(c);
    int64_t first_bad = -1;
#pragma omp parallel for schedule(static) reduction(max : first_bad) num_threads(swg_host_threads())
    for (int64_t blk = 0; blk < (int64_t)((total + 65535) / 65536); blk++) {
        const uint64_t lo = (uint64_t)blk * 65536, hi = lo + 65536 < total ? lo + 65536 : total;
        int8_t any = 0;
        for (uint64_t i = lo; i < hi; i++) {
            const int8_t v = lut[(unsigned char)s->seq[i]];
            out[i] = v;
            any |= v;
        }
        if (any < 0) first_bad = blk > first_bad ? blk : first_bad; /* some block with an illegal residue */
    }
    if (first_bad >= 0) {
        for (uint64_t i = 0; i < total; i++)
            if (lut[(unsigned char)s->seq[i]] < 0) {
                if (bad) *bad = s->seq[i];
                break;
            }
        return SWG_ERR_RESIDUE;
    }
    return SWG_OK;
}
