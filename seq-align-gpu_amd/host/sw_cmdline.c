/*
 * sw_cmdline.c -- the `smith_waterman` tool over libswg (plain C host code).
 *
 * Same command line, same stdout as the reference's tool (flags:
 * reference src/alignment_cmdline.c:205-292, output: src/tools/sw_cmdline.c:38-75
 * and src/alignment_cmdline.c:274,529-530), with the whole timed
 * fill region (src/alignment_cmdline.c:503-509) replaced by one swg_search()
 * on the GPU.  Deliberate differences, all from SURVEY A.7:
 *   - the database need not be length-sorted nor a multiple of 16 records;
 *   - "Entry #n" is always the true 0-based record index (A.7-3);
 *   - the substitution matrix is mandatory and undefined pairs score 0 (A.7-1,2);
 *   - scores above 32767 are exact instead of wrapped (A.4);
 *   - --topk K appends a ranked report, --align the alignments of those K (the traceback the
 *     reference's fork removed, re-run for the reported pairs only: swg_align_hits);
 *     --gpu N selects the device; --gpus N shards the database over devices 0..N-1 (one RCCL
 *     all-reduce merges the top-K lists).
 * There is no CPU backend: without a GPU the tool fails with a message.
 */
#define _POSIX_C_SOURCE 200809L
#include "../../include/swg.h"
#include "../../include/swg_host.h"

#include <limits.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>
#ifndef CLOCK_BOOTTIME
#define CLOCK_BOOTTIME 7
#endif

static void usage(const char *argv0, const char *err)
{
    if (err) fprintf(stderr, "Error: %s\n", err);
    fprintf(stderr,
            "usage: %s [OPTIONS] --substitution_matrix <file> --files <query> <database>\n"
            "  Smith-Waterman optimal local alignment score of one query against every\n"
            "  record of a database (FASTA/FASTQ/plain, gzip ok), on an AMD MI355X.\n\n"
            "  OPTIONS:\n"
            "    --files <f1> <f2>    query file (first record) and database file\n"
            "    --substitution_matrix <file>  scoring matrix (see data/*.txt)\n"
            "    --gapopen <score>    [default: -2]\n"
            "    --gapextend <score>  [default: -1]   gap of length N costs open + N*extend\n"
            "    --match <score> --mismatch <score>   accepted, unused with a matrix\n"
            "    --printseq           print sequences\n"
            "    --printfasta         print record names\n"
            "    --printmatrices --pretty --colour --scoring <x>   accepted, no effect\n"
            "    --topk <K>           append the K best hits (score, index, name)\n"
            "    --align              with --topk: append the alignment of every reported hit\n"
            "                         (query line over database line, '-' = gap; coordinates 0-based, end exclusive)\n"
            "    --timing             wall time of every phase (reading, packing, upload, search, printing) on stderr\n"
            "    --gpu <N>            HIP device ordinal [default: 0]\n"
            "    --gpus <N>           shard the database over GPUs 0..N-1 (RCCL top-K merge)\n"
            "    --savedb <file>      also write the packed database (sorted, binned, dword-packed)\n"
            "    --packed             the database file is such a packed database: no parsing,\n"
            "                         no sorting; record names and sequences are not in it\n"
            "    --allqueries         every record of the query file against the resident database\n"
            "                         (one block of output per query, headed `Query #n: name`)\n",
            argv0);
    exit(EXIT_FAILURE);
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* Milliseconds this process had been alive when called (its start time: /proc/self/stat field 22, in clock ticks
 * since boot): what the dynamic loader and the libraries' static initialisers took before main(). */
static double ms_since_process_start(void)
{
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1.0;
    char buf[1024];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')'); /* the command name may hold spaces */
    if (!p) return -1.0;
    unsigned long long start = 0;
    int field = 2;
    for (p++; *p && field < 22; p++)
        if (*p == ' ') field++;
    if (sscanf(p, "%llu", &start) != 1) return -1.0;
    struct timespec ts;
    clock_gettime(CLOCK_BOOTTIME, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6 - (double)start * 1e3 / (double)sysconf(_SC_CLK_TCK);
}

/* The HIP runtime's start-up (most of a one-shot run: ~200 ms) needs nothing of the input: it runs on a thread of
 * its own while the main thread reads, converts, sorts and packs the files. */
typedef struct {
    swg_config cfg;
    swg_ctx *ctx;
    int rc;
    char err[512];
    double ms;
} ctx_job;
static void *ctx_job_run(void *arg)
{
    ctx_job *j = (ctx_job *)arg;
    const double t0 = now_ms();
    j->rc = swg_create(&j->cfg, &j->ctx);
    if (j->rc != SWG_OK) snprintf(j->err, sizeof j->err, "%s", swg_global_error()); /* (thread-local text) */
    j->ms = now_ms() - t0;
    return NULL;
}

static pthread_t g_job_thread;
static int g_job_started = 0;
/* every way out of main() after the thread has started waits for it: the process must not run its exit handlers
 * while another thread is inside the HIP runtime's start-up */
static int leave(int code)
{
    if (g_job_started) {
        pthread_join(g_job_thread, NULL);
        g_job_started = 0;
    }
    return code;
}

/* --timing: wall time of the phases the reference leaves out of its `Total Time`, on stderr */
static int timing = 0;
static double phase_t0;
static void phase(const char *what)
{
    const double t = now_ms();
    if (timing) fprintf(stderr, "[timing] %-28s %9.2f ms\n", what, t - phase_t0);
    phase_t0 = t;
}

static int parse_int(const char *s, long lo, long hi, long *out)
{
    char *end = NULL;
    const long v = strtol(s, &end, 10);
    if (end == s || *end != '\0' || v < lo || v > hi) return 0;
    *out = v;
    return 1;
}

static void die_illegal(char c)
{
    /* reference src/alignment_scoring.c:78-79 */
    printf("Error: %c is not a legal character for the substitution matrix!\n", c);
    (void)leave(0);
    exit(1);
}

int main(int argc, char **argv)
{
    swg_scoring sc;
    swg_scoring_init(&sc);
    const char *qpath = NULL, *dbpath = NULL, *savedb = NULL;
    int print_seq = 0, print_fasta = 0, have_matrix = 0, packed = 0, allq = 0;
    long topk = 0, gpu = 0, gpus = 0, v;
    int align = 0;
    if (argc == 1) usage(argv[0], NULL);
    for (int i = 1; i < argc; i++)
        if (!strcasecmp(argv[i], "--help") || !strcasecmp(argv[i], "-help") || !strcasecmp(argv[i], "-h"))
            usage(argv[0], NULL);
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcasecmp(a, "--printseq")) print_seq = 1;
        else if (!strcasecmp(a, "--printfasta")) print_fasta = 1;
        else if (!strcasecmp(a, "--printmatrices") || !strcasecmp(a, "--pretty") || !strcasecmp(a, "--colour")) {
        } else if (!strcasecmp(a, "--stdin")) {
            /* reference src/alignment_cmdline.c:219-222: cmdline_set_files(cmd, "", NULL) -- a query path
             * and NO database, which its cmdline_new then refuses with "No input specified" (:303-305) */
            qpath = "";
            dbpath = NULL;
        } else if (!strcasecmp(a, "--packed")) { /* this tool's flag-only options: valid in last position too */
            packed = 1;
        } else if (!strcasecmp(a, "--align")) {
            align = 1;
        } else if (!strcasecmp(a, "--timing")) {
            timing = 1;
        } else if (!strcasecmp(a, "--allqueries")) {
            allq = 1;
        } else if (i == argc - 1) {
            char msg[256];
            snprintf(msg, sizeof msg, "Unknown argument without parameter: %s", a);
            usage(argv[0], msg);
        } else if (!strcasecmp(a, "--scoring")) {
            i++;
        } else if (!strcasecmp(a, "--substitution_matrix")) {
            char err[512];
            if (swg_scoring_load_matrix(&sc, argv[i + 1], err, sizeof err) != SWG_OK) {
                fprintf(stderr, "Error: %s\n", err);
                return EXIT_FAILURE;
            }
            have_matrix = 1;
            i++;
        } else if (!strcasecmp(a, "--match") || !strcasecmp(a, "--mismatch")) {
            if (!parse_int(argv[i + 1], INT_MIN, INT_MAX, &v)) usage(argv[0], "Invalid --match/--mismatch argument, must be an int");
            if (!strcasecmp(a, "--match")) sc.match = (int)v; else sc.mismatch = (int)v;
            i++;
        } else if (!strcasecmp(a, "--gapopen") || !strcasecmp(a, "--gapextend")) {
            /* the reference's score_t is int16 (src/alignment_cmdline.c:255-267) */
            if (!parse_int(argv[i + 1], SHRT_MIN, SHRT_MAX, &v)) usage(argv[0], "Invalid --gapopen/--gapextend argument, must be an int");
            if (!strcasecmp(a, "--gapopen")) sc.gap_open = (int)v; else sc.gap_extend = (int)v;
            i++;
        } else if (!strcasecmp(a, "--topk")) {
            if (!parse_int(argv[i + 1], 0, 1 << 20, &topk)) usage(argv[0], "Invalid --topk argument");
            i++;
        } else if (!strcasecmp(a, "--savedb")) {
            if (i >= argc - 1) usage(argv[0], "--savedb takes a file name");
            savedb = argv[++i];
        } else if (!strcasecmp(a, "--file")) {
            /* reference src/alignment_cmdline.c:268-270: cmdline_set_files(cmd, argv[argi + 1], NULL) */
            qpath = argv[i + 1];
            dbpath = NULL;
            i++;
        } else if (!strcasecmp(a, "--gpus")) {
            if (!parse_int(argv[i + 1], 1, 64, &gpus)) usage(argv[0], "Invalid --gpus argument");
            i++;
        } else if (!strcasecmp(a, "--gpu")) {
            if (!parse_int(argv[i + 1], 0, 1023, &gpu)) usage(argv[0], "Invalid --gpu argument");
            i++;
        } else if (!strcasecmp(a, "--files")) {
            if (i >= argc - 2) usage(argv[0], "--files option takes 2 arguments");
            /* reference src/alignment_cmdline.c:274 */
            printf("Query File=%s and Database File=%s\n", argv[i + 1], argv[i + 2]);
            qpath = argv[i + 1];
            dbpath = argv[i + 2];
            i += 2;
        } else {
            char msg[256];
            snprintf(msg, sizeof msg, "Unknown argument '%s'", a);
            usage(argv[0], msg);
        }
    }
    if (!qpath || !dbpath) usage(argv[0], "No input specified"); /* reference src/alignment_cmdline.c:303-305 */
    if (!have_matrix) usage(argv[0], "--substitution_matrix is required (the fill scores from the matrix only)");
    if (packed && (print_seq || print_fasta)) usage(argv[0], "--printseq/--printfasta need the FASTA database, not --packed");
    if ((packed || savedb || allq) && gpus > 0) usage(argv[0], "--packed/--savedb/--allqueries work with one GPU (--gpu)");
    if (align && topk == 0) usage(argv[0], "--align reports the alignments of the --topk hits: give --topk K");

    char err[512];
    swg_seqs q, db;
    if (timing) fprintf(stderr, "[timing] %-28s %9.2f ms\n", "process start to main()", ms_since_process_start());
    /* one GPU: the context is created beside the reading and packing (several GPUs: swg_group_create, below) */
    ctx_job job;
    memset(&job, 0, sizeof job);
    if (gpus == 0) {
        job.cfg.device = (int)gpu;
        g_job_started = pthread_create(&g_job_thread, NULL, ctx_job_run, &job) == 0;
    }
    phase_t0 = now_ms();
    if (swg_seqs_read(qpath, allq ? 0 : 1, &q, err, sizeof err) != SWG_OK) {
        fprintf(stderr, "Error: couldn't open query file %s\n", qpath);
        return leave(EXIT_SUCCESS); /* the reference returns from the driver and exits 0 */
    }
    if (q.n == 0 || q.seq_off[1] == 0) {
        fprintf(stderr, "Error: Query file %s is empty or invalid\n", qpath);
        return leave(EXIT_SUCCESS);
    }
    swg_db *pdb = NULL;
    memset(&db, 0, sizeof db);
    if (packed) {
        if (swg_db_load(dbpath, &pdb) != SWG_OK) {
            fprintf(stderr, "Error: %s\n", swg_global_error());
            return leave(EXIT_SUCCESS);
        }
        db.n = swg_db_total_count(pdb);
    } else if (swg_seqs_read(dbpath, 0, &db, err, sizeof err) != SWG_OK) {
        fprintf(stderr, "Error: couldn't open database file %s\n", dbpath);
        return leave(EXIT_SUCCESS);
    }
    phase(packed ? "read query, load packed db" : "read query and database");
    const size_t lq = (size_t)q.seq_off[1];
    int8_t *qidx = (int8_t *)malloc(lq);
    int8_t *didx = (int8_t *)malloc(db.n && !packed ? (size_t)db.seq_off[db.n] + 1 : 1);
    if (!qidx || !didx) {
        fprintf(stderr, "Error: out of memory\n");
        return leave(EXIT_FAILURE);
    }
    char bad = 0;
    {
        swg_seqs q1 = q;
        q1.n = 1;
        if (swg_seqs_to_indices(&q1, qidx, &bad) != SWG_OK) die_illegal(bad);
    }
    swg_query_sanitize(&sc, qidx, lq); /* reference src/alignment_cmdline.c:391-396 */
    if (!packed && swg_seqs_to_indices(&db, didx, &bad) != SWG_OK) die_illegal(bad);

    phase("letters to table indices");
    int32_t *scores = (int32_t *)calloc(db.n ? db.n : 1, sizeof(int32_t));
    swg_hit *hits = (swg_hit *)calloc(topk ? (size_t)topk : 1, sizeof(swg_hit));
    if (!scores || !hits) {
        fprintf(stderr, "Error: out of memory\n");
        return leave(EXIT_FAILURE);
    }
    size_t n_hits = 0;
    double total_ms = 0.0;
    swg_ctx *ctx = NULL;
    swg_group *grp = NULL;
    if (gpus > 0) {
        /* database sharded over several GPUs of this process */
        swg_stats *st = (swg_stats *)calloc((size_t)gpus, sizeof(swg_stats));
        if (!st) return leave(EXIT_FAILURE);
        int rc = swg_group_create(NULL, (int)gpus, 0, &grp);
        if (rc != SWG_OK) {
            fprintf(stderr, "Error: %s\n", swg_global_error());
            return leave(EXIT_FAILURE);
        }
        phase("create contexts");
        /* one search per query length: timing candidate geometries first would cost more than it saves */
        rc = swg_group_set_option(grp, "autotune", 0);
        if (rc == SWG_OK) rc = swg_group_set_scoring(grp, (const int8_t(*)[32])sc.sub, sc.gap_open, sc.gap_extend);
        if (rc == SWG_OK) rc = swg_group_set_query(grp, qidx, lq);
        if (rc == SWG_OK) rc = swg_group_load(grp, didx, db.seq_off, db.n);
        phase("pack, shard and upload");
        if (rc == SWG_OK) rc = swg_group_search(grp, scores, hits, (size_t)topk, &n_hits, st);
        phase("search");
        if (rc != SWG_OK) {
            fprintf(stderr, "Error: %s\n", swg_group_last_error(grp));
            return leave(EXIT_FAILURE);
        }
        for (long g = 0; g < gpus; g++)
            if (st[g].total_ms > total_ms) total_ms = st[g].total_ms; /* the GPUs run side by side */
        free(st);
    } else {
        swg_stats st;
        memset(&st, 0, sizeof st);
        int rc = SWG_OK;
        if (!packed) { /* host only: still beside the context's creation */
            rc = swg_db_pack(didx, db.seq_off, db.n, 0, 1, &pdb);
            if (rc != SWG_OK) fprintf(stderr, "Error: %s\n", swg_global_error());
            phase("sort and pack");
        }
        if (g_job_started) {
            (void)leave(0);
        } else {
            ctx_job_run(&job);
        }
        ctx = job.ctx;
        if (job.rc != SWG_OK) {
            fprintf(stderr, "Error: %s\n", job.err);
            return leave(EXIT_FAILURE);
        }
        if (timing) fprintf(stderr, "[timing] %-28s %9.2f ms (on its own thread, beside the phases above)\n", "create context", job.ms);
        phase("wait for the context");
        /* one search per query length: timing candidate geometries first would cost more than it saves */
        if (rc == SWG_OK) rc = swg_set_option(ctx, "autotune", 0);
        if (rc == SWG_OK) rc = swg_set_scoring(ctx, (const int8_t(*)[32])sc.sub, sc.gap_open, sc.gap_extend);
        if (rc == SWG_OK) rc = swg_set_query(ctx, qidx, lq);
        if (rc == SWG_OK && savedb) {
            if (swg_db_save(pdb, savedb) != SWG_OK) {
                fprintf(stderr, "Error: %s\n", swg_global_error());
                return leave(EXIT_FAILURE);
            }
            fprintf(stderr, "packed database written to %s\n", savedb);
        }
        if (rc == SWG_OK) rc = swg_db_upload(ctx, pdb);
        phase("upload");
        if (rc == SWG_OK) rc = swg_search(ctx, pdb, scores, hits, (size_t)topk, &n_hits, &st);
        phase("search (first of this database)");
        if (rc != SWG_OK) {
            fprintf(stderr, "Error: %s\n", swg_last_error(ctx));
            return leave(EXIT_FAILURE);
        }
        total_ms = st.total_ms;
    }

    /* reference src/tools/sw_cmdline.c:38-75: per 16 records the query lines, then per record */
    const char *qname = q.names + q.name_off[0];
    size_t qi = 0;
next_query:
    if (allq) printf("Query #%lu: %s\n", (unsigned long)qi, qname);
    for (size_t i = 0; i < db.n; i++) {
        if (i % 16 == 0) {
            if (print_fasta) {
                fputs(qname, stdout);
                putc('\n', stdout);
            }
            if (print_seq) {
                fwrite(q.seq + q.seq_off[qi], 1, (size_t)(q.seq_off[qi + 1] - q.seq_off[qi]), stdout);
                putc('\n', stdout);
            }
        }
        printf("Entry #%lu:\n", (unsigned long)i);
        if (print_fasta) {
            fputs(db.names + db.name_off[i], stdout);
            putc('\n', stdout);
        }
        if (print_seq) {
            fwrite(db.seq + db.seq_off[i], 1, (size_t)(db.seq_off[i + 1] - db.seq_off[i]), stdout);
            putc('\n', stdout);
        }
        printf("score: %i\n\n", scores[i]);
    }
    /* reference src/alignment_cmdline.c:529-530; the time is the device time of the fill */
    printf("Total Time: %f\n", total_ms * 1e-3);
    printf("Total Entries: %lu\n", (unsigned long)db.n);
    if (topk > 0) {
        printf("Top %lu hits (score, entry, name):\n", (unsigned long)n_hits);
        for (size_t i = 0; i < n_hits; i++)
            printf("%d\t%u\t%s\n", hits[i].score, hits[i].index, packed ? "" : db.names + db.name_off[hits[i].index]);
    }
    if (align && n_hits > 0) {
        const size_t stride = grp ? swg_group_align_ops_bound(grp) : swg_align_ops_bound(ctx, pdb);
        swg_alignment *al = (swg_alignment *)calloc(n_hits, sizeof *al);
        char *ops = (char *)malloc(n_hits * stride);
        char *line = (char *)malloc(stride);
        if (!al || !ops || !line) return leave(EXIT_FAILURE);
        if ((grp ? swg_group_align_hits(grp, hits, n_hits, al, ops, stride)
                 : swg_align_hits(ctx, pdb, hits, n_hits, al, ops, stride)) != SWG_OK) {
            fprintf(stderr, "Error: %s\n", grp ? swg_group_last_error(grp) : swg_last_error(ctx));
            return leave(EXIT_FAILURE);
        }
        for (size_t i = 0; i < n_hits; i++) {
            const swg_alignment *a = &al[i];
            const char *o = ops + i * stride;
            printf("Alignment #%lu: entry %u score %d query %u..%u entry %u..%u\n", (unsigned long)i, a->index,
                   a->score, a->q_begin, a->q_end, a->d_begin, a->d_end);
            if (packed) { /* no letters in a packed database: the path itself */
                printf("%s\n\n", o);
                continue;
            }
            const char *qs = q.seq + q.seq_off[qi] + a->q_begin, *ds = db.seq + db.seq_off[a->index] + a->d_begin;
            size_t c = 0;
            for (uint32_t k = 0; k < a->n_ops; k++) line[k] = o[k] == 'I' ? '-' : qs[c++];
            line[a->n_ops] = 0;
            printf("%s\n", line);
            c = 0;
            for (uint32_t k = 0; k < a->n_ops; k++) line[k] = o[k] == 'D' ? '-' : ds[c++];
            printf("%s\n\n", line);
        }
        free(al);
        free(ops);
        free(line);
    }
    if (allq && ++qi < q.n) {
        /* The database stays resident; the remaining queries go through swg_search_multi in chunks (one
         * launch per class for a whole chunk: a small database is filled with many queries at once),
         * results are kept per query and printed in order. */
        static int32_t *mq_scores = NULL;
        static swg_hit *mq_hits = NULL;
        static size_t *mq_nhits = NULL;
        static size_t chunk_first = 0, chunk_n = 0;
        static double chunk_ms = 0.0;
        static int8_t *qx = NULL;     /* the chunk's queries as table indices */
        static uint64_t *qoff = NULL;
        if (qi >= chunk_first + chunk_n) {
            size_t budget = ((size_t)256 << 20) / (sizeof(int32_t) * (db.n ? db.n : 1));
            if (budget < 1) budget = 1;
            if (budget > 1024) budget = 1024;
            chunk_first = qi;
            chunk_n = q.n - qi < budget ? q.n - qi : budget;
            const size_t nres = (size_t)(q.seq_off[chunk_first + chunk_n] - q.seq_off[chunk_first]);
            free(qx);
            free(qoff);
            qx = (int8_t *)malloc(nres ? nres : 1);
            qoff = (uint64_t *)malloc((chunk_n + 1) * sizeof(uint64_t));
            free(mq_scores);
            free(mq_hits);
            free(mq_nhits);
            mq_scores = (int32_t *)calloc(chunk_n * (db.n ? db.n : 1), sizeof(int32_t));
            mq_hits = (swg_hit *)calloc(chunk_n * (topk ? (size_t)topk : 1), sizeof(swg_hit));
            mq_nhits = (size_t *)calloc(chunk_n, sizeof(size_t));
            if (!qx || !qoff || !mq_scores || !mq_hits || !mq_nhits) return leave(EXIT_FAILURE);
            for (size_t i = 0; i <= chunk_n; i++) qoff[i] = q.seq_off[chunk_first + i] - q.seq_off[chunk_first];
            for (size_t i = 0; i < chunk_n; i++) {
                const size_t lqi = (size_t)(qoff[i + 1] - qoff[i]);
                if (lqi == 0) {
                    fprintf(stderr, "Error: query #%lu is empty\n", (unsigned long)(chunk_first + i));
                    return leave(EXIT_FAILURE);
                }
                for (size_t c = 0; c < lqi; c++) {
                    const char ch = q.seq[q.seq_off[chunk_first + i] + c];
                    const int v = swg_letter_index((unsigned char)ch);
                    if (v < 0) die_illegal(ch);
                    qx[qoff[i] + c] = (int8_t)v;
                }
                swg_query_sanitize(&sc, qx + qoff[i], lqi);
            }
            swg_stats st;
            memset(&st, 0, sizeof st);
            const int rc = swg_search_multi(ctx, pdb, qx, qoff, chunk_n, mq_scores, mq_hits, (size_t)topk, mq_nhits, &st);
            if (rc != SWG_OK) {
                fprintf(stderr, "Error: %s\n", swg_last_error(ctx));
                return leave(EXIT_FAILURE);
            }
            chunk_ms = st.total_ms / (double)chunk_n; /* the chunk's device time, shared out over its queries */
            if (timing)
                fprintf(stderr, "[timing] %lu queries in one pass: %.3f ms of fill, %.1f GCUPS\n", (unsigned long)chunk_n,
                        st.fill_ms, st.fill_ms > 0 ? (double)st.cells / (st.fill_ms * 1e-3) / 1e9 : 0.0);
        }
        const size_t at = qi - chunk_first;
        memcpy(scores, mq_scores + at * (db.n ? db.n : 1), db.n * sizeof(int32_t));
        n_hits = mq_nhits[at];
        memcpy(hits, mq_hits + at * (topk ? (size_t)topk : 1), n_hits * sizeof(swg_hit));
        if (align && swg_set_query(ctx, qx + qoff[at], (size_t)(qoff[at + 1] - qoff[at])) != SWG_OK) {
            /* (the alignments of the hits are made against the context's query) */
            fprintf(stderr, "Error: %s\n", swg_last_error(ctx));
            return leave(EXIT_FAILURE);
        }
        total_ms = chunk_ms;
        qname = q.names + q.name_off[qi];
        goto next_query;
    }
    fflush(stdout);
    phase("print");
    /* The run is over and its output is out: the process ends here, without walking the runtime's teardown (streams,
     * device buffers, the runtime's own exit handlers: 25-40 ms of a 200 ms run) -- the kernel driver releases a dead
     * process's device resources anyway.  SWG_CLI_RELEASE=1 takes the long way (leak checkers want it). */
    if (!getenv("SWG_CLI_RELEASE")) {
        fflush(stderr);
        _exit(leave(EXIT_SUCCESS));
    }
    swg_db_free(pdb);
    swg_destroy(ctx);
    swg_group_destroy(grp);
    swg_seqs_free(&q);
    swg_seqs_free(&db);
    free(qidx);
    free(didx);
    free(scores);
    free(hits);
    phase("release (context, buffers)");
    return leave(EXIT_SUCCESS);
}
