/*
 * grab_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of stealth/grab's
 * per-file match loop, used as the parity checker for the MI355X scan engine.
 *
 * Nothing in the product path (grab_amd/) may link, import or execute this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What is restated here (all citations relative to /root/reference):
 *   - chunk geometry + mmap loop ........ src/grab.cc:131-239
 *   - inner match loop / advance rule ... src/grab.cc:171-213
 *   - offset + line-extent formatting ... src/grab.cc:182-207
 *   - per-chunk flush, -s ............... src/grab.cc:217-234
 *   - pattern preparation ............... src/grab.cc:101-123
 *   - CLI flags, chunk-size rules ....... src/main.cc:110-173,231-265
 *   - recursive walk .................... src/grab.cc:260-279
 *
 * The regex arithmetic itself lives in a third-party dependency that is NOT under
 * /root/reference: libpcre (PCRE1), un-vendored and un-pinned by the reference
 * (src/Makefile:14 "-lpcre").  This image carries 8.39 (Ubuntu, JIT; the timing
 * build) and 8.45 (conda, no JIT); the oracle links whichever the Makefile names
 * and calls it exactly as the reference does (pcre_compile/pcre_study/
 * pcre_fullinfo/pcre_exec with options 0, ovector[3]).
 *
 * Parity pin: the reference has no tests of its own, so this
 * restatement is pinned against outputs of the reference itself, built from its
 * own sources into oracle/_ref/ (see oracle/Makefile) and frozen as fixtures under
 * tests/golden/ (tests/golden/make_golden.py).
 *
 * Besides the CLI (same flags as the reference), the file exports a small C API
 * (when built as liboracle.so) for in-memory checks:
 *   oracle_minlen()          PCRE_INFO_MINLENGTH as the reference reads it
 *   oracle_scan_chunk()      the inner loop of one chunk -> formatted bytes
 *   oracle_all_starts()      every offset p at which pcre_exec(ANCHORED) matches
 *                            with the subject starting at p (quirk Q4)
 */
#define _GNU_SOURCE
#define _XOPEN_SOURCE 700
#include <errno.h>
#include <fcntl.h>
#include <ftw.h>
#include <pcre.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#ifndef PCRE_STUDY_JIT_COMPILE
#define PCRE_STUDY_JIT_COMPILE 0
#endif

/* ---- growable output buffer (stands in for std::ostringstream, grab.cc:152) ---- */
typedef struct {
    char *p;
    size_t n, cap;
} obuf;

static void ob_put(obuf *o, const void *s, size_t n)
{
    if (o->n + n > o->cap) {
        size_t c = o->cap ? o->cap : 4096;
        while (c < o->n + n) c *= 2;
        o->p = (char *)realloc(o->p, c);
        o->cap = c;
    }
    memcpy(o->p + o->n, s, n);
    o->n += n;
}
static void ob_str(obuf *o, const char *s) { ob_put(o, s, strlen(s)); }

/* ---- state mirroring FileGrep's members (grab.h:41-53) ---- */
typedef struct {
    pcre *h;
    pcre_extra *x;
    int minlen;
    int print_line, print_offset, recursive, colored, print_path, single, low_mem;
    size_t chunk_size;
    char err[256];
} ogrep;

static void og_init(ogrep *g)
{
    memset(g, 0, sizeof(*g));
    g->minlen = 1;            /* grab.h:44 */
    g->print_line = 1;        /* grab.h:45 */
    g->chunk_size = 1u << 30; /* grab.h:48 */
}

/* grab.cc:101-123 */
static int og_prepare(ogrep *g, const char *regex)
{
    const char *errptr = NULL;
    int erroff = 0;
    g->h = pcre_compile(regex, 0, &errptr, &erroff, pcre_maketables());
    if (!g->h) {
        snprintf(g->err, sizeof g->err, "FileGrep::prepare::pcre_compile error");
        return -1;
    }
    g->x = pcre_study(g->h, PCRE_STUDY_JIT_COMPILE, &errptr);
    if (!g->x) {
        snprintf(g->err, sizeof g->err, "FileGrep::prepare::pcre_study error");
        return -1;
    }
    pcre_fullinfo(g->h, g->x, PCRE_INFO_MINLENGTH, &g->minlen);
    return 0;
}

/* How often pcre_exec gave up (PCRE_ERROR_MATCHLIMIT, PCRE_ERROR_JIT_STACKLIMIT, ...) since the library was loaded.
 * The reference treats that like "no further match" and ends the chunk silently; WHERE libpcre gives up is a property
 * of its implementation, so differential tests skip inputs on which this counter moves. */
static long g_resource_errors = 0;
long oracle_resource_errors(void) { return g_resource_errors; }

static const char start_inv[] = "\33[7m", stop_inv[] = "\33[27m"; /* grab.cc:66-67 */

/*
 * The inner loop of ONE chunk (grab.cc:171-213): content[0..clen), file offset
 * `off`, results appended to `o`.
 */
static void og_chunk(const ogrep *g, const char *path, const char *content, size_t clen,
                     long long off, obuf *o)
{
    int ovector[3];
    const char *start = content, *end = content + clen;
    char before[512], after[512];
    char num[64];

    for (; start + g->minlen < end;) { /* strict '<' : quirk Q3, grab.cc:175 */
        memset(ovector, 0, sizeof ovector);
        int rc = pcre_exec(g->h, g->x, start, (int)(end - start), 0, 0, ovector, 3);
        if (rc < PCRE_ERROR_NOMATCH) g_resource_errors++; /* match limit, JIT stack ...: see oracle_resource_errors() */
        if (rc <= 0) /* grab.cc:179 : errors and rc==0 (captures, Q5) end the chunk */
            break;

        if (g->recursive || g->print_path) { /* grab.cc:182-183 */
            ob_str(o, path);
            ob_put(o, ":", 1);
        }
        if (g->print_offset) { /* grab.cc:185-186 */
            int k = snprintf(num, sizeof num, "Match at offset %lld\n",
                             off + (long long)(start - content) + (long long)ovector[0]);
            ob_put(o, num, (size_t)k);
        }

        unsigned a = 0, b = sizeof(before) - 1; /* grab.cc:188 */
        if (g->print_line) {
            const char *ptr = start + ovector[0] - 1;
            while (ptr >= start && *ptr != '\n' && b > 0) /* grab.cc:192-193 */
                before[b--] = *ptr--;
            ptr = start + ovector[1];
            while (ptr < end && *ptr != '\n' && a < sizeof(after) - 1) /* grab.cc:195-196 */
                after[a++] = *ptr++;
            ob_put(o, before + b + 1, sizeof(before) - b - 1);
            if (g->colored) ob_str(o, start_inv);
            ob_put(o, start + ovector[0], (size_t)(ovector[1] - ovector[0]));
            if (g->colored) ob_str(o, stop_inv);
            ob_put(o, after, a);
            ob_put(o, "\n", 1);
        } else if (!g->print_offset) { /* grab.cc:204-207 */
            ob_str(o, "matches\n");
            break;
        }

        start += ovector[1] + a; /* grab.cc:209 */

        if (g->single) /* grab.cc:211-212 */
            break;
    }
}

#ifndef ORACLE_NO_MAIN
/* grab.cc:131-239 */
static int og_find_file(ogrep *g, const char *path, const struct stat *st)
{
    size_t clen = (size_t)st->st_size;
    if ((size_t)g->minlen > clen) /* grab.cc:133-135 ; minlen==-1 skips everything (Q2) */
        return 0;

    int flags = O_RDONLY | O_NOCTTY;
    uid_t me = geteuid();
    if (st->st_uid == me || me == 0) flags |= O_NOATIME; /* grab.cc:139-143 */
    int fd = open(path, flags);
    if (fd < 0) {
        snprintf(g->err, sizeof g->err, "FileGrep::find::open: %s", strerror(errno));
        return -1;
    }

    const off_t overlap = 0x1000; /* grab.cc:151 */
    obuf o = {0, 0, 0};

    for (off_t off = 0; off < st->st_size; off += (off_t)g->chunk_size - overlap) {
        if (st->st_size - off < (off_t)g->chunk_size)
            clen = (size_t)(st->st_size - off);
        else
            clen = g->chunk_size;

        char *content = (char *)mmap(NULL, clen, PROT_READ, MAP_PRIVATE | MAP_NORESERVE, fd, off);
        if (content == MAP_FAILED) {
            snprintf(g->err, sizeof g->err, "FileGrep::find::mmap: %s", strerror(errno));
            close(fd);
            free(o.p);
            return -1;
        }
        if (clen > 4 * 0x1000 && !g->single) /* grab.cc:168-169 */
            posix_madvise(content, clen, POSIX_MADV_SEQUENTIAL);

        og_chunk(g, path, content, clen, (long long)off, &o);

        munmap(content, clen);

        if (o.n > 0) { /* grab.cc:217-234 */
            fwrite(o.p, 1, o.n, stdout);
            o.n = 0;
            if (g->single) break;
        }
    }
    free(o.p);
    close(fd);
    return 0;
}

/* grab.cc:242-257 */
static int og_find_path(ogrep *g, const char *path)
{
    struct stat st;
    if (stat(path, &st) < 0) {
        snprintf(g->err, sizeof g->err, "FileGrep::find::stat: %s", strerror(errno));
        return -1;
    }
    if (S_ISREG(st.st_mode))
        return og_find_file(g, path, &st);
    if (S_ISDIR(st.st_mode))
        fputs("Clever boy! Want recursion? Add -R!\n", stderr);
    return 0;
}

static ogrep *g_walk; /* stands in for the global `FileGrep *grep`, main.cc:62 */

/* grab.cc:263-272 */
static int og_walk(const char *path, const struct stat *st, int typeflag, struct FTW *ftw)
{
    (void)ftw;
    if (typeflag == FTW_F && S_ISREG(st->st_mode)) {
        if (og_find_file(g_walk, path, st) < 0) fprintf(stderr, "%s: %s\n", path, g_walk->err);
    }
    return 0;
}

#endif /* !ORACLE_NO_MAIN */

/* ------------------------------ library API ------------------------------ */

int oracle_minlen(const char *regex, int *minlen)
{
    ogrep g;
    og_init(&g);
    if (og_prepare(&g, regex) < 0) return -1;
    *minlen = g.minlen;
    return 0;
}

/*
 * Run the inner loop over one in-memory chunk.  flags: bit0 print_offset (-O),
 * bit1 noline (-l), bit2 single (-s), bit3 path prefix, bit4 colour.
 * Returns a malloc'd buffer (caller frees with oracle_free) and its length.
 */
int oracle_scan_chunk(const char *regex, const char *path, const char *content, size_t clen,
                      long long off, unsigned flags, char **out, size_t *outlen)
{
    ogrep g;
    og_init(&g);
    if (og_prepare(&g, regex) < 0) return -1;
    g.print_offset = !!(flags & 1);
    g.print_line = !(flags & 2);
    g.single = !!(flags & 4);
    g.print_path = !!(flags & 8);
    g.colored = !!(flags & 16);
    obuf o = {0, 0, 0};
    if (!((size_t)g.minlen > clen)) og_chunk(&g, path ? path : "", content, clen, off, &o);
    *out = o.p;
    *outlen = o.n;
    return 0;
}

/*
 * Every p in [0,len) such that the pattern matches AT p when the subject is taken
 * to start at p (grab.cc:178 passes subject=start, startoffset=0: quirk Q4), with
 * the match end.  This is the definition of the candidate superset the GPU engine
 * emits.  O(len * match cost): small inputs only.
 */
long oracle_all_starts(const char *regex, const char *buf, size_t len, uint32_t *starts,
                       uint32_t *ends, size_t cap)
{
    ogrep g;
    og_init(&g);
    if (og_prepare(&g, regex) < 0) return -1;
    long n = 0;
    int ov[3];
    for (size_t p = 0; p < len; p++) {
        int rc = pcre_exec(g.h, g.x, buf + p, (int)(len - p), 0, PCRE_ANCHORED, ov, 3);
        if (rc > 0) {
            if ((size_t)n < cap) {
                starts[n] = (uint32_t)p;
                if (ends) ends[n] = (uint32_t)(p + (size_t)ov[1]);
            }
            n++;
        }
    }
    return n;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------ CLI (main.cc:103-266) ------------------------------ */
#ifndef ORACLE_NO_MAIN
static void usage(const char *p)
{
    printf("Usage: %s [-rR] [-I] [-O] [-L] [-l] [-s] [-n <cores>] <regex> <path>\n", p);
    exit(1);
}

int main(int argc, char **argv)
{
    int c, cores = 0;
    size_t chunk_size = 1u << 30;
    ogrep g;
    og_init(&g);
    int recursive = 0;

    while ((c = getopt(argc, argv, "Rrn:IOlsL")) != -1) { /* main.cc:116 */
        switch (c) {
        case 'r':
        case 'R': recursive = 1; break;
        case 's': g.single = 1; break;
        case 'O': g.print_offset = 1; break;
        case 'l': g.print_line = 0; break;
        case 'L': /* main.cc:131-136 */
            g.low_mem = 1;
            chunk_size >>= 1;
            if (chunk_size < (1u << 25)) chunk_size = 1u << 25;
            break;
        case 'I':
            if (isatty(1)) g.colored = 1;
            break;
        case 'n': cores = atoi(optarg); break;
        default: usage(argv[0]);
        }
    }
    if (argc < optind + 2) usage(argv[0]);
    const char *regex = argv[optind++];
    const char *path = argv[optind++];

    if (cores > 1) { /* main.cc:163-173 : the oracle keeps -n's geometry, runs the files serially */
        if (!recursive) {
            fputs("Multicore support only for recursive grabs.\n", stderr);
            return 255;
        }
        chunk_size >>= 2;
    }
    g.chunk_size = chunk_size;

    if (og_prepare(&g, regex) < 0) {
        if (cores > 1) return 0; /* main.cc:198 ignores prepare's result in -n mode */
        fprintf(stderr, "%s\n", g.err);
        return 255;
    }

    if (recursive) {
        g.recursive = 1;
        g_walk = &g;
        if (nftw(path, og_walk, 1024, FTW_PHYS) < 0 && cores <= 1) { /* grab.cc:278 */
            fprintf(stderr, "%s\n", g.err);
            return 255;
        }
    } else {
        if (argc - optind > 0) g.print_path = 1; /* main.cc:249-250 */
        for (;;) {
            if (og_find_path(&g, path) < 0) {
                fprintf(stderr, "%s\n", g.err);
                return 255;
            }
            if (argc > optind)
                path = argv[optind++];
            else
                break;
        }
    }
    fflush(stdout);
    if (g_resource_errors && getenv("GRAB_DIAG")) /* tests: skip comparisons on inputs where libpcre gave up */
        fprintf(stderr, "oracle: pcre_exec gave up %ld times (match limit / JIT stack)\n", g_resource_errors);
    return 0;
}
#endif
