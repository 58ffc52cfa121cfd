/*
 * gscan.h -- C ABI of the MI355X (gfx950) scan engine that replaces the regex
 * engine calls inside stealth/grab's per-file match loop.
 *
 * The reference has no plugin/FFI layer; its seam is the four libpcre calls made by
 * FileGrep (citations relative to /root/reference):
 *
 *   pcre_compile + pcre_study + pcre_fullinfo(PCRE_INFO_MINLENGTH)   src/grab.cc:106,115,120
 *       -> gscan_compile()      (pattern -> database, minlen with PCRE's meaning)
 *   pcre_exec(h, extra, start, end-start, 0, 0, ovector, 3)          src/grab.cc:178
 *       -> gscan_submit()/gscan_wait()   (one call per CHUNK instead of one per match:
 *          the kernels scan for every offset p of the chunk at which a match may start --
 *          the "candidates": exactly the offsets at which pcre_exec would report a match if
 *          asked to start there when gscan_info.exact is set, a superset of them (what a
 *          match must begin with) otherwise -- and return the start of every group of
 *          consecutive candidates)
 *       -> gscan_next_match()   (one pcre_exec call of the reference's loop: the leftmost
 *          match at or after the restart position, found by walking that list and asking the
 *          host matcher AT the offsets it yields; src/grab.cc:175-213 stays as it is around it)
 *       -> gscan_match_at() / gscan_match_info() / gscan_match_end()   (the matcher's answer
 *          for one offset: is it a match start, where does the match end (ovector[1]), did
 *          its path set a capturing group)
 *   pcre_free_study                                                   src/grab.cc:79
 *       -> gscan_free()
 *
 * Plain C: pointers and sizes only, no exceptions, no global state.  A gscan_db is
 * immutable and may be shared between threads; a gscan_ctx belongs to one thread
 * (the reference builds one FileGrep per pthread, src/main.cc:195-199).
 *
 * All functions return 0 on success, GSCAN_UNSUPPORTED (1) from gscan_compile when
 * the pattern is valid but outside the GPU engine's subset, and a negative
 * GSCAN_E* code on error.  There is NO CPU scanning path behind this interface: if
 * no HIP device can be opened, gscan_open fails.
 */
#ifndef GSCAN_H
#define GSCAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSCAN_ABI_VERSION 2

/* libgscan.so is built with -fvisibility=hidden: it exports the functions declared here (and the test / diagnostic hooks of
 * include/gscan_test.h) and nothing else -- tests/test_abi.py compares the library's dynamic symbol table with the two headers */
#ifndef GSCAN_API
#define GSCAN_API __attribute__((visibility("default")))
#endif

/* return codes */
#define GSCAN_OK 0
#define GSCAN_UNSUPPORTED 1 /* gscan_compile: valid pattern, not in the engine's subset */
#define GSCAN_EINVAL (-1)   /* bad argument / malformed pattern */
#define GSCAN_ENOMEM (-2)
#define GSCAN_EHIP (-3)     /* a HIP call failed; text via gscan_strerror(ctx) */
#define GSCAN_EBUSY (-4)    /* all in-flight slots used: call gscan_wait first */
#define GSCAN_EEMPTY (-5)   /* gscan_wait with nothing in flight */
#define GSCAN_ETOOBIG (-6)  /* chunk larger than the context was opened for */
#define GSCAN_EIO (-7)      /* gscan_submit_fd: reading the file failed; text via gscan_strerror(ctx) */

/* gscan_compile flags */
#define GSCAN_LITERAL 1u /* treat the pattern as a literal byte string (grab's documented -S) */
#define GSCAN_PCRE_CHECKED 2u /* pcre_compile has accepted this text (FileGrep::prepare asks it first): the one diagnostic the compiler
                                 mirrors by analysis rather than by syntax -- error 40, "recursive call could loop indefinitely" -- is
                                 not made, so that no corner of libpcre's rule can turn a pattern it accepts into a refusal */

/* engine tiers (gscan_info.tier) */
#define GSCAN_TIER_NULL 0    /* pattern can match the empty string: PCRE minlen -1, grab skips every file (Q2) */
#define GSCAN_TIER_LITERAL 1 /* K1: 4-byte anchor compare + class-sequence verify */
#define GSCAN_TIER_CLASSRUN 2 /* K2: LDS class table -> per-class bitmaps -> run detection */
#define GSCAN_TIER_BUCKET 3  /* K3: several alternatives (or > 4 classes): LDS bucket filter on 4 window positions + verify */
#define GSCAN_TIER_ANCHORED 4 /* ^foo, foo$ ...: a match can only sit at the restart position or at the chunk end; nothing to scan,
                                 the host's window tests (gscan_match_info, gscan_tail_positions) find everything */

typedef struct gscan_db gscan_db;
typedef struct gscan_ctx gscan_ctx;

typedef struct gscan_info {
    int tier;            /* GSCAN_TIER_* */
    int minlen;          /* == PCRE_INFO_MINLENGTH for the pattern (window length), -1 for TIER_NULL */
    int n_classes;       /* distinct byte classes in the window */
    int has_tail;        /* 1 if the last atom is a greedy variable repeat */
    uint32_t tail_extra; /* max bytes the tail may take beyond the window (UINT32_MAX = unbounded) */
    int anchor_off;      /* K1: offset of the anchor inside the window */
    int anchor_len;      /* K1: 1..4 bytes */
    int is_literal;      /* 1 if every window position is a single byte value */
    int n_alts;          /* alternatives the pattern unfolds into (priority order); the fields above describe alternative 0 */
    int has_context;     /* bit 0: some alternative looks at the byte BEFORE its match (\b ^ ...), bit 1: at the byte AFTER it */
    int lines_ok;        /* 1 if the line-extent pass applies (gscan_set_option "line_extents"): one plain alternative, no newline in its classes */
    int exact;           /* 1: the alternatives ARE the pattern.  0: they are what every match must begin with (the pattern goes on
                            with a second unbounded repeat, a repeated group ...); the host confirms each offset with its
                            backtracking matcher (gscan_next_match does) */
    int vm;              /* 1 (exact == 0 only): the DEVICE confirms the candidates itself -- the scan kernel runs the pattern as a
                            small backtracking VM at every filter hit and drops the hits at which no match can start -- and the
                            list it returns holds EVERY hit it kept (no group-start compression): gscan_next_match then asks the
                            host matcher at the listed offsets only */
    int gapped;          /* alternatives with one unbounded repeat in the middle (a+b, foo.*bar): the kernels list where the part BEHIND
                            the repeat begins (one repeat byte + the rest), gscan_next_match walks the run back to the match start */
    int textfree;        /* 1: one plain alternative of fixed length whose window cannot match at two ADJACENT offsets (two neighbouring
                            positions of it have no byte in common -- any literal that is not one repeated byte): every candidate is
                            listed and ends where its window ends, so gscan_next_match never looks at `content` (it may be NULL) */
    int ends_ok;         /* 1 if the match-end pass applies (gscan_set_option "match_ends"): one plain alternative without context that
                            ends in an unbounded greedy repeat whose class contains the window's first class (gscan_next_listed) */
    int resolve;         /* 1: the DEVICE settles the matches (round 6).  The kernels scan for what a match must BEGIN with -- START windows, one
                            per alternative -- and list every offset where one fits; a per-record pass (k_resolve) runs the pattern's VM
                            program there with the chunk's real bytes in front of the offset and keeps the offsets at which a match
                            starts, each with its end: gscan_wait's list is the list of MATCH starts, gscan_last_ends their ends, and
                            gscan_next_resolved walks it without asking the host matcher (but for the `reach` offsets behind s) */
    int reach;           /* resolve: how far in front of a match start the pattern looks (\b ^: 1, a look-behind: its length, else 0).  The
                            device's verdict at p is pcre_exec's for every restart position s <= p - reach */
    int n_windows;       /* device windows the kernels scan for (gscan_db_dev_window: 0 .. n_windows - 1); == n_alts unless resolve is set */
} gscan_info;

/* one scan unit inside a device-resident arena (gscan_scan_device) */
typedef struct gscan_seg {
    uint64_t offset; /* byte offset of the segment from dev_base; must be 16-byte aligned */
    uint32_t len;    /* bytes; <= 2^30 like a grab chunk (src/grab.h:48) */
    uint32_t _pad;
} gscan_seg;

/* one small file of a batch handed over by NAME (gscan_submit_files) */
typedef struct gscan_file {
    const char *path; /* opened by the device's reader threads with `oflags` (O_RDONLY | O_NOATIME ...) and closed after the
                         read; the string is copied at submit.  NULL: read from `fd` instead */
    int fd;           /* path == NULL: an open descriptor, the caller's; it must stay open until gscan_wait_segs has
                         returned the batch */
    int oflags;
    uint32_t len;     /* bytes to read from offset 0 (st_size as the tree walk saw it); at most gscan_block_size() */
} gscan_file;

/* result of a device-resident scan: everything stays in HBM */
typedef struct gscan_dev_result {
    const uint32_t *recs;  /* device: candidate group starts (see gscan_wait), segment-relative; runs of ascending offsets */
    const uint64_t *desc;  /* device: per tile {count:u32 | base:u32<<32}, base = index into recs;
                              tiles are in (segment, text) order, so walking desc yields ascending offsets.
                              A "tile" here is what ONE WAVE scans (8-16 KiB): every wave reserves and describes its own run */
    uint64_t n_tiles;
    uint32_t tile_bytes;
    uint64_t total;        /* number of records the scan produced (valid after gscan_dev_sync) */
    int overflow;          /* 1 if the record buffer was too small: total says how many are needed */
    const uint32_t *ends;  /* device, parallel to recs: the resolve pass's ends (databases with gscan_info.resolve: recs are match starts); else NULL */
} gscan_dev_result;

/* ---- pattern database (host only; no device needed) ---- */
GSCAN_API int gscan_compile(const char *pat, size_t len, unsigned flags, gscan_db **out, int *minlen,
                  char *err, size_t errcap);
GSCAN_API void gscan_free(gscan_db *db);
GSCAN_API int gscan_db_info(const gscan_db *db, gscan_info *info);
/* pcre_exec's verdict on a match attempt AT p when the subject starts at subject_start <= p (the position the
 * reference restarted at: src/grab.cc:178 passes subject = start, so ^ \b \B see nothing before it):
 * 0 no match starts at p;  1 match, *end = ovector[1];  2 match whose path closes a capturing group -- with the
 * reference's int ovector[3] (src/grab.cc:171) pcre_exec returns 0 for it and the chunk loop ends (src/grab.cc:179). */
GSCAN_API int gscan_match_info(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p,
                     uint32_t *end);
/*
 * The reference's inner loop, one step:  rc = pcre_exec(h, extra, start, end - start, 0, 0, ovector, 3)
 * (src/grab.cc:178) -- the leftmost match in content[s..clen) when the subject STARTS at s.  `starts` is
 * the list gscan_wait returned for this chunk; `cur` is zero-initialised by the caller once per chunk and
 * handed back on every call (s may only grow from call to call, as it does in the reference's loop).
 * Returns 0: no match (rc < 0);  1: match, [*m0, *m1) = ovector[0..1] as chunk offsets;  2: a match whose
 * path closes a capturing group -- with the reference's int ovector[3] (src/grab.cc:171) pcre_exec returns 0
 * for it and the chunk loop ends (src/grab.cc:179).
 * This is the whole rule, including what is not a function of a byte window: the restart position (nothing
 * before it), group members the kernels do not list, windows that end with the chunk.
 */
#define GSCAN_MAX_TAILS 132
#define GSCAN_MAX_ALTS 64
typedef struct gscan_cursor {
    size_t li;                      /* first list entry > the last s */
    uint32_t ntails, ready;         /* ready: set to 0 by the caller before the first call for a chunk; the rest is the callee's */
    uint32_t tails[GSCAN_MAX_TAILS]; /* offsets whose window ends with the chunk: never listed by the kernels */
    /* the next offset at which each kind of alternative can start a match (slot 0: the plain ones together, 1 + i: gapped
     * alternative i), remembered from call to call: s only grows, so an answer beyond s stays the answer */
    uint32_t next_at[GSCAN_MAX_ALTS + 1];
    uint8_t next_known[GSCAN_MAX_ALTS + 1];
} gscan_cursor;
GSCAN_API int gscan_next_match(const gscan_db *db, const void *content, size_t clen, const uint32_t *starts, size_t n,
                     gscan_cursor *cur, uint32_t s, uint32_t *m0, uint32_t *m1);
/* Match attempts this process abandoned at the matcher's resource limits (its counterpart of PCRE_ERROR_MATCHLIMIT /
 * PCRE_ERROR_JIT_STACKLIMIT): gscan_next_match then answers 0, which ends the chunk exactly as every pcre_exec error
 * does in the reference (rc <= 0: break, src/grab.cc:179).  WHERE an engine gives up is its own business -- libpcre's
 * interpreter, its JIT and this matcher all differ -- so differential tests skip inputs on which either side did.
 * Limit: 2^28 matcher steps per attempt (GSCAN_MATCH_LIMIT in the environment overrides it), 12000 nested group iterations (an attempt that outgrows 192 KiB of the caller's stack is repeated on a
 * 16 MiB stack of the matcher's own). */
GSCAN_API uint64_t gscan_resource_errors(void);

/* ---- device context: one per worker thread ---- */
GSCAN_API int gscan_open(int hip_device, size_t max_chunk, gscan_ctx **out);
GSCAN_API void gscan_close(gscan_ctx *ctx);
GSCAN_API const char *gscan_strerror(const gscan_ctx *ctx);
GSCAN_API int gscan_device_count(void);
/*
 * Where a device's host-side work belongs.  The reader threads of a device (gscan_submit_fd) run on the CPUs of the
 * device's NUMA node -- sysfs <pci root>/<bus id>/local_cpulist, e.g. "0-31,64-95" -- so that the pinned staging blocks
 * and the page cache -> pinned copy stay local to the PCIe root the GPU hangs off; `grab -n` puts worker i on the CPUs
 * of device (i mod #devices) the same way (the reference pins thread i to CPU i, src/main.cc:200-215).
 *   gscan_device_cpulist  the list for a HIP device (bus id from the runtime; root /sys/bus/pci/devices, or
 *                         $GSCAN_SYSFS_PCI); returns its length, <0 if unknown
 *   (gscan_pci_cpulist, include/gscan_test.h: the same for an explicit sysfs root + bus id -- no device needed)
 *   gscan_parse_cpulist   "0-3,8,10-11" -> CPU numbers; returns how many there are, fills at most cap
 */
GSCAN_API int gscan_device_cpulist(int hip_device, char *buf, size_t cap);
GSCAN_API long gscan_parse_cpulist(const char *list, int *cpus, size_t cap);

/*
 * Host-chunk path (what FileGrep::find uses): three ways to hand over the bytes that the
 * reference mmap()s and gives to pcre_exec (src/grab.cc:161,178).  Each starts H2D + scan of one
 * chunk (on the DEVICE's copy streams, shared by its contexts: the scan rides on the first, behind
 * the chunk's copies); up to GSCAN_SLOTS chunks may be in flight and
 * gscan_wait[_segs] returns them in submission order.  A gscan_wait that fails (a read error, a
 * device error) has dropped that chunk and freed its slot: the context stays usable.
 *
 *   gscan_submit_fd    a range of an open file.  The device's reader threads (GSCAN_READERS)
 *                      pread(2) it in pieces of gscan_block_size() bytes into a small pool of pinned
 *                      blocks and DMA every piece as soon as it is read, spread over the device's
 *                      copy streams; the reader that finishes the last piece launches the scan.
 *                      ASYNCHRONOUS: returns once the pieces are queued -- fd must stay open until
 *                      gscan_wait has returned the chunk; read errors surface there (GSCAN_EIO).
 *                      No host copy is kept: gscan_wait gives *content = NULL and the caller maps
 *                      the file itself if it has matches to print.
 *   gscan_submit_files MANY small files by name (the reference opens and maps every file on the
 *                      worker thread, src/main.cc:86-100 -> src/grab.cc:137-169): the caller only queues them.  The
 *                      device's reader threads open, read and close them -- runs of consecutive files that fit one
 *                      pinned block are read by one reader and leave in ONE DMA --, segment i is file i, packed at
 *                      16-byte aligned offsets in the order given; the reader that finishes the last piece launches
 *                      ONE scan over the segment table.  ASYNCHRONOUS like gscan_submit_fd, *content = NULL.  A
 *                      file that cannot be opened, or is shorter than `len`, does not fail the batch: its segment is
 *                      scanned zero-filled and gscan_last_file_errors says which -- the caller must ignore that
 *                      segment's records.  The batch as a whole (16-byte padding included) must fit max_chunk.
 *   gscan_acquire +    the caller fills the slot's pinned buffer (gscan_block_size() bytes come
 *   gscan_submit[_segs] from the pinned pool; more is allocated for the slot): one chunk, or MANY
 *                      small files packed at 16-byte aligned offsets and described by a segment
 *                      table -- one H2D and one launch for the lot (segments behave exactly like
 *                      separate chunks).
 *   gscan_submit       any other caller buffer: >= 1 MiB is registered with the runtime and DMA'd
 *                      in place, less is staged through the slot's pinned buffer.  The buffer must
 *                      stay valid and unchanged until gscan_wait has returned the chunk.
 */
#define GSCAN_SLOTS 3
GSCAN_API int gscan_acquire(gscan_ctx *ctx, size_t len, void **pinned);
GSCAN_API size_t gscan_block_size(void);
/* Optional, before the first other call of a process that is about to scan (the command line calls it first thing in main):
 * helper threads map and touch `blocks` staging blocks' worth of memory WHILE the HIP runtime starts; the reader pool then
 * only registers them with the runtime instead of allocating pinned memory block by block while the pipe fills (1.5 - 2 ms
 * each, one at a time).  No HIP call is made here.  GSCAN_PREFAULT=0 in the environment turns it into a no-op. */
GSCAN_API int gscan_prefault(size_t blocks);
/* The same, and the helper threads FILL the first blocks with the first bytes of the files named (in the order given, piece by
 * piece of gscan_block_size() bytes, `blocks` pieces at most): what the reader threads would pread once the runtime is up is
 * read while it starts (hipInit takes 50 ms; 256 MiB are read in 10).  A gscan_submit_fd whose range holds such a piece --
 * same file (device + inode), same offset, same length -- takes the block as it is: registered, DMA'd, nothing is read twice.
 * For callers that know their input before the first HIP call (the command line with explicit paths: BASELINE configs[0]). */
GSCAN_API int gscan_prefault_files(size_t blocks, const char *const *paths, size_t npaths);
/* the ingest configuration in force (environment: GSCAN_BLOCK_MIB, GSCAN_READERS, GSCAN_COPY_STREAMS = the copy streams of a
 * DEVICE, shared by its contexts); any pointer may be NULL */
GSCAN_API void gscan_ingest_info(size_t *block_bytes, int *readers, int *copy_streams);
GSCAN_API int gscan_submit(gscan_ctx *ctx, const gscan_db *db, const void *host_bytes, size_t len,
                 uint64_t tag);
GSCAN_API int gscan_submit_segs(gscan_ctx *ctx, const gscan_db *db, const void *pinned, const gscan_seg *segs,
                      size_t nseg, uint64_t tag);
GSCAN_API int gscan_submit_fd(gscan_ctx *ctx, const gscan_db *db, int fd, long long file_off, size_t len,
                    uint64_t tag);
GSCAN_API int gscan_submit_files(gscan_ctx *ctx, const gscan_db *db, const gscan_file *files, size_t n, uint64_t tag);
/* per-segment status of the gscan_submit_files batch the last gscan_wait_segs handed out: 0 the file was read in full, an
 * errno from open(2) / pread(2), -1 the file was shorter than `len`.  *n = number of segments; NULL (and *n = 0) if that
 * chunk was not such a batch.  Same lifetime as its starts. */
GSCAN_API const int *gscan_last_file_errors(const gscan_ctx *ctx, size_t *n);
/* starts[0..n), ascending: the START of every group of consecutive candidate offsets of the
 * chunk (offsets p at which the pattern matches), possibly with further candidates of the
 * same groups in between.  That is all pcre_exec's "leftmost match at or after s" needs:
 *     gscan_match_at(db, content, clen, s) ? s : first starts[i] > s
 * (if s is a candidate it is the answer; if not, the next candidate after s begins a group).
 * *content is the chunk's bytes on the host: the buffer the chunk was submitted from, NULL after
 * gscan_submit_fd / gscan_submit_files.  starts (and gscan_last_ext / _ends / _gather / _file_errors) stay valid until the
 * next gscan_wait*, gscan_submit* or gscan_acquire on this context, whichever comes first: they live in the pinned buffers of
 * the slot the chunk ran in (a dense list is handed out exactly as the device put it together: in text order, one linear
 * transfer, no merge on the host), and a new chunk may take that slot (it does so only when no other slot is free).
 * Consume a chunk's result before handing over the next one -- FileGrep does.  A pinned *content stays until the second
 * gscan_acquire / gscan_submit* after this call reuses the slot. */
GSCAN_API int gscan_wait(gscan_ctx *ctx, uint64_t *tag, const uint32_t **starts, size_t *n,
               const void **content);
/*
 * Line extents, orbit selection and line gather on the device, for the reference's line-printing modes (src/grab.cc:188-209:
 * print the line around the match, restart at the end of that line).  With gscan_set_option(ctx, "line_extents", 1) and a
 * pattern whose gscan_info.lines_ok is set, every chunk also carries ext[4*i .. 4*i+3] = {m1, lb, le, goff} for starts[i]:
 *   m1 == 0            starts[i] is not printed (an earlier candidate sits in the same line);
 *   lb == 0xffffffff   ask the host: from here on the reference's loop itself is needed (gscan_next_match) -- a line
 *                      that runs on past the 511 bytes of printed context, a line start or tail too far away;
 *   else               print [lb, starts[i]) + the match [starts[i], m1) + [m1, le) + '\n'; the loop restarts at le.
 *                      goff != 0xffffffff: the bytes [lb, le) of the chunk are at gscan_last_gather()[goff ..] -- the device
 *                      copied the text of every printed line into one buffer, so the caller never reads the chunk itself
 *                      (for a file range that was never mapped: no page is faulted in); 0xffffffff: take them from the chunk.
 * gscan_last_ext returns the array for the chunk the last gscan_wait / gscan_wait_segs call handed out (parallel to its
 * starts, same lifetime), or NULL if that chunk has none; gscan_last_gather the gathered text (NULL: none was fetched --
 * treat every goff as 0xffffffff), valid until the next gscan_wait / gscan_wait_segs on this context.
 */
GSCAN_API const uint32_t *gscan_last_ext(const gscan_ctx *ctx);
GSCAN_API const uint8_t *gscan_last_gather(const gscan_ctx *ctx, size_t *bytes);
/*
 * Match ends on the device, for -O -l (src/grab.cc:175-213 with d_print_line off: print the offset, restart at the match
 * end).  With gscan_set_option(ctx, "match_ends", 1) and a pattern whose gscan_info.ends_ok is set, every chunk also
 * carries ends[i] = ovector[1] of the match that starts at starts[i] (window + greedy tail, cut at the chunk end), or 0
 * where the device left it to the host (a tail that runs on for more than 4 KiB).  gscan_last_ends returns the array for
 * the chunk the last gscan_wait / gscan_wait_segs handed out (parallel to its starts, same lifetime), or NULL.
 */
GSCAN_API const uint32_t *gscan_last_ends(const gscan_ctx *ctx);
/*
 * gscan_next_match for such a chunk, WITHOUT the text: valid when s is 0 or the end of the previous match of this walk
 * (the byte that stopped its tail cannot begin a match, so the leftmost match from s is the first listed start >= s).
 * Returns 0 no match, 1 match with [*m0, *m1), -1: this match's end is the host's to find -- make this step with
 * gscan_next_match (same cursor).  `cur` as for gscan_next_match.
 */
GSCAN_API int gscan_next_listed(const gscan_db *db, size_t clen, const uint32_t *starts, const uint32_t *ends, size_t n,
                      gscan_cursor *cur, uint32_t s, uint32_t *m0, uint32_t *m1);
/*
 * The loop step for a database with gscan_info.resolve set:  rc = pcre_exec(h, extra, start, end - start, 0, 0, ovector, 3)
 * (src/grab.cc:178) answered from the device's list.  starts / ends as gscan_wait / gscan_last_ends returned them (ends may be
 * NULL: every record is then put to the host matcher, which is what the list means before k_resolve has seen it):
 *   ends[i] == GSCAN_END_ASK       the device's VM gave up at starts[i] (step / stack limit): the host matcher decides;
 *   ends[i] == GSCAN_END_CAPTURES  a match starts there whose path closes a capturing group: pcre_exec returns 0 with the
 *                                  reference's int ovector[3] (src/grab.cc:171) and the chunk loop ends (src/grab.cc:179);
 *   else                           a match [starts[i], ends[i] & ~GSCAN_END_LOOK).
 * The verdicts were reached with the chunk's real bytes in front of starts[i]; pcre_exec sees nothing in front of the restart
 * position s, so the `reach` offsets from s on are the host matcher's (content is read there, and only there
 * and at GSCAN_END_ASK records: with reach == 0 and no such record it may be NULL).  Returns as gscan_next_match; -1 if the
 * database is not of this kind.
 */
#define GSCAN_END_ASK 0u
#define GSCAN_END_CAPTURES 0xfffffffeu
/* bit 31 of a match end (ends are chunk offsets, <= 2^30 + 4096): behind THIS match's end the host has to look at the text before
 * it trusts the list again -- the pattern looks back (reach > 0), a match could begin within `reach` bytes of the end, and the
 * byte in front of the end is not one the pattern takes for the subject start.  Clear: a walk that restarts at this end
 * (-O -l: src/grab.cc:209 with a == 0) goes straight to the next record, without reading a byte of the chunk.  Set by the device
 * (k_resolve looks at the two bytes around the end); gscan_next_resolved does not need it (it looks itself) and masks it. */
#define GSCAN_END_LOOK 0x80000000u
GSCAN_API int gscan_next_resolved(const gscan_db *db, const void *content, size_t clen, const uint32_t *starts, const uint32_t *ends, size_t n,
                        gscan_cursor *cur, uint32_t s, uint32_t *m0, uint32_t *m1);
/* the bytes a match can begin with: table[b] & 1 if b can; returns 1 if that is known (0: the pattern may begin without
 * consuming a byte -- an assertion, an optional item -- and bit 0 of every entry is set).  table[b] & 2 (reach == 1 only): b in
 * front of an offset is to the pattern what the subject start is -- a non-word byte for \b \B, a newline for (?m)^, a byte
 * outside a one-byte look-behind's class -- so with such a byte in front of the restart position the device's verdict AT the
 * restart position is pcre_exec's too and gscan_next_resolved asks nobody. */
GSCAN_API int gscan_db_first(const gscan_db *db, uint8_t table[256]);
/* the same for a chunk of several segments: the records of segment i are
 * starts[seg_first[i] .. seg_first[i+1]), segment-relative; seg_first has *nseg + 1 entries
 * (a single-segment chunk reports *nseg = 1). */
GSCAN_API int gscan_wait_segs(gscan_ctx *ctx, uint64_t *tag, const uint32_t **starts, const size_t **seg_first,
                    size_t *nseg, const void **content);

/*
 * Device-resident path (bench, batching): scan nseg segments of an arena that is
 * already in HBM, one launch, on `stream` (a hipStream_t, NULL = the stream the context's
 * scans ride on).  Asynchronous; results stay on the device.
 */
GSCAN_API int gscan_scan_device(gscan_ctx *ctx, const gscan_db *db, const void *dev_base,
                      const gscan_seg *segs, size_t nseg, void *stream,
                      gscan_dev_result *res);
/* wait for the scan, fill res->total / res->overflow */
GSCAN_API int gscan_dev_sync(gscan_ctx *ctx, gscan_dev_result *res);
/* copy the records of segment `seg` to the host, ascending; returns count or <0 */
GSCAN_API long gscan_dev_fetch(gscan_ctx *ctx, const gscan_dev_result *res, size_t seg, uint32_t *out,
                     size_t cap);
/* record-buffer capacity (records, split into 8 equal shard regions) for device scans;
 * default = arena bytes / 16 */
GSCAN_API int gscan_set_capacity(gscan_ctx *ctx, size_t n_records);
/* tuning knobs for A/B runs: name in {"variant","blocks_per_cu","register_min","line_extents","match_ends","k3_depth"}; see DESIGN.md
 * ("k3_depth": 0 = the compiler's choice, 3 / 4 = K3's filter looks at that many window positions -- the records are the same) */
GSCAN_API int gscan_set_option(gscan_ctx *ctx, const char *name, long value);
/* scan-kernel time of the gscan_scan_device launches since the last reset: HIP events recorded
 * around each launch on the launch's own stream.  Waits for the launches to finish. */
GSCAN_API int gscan_kernel_time(gscan_ctx *ctx, double *sum_ms, uint64_t *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* GSCAN_H */
