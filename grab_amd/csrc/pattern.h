// pattern.h -- PCRE-subset pattern compiler for the gfx950 scan engine.
//
// Replaces the reference's pcre_compile/pcre_study/pcre_fullinfo(MINLENGTH) step
// (/root/reference/src/grab.cc:101-123).  The pattern is parsed into a tree (Node) -- single-byte
// atoms (literal, '.', escape classes, [...]), repeats (greedy / lazy / possessive), alternation,
// groups (plain, capturing, named, atomic), assertions (^ $ \b \B \A \z \Z, look-ahead, look-behind),
// back references, \K, the inline options i s m x -- and the tree serves two readers:
//
// * matcher.cc walks it: a backtracking matcher with PCRE's semantics decides "the match pcre_exec
//   reports AT offset p" (its end, the reported start, whether its path set a capturing group: the
//   reference gives pcre_exec room for one offset pair only -- int ovector[3], src/grab.cc:171 -- so
//   such a match comes back as rc == 0 and ENDS the chunk).
// * the unfolder (pattern.cc) turns it into what the KERNELS look for: a short, priority-ordered list
//   of alternatives, each a fixed window of byte classes, optionally ending in ONE variable repeat of a
//   single class ("tail"), optionally with one unbounded repeat in the middle ("gapped"), and one byte
//   of context at either end for the assertions.  Optional and bounded repeats in the middle of the
//   pattern (colou?r, [ab]{1,3}c, (?:foo|bar)?baz) unfold into alternatives in the order PCRE's
//   backtracking tries them.  When that list IS the pattern (Database::exact) "a match starts at p" is a
//   function of the bytes at p; where the unfolder has to stop -- a second unbounded repeat, a repeated
//   group, a look-around, a back reference -- the list says what every match must BEGIN with, and the
//   matcher confirms each such offset.  Either way the split "GPU lists candidate starts, host walks the
//   restart orbit" is exact.
//
// What stays outside (recursion, conditionals, Unicode properties, a few constructs libpcre itself
// treats inconsistently) is reported as GSCAN_UNSUPPORTED with the reason.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "vm.h"

namespace gscan {

struct ByteSet {
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool test(unsigned b) const { return (w[(b & 255) >> 5] >> (b & 31)) & 1u; }
    void set(unsigned b) { w[(b & 255) >> 5] |= 1u << (b & 31); }
    void set_range(unsigned lo, unsigned hi)
    {
        for (unsigned b = lo; b <= hi; b++) set(b);
    }
    void merge(const ByteSet &o)
    {
        for (int i = 0; i < 8; i++) w[i] |= o.w[i];
    }
    void negate()
    {
        for (int i = 0; i < 8; i++) w[i] = ~w[i];
    }
    bool operator==(const ByteSet &o) const
    {
        for (int i = 0; i < 8; i++)
            if (w[i] != o.w[i]) return false;
        return true;
    }
    int count() const
    {
        int c = 0;
        for (int i = 0; i < 8; i++) c += __builtin_popcount(w[i]);
        return c;
    }
    int single() const // the byte value if the set has exactly one member, else -1
    {
        if (count() != 1) return -1;
        for (int i = 0; i < 8; i++)
            if (w[i]) return i * 32 + __builtin_ctz(w[i]);
        return -1;
    }
};

// Zero-width assertions, with the inline option (?m) already folded in.
enum { A_BOS = 1, // ^ without (?m), \A, \G: the subject start (the restart position: src/grab.cc:178 passes subject = start)
       A_MBOL,    // (?m)^: subject start, or just after a newline
       A_EOL,     // $ without (?m), \Z: the very end of the chunk, or just before a newline that is its last byte
       A_MEOL,    // (?m)$: the very end, or just before any newline
       A_EOS,     // \z: the very end only
       A_WB,      // \b
       A_NWB,     // \B
       A_KEEP };  // \K: always holds; the reported match starts here (ovector[0])

// Parse tree.  SET = one byte drawn from a class; REP repeats its single child (max == UINT32_MAX: unbounded).
// LOOK = (?=..) (?!..) (?<=..) (?<!..) around its single child; ATOMIC = (?>..); BACKREF = \1 \g{2} \k<name> (?P=name).
// COND = (?(condition)yes|no): kids = [the condition's assertion (cond == C_ASSERT only),] yes [, no].
// RECURSE = (?R) (?1) (?&name) (?P>name) \g<1>: the group's pattern (group 0: the whole pattern) matched as an atomic
// subroutine; what it captures is dropped when it returns (PCRE1's rule, unlike Perl's).
struct Node {
    enum Kind { SET, CAT, ALT, REP, ASSERT, LOOK, ATOMIC, BACKREF, COND, RECURSE } kind = SET;
    enum { C_NONE, C_GROUP, C_ASSERT, C_IN_RECURSION, C_IN_RECURSION_OF, C_DEFINE };
    int cond = C_NONE;                // COND: what is tested (C_GROUP / C_IN_RECURSION_OF: `group`, by name until resolved)
    bool behind = false, neg = false; // LOOK
    int group = 0;                    // capturing CAT: its number (1..);  BACKREF: the group referred to
    bool icase = false;               // BACKREF under (?i)
    std::string refname;              // BACKREF by name, until the parser has resolved it
    int acode = 0;             // ASSERT: one of the A_* codes
    ByteSet set;
    std::vector<Node> kids;
    uint32_t min = 1, max = 1; // REP; max == kInf: unbounded
    int mode = 0;              // REP: 0 greedy, 1 lazy, 2 possessive
    bool cap = false;          // the node is the body of a capturing group
    bool newline_seq = false;  // ATOMIC: this is \R or \X (pattern.cc: what may stand in front of it is restricted)
};

constexpr int kMaxWindow = 256; // window positions a database may hold
constexpr int kMaxClasses = 64; // distinct byte classes per database
constexpr int kK2MaxClasses = 4;
constexpr int kK2MaxWindow = 49; // 16 own positions + 48 bits of look-ahead
constexpr int kK2MaxRuns = 16; // run descriptors live in the lanes of one VGPR for the whole kernel
constexpr int kMaxAlts = 64;          // alternatives a pattern may unfold into
constexpr int kAltWindowBytes = 4096; // sum of their window lengths
constexpr int kK3Buckets = 8;         // K3: alternatives share 8 filter buckets
constexpr int kK3Depth = 4;           // K3: window positions the filter looks at
constexpr int kK3Confirm = 24;        // K3: window positions the LDS confirm tables cover
constexpr uint32_t kMaxMidRepeat = 16; // a bounded repeat {n,m} before the end unfolds if m-n <= this

// POD uploaded verbatim to the device; the kernels read it from global memory /
// kernel arguments.  Keep in sync with kernels.hip.
struct DevProgram {
    uint32_t m;           // window length == minlen
    uint32_t n_classes;
    uint32_t is_literal;  // every window position is one byte value -> window[] holds the bytes
    uint32_t anchor;      // K1: little-endian packed anchor bytes
    uint32_t anchor_mask; // K1: 0xff.. over anchor_len bytes
    uint32_t anchor_off;  // K1: offset of the anchor inside the window
    uint32_t anchor_len;  // K1: 1..4, 0 = no anchor (K1 unusable)
    uint32_t nruns;       // K2: 0 = K2 unusable
    uint8_t run_cls[kK2MaxRuns];
    uint8_t run_len[kK2MaxRuns];
    uint8_t run_off[kK2MaxRuns];
    uint32_t k2_table[256];              // K2: byte -> class bits at bit 0/8/16/24
    uint32_t cls_bits[kMaxClasses][8];   // 256-bit membership bitmap per class
    uint8_t window[kMaxWindow];          // alternative 0: class id per window position (or the literal byte)
    // all alternatives, in priority order (K3 verify path)
    uint32_t n_alts;
    uint32_t k3_off;                     // K3: window offset of the 4 filtered positions
    uint16_t alt_off[kMaxAlts];          // start of alternative i inside alt_window
    uint16_t alt_len[kMaxAlts];          // its window length
    uint8_t alt_bucket[kMaxAlts];        // its K3 bucket
    uint8_t alt_window[kAltWindowBytes]; // class ids
    uint32_t k3_table[256];              // K3: byte -> 4 x 8 bucket bits (byte k: buckets that accept it at position k3_off+k)
    uint8_t k3_pos[kK3Confirm][256];     // K3 confirm: [window position][byte] -> buckets that accept it there (all, past an alternative's end)
    uint8_t k3_blen[kK3Buckets];         // K3 confirm: window length of the bucket's alternative when k3_confirm_exact
    uint32_t k3_confirm_exact;           // every alternative has its own bucket and fits kK3Confirm positions: the tables ARE the pattern
    uint32_t report_shift;               // 1 when the device windows start one byte before the match (context position)
    // line-extent pass (k_lines): the greedy repeat at the end of alternative 0, and whether the pass applies at all
    uint32_t tail_bits[8];
    uint32_t tail_extra;                 // 0: no tail
    uint32_t lines_ok;                   // one plain alternative, no context, and no class of it contains a newline
    // match-end pass (k_ends): one plain alternative without context or capturing group that ends in an UNBOUNDED greedy repeat
    // whose class contains the window's first class.  The match at a listed start p then ends at the first byte from p + m on
    // that is outside the tail class -- and that byte cannot begin a match, so after a match the leftmost next one is the next
    // LISTED start: with the ends from the device the -O -l walk (grab.cc:175-213, a == 0) never looks at the text.
    uint32_t ends_ok;
    // Candidates confirmed on the device (vm.h): K3 runs the pattern's VM program at every filter hit and drops the hits at
    // which no match can start.  For a gapped alternative the hit is the LAST byte of its unbounded repeat (device window =
    // repeat byte + the rest): the possible starts are walked back along the run of repeat bytes.
    uint32_t resolve;                    // 1: the windows are START windows and every record goes through k_resolve (Database::resolve)
    // resolve: what the host would look at behind a match's end before it trusts the list again (gscan_next_resolved) -- the device
    // looks for it (resolve_code below): Database::reach, ::first / ::first_ok, ::start_like as bitmaps
    uint32_t reach, first_ok;
    uint32_t est_permille;               // the windows' expected hits per 1000 bytes of text (pattern.cc, class_prob): sizes the record buffers of a database whose every hit is a record
    uint32_t first_bits[8], start_like_bits[8];
    uint32_t vm_filter;                  // 1: on
    // bit b0 << 8 | b1: a match may begin with the bytes b0 b1 (matcher.cc, tree_prefix_viable: the host matcher run on
    // every two-byte prefix at compile time).  The hits the filter passes are put to this table first: most die here,
    // two loads instead of a VM run.  vm_pair_ok: the table is filled in.
    uint32_t vm_pair_ok;
    uint32_t vm_pair[2048];
    uint8_t alt_gap_cls[kMaxAlts];       // class id of a gapped alternative's repeat byte; 0xff: a plain alternative
    uint16_t alt_plen[kMaxAlts];         // gapped: length of the fixed part in front of the repeat
    VmProg vm;
};

// One alternative: a fixed class window + an optional variable repeat of one class at its end.
struct AltSeq {
    std::vector<uint8_t> window; // class id per position
    bool has_tail = false;
    uint32_t tail_extra = 0;     // max bytes beyond the window (UINT32_MAX = unbounded)
    ByteSet tail;
    bool captures = false;       // the path closes a capturing group: pcre_exec with the reference's ovector[3] returns 0 for such a match
    // Zero-width assertions at the two ends of the alternative (^ $ \b \B \A \z \Z, (?m)), reduced to one byte of
    // context each.  The reference restarts pcre_exec with the subject beginning AT the restart position
    // (src/grab.cc:178), so "before the match" is either a real byte or the subject start:
    ByteSet pre;                 // bytes that may precede the match ...
    bool pre_start = true;       // ... and whether the match may sit at the subject start (the restart position)
    ByteSet post;                // bytes that may follow the window ...
    bool post_end = true;        // ... whether the window may end exactly at the chunk end ...
    bool post_final_nl = false;  // ... and whether "\n as the very last byte of the chunk" may follow ($ without (?m), \Z)
    // One unbounded repeat in the middle of the path ("gapped"):  pwindow . gap{1,} . window [. tail].
    // pre then refers to the byte before pwindow (the match start), post/tail to what follows `window`.  The kernels look
    // for  gap . window  (one repeat byte + the rest); the host finds the start by walking the run of repeat bytes back
    // to pwindow (matcher.cc).
    bool gapped = false;
    std::vector<uint8_t> pwindow; // class ids
    ByteSet gap;
    int gap_mode = 0;             // 0 greedy, 1 lazy
    int gap_id = 0;               // consecutive alternatives with the same id share ONE instance of the repeat
    size_t min_len() const { return window.size() + (gapped ? pwindow.size() + 1 : 0); }
    bool has_pre() const;        // pre / pre_start restrict anything
    bool has_post() const;
    AltSeq()
    {
        pre.negate();
        post.negate();
    }
};
inline bool AltSeq::has_pre() const { return pre.count() != 256 || !pre_start; }
inline bool AltSeq::has_post() const { return post.count() != 256 || !post_end; }

struct Database {
    int tier = 0;
    int minlen = -1;             // shortest alternative == PCRE_INFO_MINLENGTH; -1 if "" can match
    std::vector<ByteSet> classes;
    std::vector<AltSeq> alts;    // priority order: the first one whose window matches at p is PCRE's match at p
    bool exact = true;           // false: some alternative stops in front of a construct the unfolder leaves alone (a second
                                 // unbounded repeat, a repeated group, ...).  The alternatives then only say where a match
                                 // MAY start; matcher.cc's backtracking matcher confirms every such offset
    std::shared_ptr<Node> tree;  // the parse tree: matcher.cc's backtracking matcher walks it (match end, capturing groups)
    int n_groups = 0;            // capturing groups in the pattern
    bool has_backref = false;    // the matcher has to remember what the groups captured
    // What the kernels scan: when some alternative looks at the byte before (after) its window, EVERY alternative's
    // device window gets a leading (trailing) context position -- its own condition, or "any byte".  A device hit at q
    // is reported as q + dev_pre.  Matches at the restart position and windows ending at the chunk end have no such
    // byte and are the host's to find (filegrep.cc).
    bool dev_pre = false, dev_post = false;
    std::vector<std::vector<uint8_t>> dev_windows; // class ids, per alternative
    DevProgram prog;
    uint64_t id = 0; // unique per compile; contexts key their device copy on it
    // One plain alternative whose window cannot match at two ADJACENT offsets (two neighbouring positions of it have no
    // byte in common): every candidate is then the start of its own group, i.e. listed, and "the leftmost match from s" is
    // the first listed start >= s -- the host's walk never has to look at the text (matcher.cc, gscan_next_match).
    bool solitary = false;
    std::vector<const Node *> group_nodes; // [g] = the capturing group g inside `tree` (what (?g) calls), [0] = the tree
    // The device settles the matches itself (k_resolve, kernels.hip): dev_windows are START windows -- what a match must begin
    // with, one set per alternative that can sit away from the subject start -- the kernels list every offset where one of
    // them fits (no group-start compression), and a per-record pass runs the pattern's VM program there with the chunk's real
    // bytes in front: what comes back is the list of MATCH starts with their ends.  That verdict is pcre_exec's for every
    // restart position s <= p - reach; the host (gscan_next_resolved) asks its own matcher about the `reach` offsets from s on.
    bool resolve = false;
    uint32_t reach = 0;     // how far in front of a match start the pattern can look (0: not at all)
    ByteSet first;          // the bytes a match can begin with ...
    bool first_ok = false;  // ... if that is known (the pattern cannot begin without consuming a byte)
    ByteSet start_like;     // reach == 1: a byte of this set in front of p is to the pattern what the subject start is (pattern.cc, start_like_bytes)
    bool vm_ok = false; // prog.vm holds the tree as a VM program (vm.h): gscan_vm_verdict works; prog.vm_filter says whether the device uses it
};

// matcher.cc
bool tree_prefix_viable(const Database &d, const uint8_t *bytes, size_t n);
// vm_compile.cc
bool vm_compile(const Node &root, int n_groups, bool has_backref, VmProg &out);
bool vm_independent_of_subject_start(const Node &root);

// Keep the device hit at q (the start of some alternative's device window)?  false only if NO match can start there: at q
// itself for the plain alternatives, anywhere along the run of repeat bytes that ends at q for a gapped one.  Shared by
// the K3 kernel and the host (gscan_vm_filter, tests).
// may a match start at x, going by its first two bytes? (true when there is no table, or no second byte)
GSCAN_HD inline bool vm_start_viable(const DevProgram *pg, const uint8_t *seg, uint32_t slen, uint32_t x)
{
    if (!pg->vm_pair_ok || x + 1 >= slen) return true;
    const uint32_t idx = (uint32_t)seg[x] << 8 | seg[x + 1];
    return (pg->vm_pair[idx >> 5] >> (idx & 31)) & 1u;
}
GSCAN_HD inline bool vm_keep_hit(const DevProgram *pg, const VmProg *vm, const uint8_t *seg, uint32_t slen, uint32_t q)
{
    if (vm_start_viable(pg, seg, slen, q) && vm_run(vm, seg, slen, q, 0) != 0) return true;
    const uint32_t n = pg->n_alts;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t gc = pg->alt_gap_cls[i];
        if (gc == 0xffu) continue;
        const uint32_t m = pg->alt_len[i];
        if (q + m > slen) continue;
        const uint8_t *w = pg->alt_window + pg->alt_off[i];
        uint32_t k = 0;
        for (; k < m; k++) {
            const uint32_t b = seg[q + k];
            if (!((pg->cls_bits[w[k]][b >> 5] >> (b & 31)) & 1u)) break;
        }
        if (k < m) continue;
        // the run of repeat bytes that ends at q reaches back to r0; a match of this alternative starts plen bytes in
        // front of a position of that run
        uint32_t r0 = q, walked = 0;
        while (r0 > 0) {
            const uint32_t b = seg[r0 - 1];
            if (!((pg->cls_bits[gc][b >> 5] >> (b & 31)) & 1u)) break;
            r0--;
            if (++walked > 255u) return true; // a long run: the host looks at it
        }
        const uint32_t plen = pg->alt_plen[i];
        for (uint32_t g = r0; g <= q; g++) {
            if (g < plen || g - plen == q) continue;
            if (vm_start_viable(pg, seg, slen, g - plen) && vm_run(vm, seg, slen, g - plen, 0) != 0) return true;
        }
    }
    return false;
}

// What k_resolve writes next to a record at which the VM (verdict v: 1 match, 2 gave up; out: its end / captured) did not say
// "no match": the match's end, GSCAN_END_CAPTURES, or GSCAN_END_ASK -- and, in bit 31 of an end (GSCAN_END_LOOK), whether the
// host has to LOOK at the text when it restarts behind this match: pcre_exec sees nothing in front of the restart position
// (src/grab.cc:178), so with a pattern that looks back (reach > 0) the list is good only from `reach` bytes further on and
// the offsets in between are the host matcher's -- unless no match can begin there anyway (the byte at the end cannot begin
// one) or the byte in front of the end is to the pattern what the subject start is (start_like).  Both are a look at two bytes
// the device has at hand; the host's -O -l walk then touches the text for a few matches in a thousand instead of for each.
// Shared by the kernel and its host mirror (gscan_vm_resolve).
GSCAN_HD inline uint32_t resolve_code(const DevProgram *pg, const uint8_t *seg, uint32_t slen, uint32_t p, int v, const VmOut &o)
{
    if (v != 1 || o.end <= p) return 0u;              // GSCAN_END_ASK
    if (o.cap) return 0xfffffffeu;                    // GSCAN_END_CAPTURES
    uint32_t look = 0;
    const uint32_t e = o.end;
    if (pg->reach > 1u) {
        look = 1;
    } else if (pg->reach == 1u && e < slen) {
        const uint32_t before = seg[e - 1], at = seg[e];
        const bool same = (pg->start_like_bits[before >> 5] >> (before & 31u)) & 1u;
        const bool can_begin = !pg->first_ok || ((pg->first_bits[at >> 5] >> (at & 31u)) & 1u);
        look = !same && can_begin;
    }
    return e | (look << 31);
}

// rc: 0 ok, 1 unsupported, -1 malformed.  `why` gets a short reason.
int compile_pattern(const char *pat, size_t len, unsigned flags, Database &db, std::string &why);

} // namespace gscan
