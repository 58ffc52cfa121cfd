// matcher.cc -- the host half of the match rule.  Host only; no HIP.
//
// The kernels answer "where are the device windows of this pattern in the chunk" (an ascending list of hit
// offsets, one per group of consecutive hits).  This file turns that list into what the reference's loop
// needs at every step: "the leftmost match in content[s..clen) when the subject starts at s, and its end"
// -- exactly what pcre_exec(d_pcreh, d_extra, start, end - start, 0, 0, ovector, 3) returns in
// /root/reference/src/grab.cc:178 -- including everything that is not a pure function of a byte window:
// the restart position (nothing before it: SURVEY.md Q4), windows that end with the chunk, priority among
// alternatives, the one-pair ovector (Q5).
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/gscan.h"
#include "db.h"

using gscan::AltSeq;
using gscan::Database;

namespace {

// does alternative a match AT p?  at_start: p is the subject start (the position the reference restarted
// pcre_exec at), so there is no byte before it.
bool alt_matches(const Database &d, const AltSeq &a, const uint8_t *content, size_t clen, size_t p, bool at_start)
{
    const size_t m = a.window.size();
    if (p + m > clen) return false;
    const uint8_t *t = content + p;
    size_t i = 0;
    while (i < m && d.classes[a.window[i]].test(t[i])) i++;
    if (i < m) return false;
    if (at_start) {
        if (!a.pre_start) return false;
    } else if (p == 0 || !a.pre.test(content[p - 1])) {
        return false;
    }
    const size_t e = p + m;
    if (e == clen) return a.post_end;
    return a.post.test(content[e]) || (a.post_final_nl && content[e] == '\n' && e + 1 == clen);
}

// The alternative pcre_exec's match at p goes through: the first one, in priority order, that matches there
// (pattern.h).  nullptr: no match starts at p.
const AltSeq *alt_at(const Database &d, const uint8_t *content, size_t clen, size_t p, bool at_start)
{
    for (const AltSeq &a : d.alts)
        if (alt_matches(d, a, content, clen, p, at_start)) return &a;
    return nullptr;
}

uint32_t end_of(const AltSeq &a, const uint8_t *t, size_t clen, size_t start)
{
    size_t e = start + a.window.size();
    if (a.has_tail) {
        uint64_t extra = 0;
        while (e < clen && extra < (uint64_t)a.tail_extra && a.tail.test(t[e])) {
            e++;
            extra++;
        }
    }
    return (uint32_t)e;
}

// is x an offset the kernels report (before group-start suppression): some alternative's DEVICE window -- the
// window plus its context positions -- fits into the chunk and matches, starting at x - shift
bool dev_hit(const Database &d, const uint8_t *content, size_t clen, size_t x)
{
    const size_t shift = d.dev_pre ? 1 : 0;
    if (x < shift) return false;
    const size_t q = x - shift;
    for (const std::vector<uint8_t> &w : d.dev_windows) {
        if (q + w.size() > clen) continue;
        size_t i = 0;
        while (i < w.size() && d.classes[w[i]].test(content[q + i])) i++;
        if (i == w.size()) return true;
    }
    return false;
}

// Ascending walk over the offsets where a match may start (or, later, where a device window of any kind sits):
// the listed group starts, the unlisted members of those groups (found by testing the successor of every hit),
// and the few tail offsets whose windows end with the chunk and which the kernels therefore never list.
struct Walk {
    const Database &d;
    const uint8_t *content;
    size_t clen;
    const uint32_t *starts;
    size_t n, li;
    const uint32_t *tails;
    size_t nt, ti;
    size_t x;      // next offset to look at
    bool in_group; // x - 1 was a device hit: x may belong to the same group without being listed

    static constexpr size_t kEnd = SIZE_MAX;

    Walk(const Database &db, const uint8_t *c, size_t cl, const uint32_t *st, size_t nst, size_t li0, const uint32_t *tl, size_t ntl, size_t from)
        : d(db), content(c), clen(cl), starts(st), n(nst), li(li0), tails(tl), nt(ntl), ti(0), x(from)
    {
        in_group = from > 0 && dev_hit(d, content, clen, from - 1);
    }

    size_t next()
    {
        for (;;) {
            if (in_group) {
                if (x < clen && dev_hit(d, content, clen, x)) return x++;
                in_group = false;
            }
            while (li < n && (size_t)starts[li] < x) li++;
            while (ti < nt && (size_t)tails[ti] < x) ti++;
            const size_t a = li < n ? (size_t)starts[li] : kEnd, b = ti < nt ? (size_t)tails[ti] : kEnd;
            const size_t c = std::min(a, b);
            if (c == kEnd) return kEnd;
            x = c + 1;
            in_group = c == a; // a listed offset is a device hit
            return c;
        }
    }
};

size_t tail_positions(const Database &d, size_t clen, uint32_t *out, size_t cap)
{
    if (!d.dev_post || d.minlen <= 0) return 0;
    // windows that end at the chunk end, or one byte before it ($ in front of a final newline): the kernels ask for a
    // real byte after the window, so these positions are never in their lists
    std::vector<uint32_t> v;
    for (const AltSeq &a : d.alts)
        for (size_t back = 0; back < 2; back++) {
            const size_t need = a.window.size() + back;
            if (need <= clen) v.push_back((uint32_t)(clen - need));
        }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}

} // namespace

extern "C" {

int gscan_match_at(const gscan_db *db, const void *content, size_t clen, uint32_t p)
{
    const Database &d = db->db;
    if (d.minlen <= 0) return 0;
    return alt_at(d, (const uint8_t *)content, clen, p, true) != nullptr;
}

int gscan_match_info(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p, uint32_t *end)
{
    const Database &d = db->db;
    const uint8_t *t = (const uint8_t *)content;
    if (d.minlen <= 0 || p < subject_start) return 0;
    const AltSeq *a = alt_at(d, t, clen, p, p == subject_start);
    if (!a) return 0;
    if (end) *end = end_of(*a, t, clen, p);
    return a->captures ? 2 : 1;
}

uint32_t gscan_match_end(const gscan_db *db, const void *content, size_t clen, uint32_t start)
{
    const Database &d = db->db;
    const uint8_t *t = (const uint8_t *)content;
    const AltSeq *a = d.minlen > 0 ? alt_at(d, t, clen, start, true) : nullptr;
    return a ? end_of(*a, t, clen, start) : start; // start itself: not a match start
}

size_t gscan_tail_positions(const gscan_db *db, size_t clen, uint32_t *out, size_t cap) { return tail_positions(db->db, clen, out, cap); }

int gscan_next_match(const gscan_db *db, const void *content_, size_t clen, const uint32_t *starts, size_t n, gscan_cursor *cur,
                     uint32_t s, uint32_t *m0, uint32_t *m1)
{
    const Database &d = db->db;
    const uint8_t *content = (const uint8_t *)content_;
    if (d.minlen <= 0 || !cur || (size_t)s >= clen) return 0;
    if (!cur->ready) { // first call for this chunk
        cur->li = 0;
        cur->ntails = (uint32_t)std::min(tail_positions(d, clen, cur->tails, GSCAN_MAX_TAILS), (size_t)GSCAN_MAX_TAILS);
        cur->ready = 1;
    }
    while (cur->li < n && starts[cur->li] <= s) cur->li++; // s only moves forward: the cursor is kept across calls

    // the subject start itself: nothing before it
    const AltSeq *a = alt_at(d, content, clen, s, true);
    size_t at = s;
    if (!a) {
        Walk w(d, content, clen, starts, n, cur->li, cur->tails, cur->ntails, (size_t)s + 1);
        for (;;) {
            at = w.next();
            if (at == Walk::kEnd) return 0;
            a = alt_at(d, content, clen, at, false);
            if (a) break;
        }
    }
    *m0 = (uint32_t)at;
    *m1 = end_of(*a, content, clen, at);
    return a->captures ? 2 : 1;
}

int gscan_db_dev_window(const gscan_db *db, int alt, int pos, uint8_t table[256], int *len, int *shift)
{
    if (!db) return GSCAN_EINVAL;
    const Database &d = db->db;
    if (alt < 0 || (size_t)alt >= d.dev_windows.size()) return GSCAN_EINVAL;
    const std::vector<uint8_t> &w = d.dev_windows[(size_t)alt];
    if (len) *len = (int)w.size();
    if (shift) *shift = d.dev_pre ? 1 : 0;
    if (table) {
        if (pos < 0 || (size_t)pos >= w.size()) return GSCAN_EINVAL;
        for (int b = 0; b < 256; b++) table[b] = d.classes[w[(size_t)pos]].test((unsigned)b);
    }
    return GSCAN_OK;
}

} // extern "C"
