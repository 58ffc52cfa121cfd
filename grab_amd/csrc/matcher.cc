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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/mman.h>
#include <ucontext.h>

#include "../../include/gscan.h"
#include "../../include/gscan_test.h"
#include "db.h"

using gscan::AltSeq;
using gscan::Database;

namespace {

// Context in front of a match that starts at p.  at_start: p is the subject start (the position the reference
// restarted pcre_exec at), so there is no byte before it.
bool pre_ok(const AltSeq &a, const uint8_t *content, size_t p, bool at_start)
{
    if (at_start) return a.pre_start;
    return p > 0 && a.pre.test(content[p - 1]);
}

bool window_at(const Database &d, const std::vector<uint8_t> &w, const uint8_t *content, size_t clen, size_t p)
{
    const size_t m = w.size();
    if (p + m > clen) return false;
    const uint8_t *t = content + p;
    size_t i = 0;
    while (i < m && d.classes[w[i]].test(t[i])) i++;
    return i == m;
}

// a.window (for a gapped path: the part behind the repeat) at pos, with the context behind it
bool rest_at(const Database &d, const AltSeq &a, const uint8_t *content, size_t clen, size_t pos)
{
    if (!window_at(d, a.window, content, clen, pos)) return false;
    const size_t e = pos + a.window.size();
    if (e == clen) return a.post_end;
    return a.post.test(content[e]) || (a.post_final_nl && content[e] == '\n' && e + 1 == clen);
}

// does the plain (not gapped) alternative a match AT p?
bool alt_matches(const Database &d, const AltSeq &a, const uint8_t *content, size_t clen, size_t p, bool at_start)
{
    return rest_at(d, a, content, clen, p) && pre_ok(a, content, p, at_start);
}

// end of the match whose window part a.window sits at pos: + the greedy repeat at the very end, if any
uint32_t end_of(const AltSeq &a, const uint8_t *t, size_t clen, size_t pos)
{
    size_t e = pos + a.window.size();
    if (a.has_tail) {
        uint64_t extra = 0;
        while (e < clen && extra < (uint64_t)a.tail_extra && a.tail.test(t[e])) {
            e++;
            extra++;
        }
    }
    return (uint32_t)e;
}

// The match pcre_exec reports AT p, if any: its end and whether its path closes a capturing group.
// Alternatives are tried in priority order (pattern.h); consecutive gapped alternatives that share one instance of
// the unbounded repeat (gap_id) are tried the way PCRE backtracks -- repeat count first (longest first when greedy),
// then the paths behind it in order.
struct MatchAt {
    uint32_t end;
    bool captures;
    bool gave_up = false; // the attempt ran into the matcher's resource limits
    uint32_t start = UINT32_MAX; // where the match is reported to start when the pattern says so (\K); UINT32_MAX: at the offset tried
};
bool match_at_alts(const Database &d, const uint8_t *content, size_t clen, size_t p, bool at_start, MatchAt &out)
{
    const size_t na = d.alts.size();
    for (size_t i = 0; i < na; i++) {
        const AltSeq &a = d.alts[i];
        if (!a.gapped) {
            if (alt_matches(d, a, content, clen, p, at_start)) {
                out = {end_of(a, content, clen, p), a.captures, false};
                return true;
            }
            continue;
        }
        size_t j = i + 1;
        while (j < na && d.alts[j].gapped && d.alts[j].gap_id == a.gap_id) j++;
        if (window_at(d, a.pwindow, content, clen, p) && pre_ok(a, content, p, at_start)) {
            const size_t g = p + a.pwindow.size();
            size_t kmax = 0;
            while (g + kmax < clen && a.gap.test(content[g + kmax])) kmax++;
            for (size_t step = 0; step < kmax; step++) {
                const size_t k = a.gap_mode == 1 ? step + 1 : kmax - step;
                for (size_t r = i; r < j; r++)
                    if (rest_at(d, d.alts[r], content, clen, g + k)) {
                        out = {end_of(d.alts[r], content, clen, g + k), d.alts[r].captures, false};
                        return true;
                    }
            }
        }
        i = j - 1;
    }
    return false;
}

// ---- the backtracking matcher over the parse tree -----------------------------------------------------------------
// PCRE's semantics by construction for everything the parser takes: alternatives left to right, greedy repeats longest
// first, lazy ones shortest first, possessive ones never given back, depth first -- so "the match reported AT p" (its
// end, and whether its path closed a capturing group: ovector[3], src/grab.cc:171,179) does not depend on how the
// pattern was unfolded for the kernels.  A repeat of a single byte class is a loop over counts, not a recursion, so
// the stack depth is bounded by the pattern's size (times the iteration count of repeated GROUPS, which are small).
using gscan::Node;
std::atomic<uint64_t> g_given_up{0}; // match attempts abandoned at a resource limit (gscan_resource_errors)

struct TreeMatch {
    const uint8_t *c;
    size_t clen, s0; // s0: the subject start (nothing before it)
    size_t end = 0;
    bool captured = false;
    // Resource limits, in the spirit of libpcre's match_limit / JIT stack: a match attempt that exceeds them is given up
    // and ENDS the chunk, as every pcre_exec error does in the reference (rc <= 0: break, src/grab.cc:178-180).  The
    // thresholds are this matcher's own.
    static uint64_t max_steps()
    {
        static const uint64_t v = getenv("GSCAN_MATCH_LIMIT") ? strtoull(getenv("GSCAN_MATCH_LIMIT"), nullptr, 10) : (uint64_t)1 << 28;
        return v;
    }
    static constexpr uint32_t kMaxDepth = 12000; // nested group iterations (each costs a few stack frames) ...
    // ... and the stack the recursion may use, whichever is hit first: the matcher runs on the CALLER's stack -- a worker of
    // grab_cli, a Python thread, a JNI / cgo thread with far less than 8 MiB
    static constexpr size_t kCallerStack = 192 * 1024;
    size_t max_stack = kCallerStack;
    bool out_of_stack = false; // gave up for THIS reason: tree_match_at repeats the attempt on a stack of its own
    const char *stack_base = (const char *)__builtin_frame_address(0);
    uint64_t steps = 0;
    uint32_t depth = 0;
    bool gave_up = false;
    // What the groups captured: kept only for patterns with back references ([2g], [2g+1] = start, end; SIZE_MAX = unset).
    // Shared with the matchers of look-arounds / atomic groups; every setter restores the old value when what follows fails.
    std::vector<size_t> *caps = nullptr;
    size_t keep = SIZE_MAX; // \K: the reported start of the match (restored when what follows fails)
    // subroutine calls: the groups of the pattern ([0]: the pattern itself), and the call this matcher runs inside
    // (-1: none; what (?(R)..) and (?(Rn)..) ask about)
    // Prefix probe (tree_prefix_viable): the text is only the first bytes of some longer subject.  Whatever the matcher decides
    // by looking AT OR BEYOND the end of what it was given could come out differently on the real subject: the first such look
    // sets touched_end and abandons the attempt (via gave_up) -- "a match may start here" is then the only safe answer.
    bool probe = false, touched_end = false;
    bool beyond(size_t pos)
    {
        if (probe && pos >= clen) touched_end = gave_up = true;
        return pos >= clen;
    }
    const std::vector<const Node *> *groups = nullptr;
    int rec_group = -1;
    uint32_t rec_depth = 0;
    static constexpr uint32_t kMaxRecursion = 2000; // nested subroutine calls (each is a matcher of its own on the stack)

    // what is still to be matched after the current node
    struct Cont {
        enum Kind { SEQ, REPG, ATOMIC_END } kind;
        const Node *n;
        size_t i;     // SEQ: next child;  REPG: iterations done
        size_t start; // REPG: where the iteration that just ended began
        const Cont *next;
    };

    // the fixed length of a look-behind alternative (the parser has checked that it has one)
    static long look_len(const Node &nd)
    {
        switch (nd.kind) {
        case Node::SET: return 1;
        case Node::ASSERT:
        case Node::LOOK: return 0;
        case Node::BACKREF:
        case Node::COND:
        case Node::RECURSE: return -1;
        case Node::ATOMIC: return look_len(nd.kids[0]);
        case Node::CAT: {
            long t = 0;
            for (const Node &k : nd.kids) {
                const long l = look_len(k);
                if (l < 0) return -1;
                t += l;
            }
            return t;
        }
        case Node::ALT: return nd.kids.empty() ? 0 : look_len(nd.kids[0]);
        case Node::REP: {
            const long l = look_len(nd.kids[0]);
            return l < 0 || nd.min != nd.max ? -1 : l * (long)nd.min;
        }
        }
        return -1;
    }

    static uint8_t lower(uint8_t b) { return b >= 'A' && b <= 'Z' ? (uint8_t)(b + 32) : b; }
    static bool is_word(uint8_t b) { return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_'; }

    bool holds(int code, size_t pos)
    {
        if (probe) { // (every one of these but \A looks at the byte at pos or asks whether the text ends there or one byte on)
            if (code != gscan::A_BOS && pos + 1 >= clen) {
                touched_end = gave_up = true;
                return false;
            }
        }
        switch (code) {
        case gscan::A_BOS: return pos == s0;
        case gscan::A_MBOL: return pos == s0 || (pos > s0 && c[pos - 1] == '\n' && pos < clen);
        case gscan::A_EOL: return pos == clen || (c[pos] == '\n' && pos + 1 == clen);
        case gscan::A_MEOL: return pos == clen || c[pos] == '\n';
        case gscan::A_EOS: return pos == clen;
        case gscan::A_WB:
        case gscan::A_NWB: {
            const bool l = pos > s0 && is_word(c[pos - 1]), r = pos < clen && is_word(c[pos]);
            return (l != r) == (code == gscan::A_WB);
        }
        }
        return false;
    }

    bool run(const Cont *k, size_t pos, bool cap)
    {
        if (!k) {
            end = pos;
            captured = cap;
            return true;
        }
        switch (k->kind) {
        case Cont::SEQ:
            if (k->i == k->n->kids.size()) { // a capturing group closes here
                if (caps && k->n->cap && k->n->group > 0) {
                    size_t *slot = caps->data() + 2 * (size_t)k->n->group;
                    const size_t o0 = slot[0], o1 = slot[1];
                    slot[0] = k->start;
                    slot[1] = pos;
                    if (run(k->next, pos, true)) return true;
                    slot = caps->data() + 2 * (size_t)k->n->group;
                    slot[0] = o0;
                    slot[1] = o1;
                    return false;
                }
                return run(k->next, pos, cap || k->n->cap);
            }
            {
                const Cont f{Cont::SEQ, k->n, k->i + 1, k->start, k->next};
                return m(&k->n->kids[k->i], pos, cap, &f);
            }
        case Cont::REPG:
            // An iteration of the UNBOUNDED part of a repeat that matched "": PCRE leaves the loop (OP_KETRMAX / OP_KETRMIN
            // with eptr == saved_eptr).  The copies a {n,m} count is compiled into have no such check: an empty iteration
            // is followed by the next one like any other.
            if (pos == k->start && k->n->max == UINT32_MAX && k->i >= (size_t)k->n->min) return run(k->next, pos, cap);
            return rep_group(k->n, k->i, pos, cap, k->next);
        case Cont::ATOMIC_END: // the atomic (possessive) part is matched: remember where, do not continue from inside it
            end = pos;
            captured = cap;
            return true;
        }
        return false;
    }

    bool rep_group(const Node *n, size_t count, size_t pos, bool cap, const Cont *k)
    {
        if (depth >= kMaxDepth) gave_up = true;
        if (gave_up) return false;
        struct Nest {
            uint32_t &d;
            explicit Nest(uint32_t &x) : d(x) { d++; }
            ~Nest() { d--; }
        } nest(depth);
        const Node *kid = &n->kids[0];
        const bool can_more = count < (size_t)n->max, can_stop = count >= (size_t)n->min;
        if (n->mode == 1) { // lazy: stop first
            if (can_stop && run(k, pos, cap)) return true;
            if (!can_more) return false;
            const Cont f{Cont::REPG, n, count + 1, pos, k};
            return m(kid, pos, cap, &f);
        }
        if (can_more) {
            const Cont f{Cont::REPG, n, count + 1, pos, k};
            if (m(kid, pos, cap, &f)) return true;
        }
        return can_stop && run(k, pos, cap);
    }

    // `what` matched on its own with a matcher of its own (same limits, same captures), to its first success: the end of
    // that match and whether its path closed a capturing group.  as_repeat: `what` is a REP node entered at count 0.
    bool atomic_run(const Node *what, bool as_repeat, size_t at, bool cap, size_t &end_out, bool &cap_out)
    {
        const Cont stop{Cont::ATOMIC_END, nullptr, 0, 0, nullptr};
        TreeMatch inner{c, clen, s0};
        inner.steps = steps;
        inner.depth = depth;
        inner.stack_base = stack_base;
        inner.max_stack = max_stack;
        inner.caps = caps;
        inner.keep = keep;
        inner.groups = groups;
        inner.rec_group = rec_group;
        inner.rec_depth = rec_depth;
        inner.probe = probe;
        const bool got = as_repeat ? inner.rep_group(what, 0, at, cap, &stop) : inner.m(what, at, cap, &stop);
        steps = inner.steps;
        gave_up = gave_up || inner.gave_up;
        touched_end = touched_end || inner.touched_end;
        out_of_stack = out_of_stack || inner.out_of_stack;
        if (!got || gave_up) return false;
        end_out = inner.end;
        cap_out = inner.captured;
        keep = inner.keep; // (the callers put the old value back when what follows the atomic part fails)
        return true;
    }

    // Does the body of the look-around n match at pos (before its negation is applied)?  The body is matched on its own, to
    // its first success (assertions are atomic).  A look-behind body has a fixed length per top-level alternative and may
    // not reach back over the subject start (the restart position: src/grab.cc:178 hands pcre_exec the subject FROM
    // there).  What the body captured stays in *caps: the caller decides whether it may.
    bool look_holds(const Node *n, size_t pos, bool cap, bool &inner_cap)
    {
        const Node *body = &n->kids[0];
        bool ok = false;
        size_t e = 0;
        if (!n->behind) {
            ok = atomic_run(body, false, pos, cap, e, inner_cap);
        } else if (body->kind == Node::ALT) {
            for (const Node &alt : body->kids) {
                const long len = look_len(alt);
                if (len >= 0 && pos >= s0 + (size_t)len) ok = atomic_run(&alt, false, pos - (size_t)len, cap, e, inner_cap) && e == pos;
                if (ok || gave_up) break;
            }
        } else {
            const long len = look_len(*body);
            if (len >= 0 && pos >= s0 + (size_t)len) ok = atomic_run(body, false, pos - (size_t)len, cap, e, inner_cap) && e == pos;
        }
        return ok;
    }

    bool m(const Node *n, size_t pos, bool cap, const Cont *k)
    {
        if (++steps > max_steps()) gave_up = true;
        if ((size_t)(stack_base - (const char *)__builtin_frame_address(0)) > max_stack) gave_up = out_of_stack = true;
        if (gave_up) return false;
        switch (n->kind) {
        case Node::SET:
            return !beyond(pos) && n->set.test(c[pos]) && run(k, pos + 1, cap);
        case Node::ASSERT:
            if (n->acode == gscan::A_KEEP) {
                const size_t old = keep;
                keep = pos;
                if (run(k, pos, cap)) return true;
                keep = old;
                return false;
            }
            return holds(n->acode, pos) && run(k, pos, cap);
        case Node::COND: {
            const size_t base = n->cond == Node::C_ASSERT ? 1 : 0;
            const std::vector<size_t> saved = caps ? *caps : std::vector<size_t>();
            bool truth = false, inner_cap = cap;
            switch (n->cond) {
            case Node::C_GROUP: truth = caps && (*caps)[2 * (size_t)n->group + 1] != SIZE_MAX; break;
            case Node::C_IN_RECURSION: truth = rec_group >= 0; break;
            case Node::C_IN_RECURSION_OF: truth = rec_group == n->group; break;
            case Node::C_ASSERT: {
                const Node *look = &n->kids[0];
                const bool ok = look_holds(look, pos, cap, inner_cap);
                if (gave_up) return false;
                truth = ok != look->neg;
                // What the body captured stays captured whenever the body matched -- also under a NEGATIVE condition, where
                // that means "condition false" (libpcre's JIT, the reference's build: a(?(?!(b))) on "ab" comes back as 0,
                // the one-pair ovector overflowing; an ordinary (?!(b)) unsets the group again).
                if (!ok) {
                    if (caps) *caps = saved;
                    inner_cap = cap;
                }
                break;
            }
            default: break; // DEFINE: never true
            }
            const Node *branch = truth ? &n->kids[base] : n->kids.size() > base + 1 ? &n->kids[base + 1] : nullptr;
            if (branch ? m(branch, pos, inner_cap, k) : run(k, pos, inner_cap)) return true;
            if (caps && n->cond == Node::C_ASSERT) *caps = saved;
            return false;
        }
        case Node::RECURSE: {
            // The called group's pattern on its own, to its first success, never re-entered (PCRE1: "a recursive subpattern
            // call is always treated as an atomic group"); what was captured during the call is dropped when it returns.
            const Node *target = groups && (size_t)n->group < groups->size() ? (*groups)[(size_t)n->group] : nullptr;
            if (!target) return false;
            if (rec_depth >= kMaxRecursion) gave_up = true;
            if (gave_up) return false;
            const Node *body = n->group == 0 ? target : &target->kids[0]; // (the group's content: a called group does not capture)
            const std::vector<size_t> saved = caps ? *caps : std::vector<size_t>();
            const Cont stop{Cont::ATOMIC_END, nullptr, 0, 0, nullptr};
            TreeMatch inner{c, clen, s0};
            inner.steps = steps;
            inner.depth = depth;
            inner.stack_base = stack_base;
            inner.max_stack = max_stack;
            inner.caps = caps;
            inner.keep = keep;
            inner.groups = groups;
            inner.rec_group = n->group;
            inner.rec_depth = rec_depth + 1;
            inner.probe = probe;
            const bool got = inner.m(body, pos, false, &stop);
            steps = inner.steps;
            gave_up = gave_up || inner.gave_up;
            touched_end = touched_end || inner.touched_end;
            out_of_stack = out_of_stack || inner.out_of_stack;
            if (caps) *caps = saved;
            if (!got || gave_up) return false;
            return run(k, inner.end, cap);
        }
        case Node::LOOK: {
            const std::vector<size_t> saved = caps ? *caps : std::vector<size_t>();
            bool inner_cap = cap;
            const bool ok = look_holds(n, pos, cap, inner_cap);
            if (gave_up) return false;
            if (n->neg) { // (groups set inside a failed -- or a negative -- assertion are unset again)
                if (caps) *caps = saved;
                return !ok && run(k, pos, cap);
            }
            if (ok && run(k, pos, inner_cap)) return true;
            if (caps) *caps = saved;
            return false;
        }
        case Node::ATOMIC: { // matched on its own to its first success; no way back into it
            const std::vector<size_t> saved = caps ? *caps : std::vector<size_t>();
            const size_t saved_keep = keep;
            size_t e = 0;
            bool inner_cap = cap;
            if (atomic_run(&n->kids[0], false, pos, cap, e, inner_cap) && run(k, e, inner_cap)) return true;
            if (caps) *caps = saved;
            keep = saved_keep;
            return false;
        }
        case Node::CAT: {
            const Cont f{Cont::SEQ, n, 0, pos, k}; // (start: where the group begins -- what a capturing group records)
            return run(&f, pos, cap);
        }
        case Node::BACKREF: {
            if (!caps) return false;
            const size_t lo = (*caps)[2 * (size_t)n->group], hi = (*caps)[2 * (size_t)n->group + 1];
            if (lo == SIZE_MAX) return false; // a reference to a group that has not been set fails
            const size_t len = hi - lo;
            if (pos + len > clen) {
                beyond(clen);
                return false;
            }
            if (n->icase) {
                for (size_t q = 0; q < len; q++)
                    if (lower(c[lo + q]) != lower(c[pos + q])) return false;
            } else if (memcmp(c + lo, c + pos, len) != 0) {
                return false;
            }
            return run(k, pos + len, cap);
        }
        case Node::ALT:
            for (const Node &kid : n->kids)
                if (m(&kid, pos, cap, k)) return true;
            return false;
        case Node::REP: {
            const Node *kid = &n->kids[0];
            if (n->max == 0) return run(k, pos, cap);
            if (kid->kind == Node::SET) { // a loop over counts
                size_t kmax = 0;
                while (kmax < (size_t)n->max && pos + kmax < clen && kid->set.test(c[pos + kmax])) kmax++;
                if (kmax < (size_t)n->max && pos + kmax >= clen) beyond(clen); // (the repeat stopped because the text did)
                if (gave_up) return false;
                if (kmax < (size_t)n->min) return false;
                if (n->mode == 2) return run(k, pos + kmax, cap); // possessive: all of it, no giving back
                if (n->mode == 1) {
                    for (size_t j = n->min; j <= kmax; j++)
                        if (run(k, pos + j, cap)) return true;
                    return false;
                }
                for (size_t j = kmax + 1; j-- > (size_t)n->min;)
                    if (run(k, pos + j, cap)) return true;
                return false;
            }
            if (n->mode == 2) { // possessive group repeat: match the repeat on its own (greedily), then never re-enter it
                Node greedy = *n;
                greedy.mode = 0;
                const std::vector<size_t> saved = caps ? *caps : std::vector<size_t>();
                const size_t saved_keep = keep;
                size_t e = 0;
                bool inner_cap = cap;
                if (atomic_run(&greedy, true, pos, cap, e, inner_cap) && run(k, e, inner_cap)) return true;
                if (caps) *caps = saved;
                keep = saved_keep;
                return false;
            }
            return rep_group(n, 0, pos, cap, k);
        }
        }
        return false;
    }
};

// A match attempt that ran out of the caller's stack budget (hundreds of nested group iterations: (?:ab)+ over a long
// run) is repeated on a stack of the matcher's own -- 16 MiB per thread, mapped on first use, entered with
// makecontext / swapcontext.  kMaxDepth (12 000 nested iterations) is what ends an attempt there; libpcre's JIT gives up
// between ~1 400 and ~4 100 iterations of such a group on its 32 KiB stack.
constexpr size_t kOwnStack = 16u << 20;
struct OwnStack {
    void *mem = nullptr;
    ucontext_t caller, callee;
    ~OwnStack()
    {
        if (mem) munmap(mem, kOwnStack);
    }
};
struct OwnCall {
    TreeMatch *t;
    const Node *root;
    size_t p;
    bool hit;
};
void own_stack_entry(unsigned lo, unsigned hi)
{
    OwnCall *c = reinterpret_cast<OwnCall *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    c->t->stack_base = (const char *)__builtin_frame_address(0);
    c->hit = c->t->m(c->root, c->p, false, nullptr);
}
bool run_on_own_stack(TreeMatch &t, const Node *root, size_t p)
{
    static thread_local OwnStack st;
    if (!st.mem) {
        void *m = mmap(nullptr, kOwnStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
        if (m == MAP_FAILED) {
            t.gave_up = true;
            return false;
        }
        st.mem = m;
    }
    OwnCall call{&t, root, p, false};
    getcontext(&st.callee);
    st.callee.uc_stack.ss_sp = st.mem;
    st.callee.uc_stack.ss_size = kOwnStack;
    st.callee.uc_link = &st.caller;
    const uintptr_t a = reinterpret_cast<uintptr_t>(&call);
    makecontext(&st.callee, reinterpret_cast<void (*)()>(own_stack_entry), 2, (unsigned)(a & 0xffffffffu), (unsigned)(a >> 32));
    swapcontext(&st.caller, &st.callee);
    return call.hit;
}

bool tree_match_at(const Database &d, const uint8_t *content, size_t clen, size_t p, size_t subject_start, MatchAt &out)
{
    if (!d.tree) return false;
    std::vector<size_t> caps;
    if (d.has_backref) caps.assign(2 * (size_t)d.n_groups + 2, SIZE_MAX);
    TreeMatch t{content, clen, subject_start};
    t.caps = d.has_backref ? &caps : nullptr;
    t.groups = &d.group_nodes;
    bool hit = t.m(d.tree.get(), p, false, nullptr);
    if (t.out_of_stack) { // again, with room
        t = TreeMatch{content, clen, subject_start};
        if (d.has_backref) caps.assign(2 * (size_t)d.n_groups + 2, SIZE_MAX);
        t.caps = d.has_backref ? &caps : nullptr;
        t.groups = &d.group_nodes;
        t.max_stack = kOwnStack - (256u << 10);
        hit = run_on_own_stack(t, d.tree.get(), p);
    }
    out.gave_up = t.gave_up;
    if (t.gave_up) g_given_up.fetch_add(1, std::memory_order_relaxed);
    if (!hit || t.gave_up) return false;
    if (t.end == p) return false; // (patterns that can match "" never get here: minlen -1, every file skipped)
    out = {(uint32_t)t.end, t.captured, false, t.keep == SIZE_MAX ? UINT32_MAX : (uint32_t)t.keep};
    return true;
}

} // namespace

// Can a match start at offset 0 of ANY subject that begins with these n bytes (no byte in front of them)?  false only if the
// matcher fails without ever looking at or beyond byte n.  (pattern.cc builds the device VM's two-byte table from it.)
bool gscan::tree_prefix_viable(const Database &d, const uint8_t *bytes, size_t n)
{
    if (!d.tree) return true;
    std::vector<size_t> caps;
    if (d.has_backref) caps.assign(2 * (size_t)d.n_groups + 2, SIZE_MAX);
    TreeMatch t{bytes, n, 0};
    t.caps = d.has_backref ? &caps : nullptr;
    t.groups = &d.group_nodes;
    t.probe = true;
    t.max_stack = 64 * 1024; // (a probe that needs more than that is called viable)
    const bool hit = t.m(d.tree.get(), 0, false, nullptr);
    return hit || t.gave_up || t.touched_end;
}

namespace {

// "The match reported AT p": the tree matcher decides; GSCAN_CHECK_TREE=1 (tests) also runs the rule on the unfolded
// alternatives and aborts on any difference between the two.
bool match_at(const Database &d, const uint8_t *content, size_t clen, size_t p, size_t subject_start, MatchAt &out)
{
    static const bool check = getenv("GSCAN_CHECK_TREE") != nullptr;
    const bool at_start = p == subject_start;
    const bool hit = tree_match_at(d, content, clen, p, subject_start, out);
    if (check && d.exact && !out.gave_up) { // (an inexact database's alternatives are necessary conditions only)
        MatchAt o2{0, false};
        const bool h2 = match_at_alts(d, content, clen, p, at_start, o2);
        if (h2 != hit || (hit && (o2.end != out.end || o2.captures != out.captures))) {
            fprintf(stderr, "gscan: tree matcher and alternatives disagree at %zu (at_start %d): tree %d end %u cap %d / alts %d end %u cap %d\n", p,
                    (int)at_start, (int)hit, out.end, (int)out.captures, (int)h2, o2.end, (int)o2.captures);
            abort();
        }
    }
    return hit;
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
    bool first0;   // offset 0 is still to be handed out (see the constructor)
    bool probe;    // look for unlisted members of a listed hit's group

    static constexpr size_t kEnd = SIZE_MAX;

    Walk(const Database &db, const uint8_t *c, size_t cl, const uint32_t *st, size_t nst, size_t li0, const uint32_t *tl, size_t ntl, size_t from)
        : d(db), content(c), clen(cl), starts(st), n(nst), li(li0), tails(tl), nt(ntl), ti(0), x(from)
    {
        // (a database whose candidates the device has confirmed itself lists EVERY hit it kept -- no group-start compression --
        // so there is nothing to find between the listed offsets; probing would only dig up what the device has dropped)
        probe = !d.prog.vm_filter;
        in_group = probe && from > 0 && dev_hit(d, content, clen, from - 1);
        // device windows with a leading context position start one byte before the offset they report: offset 0 can
        // never be listed.  A walk that starts there has to look at it itself.
        first0 = from == 0 && d.dev_pre;
    }

    size_t next()
    {
        if (first0) {
            first0 = false;
            x = 1;
            in_group = false; // offset 1, if it is a hit, has no listed predecessor: it is listed itself
            return 0;
        }
        for (;;) {
            if (in_group) {
                if (x < clen && dev_hit(d, content, clen, x)) return x++;
                in_group = false;
            }
            while (li < n && (size_t)starts[li] < x) li++;
            if (li + 16 < n) __builtin_prefetch(content + starts[li + 16]); // (see gscan_next_match's fast path)
            while (ti < nt && (size_t)tails[ti] < x) ti++;
            const size_t a = li < n ? (size_t)starts[li] : kEnd, b = ti < nt ? (size_t)tails[ti] : kEnd;
            const size_t c = std::min(a, b);
            if (c == kEnd) return kEnd;
            x = c + 1;
            in_group = probe && c == a; // a listed offset is a device hit
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
            const size_t need = a.window.size() + (a.gapped ? 1 : 0) + back; // a gapped path's device window: one repeat byte + the rest
            if (need <= clen) v.push_back((uint32_t)(clen - need));
        }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}

// The leftmost offset p >= x at which some alternative's condition holds (s: the subject start -- nothing lies before it).
// Every kind of alternative keeps its own answer in the cursor: it is computed by walking the device hits from x on until
// that kind has a candidate -- possibly far ahead -- and stays valid until the search position passes it.
constexpr uint32_t kNoCand = UINT32_MAX;

// the plain alternatives: the subject start itself (nothing before it), else the first offset of the walk at which one of
// them matches
uint32_t next_plain(const Database &d, const uint8_t *content, size_t clen, const uint32_t *starts, size_t n, size_t li, const gscan_cursor *cur,
                    size_t s, size_t x)
{
    size_t from = x;
    if (x == s) {
        for (const AltSeq &a : d.alts)
            if (!a.gapped && alt_matches(d, a, content, clen, s, true)) return (uint32_t)s;
        from = s + 1;
    }
    Walk w(d, content, clen, starts, n, li, cur->tails, cur->ntails, from);
    for (size_t at = w.next(); at != Walk::kEnd; at = w.next())
        for (const AltSeq &a : d.alts)
            if (!a.gapped && alt_matches(d, a, content, clen, at, false)) return (uint32_t)at;
    return kNoCand;
}

// a gapped alternative  P . C{1,} . R : the device window is  C . R  (one repeat byte + the rest).  For every such hit h, in
// ascending order: the run of C bytes that ends at h reaches back to r0; a match starts wherever P ends inside [r0, h].
// The first hit that has such a start gives the leftmost start of the alternative (a later hit lies in the same run, or
// in a later one).
uint32_t next_gapped(const Database &d, const AltSeq &a, const uint8_t *content, size_t clen, const uint32_t *starts, size_t n, size_t li,
                     const gscan_cursor *cur, size_t s, size_t x)
{
    const size_t plen = a.pwindow.size(), t = x + plen;
    if (t >= clen) return kNoCand;
    Walk w(d, content, clen, starts, n, li, cur->tails, cur->ntails, t);
    for (size_t h = w.next(); h != Walk::kEnd; h = w.next()) {
        if (h < t || !a.gap.test(content[h]) || !rest_at(d, a, content, clen, h + 1)) continue;
        size_t r0 = h;
        while (r0 > t && a.gap.test(content[r0 - 1])) r0--;
        for (size_t g = r0; g <= h; g++) {
            const size_t p = g - plen;
            if (window_at(d, a.pwindow, content, clen, p) && pre_ok(a, content, p, p == s)) return (uint32_t)p;
        }
    }
    return kNoCand;
}

size_t leftmost(const Database &d, const uint8_t *content, size_t clen, const uint32_t *starts, size_t n, size_t li, gscan_cursor *cur, size_t s,
                size_t x, bool any_plain)
{
    // an answer is still good if it lies at or beyond x -- except AT the subject start, where "nothing before it" replaces
    // the byte before (the answer was found for an earlier subject start)
    if (x == s) { // the subject start has its own rule ("nothing before it"): the remembered answers know nothing about it
        for (const AltSeq &a : d.alts) {
            if (!a.gapped) {
                if (alt_matches(d, a, content, clen, s, true)) return s;
                continue;
            }
            if (!window_at(d, a.pwindow, content, clen, s) || !pre_ok(a, content, s, true)) continue;
            for (size_t g = s + a.pwindow.size(); g < clen && a.gap.test(content[g]); g++)
                if (rest_at(d, a, content, clen, g + 1)) return s;
        }
    }
    auto good = [&](size_t slot) { return cur->next_known[slot] && (cur->next_at[slot] == kNoCand || (cur->next_at[slot] >= x && cur->next_at[slot] != s)); };
    uint32_t best = kNoCand;
    if (any_plain) {
        if (!good(0)) {
            cur->next_at[0] = next_plain(d, content, clen, starts, n, li, cur, s, x);
            cur->next_known[0] = 1;
        }
        best = cur->next_at[0];
    }
    for (size_t i = 0; i < d.alts.size(); i++) {
        if (!d.alts[i].gapped) continue;
        if (!good(1 + i)) {
            cur->next_at[1 + i] = next_gapped(d, d.alts[i], content, clen, starts, n, li, cur, s, x);
            cur->next_known[1 + i] = 1;
        }
        best = std::min(best, cur->next_at[1 + i]);
    }
    return best == kNoCand ? Walk::kEnd : (size_t)best;
}

} // namespace

extern "C" {

int gscan_match_at(const gscan_db *db, const void *content, size_t clen, uint32_t p)
{
    const Database &d = db->db;
    MatchAt m;
    return d.minlen > 0 && match_at(d, (const uint8_t *)content, clen, p, p, m);
}

int gscan_match_info(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p, uint32_t *end)
{
    const Database &d = db->db;
    MatchAt m;
    if (d.minlen <= 0 || p < subject_start || !match_at(d, (const uint8_t *)content, clen, p, subject_start, m)) return 0;
    if (end) *end = m.end;
    return m.captures ? 2 : 1;
}

uint32_t gscan_match_end(const gscan_db *db, const void *content, size_t clen, uint32_t start)
{
    const Database &d = db->db;
    MatchAt m;
    return d.minlen > 0 && match_at(d, (const uint8_t *)content, clen, start, start, m) ? m.end : start; // start itself: not a match start
}

uint64_t gscan_resource_errors(void) { return g_given_up.load(std::memory_order_relaxed); }

size_t gscan_tail_positions(const gscan_db *db, size_t clen, uint32_t *out, size_t cap) { return tail_positions(db->db, clen, out, cap); }

int gscan_next_match(const gscan_db *db, const void *content_, size_t clen, const uint32_t *starts, size_t n, gscan_cursor *cur,
                     uint32_t s, uint32_t *m0, uint32_t *m1)
{
    const Database &d = db->db;
    const uint8_t *content = (const uint8_t *)content_;
    if (d.minlen <= 0 || !cur || (size_t)s >= clen) return 0;
    if (d.resolve) return gscan_next_resolved(db, content_, clen, starts, nullptr, n, cur, s, m0, m1); // (a list nobody has resolved: every record is the matcher's)
    if (!cur->ready) { // first call for this chunk
        cur->li = 0;
        cur->ntails = (uint32_t)std::min(tail_positions(d, clen, cur->tails, GSCAN_MAX_TAILS), (size_t)GSCAN_MAX_TAILS);
        memset(cur->next_known, 0, sizeof cur->next_known);
        cur->ready = 1;
    }
    while (cur->li < n && starts[cur->li] < s) cur->li++; // first entry >= s; s only moves forward: the cursor is kept across calls

    // The common case -- one alternative, no context, no gap -- needs none of the machinery below: s itself if the
    // window matches there, else the first listed start after s.  (If s is no match, the next candidate after it begins
    // a group and is therefore listed; the kernels list candidates only, so it is not tested again.)
    if (d.exact && d.alts.size() == 1 && !d.alts[0].gapped && !d.dev_pre && !d.dev_post) {
        const AltSeq &a0 = d.alts[0];
        // (the walk touches the text once per match, a few hundred bytes apart: every touch a cache and TLB miss -- 100 of the
        // loop's 105 ns per match.  The list says where the next ones will be.)
        if (cur->li + 16 < n) { // (the line in front as well: with lines printed the walk copies from the last newline on)
            const uint8_t *ahead = content + starts[cur->li + 16];
            __builtin_prefetch(ahead - 64);
            __builtin_prefetch(ahead);
            __builtin_prefetch(ahead + 64);
        }
        size_t at = s;
        if (d.solitary) { // every candidate is listed: the first one at or after s, without a look at the text (a fresh
                          // mapping costs a page fault per match there -- half a second per 500 000 matches)
            if (cur->li >= n) return 0;
            at = starts[cur->li];
        } else if (!window_at(d, a0.window, content, clen, s)) {
            size_t li = cur->li;
            while (li < n && starts[li] <= s) li++;
            if (li >= n) return 0;
            at = starts[li];
        }
        *m0 = (uint32_t)at;
        *m1 = end_of(a0, content, clen, at);
        return a0.captures ? 2 : 1;
    }
    bool any_plain = false;
    for (const AltSeq &a : d.alts) any_plain = any_plain || !a.gapped;
    // The leftmost offset >= x at which some alternative holds, then the matcher's verdict AT that offset.  For an exact
    // database the verdict is always "match" (the alternatives are the pattern) and the loop runs once; for an inexact
    // one the alternatives only say where a match may begin, and a refused offset sends the search on behind it.
    size_t li = cur->li;
    for (size_t x = s;;) {
        const size_t best = leftmost(d, content, clen, starts, n, li, cur, s, x, any_plain);
        if (best == Walk::kEnd) return 0;
        MatchAt m;
        if (match_at(d, content, clen, best, (size_t)s, m)) {
            *m0 = m.start == UINT32_MAX ? (uint32_t)best : m.start;
            *m1 = m.end;
            return m.captures ? 2 : 1;
        }
        if (d.exact || m.gave_up) return 0; // (exact: cannot happen.)  A given-up attempt ends the chunk: rc <= 0, src/grab.cc:179
        x = best + 1;
        if (x >= clen) return 0;
        while (li < n && (size_t)starts[li] < x) li++;
    }
}

int gscan_next_listed(const gscan_db *db, size_t clen, const uint32_t *starts, const uint32_t *ends, size_t n, gscan_cursor *cur, uint32_t s,
                      uint32_t *m0, uint32_t *m1)
{
    const Database &d = db->db;
    if (d.minlen <= 0 || !cur || (size_t)s >= clen) return 0;
    if (!d.prog.ends_ok || !ends) return -1;
    if (!cur->ready) { // first call for this chunk (as gscan_next_match; such a pattern has no tail positions)
        cur->li = 0;
        cur->ntails = (uint32_t)std::min(tail_positions(d, clen, cur->tails, GSCAN_MAX_TAILS), (size_t)GSCAN_MAX_TAILS);
        memset(cur->next_known, 0, sizeof cur->next_known);
        cur->ready = 1;
    }
    while (cur->li < n && starts[cur->li] < s) cur->li++; // first entry >= s
    if (cur->li >= n) return 0;
    if (ends[cur->li] == 0) return -1;
    *m0 = starts[cur->li];
    *m1 = ends[cur->li];
    return 1;
}

// The reference's loop step for a database whose matches the device settles (Database::resolve).  starts: every offset at which
// a start window fits (ends == NULL: nobody has looked at them yet), or what k_resolve left of them with ends[i] = ovector[1],
// GSCAN_END_ASK (ask the host matcher at starts[i]) or GSCAN_END_CAPTURES (a match that sets a capturing group).
int gscan_next_resolved(const gscan_db *db, const void *content_, size_t clen, const uint32_t *starts, const uint32_t *ends, size_t n, gscan_cursor *cur,
                        uint32_t s, uint32_t *m0, uint32_t *m1)
{
    const Database &d = db->db;
    const uint8_t *content = (const uint8_t *)content_;
    if (d.minlen <= 0 || !cur || (size_t)s >= clen) return 0;
    if (!d.resolve) return -1;
    if (!cur->ready) {
        cur->li = 0;
        cur->ntails = 0;
        memset(cur->next_known, 0, sizeof cur->next_known);
        cur->ready = 1;
    }
    // (1) the offsets the pattern can look back over the restart position from: the device's verdicts there were reached with
    // bytes in front that pcre_exec, handed the subject FROM s (src/grab.cc:178), does not see -- the host's matcher decides
    // (... unless the byte in front of s is, to this pattern, what the subject start is: a non-word byte for \b, a newline for (?m)^)
    const bool same = d.reach == 1 && s > 0 && d.start_like.test(content[s - 1]);
    const size_t near_end = same ? (size_t)s : std::min(clen, (size_t)s + d.reach);
    for (size_t q = s; q < near_end; q++) {
        if (d.first_ok && !d.first.test(content[q])) continue;
        MatchAt m;
        if (match_at(d, content, clen, q, (size_t)s, m)) {
            *m0 = (uint32_t)q; // (no \K in a database of this kind)
            *m1 = m.end;
            return m.captures ? 2 : 1;
        }
        if (m.gave_up) return 0; // rc <= 0 ends the chunk: src/grab.cc:179
    }
    // (2) from there on the list is the truth: the first record is the leftmost match
    while (cur->li < n && (size_t)starts[cur->li] < near_end) cur->li++;
    for (size_t i = cur->li; i < n; i++) {
        const uint32_t e = ends ? ends[i] : GSCAN_END_ASK;
        if (e == GSCAN_END_CAPTURES) return 2;
        if (e != GSCAN_END_ASK) {
            *m0 = starts[i];
            *m1 = e & ~GSCAN_END_LOOK;
            return 1;
        }
        MatchAt m;
        if (match_at(d, content, clen, starts[i], (size_t)s, m)) {
            *m0 = starts[i];
            *m1 = m.end;
            return m.captures ? 2 : 1;
        }
        if (m.gave_up) return 0;
    }
    return 0;
}

int gscan_db_first(const gscan_db *db, uint8_t table[256])
{
    if (!db || !table) return GSCAN_EINVAL;
    for (int b = 0; b < 256; b++) table[b] = (uint8_t)((db->db.first_ok ? db->db.first.test((unsigned)b) : 1) | (db->db.start_like.test((unsigned)b) ? 2 : 0));
    return db->db.first_ok ? 1 : 0;
}

int gscan_vm_verdict(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p)
{
    if (!db || !db->db.vm_ok || p < subject_start || (size_t)p > clen || clen > 0xfffffff0u) return -1;
    return gscan::vm_run(&db->db.prog.vm, (const uint8_t *)content, (uint32_t)clen, p, subject_start);
}

int gscan_vm_match(const gscan_db *db, const void *content, size_t clen, uint32_t subject_start, uint32_t p, uint32_t *end, int *captures)
{
    if (!db || !db->db.vm_ok || p < subject_start || (size_t)p > clen || clen > 0xfffffff0u) return -1;
    gscan::VmOut o{0, 0};
    const int v = gscan::vm_run(&db->db.prog.vm, (const uint8_t *)content, (uint32_t)clen, p, subject_start, o);
    if (v == 1) {
        if (end) *end = o.end;
        if (captures) *captures = (int)o.cap;
    }
    return v;
}

long gscan_vm_resolve(const gscan_db *db, const void *content, size_t clen, const uint32_t *hits, size_t n, uint32_t *starts, uint32_t *ends)
{
    if (!db || !db->db.vm_ok || (!hits && n) || !starts || !ends || clen > 0xfffffff0u) return -1;
    size_t k = 0;
    for (size_t i = 0; i < n; i++) { // (what k_resolve does with one record, same source)
        if ((size_t)hits[i] >= clen) continue;
        gscan::VmOut o{0, 0};
        const int v = gscan::vm_run(&db->db.prog.vm, (const uint8_t *)content, (uint32_t)clen, hits[i], 0, o);
        if (v == 0) continue;
        starts[k] = hits[i];
        ends[k] = gscan::resolve_code(&db->db.prog, (const uint8_t *)content, (uint32_t)clen, hits[i], v, o);
        k++;
    }
    return (long)k;
}

int gscan_vm_pair(const gscan_db *db, unsigned b0, unsigned b1)
{
    if (!db || !db->db.prog.vm_filter || !db->db.prog.vm_pair_ok || b0 > 255 || b1 > 255) return -1;
    const uint32_t idx = b0 << 8 | b1;
    return (db->db.prog.vm_pair[idx >> 5] >> (idx & 31)) & 1u;
}

int gscan_prefix_viable(const gscan_db *db, const void *bytes, size_t n)
{
    if (!db || (!bytes && n)) return -1;
    return gscan::tree_prefix_viable(db->db, (const uint8_t *)bytes, n) ? 1 : 0;
}

long gscan_vm_filter(const gscan_db *db, const void *content, size_t clen, const uint32_t *hits, size_t n, uint32_t *kept)
{
    if (!db || !db->db.prog.vm_filter || (!hits && n) || clen > 0xfffffff0u) return -1;
    size_t k = 0;
    for (size_t i = 0; i < n; i++)
        if (gscan::vm_keep_hit(&db->db.prog, &db->db.prog.vm, (const uint8_t *)content, (uint32_t)clen, hits[i])) {
            if (kept) kept[k] = hits[i];
            k++;
        }
    return (long)k;
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
