// vm.h -- a small backtracking regex VM that runs on the DEVICE (and, same source, on the host for its tests).
//
// What it is for (task 7; DESIGN.md 2 "Beyond the unfoldable subset"): for patterns whose alternatives only
// say what a match must BEGIN with (gscan_info.exact == 0) the kernels used to list every such offset and the host's
// backtracking matcher (matcher.cc, TreeMatch) was asked at each of them -- one CPU thread per worker in front of a
// common prefix.  With this VM the scan kernel asks the question itself, per candidate, before it writes the record:
// the parse tree is compiled into a short program (vm_compile, pattern.cc) and run AT the candidate offset; a candidate
// at which the program finds no match is dropped on the device and never reaches the host.
//
// The VM is a FILTER, not the judge: it answers  0 = no match starts at p,  1 = a match starts at p,  2 = gave up
// (step or stack limit, anything it is not sure about) -- and only 0 drops a candidate.  For everything that is kept the
// host's matcher still decides the match end, the reported start (\K) and the one-pair ovector rule (src/grab.cc:171,179)
// as before.  So the VM has to be SOUND in one direction only: never say 0 where TreeMatch finds a match.  It follows
// TreeMatch's order of exploration construct by construct (alternatives left to right, greedy longest first, lazy shortest
// first, possessive / atomic / look-around bodies matched once and cut, the empty-iteration rule of unbounded group
// repeats, captures restored on backtracking), which is PCRE's; tests/test_vm.py checks verdict against TreeMatch and
// libpcre offset by offset on random patterns.
//
// Device use is limited to patterns that never look BEHIND the match start (no ^ \A \b \B (?m)^ look-behind): then "a
// match starts at p" does not depend on where the reference restarted pcre_exec and one verdict per
// offset serves every restart position.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GSCAN_HD __host__ __device__
#else
#define GSCAN_HD
#endif

namespace gscan {

constexpr int kVmMaxIns = 192;   // instructions per program
constexpr int kVmMaxCls = 32;    // byte classes per program
constexpr int kVmMaxSlots = 40;  // capture offsets (3 per group) + repeat counters and iteration marks
constexpr int kVmStack = 80;     // backtrack entries (2 words each) per attempt
constexpr uint32_t kVmMaxSteps = 1536;
constexpr uint32_t kVmUnset = 0xffffffffu;
constexpr uint32_t kVmInf = 0xffffffffu;

enum : uint32_t {
    V_SET = 1,     // a: class                                   one byte of the class
    V_REPSET,      // a: class, b: min, c: max; op bits 8-9: mode  a repeat of one class (0 greedy, 1 lazy, 2 possessive)
    V_JMP,         // a: pc
    V_SPLIT,       // a: pc tried first, b: pc tried second; c: class of the bytes the first branch can begin with | the second's << 16
                   //    (0xffff: unknown / the branch may match ""): a branch the next byte rules out is not entered, and its
                   //    choice point never pushed
    V_SAVE,        // a: slot                                    slot = pos (undone on backtracking)
    V_CLOSE,       // a: first capture slot of the group, b: slot holding its start   a capturing group closes
    V_REP_ENTER,   // a: counter slot                            a repeated group is entered: counter = 0
    V_REP_TOP,     // a: counter slot, b: min, c: max; op bits 8-9: mode, bits 16-31: exit pc; the body starts at pc + 1
    V_REP_END,     // a: counter slot, b: mark slot, c: min; op bit 8: unbounded, bits 16-31: pc of the REP_TOP
    V_ASSERT,      // a: A_* code
    V_BACKREF,     // a: first capture slot of the group; op bit 8: caseless
    V_BAR_BEGIN,   // op bits 8-9: 0 atomic group, 1 positive look-around, 2 negative; bits 16-31: pc behind the matching BAR_END
    V_BAR_END,     // op bits 8-9: the same
    V_BACK,        // a: length                                  look-behind: step back (fails in front of the subject start)
    V_ISSET,       // a: first capture slot of a group; op bit 8: negated   holds iff the group is (is not) set: the two guards a
                   //    conditional group (?(1)yes|no) is compiled into (vm_compile.cc)
    V_MATCH,
    V_FAIL,
};

struct VmIns {
    uint32_t op, a, b, c;
};

struct VmProg {
    uint32_t n_ins, n_slots, n_cls, ok; // ok: the pattern compiled into a program within the limits
    uint32_t n_groups;                  // capturing groups whose spans the program records (slots [2g, 2g + 1], g = 1 ..): vm_run reports
                                        // whether the match set one (the reference's one-pair ovector, src/grab.cc:171,179); 0: none recorded
    uint32_t pad_[3];
    VmIns ins[kVmMaxIns];
    uint32_t cls[kVmMaxCls][8];
};

// backtrack stack entry kinds (low 4 bits of word 0; the rest of word 0 is the payload, word 1 a value)
enum : uint32_t {
    VK_CHOICE = 1, // payload: pc, value: pos
    VK_UNDO,       // payload: slot, value: old content
    VK_RANGE_DN,   // header of a 2-entry frame; payload: pc to continue at; the entry below holds {lo, cur}: give back one byte at a time
    VK_RANGE_UP,   // the same, taking one more byte at a time: {hi, cur}
    VK_BAR,        // payload: kind (2 bits) | pc behind the group << 2, value: pos where the group began
    VK_DEAD,       // a choice point that an atomic group / look-around has cut
    VK_DEAD2,      // ... a 2-entry one
};

GSCAN_HD inline bool vm_is_word(uint32_t b) { return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_'; }
GSCAN_HD inline uint32_t vm_lower(uint32_t b) { return b >= 'A' && b <= 'Z' ? b + 32 : b; }

// The text as the VM reads it: eight bytes per load, kept in a register pair.  On the device a byte of text is a round trip
// to the L2 (the VM comes to it long after the scan streamed it through), and a class repeat or a literal walks the text
// byte by byte: a dependent chain of such round trips per byte was most of what a VM run cost (k_resolve: 2.2 ms per 64 MiB
// window of a pattern with a candidate every five bytes).  The window follows the position: a miss refills it FROM the byte
// asked for, so a forward walk loads once per eight bytes and a look back costs one load.
struct VmText {
    const uint8_t *c;
    uint32_t clen;
    uint32_t base;          // the window holds the bytes [base, base + 8) of the text (beyond clen: zeros)
    unsigned long long v;
    GSCAN_HD inline VmText(const uint8_t *c_, uint32_t clen_) : c(c_), clen(clen_), base(0xfffffff0u), v(0) {}
    GSCAN_HD inline uint32_t at(uint32_t pos) // pos < clen
    {
        uint32_t d = pos - base;
        if (d >= 8u) {
            base = pos;
            d = 0;
            if (pos + 8u <= clen) {
                __builtin_memcpy(&v, c + pos, 8);
            } else {
                v = 0;
                for (uint32_t k = 0; pos + k < clen; k++) v |= (unsigned long long)c[pos + k] << (8u * k);
            }
        }
        return (uint32_t)(v >> (8u * d)) & 0xffu;
    }
};

// Assertion codes: pattern.h (A_BOS = 1 ... A_KEEP = 8).  s0: the subject start -- nothing lies before it.
GSCAN_HD inline bool vm_holds(uint32_t code, VmText &c, uint32_t clen, uint32_t s0, uint32_t pos)
{
    switch (code) {
    case 1: return pos == s0;                                                        // A_BOS
    case 2: return pos == s0 || (pos > s0 && pos < clen && c.at(pos - 1) == '\n');   // A_MBOL
    case 3: return pos == clen || (pos + 1 == clen && c.at(pos) == '\n');            // A_EOL
    case 4: return pos == clen || c.at(pos) == '\n';                                 // A_MEOL
    case 5: return pos == clen;                                                      // A_EOS
    case 6:
    case 7: {                                                                        // A_WB, A_NWB
        const bool l = pos > s0 && vm_is_word(c.at(pos - 1)), r = pos < clen && vm_is_word(c.at(pos));
        return (l != r) == (code == 6);
    }
    case 8: return true;                                                             // A_KEEP: where the match is REPORTED to start is the host's business
    }
    return false;
}

// The VM's slots (capture offsets, repeat counters, iteration marks).  NREG > 0: at most NREG of them, held in registers --
// every access is a compare-and-select chain over constant indices, which is a handful of VALU operations where a
// dynamically indexed array is a round trip to scratch memory (on the device that is what bounds a pattern whose every
// candidate runs the VM).  NREG == 0: any number, in an array.
template <int NREG>
struct VmSlots {
    uint32_t r[NREG > 0 ? NREG : kVmMaxSlots];
    GSCAN_HD inline void init(uint32_t n)
    {
        if (NREG > 0) {
#pragma unroll
            for (int k = 0; k < NREG; k++) r[k] = kVmUnset;
        } else {
            for (uint32_t i = 0; i < n; i++) r[i] = kVmUnset;
        }
    }
    GSCAN_HD inline uint32_t get(uint32_t i) const
    {
        if (NREG > 0) {
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < NREG; k++) v = i == (uint32_t)k ? r[k] : v;
            return v;
        }
        return r[i];
    }
    GSCAN_HD inline void set(uint32_t i, uint32_t x)
    {
        if (NREG > 0) {
#pragma unroll
            for (int k = 0; k < NREG; k++) r[k] = i == (uint32_t)k ? x : r[k];
        } else {
            r[i] = x;
        }
    }
};

GSCAN_HD inline uint32_t vm_in_class(const VmProg *pg, uint32_t cls, uint32_t b) { return (pg->cls[cls][b >> 5] >> (b & 31u)) & 1u; }

// What vm_run found when it answers 1: the match's end (ovector[1]) and whether its path closed a capturing group.
struct VmOut {
    uint32_t end, cap;
};

// 0: no match starts at p;  1: a match starts at p;  2: gave up.  c[0..clen) is the chunk (segment), s0 its subject start.
template <int NREG>
GSCAN_HD inline int vm_run_t(const VmProg *pg, const uint8_t *text, uint32_t clen, uint32_t p, uint32_t s0, VmOut &out)
{
    VmText c(text, clen);
    VmSlots<NREG> slots;
    uint32_t stk[2 * kVmStack];
    slots.init(pg->n_slots);
    uint32_t sp = 0, pc = 0, pos = p, steps = 0;
    uint32_t live = 0; // choice points (and negative look-around barriers) on the stack: with none, a failure is final and
                       // nothing that is only there to be undone on backtracking needs to be pushed

#define VM_PUSH(w0_, w1_)                      \
    do {                                       \
        if (sp >= (uint32_t)kVmStack) return 2; \
        stk[2 * sp] = (w0_);                   \
        stk[2 * sp + 1] = (w1_);               \
        sp++;                                  \
    } while (0)
#define VM_TEST(cls_, b_) vm_in_class(pg, (cls_), (b_))

    for (;;) {
        bool fail = false;
        if (++steps > kVmMaxSteps) return 2;
        const VmIns in = pg->ins[pc];
        switch (in.op & 0xffu) {
        case V_SET:
            if (pos < clen && VM_TEST(in.a, c.at(pos))) pos++, pc++;
            else fail = true;
            break;
        case V_REPSET: {
            const uint32_t mode = (in.op >> 8) & 3u;
            uint32_t k = 0;
            while (k < in.c && pos + k < clen && VM_TEST(in.a, c.at(pos + k))) k++;
            steps += k >> 3;
            if (k < in.b) {
                fail = true;
                break;
            }
            if (mode == 0) { // greedy: all of it, give back one at a time
                if (k > in.b) {
                    VM_PUSH(pos + in.b, pos + k);
                    VM_PUSH(VK_RANGE_DN | ((pc + 1) << 4), 0u);
                    live++;
                }
                pos += k;
            } else if (mode == 1) { // lazy: the minimum, take one more at a time
                if (k > in.b) {
                    VM_PUSH(pos + k, pos + in.b);
                    VM_PUSH(VK_RANGE_UP | ((pc + 1) << 4), 0u);
                    live++;
                }
                pos += in.b;
            } else {
                pos += k;
            }
            pc++;
            break;
        }
        case V_JMP: pc = in.a; break;
        case V_SPLIT: {
            const uint32_t ca = in.c & 0xffffu, cb = in.c >> 16;
            const uint32_t nb = pos < clen && (ca != 0xffffu || cb != 0xffffu) ? c.at(pos) : 0u; // the next byte, if anyone asks
            const bool oka = ca == 0xffffu || (pos < clen && VM_TEST(ca, nb));
            const bool okb = cb == 0xffffu || (pos < clen && VM_TEST(cb, nb));
            if (oka && okb) {
                VM_PUSH(VK_CHOICE | (in.b << 4), pos);
                live++;
                pc = in.a;
            } else if (oka) {
                pc = in.a;
            } else if (okb) {
                pc = in.b;
            } else {
                fail = true;
            }
            break;
        }
        case V_SAVE:
            if (live) VM_PUSH(VK_UNDO | (in.a << 4), slots.get(in.a));
            slots.set(in.a, pos);
            pc++;
            break;
        case V_CLOSE:
            if (live) {
                VM_PUSH(VK_UNDO | (in.a << 4), slots.get(in.a));
                VM_PUSH(VK_UNDO | ((in.a + 1) << 4), slots.get(in.a + 1));
            }
            slots.set(in.a, slots.get(in.b));
            slots.set(in.a + 1, pos);
            pc++;
            break;
        case V_REP_ENTER:
            if (live) VM_PUSH(VK_UNDO | (in.a << 4), slots.get(in.a));
            slots.set(in.a, 0);
            pc++;
            break;
        case V_REP_TOP: {
            const uint32_t mode = (in.op >> 8) & 3u, exit_pc = in.op >> 16, count = slots.get(in.a);
            const bool can_more = count < in.c, can_stop = count >= in.b;
            if (mode == 1) { // lazy: stop first
                if (can_stop) {
                    if (can_more) {
                        VM_PUSH(VK_CHOICE | ((pc + 1) << 4), pos);
                        live++;
                    }
                    pc = exit_pc;
                } else if (can_more) {
                    pc++;
                } else {
                    fail = true;
                }
            } else {
                if (can_more) {
                    if (can_stop) {
                        VM_PUSH(VK_CHOICE | (exit_pc << 4), pos);
                        live++;
                    }
                    pc++;
                } else if (can_stop) {
                    pc = exit_pc;
                } else {
                    fail = true;
                }
            }
            break;
        }
        case V_REP_END: {
            const uint32_t top = in.op >> 16, old = slots.get(in.a), count = old + 1;
            if (live) VM_PUSH(VK_UNDO | (in.a << 4), old);
            slots.set(in.a, count);
            // an iteration of the UNBOUNDED part that matched "" leaves the loop (PCRE's OP_KETRMAX rule; TreeMatch, Cont::REPG)
            if (pos == slots.get(in.b) && ((in.op >> 8) & 1u) && count >= in.c) pc = pg->ins[top].op >> 16;
            else pc = top;
            break;
        }
        case V_ASSERT:
            if (vm_holds(in.a, c, clen, s0, pos)) pc++;
            else fail = true;
            break;
        case V_ISSET: {
            const bool set = slots.get(in.a) != kVmUnset && slots.get(in.a + 1) != kVmUnset;
            if (set != (((in.op >> 8) & 1u) != 0)) pc++;
            else fail = true;
            break;
        }
        case V_BACKREF: {
            const uint32_t lo = slots.get(in.a), hi = slots.get(in.a + 1);
            if (lo == kVmUnset || hi == kVmUnset || hi < lo) { // a reference to a group that has not been set fails
                fail = true;
                break;
            }
            const uint32_t len = hi - lo;
            if (pos + len > clen) {
                fail = true;
                break;
            }
            steps += len >> 2;
            const bool icase = (in.op >> 8) & 1u;
            for (uint32_t q = 0; q < len; q++) {
                const uint32_t x = text[lo + q], y = text[pos + q];
                if (icase ? vm_lower(x) != vm_lower(y) : x != y) {
                    fail = true;
                    break;
                }
            }
            if (!fail) pos += len, pc++;
            break;
        }
        case V_BAR_BEGIN:
            VM_PUSH(VK_BAR | (((in.op >> 8) & 3u) << 4) | ((in.op >> 16) << 6), pos);
            if (((in.op >> 8) & 3u) == 2) live++; // a failure inside a negative look-around resumes behind it
            pc++;
            break;
        case V_BAR_END: {
            const uint32_t kind = (in.op >> 8) & 3u;
            if (kind == 2) { // the body of a NEGATIVE look-around matched: the assertion fails; what the body captured is undone
                for (;;) {
                    if (sp == 0) return 2;
                    sp--;
                    const uint32_t w0 = stk[2 * sp], k = w0 & 15u;
                    if (k == VK_UNDO) slots.set(w0 >> 4, stk[2 * sp + 1]);
                    else if (k == VK_CHOICE) live--;
                    else if (k == VK_RANGE_DN || k == VK_RANGE_UP) sp--, live--;
                    else if (k == VK_DEAD2) sp--;
                    else if (k == VK_BAR) {
                        live--; // (the negative barrier itself)
                        break;
                    }
                }
                fail = true;
                break;
            }
            // atomic group / positive look-around matched: no way back into it -- its choice points die, its undo records stay
            uint32_t i = sp;
            for (;;) {
                if (i == 0) return 2;
                i--;
                const uint32_t w0 = stk[2 * i], k = w0 & 15u;
                if (k == VK_CHOICE) {
                    stk[2 * i] = VK_DEAD;
                    live--;
                } else if (k == VK_RANGE_DN || k == VK_RANGE_UP) {
                    stk[2 * i] = VK_DEAD2;
                    live--;
                    i--;
                } else if (k == VK_DEAD2) {
                    i--;
                } else if (k == VK_BAR) {
                    if (kind == 1) pos = stk[2 * i + 1]; // an assertion consumes nothing
                    stk[2 * i] = VK_DEAD;
                    break;
                }
            }
            pc++;
            break;
        }
        case V_BACK:
            if (pos >= s0 + in.a) pos -= in.a, pc++;
            else fail = true;
            break;
        case V_MATCH: { // (an empty match AT p is "no match" for tree_match_at: the callers that use `end` ask the host about it)
            out.end = pos;
            uint32_t cap = 0;
            for (uint32_t g = 1; g <= pg->n_groups; g++) cap |= slots.get(2 * g + 1) != kVmUnset ? 1u : 0u;
            out.cap = cap;
            return 1;
        }
        default: // V_FAIL and anything unknown
            fail = true;
            break;
        }
        while (fail) { // backtrack
            if (sp == 0 || live == 0) return 0; // nothing left to try
            sp--;
            const uint32_t w0 = stk[2 * sp], w1 = stk[2 * sp + 1], k = w0 & 15u;
            if (k == VK_CHOICE) {
                pc = w0 >> 4;
                pos = w1;
                live--;
                fail = false;
            } else if (k == VK_UNDO) {
                slots.set(w0 >> 4, w1);
            } else if (k == VK_RANGE_DN) { // the entry below: {lo, cur}
                const uint32_t lo = stk[2 * (sp - 1)], cur = stk[2 * (sp - 1) + 1] - 1;
                pos = cur;
                pc = w0 >> 4;
                if (cur > lo) {
                    stk[2 * (sp - 1) + 1] = cur;
                    sp++; // the frame stays
                } else {
                    sp--;
                    live--;
                }
                fail = false;
            } else if (k == VK_RANGE_UP) { // {hi, cur}
                const uint32_t hi = stk[2 * (sp - 1)], cur = stk[2 * (sp - 1) + 1] + 1;
                pos = cur;
                pc = w0 >> 4;
                if (cur < hi) {
                    stk[2 * (sp - 1) + 1] = cur;
                    sp++;
                } else {
                    sp--;
                    live--;
                }
                fail = false;
            } else if (k == VK_BAR) {
                if (((w0 >> 4) & 3u) == 2) { // the body of a negative look-around found no match: the assertion holds
                    pos = w1;
                    pc = w0 >> 6;
                    live--;
                    fail = false;
                } // else: an atomic group / positive look-around that cannot match -- keep failing
            } else if (k == VK_DEAD2) {
                sp--;
            } // VK_DEAD: skip
            if (!fail && ++steps > kVmMaxSteps) return 2;
        }
    }
#undef VM_PUSH
#undef VM_TEST
}

GSCAN_HD inline int vm_run(const VmProg *pg, const uint8_t *c, uint32_t clen, uint32_t p, uint32_t s0, VmOut &out)
{
    return pg->n_slots <= 8u ? vm_run_t<8>(pg, c, clen, p, s0, out) : vm_run_t<0>(pg, c, clen, p, s0, out);
}
GSCAN_HD inline int vm_run(const VmProg *pg, const uint8_t *c, uint32_t clen, uint32_t p, uint32_t s0)
{
    VmOut out;
    return vm_run(pg, c, clen, p, s0, out);
}

} // namespace gscan
