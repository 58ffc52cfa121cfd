// vm_compile.cc -- the parse tree (pattern.h, Node) as a program for the backtracking VM of vm.h.
//
// Construct by construct this mirrors TreeMatch (matcher.cc), the host's authority on "the match AT an offset", so that
// the VM explores the same paths in the same order:
//   CAT            its items in sequence; a capturing group records its span when it CLOSES (start kept in a temporary slot)
//   ALT            SPLIT chain, left to right
//   REP of a class one V_REPSET (a loop over counts: longest first when greedy, shortest first when lazy, all of it when
//                  possessive)
//   REP of a group counter + iteration mark in slots; V_REP_TOP decides "one more or leave" the way rep_group does, V_REP_END
//                  applies the empty-iteration rule of the unbounded part; a possessive one is the greedy one inside an
//                  atomic bracket
//   LOOK / ATOMIC  a barrier on the backtrack stack: the body is matched to its first success, then its choice points are
//                  cut; a look-behind steps back by the fixed length of each of its top-level alternatives
//   BACKREF        compares with what the group captured (fails while the group is unset)
//   \K             nothing: where the match is reported to start is the host's business
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pattern.h"
#include "vm.h"

namespace gscan {

namespace {

struct Builder {
    VmProg &pg;
    bool ok = true;
    bool use_caps;
    int n_groups;
    uint32_t next_slot;

    Builder(VmProg &p, int groups, bool caps) : pg(p), use_caps(caps), n_groups(groups)
    {
        memset(&pg, 0, sizeof pg);
        // slots: [2g, 2g+1] what group g captured (g = 1 ..), [2 (n+1) + g] where it was entered
        next_slot = caps ? (uint32_t)(3 * (groups + 1)) : 0u;
        if (next_slot > (uint32_t)kVmMaxSlots) ok = false;
    }
    uint32_t cap_slot(int g) const { return 2u * (uint32_t)g; }
    uint32_t tmp_slot(int g) const { return 2u * (uint32_t)(n_groups + 1) + (uint32_t)g; }
    uint32_t new_slot()
    {
        if (next_slot >= (uint32_t)kVmMaxSlots) {
            ok = false;
            return 0;
        }
        return next_slot++;
    }
    uint32_t emit(uint32_t op, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0)
    {
        if (pg.n_ins >= (uint32_t)kVmMaxIns) {
            ok = false;
            return 0;
        }
        pg.ins[pg.n_ins] = {op, a, b, c};
        return pg.n_ins++;
    }
    uint32_t here() const { return pg.n_ins; }
    uint32_t cls_id(const ByteSet &s)
    {
        for (uint32_t i = 0; i < pg.n_cls; i++)
            if (!memcmp(pg.cls[i], s.w, sizeof s.w)) return i;
        if (pg.n_cls >= (uint32_t)kVmMaxCls) {
            ok = false;
            return 0;
        }
        memcpy(pg.cls[pg.n_cls], s.w, sizeof s.w);
        return pg.n_cls++;
    }

    // the fixed length of a look-behind alternative (TreeMatch::look_len)
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

    // What a match of the node can begin with: the set of first bytes, and whether it can match without consuming one (then
    // what follows decides and nothing can be predicted).
    struct First {
        ByteSet set;
        bool nullable = false;
    };
    static First first_of(const Node &n)
    {
        First f;
        switch (n.kind) {
        case Node::SET: f.set = n.set; break;
        case Node::CAT:
            f.nullable = true;
            for (const Node &k : n.kids) {
                const First g = first_of(k);
                f.set.merge(g.set);
                if (!g.nullable) {
                    f.nullable = false;
                    break;
                }
            }
            break;
        case Node::ALT:
            if (n.kids.empty()) f.nullable = true;
            for (const Node &k : n.kids) {
                const First g = first_of(k);
                f.set.merge(g.set);
                f.nullable = f.nullable || g.nullable;
            }
            break;
        case Node::REP: {
            if (n.max == 0) {
                f.nullable = true;
                break;
            }
            f = first_of(n.kids[0]);
            if (n.min == 0) f.nullable = true;
            break;
        }
        case Node::ATOMIC: f = first_of(n.kids[0]); break;
        case Node::ASSERT:
            f.nullable = true;
            break;
        case Node::LOOK:
            // a positive look-ahead consumes nothing but says what the NEXT byte has to be: for what these sets are used for -- a
            // split's first-byte prediction, "can giving back a byte of the repeat in front ever help" -- that is an answer
            // (\w+(?=\() : the repeat is compiled possessive).  Negative ones and look-behinds say nothing about the next byte.
            if (!n.behind && !n.neg) {
                const First g = first_of(n.kids[0]);
                if (!g.nullable) {
                    f.set = g.set;
                    break;
                }
            }
            f.nullable = true;
            break;
        case Node::COND:
        case Node::RECURSE:
        case Node::BACKREF: // what a reference repeats (a condition picks, a call matches) is not known here: any byte, or ""
            for (int k = 0; k < 8; k++) f.set.w[k] = 0xffffffffu;
            f.nullable = true;
            break;
        }
        return f;
    }
    // "nothing is known about what comes next" (the end of the pattern, of a look-around's or an atomic group's body)
    static First unknown()
    {
        First f;
        f.set.negate();
        f.nullable = true;
        return f;
    }
    // what can follow item i of a sequence: the first bytes of the items behind it, and of `follow` if they can all match ""
    static First behind(const std::vector<Node> &kids, size_t i, const First &follow)
    {
        First f;
        f.nullable = true;
        for (size_t j = i + 1; j < kids.size() && f.nullable; j++) {
            const First g = first_of(kids[j]);
            f.set.merge(g.set);
            f.nullable = g.nullable;
        }
        if (f.nullable) {
            f.set.merge(follow.set);
            f.nullable = follow.nullable;
        }
        return f;
    }
    static bool disjoint(const ByteSet &a, const ByteSet &b)
    {
        for (int k = 0; k < 8; k++)
            if (a.w[k] & b.w[k]) return false;
        return true;
    }

    void group_repeat(const Node &n, int mode, const First &follow)
    {
        const uint32_t cnt = new_slot(), mark = new_slot();
        emit(V_REP_ENTER, cnt);
        const uint32_t top = emit(V_REP_TOP | ((uint32_t)mode << 8), cnt, n.min, n.max);
        emit(V_SAVE, mark);
        // behind an iteration comes another one, or what follows the repeat
        First f = first_of(n.kids[0]);
        f.set.merge(follow.set);
        f.nullable = f.nullable || follow.nullable;
        gen(n.kids[0], f);
        emit(V_REP_END | ((n.max == kVmInf ? 1u : 0u) << 8) | (top << 16), cnt, mark, n.min);
        if (here() > 0xffffu) ok = false;
        if (ok) pg.ins[top].op |= here() << 16; // the exit
    }

    void behind_alt(const Node &alt)
    {
        const long len = look_len(alt);
        if (len < 0) {
            emit(V_FAIL);
            return;
        }
        if (len > 0) emit(V_BACK, (uint32_t)len);
        gen(alt, unknown());
    }

    // class id of what the alternatives kids[from..] can begin with, 0xffff if one of them may match ""
    uint32_t first_class(const std::vector<const Node *> &kids, size_t from, size_t to)
    {
        First f;
        for (size_t i = from; i < to; i++) {
            const First g = first_of(*kids[i]);
            if (g.nullable) return 0xffffu;
            f.set.merge(g.set);
        }
        const uint32_t id = cls_id(f.set);
        return ok ? id : 0xffffu;
    }

    void alternatives(const std::vector<const Node *> &kids, bool behind, const First &follow)
    {
        std::vector<uint32_t> jumps;
        for (size_t i = 0; i < kids.size(); i++) {
            uint32_t split = 0;
            const bool last = i + 1 == kids.size();
            // (look-behind alternatives step BACK first: the byte at pos says nothing about them)
            if (!last) {
                const uint32_t ca = behind ? 0xffffu : first_class(kids, i, i + 1), cb = behind ? 0xffffu : first_class(kids, i + 1, kids.size());
                split = emit(V_SPLIT, here() + 1, 0, ca | (cb << 16));
            }
            if (behind) behind_alt(*kids[i]);
            else gen(*kids[i], follow);
            if (!last) {
                jumps.push_back(emit(V_JMP, 0));
                if (ok) pg.ins[split].b = here();
            }
        }
        for (uint32_t j : jumps)
            if (ok) pg.ins[j].a = here();
    }

    // follow: what can come right behind the node -- the bytes the rest of the pattern can begin with, nullable if the match may
    // end there (or nothing is known).  It serves one decision: a greedy repeat of a class that what follows can neither begin
    // with nor do without (\w* in front of \s*\( , [^()]* in front of \) , \d+ in front of \.) is compiled as a POSSESSIVE one --
    // giving a byte back puts a byte of the class in front of something that cannot begin with it, so no such path ever
    // matches, and a failing candidate costs a handful of steps instead of three per byte of its run (pcre_compile does the same:
    // auto-possessification).  k_resolve runs the program at every candidate: that is most of what the pass costs.
    void gen(const Node &n, const First &follow)
    {
        if (!ok) return;
        switch (n.kind) {
        case Node::SET: emit(V_SET, cls_id(n.set)); break;
        case Node::CAT: {
            const bool cap = use_caps && n.cap && n.group > 0 && n.group <= n_groups;
            if (cap) emit(V_SAVE, tmp_slot(n.group));
            for (size_t i = 0; i < n.kids.size(); i++) gen(n.kids[i], behind(n.kids, i, follow));
            if (cap) emit(V_CLOSE, cap_slot(n.group), tmp_slot(n.group));
            break;
        }
        case Node::ALT: {
            std::vector<const Node *> kids;
            for (const Node &k : n.kids) kids.push_back(&k);
            if (kids.empty()) break;
            alternatives(kids, false, follow);
            break;
        }
        case Node::REP: {
            if (n.max == 0) break;
            const Node &kid = n.kids[0];
            if (kid.kind == Node::SET) {
                int mode = n.mode;
                if (mode == 0 && n.max > n.min && !follow.nullable && disjoint(follow.set, kid.set) && !getenv("GSCAN_VM_NO_POSSESSIFY")) mode = 2;
                emit(V_REPSET | ((uint32_t)mode << 8), cls_id(kid.set), n.min, n.max);
                break;
            }
            if (n.mode == 2) { // possessive: the greedy repeat matched on its own, never re-entered
                const uint32_t b = emit(V_BAR_BEGIN | (0u << 8));
                group_repeat(n, 0, unknown());
                emit(V_BAR_END | (0u << 8));
                if (ok) pg.ins[b].op |= here() << 16;
                break;
            }
            group_repeat(n, n.mode, follow);
            break;
        }
        case Node::ASSERT:
            if (n.acode != A_KEEP) emit(V_ASSERT, (uint32_t)n.acode);
            break;
        case Node::LOOK: {
            const uint32_t kind = n.neg ? 2u : 1u;
            const uint32_t b = emit(V_BAR_BEGIN | (kind << 8));
            const Node &body = n.kids[0];
            if (!n.behind) {
                gen(body, unknown());
            } else if (body.kind == Node::ALT) {
                std::vector<const Node *> kids;
                for (const Node &k : body.kids) kids.push_back(&k);
                alternatives(kids, true, unknown());
            } else {
                behind_alt(body);
            }
            emit(V_BAR_END | (kind << 8));
            if (here() > 0xffffu) ok = false;
            if (ok) pg.ins[b].op |= here() << 16; // where a NEGATIVE assertion goes on when its body finds no match
            break;
        }
        case Node::ATOMIC: {
            const uint32_t b = emit(V_BAR_BEGIN | (0u << 8));
            gen(n.kids[0], unknown());
            emit(V_BAR_END | (0u << 8));
            if (ok) pg.ins[b].op |= here() << 16;
            break;
        }
        case Node::RECURSE: ok = false; break; // not a program for this VM: the pattern stays with the host matcher
        case Node::COND: {
            // (?(c)yes|no) as  (?: <c holds> yes | <c does not hold> no ): exactly one guard passes, so the second branch is never
            // a way out of a failed first one.  A group condition: V_ISSET; an assertion: the look-around itself and its
            // negation (what it captured stays captured whenever its body matched -- also for (?(?!..)..), see matcher.cc).
            // Inside a program for this VM there are no subroutine calls: (?(R)..) and (?(Rn)..) never hold, DEFINE never does.
            const size_t base = n.cond == Node::C_ASSERT ? 1 : 0;
            const Node *yes = &n.kids[base];
            const Node *no = n.kids.size() > base + 1 ? &n.kids[base + 1] : nullptr;
            if (n.cond == Node::C_DEFINE) break;
            if (n.cond == Node::C_IN_RECURSION || n.cond == Node::C_IN_RECURSION_OF) {
                if (no) gen(*no, follow);
                break;
            }
            if (n.cond == Node::C_GROUP && (!use_caps || n.group <= 0 || n.group > n_groups)) {
                ok = false;
                break;
            }
            auto guard = [&](bool holds) {
                if (n.cond == Node::C_GROUP) {
                    emit(V_ISSET | ((holds ? 0u : 1u) << 8), cap_slot(n.group));
                } else {
                    Node look = n.kids[0];
                    if (!holds) look.neg = !look.neg;
                    gen(look, unknown());
                }
            };
            const uint32_t split = emit(V_SPLIT, here() + 1, 0, 0xffffu | (0xffffu << 16));
            guard(true);
            gen(*yes, follow);
            const uint32_t jump = emit(V_JMP, 0);
            if (ok) pg.ins[split].b = here();
            guard(false);
            if (no) gen(*no, follow);
            if (ok) pg.ins[jump].a = here();
            break;
        }
        case Node::BACKREF:
            if (!use_caps || n.group <= 0 || n.group > n_groups) {
                emit(V_FAIL); // (a reference with no captures kept can only fail: TreeMatch, `if (!caps) return false`)
                break;
            }
            emit(V_BACKREF | ((n.icase ? 1u : 0u) << 8), cap_slot(n.group));
            break;
        }
    }
};

bool looks_behind(const Node &n)
{
    if (n.kind == Node::ASSERT && (n.acode == A_BOS || n.acode == A_MBOL || n.acode == A_WB || n.acode == A_NWB)) return true;
    if (n.kind == Node::LOOK && n.behind) return true;
    for (const Node &k : n.kids)
        if (looks_behind(k)) return true;
    return false;
}

} // namespace

// The tree as a VM program; false if it does not fit the VM's limits (the pattern then stays with the host matcher).
bool vm_compile(const Node &root, int n_groups, bool has_backref, VmProg &out)
{
    // (the groups' spans are recorded whenever there are groups, not only for back references and conditions: "did the match
    // set a capturing group" is part of the verdict -- the reference's ovector holds one pair, src/grab.cc:171,179)
    (void)has_backref;
    Builder b(out, n_groups, n_groups > 0);
    b.gen(root, Builder::unknown());
    b.emit(V_MATCH);
    out.n_slots = b.next_slot;
    out.n_groups = n_groups > 0 ? (uint32_t)n_groups : 0u;
    out.ok = b.ok ? 1u : 0u;
    return b.ok;
}

// May the device drop candidates on the VM's verdict?  Only if "a match starts at p" does not depend on the restart position.
bool vm_independent_of_subject_start(const Node &root) { return !looks_behind(root); }

} // namespace gscan
