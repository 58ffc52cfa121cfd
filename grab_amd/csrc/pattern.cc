// pattern.cc -- see pattern.h.  Host only; no HIP.
#include "pattern.h"
#include "ucp_latin1.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstring>
#include <functional>
#include <map>

#include "../../include/gscan.h"

namespace gscan {
namespace {

constexpr uint32_t kInf = UINT32_MAX;

ByteSet set_all()
{
    ByteSet s;
    s.negate();
    return s;
}
ByteSet set_and(ByteSet a, const ByteSet &b)
{
    for (int i = 0; i < 8; i++) a.w[i] &= b.w[i];
    return a;
}

// One path through the pattern: window classes + optional variable repeat at the end.
struct Seq {
    std::vector<ByteSet> win;
    bool has_tail = false;
    ByteSet tail;
    uint32_t tail_extra = 0;
    int tail_mode = 0;
    bool cap = false; // the path runs through a capturing group
    std::vector<std::pair<uint32_t, int>> asserts; // (window position the assertion stands in front of, A_* code)
    // the assertions at the two ends, resolved into one byte of context each (see AltSeq)
    ByteSet pre = set_all(), post = set_all();
    bool pre_start = true, post_end = true, post_final_nl = false;
    // one unbounded repeat in the middle:  pwin . gap{1,} . win   (win, tail, asserts, post then describe the part AFTER the gap)
    bool gapped = false;
    std::vector<ByteSet> pwin;
    ByteSet gap;
    int gap_mode = 0; // 0 greedy, 1 lazy
    int gap_id = 0;   // paths that share one instance of the repeat: PCRE tries them count-major (see matcher.cc)
    std::vector<int> p_asserts; // assertions in front of pwin (A_* codes)
    bool frozen = false;        // the unfolder stopped here: what follows in the pattern is NOT part of this path's windows.
                                // The path is then a necessary condition only ("inexact"): the host confirms candidates
                                // with the backtracking matcher (matcher.cc)
    bool inexact = false;       // frozen, or an assertion of the path was left to the matcher
    bool needs_cap = false;     // the path BEGINS with a back reference and has not closed any group before it: if that is
                                // still so when the whole pattern is unfolded, the reference is to an unset group and fails
    bool settled = false;       // assertions behind the tail were dropped because they hold wherever the greedy repeat stops:
                                // true only as long as NOTHING follows (with more pattern behind, PCRE backtracks into the repeat)
    bool empty() const { return win.empty() && !has_tail && !cap && asserts.empty() && !gapped && !frozen && !inexact; }
};

// C-locale character tables, as pcre_maketables() builds them without setlocale()
// (/root/reference/src/grab.cc:106).
ByteSet set_digit()
{
    ByteSet s;
    s.set_range('0', '9');
    return s;
}
ByteSet set_word()
{
    ByteSet s;
    s.set_range('0', '9');
    s.set_range('A', 'Z');
    s.set_range('a', 'z');
    s.set('_');
    return s;
}
ByteSet set_space() // PCRE >= 8.34: \s includes VT
{
    ByteSet s;
    s.set_range(9, 13);
    s.set(' ');
    return s;
}
ByteSet set_hspace()
{
    ByteSet s;
    s.set(9);
    s.set(' ');
    s.set(0xa0);
    return s;
}
ByteSet set_vspace()
{
    ByteSet s;
    s.set_range(10, 13);
    s.set(0x85);
    return s;
}
ByteSet set_not(ByteSet s)
{
    s.negate();
    return s;
}
ByteSet set_dot() // options 0: '.' is anything but LF
{
    ByteSet s;
    s.negate();
    s.w[0] &= ~(1u << '\n');
    return s;
}

bool posix_class(const std::string &name, ByteSet &out)
{
    ByteSet s;
    for (int c = 0; c < 128; c++) { // C locale: nothing above 127 is in any class
        bool in = false;
        if (name == "alpha") in = isalpha(c);
        else if (name == "lower") in = islower(c);
        else if (name == "upper") in = isupper(c);
        else if (name == "alnum") in = isalnum(c);
        else if (name == "ascii") in = true;
        else if (name == "blank") in = (c == ' ' || c == '\t');
        else if (name == "cntrl") in = iscntrl(c);
        else if (name == "digit") in = isdigit(c);
        else if (name == "graph") in = isgraph(c);
        else if (name == "print") in = isprint(c);
        else if (name == "punct") in = ispunct(c);
        else if (name == "space") in = isspace(c);
        else if (name == "word") in = isalnum(c) || c == '_';
        else if (name == "xdigit") in = isxdigit(c);
        else return false;
        if (in) s.set((unsigned)c);
    }
    out = s;
    return true;
}

struct Parser {
    const unsigned char *p;
    size_t n, i = 0;
    std::string why;
    int rc = 0; // 0 ok, 1 unsupported, -1 malformed
    // inline options in force (PCRE: a change made inside a group lasts to the end of that
    // group, and carries into the alternatives that follow it there)
    bool caseless = false, dotall = false, multiline = false, extended = false;
    bool ungreedy = false; // (?U): quantifiers are lazy unless followed by ?
    bool dupnames = false; // (?J): two groups may share a name -- taken as long as none do
    bool has_accept = false; // (*ACCEPT) somewhere: pcre_study gives no minimum length
    bool pcre_checked = false; // GSCAN_PCRE_CHECKED

    // (?x): white space and #-comments between the items of a pattern mean nothing (not inside [...] or \Q..\E)
    void skip_extended()
    {
        while (extended && !quoting && !eof()) {
            const int c = p[i];
            if (c == ' ' || (c >= 9 && c <= 13)) {
                i++;
            } else if (c == '#') {
                while (!eof() && p[i] != '\n') i++;
                if (!eof()) i++;
            } else {
                break;
            }
        }
    }
    bool quoting = false; // inside \Q...\E
    int depth = 0;

    bool fail(int code, const char *msg)
    {
        if (rc == 0) {
            rc = code;
            why = msg;
        }
        return false;
    }
    bool eof() const { return i >= n; }

    static int hexval(int c)
    {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }

    // (?i): every member of the set also matches in its other case (C locale: ASCII letters only)
    static void fold_case(ByteSet &s)
    {
        for (unsigned c = 'a'; c <= 'z'; c++) {
            const unsigned u = c - 32;
            if (s.test(c) || s.test(u)) {
                s.set(c);
                s.set(u);
            }
        }
    }

    // After a backslash (i points at the escape letter).  Yields a set.  in_class
    // changes the meaning of \b and of \1..\7.
    bool no_fold = false; // set by escape(): the set is a Unicode property, which (?i) leaves alone
    bool escape(ByteSet &out, bool in_class, bool &is_set_escape)
    {
        is_set_escape = false;
        no_fold = false;
        if (eof()) return fail(-1, "\\ at end of pattern");
        int c = p[i++];
        ByteSet s;
        switch (c) {
        case 'd': out = set_digit(); is_set_escape = true; return true;
        case 'D': out = set_not(set_digit()); is_set_escape = true; return true;
        case 'w': out = set_word(); is_set_escape = true; return true;
        case 'W': out = set_not(set_word()); is_set_escape = true; return true;
        case 's': out = set_space(); is_set_escape = true; return true;
        case 'S':
            out = set_not(set_space());
            is_set_escape = true;
            saw_bare_S = saw_bare_S || !in_class;
            return true;
        case 'h':
            out = set_hspace();
            is_set_escape = true;
            saw_bare_hv = saw_bare_hv || !in_class;
            return true;
        case 'H': out = set_not(set_hspace()); is_set_escape = true; return true;
        case 'v':
            out = set_vspace();
            is_set_escape = true;
            saw_bare_hv = saw_bare_hv || !in_class;
            return true;
        case 'V': out = set_not(set_vspace()); is_set_escape = true; return true;
        case 'R':
        case 'X':
        case 'B': // (outside a class these never get here: parse_cat deals with them)
            if (!in_class) return fail(1, "escape sequence outside the engine's subset");
            s.set((unsigned)c);
            break;
        case 'p':
        case 'P': {
            // \pL \p{Lu} \p{^Lu} \P{Latin}: without UTF every byte is the Latin-1 code point of the same value
            // (ucp_latin1.h).  (?i) does not touch them.
            if (eof()) return fail(-1, "malformed \\P or \\p sequence");
            bool neg = c == 'P';
            std::string name;
            if (p[i] == '{') {
                size_t j = i + 1;
                if (j < n && p[j] == '^') neg = !neg, j++;
                const size_t from = j;
                while (j < n && p[j] != '}') j++;
                if (j >= n) return fail(-1, "malformed \\P or \\p sequence");
                name.assign((const char *)p + from, j - from);
                i = j + 1;
            } else {
                name.assign(1, (char)p[i++]);
            }
            const UcpLatin1 *hit = nullptr;
            for (const UcpLatin1 &e : kUcpLatin1)
                if (name == e.name) hit = &e;
            if (!hit) return fail(1, "Unicode property or script the engine has no Latin-1 table for");
            for (int k = 0; k < 8; k++) out.w[k] = neg ? ~hit->w[k] : hit->w[k];
            is_set_escape = true;
            no_fold = true;
            return true;
        }
        case 'C': // one data unit: any byte, newline included (no UTF here)
            if (in_class) { // (libpcre: inside a class \C, like \R \X \B, is the letter)
                s.set('C');
                break;
            }
            out = set_all();
            is_set_escape = true;
            return true;
        case 'N':
            if (in_class) return fail(-1, "\\N in class");
            if (!eof() && p[i] == '{') { // \N{3} is three non-newlines; \N{name} / \N{U+41} is Perl's named character: pcre_compile's error 37
                uint32_t mn, mx;
                size_t end;
                if (!counted(mn, mx, end)) return fail(-1, "PCRE does not support \\L, \\l, \\N{name}, \\U, or \\u");
            }
            out = set_dot();
            is_set_escape = true;
            return true;
        case 'a': s.set(7); break;
        case 'e': s.set(27); break;
        case 'f': s.set(12); break;
        case 'n': s.set(10); break;
        case 'r': s.set(13); break;
        case 't': s.set(9); break;
        case 'b':
            if (!in_class) return fail(1, "\\b word boundary");
            s.set(8);
            break;
        case 'c': {
            if (eof()) return fail(-1, "\\c at end of pattern");
            int d = p[i++];
            if (d >= 128) return fail(-1, "\\c followed by non-ASCII");
            s.set((unsigned)(toupper(d) ^ 0x40));
            break;
        }
        case 'x': {
            unsigned v = 0;
            if (!eof() && p[i] == '{') {
                size_t j = i + 1;
                int digits = 0;
                while (j < n && hexval(p[j]) >= 0) {
                    v = v * 16 + (unsigned)hexval(p[j]);
                    if (v > 0xffffff) v = 0xffffff;
                    j++;
                    digits++;
                }
                if (j < n && p[j] == '}' && digits > 0) {
                    if (v > 255) return fail(-1, "\\x{} value too large without UTF");
                    i = j + 1;
                    s.set(v);
                    break;
                }
                return fail(-1, "non-hex character in \\x{} (closing brace missing?)"); // (libpcre >= 8.34: error 79; before that \x with no digits and a literal '{')
            }
            int k = 0;
            while (k < 2 && !eof() && hexval(p[i]) >= 0) {
                v = v * 16 + (unsigned)hexval(p[i]);
                i++;
                k++;
            }
            s.set(v);
            break;
        }
        case 'o': {
            if (eof() || p[i] != '{') return fail(-1, "missing opening brace after \\o");
            size_t j = i + 1;
            unsigned v = 0;
            int digits = 0;
            while (j < n && p[j] >= '0' && p[j] <= '7') {
                v = v * 8 + (unsigned)(p[j] - '0');
                if (v > 0xffffff) v = 0xffffff;
                j++;
                digits++;
            }
            if (j >= n || p[j] != '}' || digits == 0) return fail(-1, "non-octal character in \\o{} (closing brace missing?)");
            if (v > 255) return fail(-1, "octal value is greater than \\377 in 8-bit non-UTF-8 mode");
            i = j + 1;
            s.set(v);
            break;
        }
        case '0': {
            unsigned v = 0;
            int k = 0;
            while (k < 2 && !eof() && p[i] >= '0' && p[i] <= '7') {
                v = v * 8 + (unsigned)(p[i] - '0');
                i++;
                k++;
            }
            s.set(v & 255);
            break;
        }
        case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9':
            // (outside a class this is reached for \NN with NN >= 10 and fewer than NN groups in the pattern -- parse_backref has
            // passed -- and pcre_compile then reads an octal escape of up to three digits, as inside a class)
            if (c >= '8') { // \8 \9 that are no back references: the digit itself
                s.set((unsigned)c);
                break;
            }
            {
                unsigned v = (unsigned)(c - '0');
                int k = 1;
                while (k < 3 && !eof() && p[i] >= '0' && p[i] <= '7') {
                    v = v * 8 + (unsigned)(p[i] - '0');
                    i++;
                    k++;
                }
                if (v > 255) return fail(-1, "octal value is greater than \\377 in 8-bit non-UTF-8 mode");
                s.set(v);
            }
            break;
        default:
            if (isalnum(c)) return fail(1, "escape sequence outside the engine's subset");
            s.set((unsigned)c); // escaped punctuation / high byte: the byte itself
        }
        out = s;
        return true;
    }

    // pcre_compile's check_posix_syntax: from a '[' followed by ':' '.' or '=' (j points at that character), is there the same
    // character followed by ']' before a ']' or another "[:"?
    bool posix_syntax(size_t j) const
    {
        const int term = p[j];
        for (j++; j < n; j++) {
            if (p[j] == '\\' && j + 1 < n && p[j + 1] == ']') j++;
            else if ((p[j] == '[' && j + 1 < n && p[j + 1] == term) || p[j] == ']') return false;
            else if (p[j] == term && j + 1 < n && p[j + 1] == ']') return true;
        }
        return false;
    }

    bool bracket(ByteSet &out) // i points just past '['
    {
        ByteSet s;
        bool neg = false;
        // "[:alpha:]" on its own: pcre_compile's error 12
        if (!eof() && (p[i] == ':' || p[i] == '.' || p[i] == '=') && posix_syntax(i)) return fail(-1, "POSIX named classes are supported only within a class");
        if (!eof() && p[i] == '^') {
            neg = true;
            i++;
        }
        bool first = true, in_quote = false;
        ByteSet props;
        for (;;) {
            if (eof()) return fail(-1, "missing terminating ] for character class");
            int c = p[i];
            ByteSet lo;
            bool lo_is_set = false;
            if (in_quote) { // \Q..\E inside a class: every byte up to \E is a member, ']' and '-' included
                if (c == '\\' && i + 1 < n && p[i + 1] == 'E') {
                    in_quote = false;
                    i += 2;
                    continue;
                }
                lo.set((unsigned)c);
                first = false;
                i++;
                while (i + 1 < n && p[i] == '\\' && p[i + 1] == 'E') { // the quote ends right behind this byte: it may begin a range
                    in_quote = false;
                    i += 2;
                }
                if (in_quote) {
                    s.merge(lo);
                    continue;
                }
                goto range_check;
            }
            if (c == '\\' && i + 1 < n && (p[i + 1] == 'Q' || p[i + 1] == 'E')) { // (a stray \E means nothing)
                in_quote = p[i + 1] == 'Q';
                i += 2;
                continue;
            }
            if (c == ']' && !first) {
                i++;
                break;
            }
            first = false;
            if (c == '[' && i + 1 < n && (p[i + 1] == ':' || p[i + 1] == '.' || p[i + 1] == '=')) {
                int kind = p[i + 1];
                size_t j = i + 2;
                while (j + 1 < n && !(p[j] == kind && p[j + 1] == ']')) j++;
                if (j + 1 < n) {
                    if (kind != ':') return fail(1, "POSIX collating element");
                    std::string name((const char *)p + i + 2, j - (i + 2));
                    bool pneg = false;
                    if (!name.empty() && name[0] == '^') {
                        pneg = true;
                        name.erase(0, 1);
                    }
                    ByteSet ps;
                    if (!posix_class(name, ps)) return fail(-1, "unknown POSIX class name");
                    // (pcre_compile: "if matching is caseless, upper and lower are converted to alpha" -- before the negation)
                    if (caseless && (name == "upper" || name == "lower")) posix_class("alpha", ps);
                    if (pneg) ps.negate();
                    s.merge(ps);
                    i = j + 2;
                    continue;
                }
                // no terminator: '[' is a plain member
            }
            if (c == '\\') {
                i++;
                if (!escape(lo, true, lo_is_set)) return false;
                if (no_fold) { // a Unicode property: (?i) does not fold it
                    props.merge(lo);
                    continue;
                }
            } else {
                lo.set((unsigned)c);
                i++;
            }
        range_check:
            if (!lo_is_set && i + 1 < n && p[i] == '-' && p[i + 1] != ']') {
                i++;
                ByteSet hi;
                bool hi_is_set = false;
                if (p[i] == '\\' && i + 1 < n && (p[i + 1] == 'Q' || p[i + 1] == 'E')) {
                    return fail(1, "\\Q or \\E as the end of a class range");
                } else if (p[i] == '\\') {
                    i++;
                    if (!escape(hi, true, hi_is_set)) return false;
                } else if (p[i] == '[' && i + 1 < n && (p[i + 1] == ':' || p[i + 1] == '.' || p[i + 1] == '=') && posix_syntax(i + 1)) {
                    return fail(-1, "invalid range in character class"); // "a-[:digit:]" (libpcre >= 8.34: error 83)
                } else {
                    hi.set(p[i]);
                    i++;
                }
                if (hi_is_set) return fail(-1, "invalid range in character class"); // [a-\d] (libpcre >= 8.34: error 83; before that the '-' was a literal)
                int l = lo.single(), h = hi.single();
                if (h < l) return fail(-1, "range out of order in character class");
                s.set_range((unsigned)l, (unsigned)h);
                continue;
            }
            s.merge(lo);
        }
        if (caseless) fold_case(s); // before negation: (?i)[^a] excludes 'A' too
        s.merge(props);
        if (neg) s.negate();
        out = s;
        return true;
    }

    // "{n}", "{n,}", "{n,m}" at i (pointing at '{')?  PCRE treats anything else as a literal '{'.
    bool counted(uint32_t &mn, uint32_t &mx, size_t &end)
    {
        size_t j = i + 1;
        if (j >= n || !isdigit(p[j])) return false;
        uint64_t a = 0;
        while (j < n && isdigit(p[j])) {
            a = a * 10 + (uint64_t)(p[j] - '0');
            if (a > 70000) a = 70000;
            j++;
        }
        uint64_t b = a;
        if (j < n && p[j] == ',') {
            j++;
            if (j < n && p[j] == '}') {
                b = kInf;
            } else {
                if (j >= n || !isdigit(p[j])) return false;
                b = 0;
                while (j < n && isdigit(p[j])) {
                    b = b * 10 + (uint64_t)(p[j] - '0');
                    if (b > 70000) b = 70000;
                    j++;
                }
            }
        }
        if (j >= n || p[j] != '}') return false;
        mn = (uint32_t)a;
        mx = (uint32_t)b;
        end = j + 1;
        return true;
    }

    // "(?" just consumed.  Either an option setting "(?i)" (returns with is_group = false), or the
    // opening of a non-capturing group "(?:" / "(?i:" (is_group = true; options already applied,
    // the caller restores them at the closing parenthesis).
    int look_depth = 0; // nesting depth of look-arounds at the current position
    int special = 0; // set by group_head: 1 (?=  2 (?!  3 (?<=  4 (?<!  5 (?>  6 (?P=name) (a back reference, not a group)
                     // 7 (?C) callout  8 (?( conditional  9 subroutine call (rec_num / rec_name)  10 (?| branch reset
    int reset_depth = 0; // inside this many branch-reset groups
    // libpcre's auto-possessification table calls \S disjoint from \h and from \v; in the C locale it is not (0xa0 is \h
    // and 0x85 is \v, neither is isspace()): \S+\h and \v*\S keep bytes they would have to give back.  A pattern that
    // uses both kinds of escape outside classes is refused.
    bool saw_bare_S = false, saw_bare_hv = false;
    int ngroups = 0; // capturing groups opened so far
    // capturing groups whose closing parenthesis has not been seen yet, with "a back reference inside it refers to it":
    // pcre_compile wraps such a group in atomic brackets (the value \N repeats must not change under it), and so does
    // parse_cat
    std::vector<std::pair<int, bool>> open_groups;
    void note_reference(int num)
    {
        for (auto &og : open_groups)
            if (og.first == num) og.second = true;
    }
    bool has_backref = false;
    std::vector<std::pair<std::string, int>> names; // named groups
    std::string group_name, ref_name;              // set by group_head
    int rec_num = 0;                               // group_head, special == 9: the group a subroutine call names (0: the pattern) ...
    std::string rec_name;                          // ... or its name
    bool has_recursion = false;

    int named_group(const std::string &nm) const
    {
        for (const auto &kv : names)
            if (kv.first == nm) return kv.second;
        return 0;
    }

    // "\" consumed, i at the character behind it.  A back reference?  (\1-\9, \10.. when that many groups are open,
    // \g1 \g{1} \g{-1} \g{name}, \k<name> \k'name' \k{name}.)  Returns 0 no, 1 yes (node filled), -1 error.
    int bad(int code, const char *msg)
    {
        fail(code, msg);
        return -1;
    }
    int backref(Node &a)
    {
        if (eof()) return 0;
        const int c = p[i];
        auto name_until = [&](size_t from, int close, std::string &out) -> size_t { // index behind the close, 0 if malformed
            size_t j = from;
            while (j < n && (isalnum(p[j]) || p[j] == '_')) j++;
            if (j == from || j >= n || p[j] != close) return 0;
            out.assign((const char *)p + from, j - from);
            return j + 1;
        };
        int num = 0;
        std::string nm;
        if (c >= '1' && c <= '9') {
            size_t j = i;
            long v = 0;
            while (j < n && isdigit(p[j]) && v < 100000) v = v * 10 + (p[j++] - '0');
            if (v >= 8 && v > ngroups) return 0; // (pcre_compile: a back reference if the number is < 8 or there are that many groups) an octal escape, or a literal 8 / 9: escape() deals with it
            num = (int)v;
            i = j;
        } else if (c == 'g') {
            size_t j = i + 1;
            if (j < n && (p[j] == '<' || p[j] == '\'')) { // \g<n> \g<+n> \g<-n> \g<name>, or quotes: Oniguruma's subroutine call
                const int close = p[j] == '<' ? '>' : '\'';
                size_t q = j + 1;
                int sign = 0;
                if (q < n && (p[q] == '+' || p[q] == '-')) sign = p[q++] == '+' ? 1 : -1;
                a = Node();
                a.kind = Node::RECURSE;
                if (q < n && isdigit(p[q])) {
                    long v = 0;
                    while (q < n && isdigit(p[q]) && v < 100000) v = v * 10 + (p[q++] - '0');
                    if (q >= n || p[q] != close) return bad(-1, "\\g is not followed by a braced, angle-bracketed, or quoted name/number or by a plain number");
                    if (sign && v == 0) return bad(-1, "a numbered reference must not be zero");
                    if (sign > 0) v = ngroups + v;
                    else if (sign < 0) v = ngroups - v + 1;
                    if (v < 0 || (sign < 0 && v <= 0)) return bad(-1, "reference to non-existent subpattern");
                    a.group = (int)v;
                    i = q + 1;
                } else if (!sign) {
                    std::string nm;
                    const size_t e = name_until(q, close, nm);
                    if (!e) return bad(-1, "\\g is not followed by a braced, angle-bracketed, or quoted name/number or by a plain number");
                    a.group = -1;
                    a.refname = nm;
                    i = e;
                } else {
                    return bad(-1, "\\g is not followed by a braced, angle-bracketed, or quoted name/number or by a plain number");
                }
                has_recursion = true;
                return 1;
            }
            const bool braced = j < n && p[j] == '{';
            if (braced) j++;
            bool neg = false;
            if (j < n && p[j] == '-') neg = true, j++;
            if (j < n && isdigit(p[j])) {
                long v = 0;
                while (j < n && isdigit(p[j]) && v < 100000) v = v * 10 + (p[j++] - '0');
                if (braced) {
                    if (j >= n || p[j] != '}') return bad(-1, "\\g is not followed by a braced, angle-bracketed, or quoted name/number or by a plain number");
                    j++;
                }
                if (v == 0) return bad(-1, "a numbered reference must not be zero");
                num = neg ? ngroups - (int)v + 1 : (int)v;
                if (num <= 0) return bad(-1, "reference to non-existent subpattern");
                i = j;
            } else if (braced && !neg) {
                const size_t e = name_until(j, '}', nm);
                if (!e) return bad(-1, "\\g is not followed by a braced, angle-bracketed, or quoted name/number or by a plain number");
                i = e;
            } else {
                return bad(-1, "\\g is not followed by a braced, angle-bracketed, or quoted name/number or by a plain number");
            }
        } else if (c == 'k') {
            const size_t j = i + 1;
            if (j >= n || (p[j] != '<' && p[j] != '\'' && p[j] != '{')) return bad(-1, "\\k is not followed by a braced, angle-bracketed, or quoted name");
            const size_t e = name_until(j + 1, p[j] == '<' ? '>' : p[j] == '{' ? '}' : '\'', nm);
            if (!e) return bad(-1, "\\k is not followed by a braced, angle-bracketed, or quoted name");
            i = e;
        } else {
            return 0;
        }
        a = Node();
        a.kind = Node::BACKREF;
        a.group = num;
        a.refname = nm;
        a.icase = caseless;
        has_backref = true;
        note_reference(nm.empty() ? num : named_group(nm));
        return 1;
    }

    // names -> numbers, and every reference must name a group of the pattern (forward references are fine)
    bool resolve_refs(Node &nd)
    {
        if (nd.kind == Node::RECURSE || (nd.kind == Node::COND && (nd.cond == Node::C_GROUP || nd.cond == Node::C_IN_RECURSION_OF))) {
            if (!nd.refname.empty()) {
                nd.group = named_group(nd.refname);
                nd.refname.clear();
                if (!nd.group) return fail(-1, "reference to non-existent subpattern");
            }
            // ((?(Rn)..) is not checked by pcre_compile: without a group n the condition is never true)
            if ((nd.group > ngroups || nd.group < 0) && !(nd.kind == Node::COND && nd.cond == Node::C_IN_RECURSION_OF)) return fail(-1, "reference to non-existent subpattern");
        }
        if (nd.kind == Node::BACKREF) {
            if (!nd.refname.empty()) {
                nd.group = named_group(nd.refname);
                nd.refname.clear();
                if (!nd.group) return fail(-1, "reference to non-existent subpattern");
            }
            if (nd.group > ngroups) return fail(-1, "reference to non-existent subpattern");
        }
        for (Node &k : nd.kids)
            if (!resolve_refs(k)) return false;
        return true;
    }

    // the length of everything the node can match, -1 if it varies (look-behind bodies must not: PCRE's rule -- the
    // top-level alternatives of the body may differ from each other, nested ones may not)
    static long fixed_len(const Node &nd)
    {
        switch (nd.kind) {
        case Node::SET: return 1;
        case Node::ASSERT:
        case Node::LOOK: return 0;
        case Node::BACKREF:
        case Node::COND:
        case Node::RECURSE: return -1;
        case Node::ATOMIC: return fixed_len(nd.kids[0]);
        case Node::CAT: {
            long t = 0;
            for (const Node &k : nd.kids) {
                const long l = fixed_len(k);
                if (l < 0) return -1;
                t += l;
            }
            return t;
        }
        case Node::ALT: {
            long t = -2;
            for (const Node &k : nd.kids) {
                const long l = fixed_len(k);
                if (l < 0 || (t != -2 && l != t)) return -1;
                t = l;
            }
            return t;
        }
        case Node::REP: {
            if (nd.min != nd.max) return -1;
            const long l = fixed_len(nd.kids[0]);
            return l < 0 ? -1 : l * (long)nd.min;
        }
        }
        return -1;
    }

    bool group_head(bool &is_group, bool &named)
    {
        is_group = false;
        named = false;
        special = 0;
        if (eof()) return fail(-1, "unrecognized character after (?");
        int c = p[i];
        if (c == '#') { // comment
            while (!eof() && p[i] != ')') i++;
            if (eof()) return fail(-1, "missing ) after comment");
            i++;
            return true;
        }
        if (c == ':') {
            i++;
            is_group = true;
            return true;
        }
        special = 0;
        if (c == '=' || c == '!') { // look-ahead
            special = c == '=' ? 1 : 2;
            i++;
            is_group = true;
            return true;
        }
        if (c == '<' && i + 1 < n && (p[i + 1] == '=' || p[i + 1] == '!')) { // look-behind
            special = p[i + 1] == '=' ? 3 : 4;
            i += 2;
            is_group = true;
            return true;
        }
        if (c == '<' || c == '\'' || (c == 'P' && i + 1 < n && p[i + 1] == '<')) { // (?<name>  (?'name'  (?P<name> : capturing
            if (c == 'P') i++;
            const int close = p[i] == '<' ? '>' : '\'';
            size_t j = i + 1;
            if (j < n && isdigit(p[j])) return fail(-1, "group name must not start with a digit");
            while (j < n && (isalnum(p[j]) || p[j] == '_')) j++;
            if (j == i + 1 || j >= n || p[j] != close) return fail(-1, "syntax error in group name");
            group_name.assign((const char *)p + i + 1, j - i - 1);
            if (named_group(group_name)) {
                if (dupnames) return fail(1, "two groups with the same name ((?J))");
                if (reset_depth > 0) return fail(1, "two groups with the same name (allowed inside (?| when their numbers agree)");
                return fail(-1, "two named subpatterns have the same name");
            }
            i = j + 1;
            is_group = true;
            named = true;
            return true;
        }
        if (c == '>') {
            special = 5;
            i++;
            is_group = true;
            return true;
        }
        if (c == '|') { // (?|..|..): every alternative numbers its groups from the same start
            i++;
            special = 10;
            is_group = true;
            return true;
        }
        if (c == 'P' && i + 1 < n && p[i + 1] == '=') { // (?P=name): a back reference in group clothing
            size_t j = i + 2;
            while (j < n && (isalnum(p[j]) || p[j] == '_')) j++;
            if (j == i + 2 || j >= n || p[j] != ')') return fail(-1, "syntax error in subpattern name (missing terminator)");
            ref_name.assign((const char *)p + i + 2, j - i - 2);
            i = j + 1;
            special = 6;
            return true;
        }
        if (c == '(') { // (?(condition)yes|no): parse_cond takes over, i stays at the condition's parenthesis
            special = 8;
            is_group = true;
            return true;
        }
        // subroutine calls: (?R) (?0) (?N) (?+N) (?-N) (?&name) (?P>name)
        if (c == 'R' || (c >= '0' && c <= '9') || ((c == '+' || c == '-') && i + 1 < n && isdigit(p[i + 1]))) {
            size_t j = i;
            long v = 0;
            if (c == 'R') {
                j++;
            } else {
                const int sign = c == '+' ? 1 : c == '-' ? -1 : 0;
                if (sign) j++;
                while (j < n && isdigit(p[j]) && v < 100000) v = v * 10 + (p[j++] - '0');
                if (sign && v == 0) return fail(-1, "a numbered reference must not be zero");
                if (sign > 0) v = ngroups + v;
                else if (sign < 0) v = ngroups - v + 1;
                if (v < 0 || (sign < 0 && v <= 0)) return fail(-1, "reference to non-existent subpattern");
            }
            if (j >= n || p[j] != ')') return fail(-1, c == 'R' ? "(?R or (?[+-]digits must be followed by )" : "(?R or (?[+-]digits must be followed by )");
            i = j + 1;
            rec_num = (int)v;
            rec_name.clear();
            special = 9;
            return true;
        }
        if (c == '&' || (c == 'P' && i + 1 < n && p[i + 1] == '>')) {
            size_t j = i + (c == '&' ? 1 : 2);
            const size_t from = j;
            while (j < n && (isalnum(p[j]) || p[j] == '_')) j++;
            if (j == from || j >= n || p[j] != ')') return fail(-1, "syntax error in subpattern name (missing terminator)");
            rec_name.assign((const char *)p + from, j - from);
            rec_num = -1;
            i = j + 1;
            special = 9;
            return true;
        }
        if (c == 'P') return fail(-1, "unrecognized character after (?P");
        if (c == 'C') { // (?C) (?Cn): a callout point; the reference sets no callout function, so nothing happens there
            size_t j = i + 1;
            long v = 0;
            while (j < n && isdigit(p[j]) && v < 1000) v = v * 10 + (p[j++] - '0');
            if (j >= n || p[j] != ')') return fail(-1, "closing ) for (?C expected");
            if (v > 255) return fail(-1, "number after (?C is > 255");
            i = j + 1;
            special = 7;
            return true;
        }
        bool on = true, ci = caseless, da = dotall, ml = multiline, ex = extended, ug = ungreedy;
        for (;; i++) {
            if (eof()) return fail(-1, "missing ) after option setting");
            c = p[i];
            if (c == '-') {
                if (!on) return fail(-1, "unrecognized character after (?");
                on = false;
            } else if (c == 'i') {
                ci = on;
            } else if (c == 's') {
                da = on;
            } else if (c == 'm') {
                ml = on;
            } else if (c == 'x') {
                ex = on;
            } else if (c == 'U') {
                ug = on;
            } else if (c == 'J') {
                dupnames = dupnames || on; // (nothing to do: duplicate names are still refused where they occur)
            } else if (c == 'X') {
                // PCRE_EXTRA: unknown escapes are errors instead of literals -- libpcre has validated the pattern text already
            } else if (c == ')' || c == ':') {
                break;
            } else {
                return fail(-1, "unrecognized character after (?");
            }
        }
        caseless = ci;
        dotall = da;
        multiline = ml;
        extended = ex;
        ungreedy = ug;
        is_group = (p[i] == ':');
        i++;
        return true;
    }

    static bool has_kind_p(const Node &nd, Node::Kind kind)
    {
        if (nd.kind == kind) return true;
        for (const Node &k : nd.kids)
            if (has_kind_p(k, kind)) return true;
        return false;
    }

    // "(?" consumed, i at the parenthesis that opens the condition.  (?(1)..) (?(+1)..) (?(<name>)..) (?('name')..) (?(name)..)
    // (?(R)..) (?(R1)..) (?(R&name)..) (?(DEFINE)..) (?(?=..)..) (?(?!..)..) (?(?<=..)..) (?(?<!..)..); then yes [| no] and ")".
    bool parse_cond(Node &a)
    {
        a = Node();
        a.kind = Node::COND;
        i++; // the condition's "("
        if (eof()) return fail(-1, "malformed number or name after (?(");
        if (p[i] == '?') { // an assertion
            int sp = 0;
            if (i + 1 < n && p[i + 1] == '=') sp = 1, i += 2;
            else if (i + 1 < n && p[i + 1] == '!') sp = 2, i += 2;
            else if (i + 2 < n && p[i + 1] == '<' && p[i + 2] == '=') sp = 3, i += 3;
            else if (i + 2 < n && p[i + 1] == '<' && p[i + 2] == '!') sp = 4, i += 3;
            else return fail(-1, "assertion expected after (?(");
            const bool ci = caseless, da = dotall, ml = multiline, ex = extended, ug = ungreedy;
            Node body;
            depth++;
            look_depth++;
            const bool ok = parse_alt(body);
            look_depth--;
            if (!ok) return false;
            depth--;
            if (eof() || p[i] != ')') return fail(-1, "missing )");
            i++;
            caseless = ci, dotall = da, multiline = ml, extended = ex, ungreedy = ug;
            if (sp >= 3) {
                if (has_kind_p(body, Node::COND) || has_kind_p(body, Node::RECURSE)) return fail(1, "conditional group or subroutine call inside a look-behind");
                bool fixed = true;
                if (body.kind == Node::ALT)
                    for (const Node &k : body.kids) fixed = fixed && fixed_len(k) >= 0;
                else
                    fixed = fixed_len(body) >= 0;
                if (!fixed) return fail(-1, "lookbehind assertion is not fixed length");
            }
            Node w;
            w.kind = Node::LOOK;
            w.behind = sp >= 3;
            w.neg = sp == 2 || sp == 4;
            w.kids.push_back(std::move(body));
            a.cond = Node::C_ASSERT;
            a.kids.push_back(std::move(w));
        } else {
            size_t j = i;
            while (j < n && p[j] != ')') j++;
            if (j >= n) return fail(-1, "missing )");
            std::string t((const char *)p + i, j - i);
            i = j + 1;
            auto all_digits = [](const std::string &x, size_t from) {
                if (from >= x.size()) return false;
                for (size_t k = from; k < x.size(); k++)
                    if (!isdigit((unsigned char)x[k])) return false;
                return true;
            };
            auto is_name = [](const std::string &x) {
                if (x.empty() || isdigit((unsigned char)x[0])) return false;
                for (char ch : x)
                    if (!isalnum((unsigned char)ch) && ch != '_') return false;
                return true;
            };
            if (all_digits(t, 0) || ((t[0] == '+' || t[0] == '-') && all_digits(t, 1))) {
                const int sign = t[0] == '+' ? 1 : t[0] == '-' ? -1 : 0;
                long v = atol(t.c_str() + (sign ? 1 : 0));
                if (v == 0) return fail(-1, sign ? "a numbered reference must not be zero" : "invalid condition (?(0)");
                if (sign > 0) v = ngroups + v;
                else if (sign < 0) v = ngroups - v + 1;
                if (v <= 0 || v > 100000) return fail(-1, "reference to non-existent subpattern");
                a.cond = Node::C_GROUP;
                a.group = (int)v;
            } else if (t == "R") {
                a.cond = Node::C_IN_RECURSION;
            } else if (t.size() > 1 && t[0] == 'R' && all_digits(t, 1)) {
                a.cond = Node::C_IN_RECURSION_OF;
                a.group = atoi(t.c_str() + 1);
            } else if (t.size() > 2 && t[0] == 'R' && t[1] == '&' && is_name(t.substr(2))) {
                a.cond = Node::C_IN_RECURSION_OF;
                a.group = -1;
                a.refname = t.substr(2);
            } else if (t == "DEFINE") {
                a.cond = Node::C_DEFINE;
            } else {
                std::string nm = t;
                if (nm.size() >= 2 && ((nm.front() == '<' && nm.back() == '>') || (nm.front() == '\'' && nm.back() == '\''))) nm = nm.substr(1, nm.size() - 2);
                if (!is_name(nm)) return fail(-1, "malformed number or name after (?(");
                a.cond = Node::C_GROUP;
                a.group = -1;
                a.refname = nm;
            }
            if (a.cond == Node::C_GROUP) has_backref = true; // (the matcher has to keep track of what the groups captured)
        }
        // the branches
        const bool ci = caseless, da = dotall, ml = multiline, ex = extended, ug = ungreedy;
        Node body;
        depth++;
        if (!parse_alt(body)) return false;
        depth--;
        if (eof() || p[i] != ')') return fail(-1, "missing )");
        i++;
        caseless = ci, dotall = da, multiline = ml, extended = ex, ungreedy = ug;
        if (body.kind == Node::ALT && !body.kids.empty()) {
            if (body.kids.size() > 2) return fail(-1, "conditional group contains more than two branches");
            if (a.cond == Node::C_DEFINE) return fail(-1, "DEFINE group contains more than one branch");
            for (Node &k : body.kids) a.kids.push_back(std::move(k));
        } else {
            a.kids.push_back(std::move(body));
        }
        return true;
    }

    // alternation := cat ('|' cat)*
    bool parse_alt(Node &out, bool reset_numbers = false)
    {
        Node alt;
        alt.kind = Node::ALT;
        const int base = ngroups;
        int highest = ngroups;
        for (;;) {
            Node cat;
            if (reset_numbers) ngroups = base;
            if (!parse_cat(cat)) return false;
            highest = std::max(highest, ngroups);
            if (reset_numbers) ngroups = highest;
            alt.kids.push_back(std::move(cat));
            if (!eof() && !quoting && p[i] == '|') {
                i++;
                continue;
            }
            break;
        }
        if (alt.kids.size() == 1) out = std::move(alt.kids[0]);
        else out = std::move(alt);
        return true;
    }

    // cat := piece*   (stops at '|', ')' or the end)
    bool parse_cat(Node &out)
    {
        out = Node();
        out.kind = Node::CAT;
        for (;;) {
            skip_extended();
            if (eof()) break;
            int c = p[i];
            Node a;
            bool look_quant = false;
            if (quoting) {
                if (c == '\\' && i + 1 < n && p[i + 1] == 'E') {
                    quoting = false;
                    i += 2;
                    continue;
                }
                a.set.set((unsigned)c);
                if (caseless) fold_case(a.set);
                i++;
                if (i + 1 < n && p[i] == '\\' && p[i + 1] == 'E') { // the quote ends here: a quantifier after \E applies to this char
                    quoting = false;
                    i += 2;
                }
            } else {
                if (c == '|') break;
                if (c == ')') {
                    if (depth == 0) return fail(-1, "unmatched parentheses");
                    break;
                }
                int acode = 0;
                if (c == '^') acode = multiline ? A_MBOL : A_BOS;
                else if (c == '$') acode = multiline ? A_MEOL : A_EOL;
                else if (c == '\\' && i + 1 < n) {
                    switch (p[i + 1]) {
                    case 'b': acode = A_WB; break;
                    case 'B': acode = A_NWB; break;
                    case 'A': case 'G': acode = A_BOS; break;
                    case 'Z': acode = A_EOL; break;
                    case 'z': acode = A_EOS; break;
                    case 'K':
                        if (look_depth > 0) return fail(1, "\\K inside a look-around");
                        acode = A_KEEP;
                        break;
                    }
                }
                if (acode) {
                    i += (c == '\\') ? 2 : 1;
                    skip_extended();
                    if (!eof() && (p[i] == '*' || p[i] == '+' || p[i] == '?')) return fail(-1, "nothing to repeat");
                    uint32_t mn, mx;
                    size_t end;
                    if (!eof() && p[i] == '{' && counted(mn, mx, end)) return fail(-1, "nothing to repeat");
                    a.kind = Node::ASSERT;
                    a.acode = acode;
                    out.kids.push_back(std::move(a));
                    continue;
                }
                switch (c) {
                case '\\':
                    if (i + 1 < n && p[i + 1] == 'Q') {
                        quoting = true;
                        i += 2;
                        continue;
                    }
                    if (i + 1 < n && p[i + 1] == 'E') { // stray \E is ignored by PCRE
                        i += 2;
                        continue;
                    }
                    {
                        const size_t save = i;
                        i++;
                        const int br = backref(a);
                        if (br < 0) return false;
                        if (br > 0) break;
                        i = save;
                    }
                    if (i + 1 < n && (p[i + 1] == 'R' || p[i + 1] == 'X')) {
                        // \R = (?>\r\n|\n|\x0b|\f|\r|\x85); \X, an extended grapheme cluster, is (?>\r\n|any byte) over Latin-1 (no
                        // byte below 256 extends another; CR LF is the only pair that stays together).  libpcre's
                        // auto-possessification treats \R as disjoint from \s and from "." -- \R?\s misses "\n", \N+\R never
                        // ends in \r, VT, FF or NEL -- so a variable count of it, or a greedy class repeat in front of it, is
                        // refused (compile_pattern: ends_in_greedy_repeat).
                        const bool any = p[i + 1] == 'X';
                        i += 2;
                        Node crlf, cr, lf, one;
                        crlf.kind = Node::CAT;
                        cr.set.set('\r');
                        lf.set.set('\n');
                        crlf.kids.push_back(cr);
                        crlf.kids.push_back(lf);
                        if (any) {
                            one.set = set_all();
                        } else {
                            one.set.set('\n'), one.set.set(0x0b), one.set.set('\f'), one.set.set('\r'), one.set.set(0x85);
                        }
                        Node alt;
                        alt.kind = Node::ALT;
                        alt.kids.push_back(std::move(crlf));
                        alt.kids.push_back(std::move(one));
                        a = Node();
                        a.kind = Node::ATOMIC;
                        a.newline_seq = true;
                        a.kids.push_back(std::move(alt));
                        break;
                    }
                    i++;
                    {
                        bool is_set;
                        if (!escape(a.set, false, is_set)) return false;
                        if (caseless && !no_fold) fold_case(a.set);
                    }
                    break;
                case '.':
                    a.set = set_dot();
                    if (dotall) a.set.set('\n');
                    i++;
                    break;
                case '[':
                    i++;
                    if (!bracket(a.set)) return false;
                    break;
                case '(': {
                    i++;
                    if (eof()) return fail(-1, "missing )");
                    if (p[i] == '*') {
                        // (*FAIL) / (*F): an empty negative look-ahead.  (*ACCEPT): pcre_study finds no minimum length
                        // ("ACCEPT makes things far too complicated"), PCRE_INFO_MINLENGTH is -1 and the reference skips every
                        // file -- nothing is ever matched, so the verb needs no meaning here.  The verbs that
                        // steer the search over start offsets (COMMIT, PRUNE, SKIP, THEN) and the (*UTF8) / newline settings
                        // are outside the subset.
                        size_t j = i + 1;
                        while (j < n && p[j] != ')') j++;
                        if (j >= n) return fail(-1, "missing )");
                        const std::string verb((const char *)p + i + 1, j - i - 1);
                        if (verb == "FAIL" || verb == "F" || verb == "ACCEPT") {
                            if (verb == "ACCEPT") has_accept = true;
                            a = Node();
                            a.kind = Node::LOOK;
                            a.neg = true;
                            Node none;
                            none.kind = Node::CAT;
                            a.kids.push_back(std::move(none));
                            i = j + 1;
                            {
                                uint32_t mn, mx;
                                size_t end;
                                if (!eof() && (p[i] == '*' || p[i] == '+' || p[i] == '?' || (p[i] == '{' && counted(mn, mx, end)))) return fail(-1, "nothing to repeat");
                            }
                            break;
                        }
                        return fail(1, "backtracking control verb / start-of-pattern setting");
                    }
                    const bool ci = caseless, da = dotall, ml = multiline, ex = extended, ug = ungreedy;
                    bool capture = true, named = false;
                    if (p[i] == '?') {
                        i++;
                        bool is_group;
                        if (!group_head(is_group, named)) return false;
                        if (special == 6) { // (?P=name)
                            special = 0;
                            a = Node();
                            a.kind = Node::BACKREF;
                            a.refname = ref_name;
                            a.icase = caseless;
                            has_backref = true;
                            note_reference(named_group(ref_name));
                            caseless = ci, dotall = da, multiline = ml;
                            break;
                        }
                        if (special == 9) { // a subroutine call
                            special = 0;
                            a = Node();
                            a.kind = Node::RECURSE;
                            a.group = rec_num;
                            a.refname = rec_name;
                            has_recursion = true;
                            caseless = ci, dotall = da, multiline = ml;
                            break;
                        }
                        if (special == 8) { // a conditional group
                            special = 0;
                            if (!parse_cond(a)) return false;
                            caseless = ci, dotall = da, multiline = ml, extended = ex, ungreedy = ug;
                            break;
                        }
                        if (special == 7) { // (?C): nothing
                            special = 0;
                            if (!eof() && (p[i] == '*' || p[i] == '+' || p[i] == '?')) return fail(-1, "nothing to repeat");
                            continue;
                        }
                        if (!is_group) { // "(?i)": stays in force to the end of the enclosing group
                            if (!eof() && (p[i] == '*' || p[i] == '+' || p[i] == '?')) return fail(-1, "nothing to repeat");
                            continue;
                        }
                        capture = named;
                    }
                    const int sp = special;
                    special = 0;
                    int gno = 0;
                    if (capture) { // groups are numbered by their opening parenthesis
                        gno = ++ngroups;
                        if (named) names.emplace_back(group_name, gno);
                        open_groups.emplace_back(gno, false);
                    }
                    depth++;
                    if (sp >= 1 && sp <= 4) look_depth++;
                    if (sp == 10) reset_depth++;
                    const bool body_ok = sp == 10 ? parse_alt(a, true) : parse_alt(a);
                    if (sp == 10) reset_depth--;
                    if (sp >= 1 && sp <= 4) look_depth--;
                    if (!body_ok) return false;
                    depth--;
                    if (eof() || p[i] != ')') return fail(-1, "missing )");
                    i++;
                    caseless = ci;
                    dotall = da;
                    multiline = ml;
                    extended = ex;
                    ungreedy = ug;
                    if (sp == 10) { // a plain group as far as matching goes
                        capture = false;
                    } else if (sp) { // look-around / atomic group: a wrapper node around the body
                        if ((sp == 3 || sp == 4) && (has_kind_p(a, Node::COND) || has_kind_p(a, Node::RECURSE)))
                            return fail(1, "conditional group or subroutine call inside a look-behind");
                        if (sp == 3 || sp == 4) {
                            bool fixed = true;
                            if (a.kind == Node::ALT)
                                for (const Node &k : a.kids) fixed = fixed && fixed_len(k) >= 0;
                            else
                                fixed = fixed_len(a) >= 0;
                            if (!fixed) return fail(-1, "lookbehind assertion is not fixed length");
                        }
                        Node w;
                        w.kind = sp == 5 ? Node::ATOMIC : Node::LOOK;
                        w.behind = sp == 3 || sp == 4;
                        w.neg = sp == 2 || sp == 4;
                        w.kids.push_back(std::move(a));
                        a = std::move(w);
                        look_quant = sp != 5; // a quantifier behind an assertion only says whether it may be skipped (below)
                        capture = false;
                    }
                    if (capture) { // wrap: the body keeps its own kind, the wrapper carries the flag
                        Node w;
                        w.kind = Node::CAT;
                        w.cap = true;
                        w.group = gno;
                        w.kids.push_back(std::move(a));
                        a = std::move(w);
                        const bool self_ref = open_groups.back().second;
                        open_groups.pop_back();
                        if (self_ref) { // see open_groups
                            Node at;
                            at.kind = Node::ATOMIC;
                            at.kids.push_back(std::move(a));
                            a = std::move(at);
                        }
                    }
                    break;
                }
                case '*':
                case '+':
                case '?': return fail(-1, "nothing to repeat");
                case '{': {
                    uint32_t mn, mx;
                    size_t end;
                    if (counted(mn, mx, end)) return fail(-1, "nothing to repeat");
                    a.set.set('{');
                    i++;
                    break;
                }
                default:
                    a.set.set((unsigned)c);
                    if (caseless) fold_case(a.set);
                    i++;
                }
            }
            // quantifier (a quoted \Q..\E char may be quantified once the quote ended; PCRE
            // applies a quantifier after \E to the last quoted char -- same thing here)
            skip_extended();
            while (!quoting && i + 1 < n && p[i] == '\\' && (p[i + 1] == 'E' || (p[i + 1] == 'Q' && i + 3 < n && p[i + 2] == '\\' && p[i + 3] == 'E'))) {
                // (a stray \E means nothing, and neither does an empty \Q\E: a quantifier behind them is this item's)
                i += p[i + 1] == 'E' ? 2 : 4;
                skip_extended();
            }
            if (!quoting && !eof()) {
                int q = p[i];
                bool have = false;
                uint32_t qmin = 1, qmax = 1;
                if (q == '*') { qmin = 0; qmax = kInf; i++; have = true; }
                else if (q == '+') { qmin = 1; qmax = kInf; i++; have = true; }
                else if (q == '?') { qmin = 0; qmax = 1; i++; have = true; }
                else if (q == '{') {
                    uint32_t mn, mx;
                    size_t end;
                    if (counted(mn, mx, end)) {
                        if (mx != kInf && mx < mn) return fail(-1, "numbers out of order in {} quantifier");
                        if (mn > 65535 || (mx != kInf && mx > 65535)) return fail(-1, "number too big in {} quantifier");
                        qmin = mn;
                        qmax = mx;
                        i = end;
                        have = true;
                    }
                }
                if (have) {
                    int mode = 0;
                    skip_extended(); // (?x): white space and comments may stand between a quantifier and its ? / + suffix
                    if (!eof() && p[i] == '?') { mode = 1; i++; }
                    else if (!eof() && p[i] == '+') { mode = 2; i++; }
                    if (ungreedy && mode != 2) mode ^= 1; // (?U): lazy by default, ? makes it greedy
                    if (look_quant) { // PCRE: {0} drops the assertion, a minimum of 0 makes it optional, anything else means once
                        if (qmax == 0) {
                            a = Node();
                            a.kind = Node::CAT;
                        }
                        qmax = qmax == 0 ? 0 : 1;
                        qmin = qmin == 0 ? 0 : 1;
                        if (mode == 2) mode = 0;
                    }
                    skip_extended();
                    if (!eof()) {
                        int r = p[i];
                        uint32_t mn, mx;
                        size_t end;
                        if (r == '*' || r == '+' || r == '?' || (r == '{' && counted(mn, mx, end))) return fail(-1, "nothing to repeat");
                    }
                    // (x)*+ : libpcre's JIT (the reference's timing build) leaves the group set when an attempt that went through
                    // it fails, so a later group-free match of the same pcre_exec call comes back as 0 -- x|(a)*+b on "a x".
                    // Only this form does ((a)++, ((a))*+, (?:(a))*+ do not); what it depends on -- every start offset
                    // pcre_exec tries, candidate or not -- is not something the engine looks at.  Refused.
                    if (a.kind == Node::COND && a.cond == Node::C_DEFINE) return fail(1, "a quantified (?(DEFINE)..) group (libpcre never finds a match behind one)");
                    if (a.kind == Node::ATOMIC && a.newline_seq && qmin != qmax) return fail(1, "a variable count of \\R or \\X (libpcre's auto-possessification misjudges what may follow it)");
                    if (mode == 2 && qmin == 0 && qmax == kInf && a.kind == Node::CAT && a.cap)
                        return fail(1, "possessive * directly on a capturing group (libpcre's JIT reports the group as set after failed attempts)");
                    Node rep;
                    rep.kind = Node::REP;
                    rep.min = qmin;
                    rep.max = qmax;
                    rep.mode = mode;
                    rep.kids.push_back(std::move(a));
                    a = std::move(rep);
                }
            }
            out.kids.push_back(std::move(a));
        }
        return true;
    }

    // pcre_compile's error 40, "recursive call could loop indefinitely": a call of group g (0: the pattern) from inside g that can
    // be reached from g's start without a byte being consumed -- (?R), ((?1)), (a|(?R)), ^(?R), (?=(?R))a, (x)(\\1(?2)).  Returns "the
    // node may match nothing" the way could_be_empty_branch sees it: assertions are empty, a back reference always may be, a call
    // may if the group it calls may, a condition if one of its branches (or the else it does not have) may.  `open`: everything in
    // front of the node, back to g's start, may.
    // A call of a group that is not complete where the call stands -- a forward reference, or a group the call sits in -- ends
    // could_be_empty_branch's scan of its branch with "may be empty", whatever follows: ((?2)b(?1))(a) is an error, (b(?2)(?1))(a)
    // is not.  Hence the nodes' positions (preorder numbers; a group's span is [its own, its last descendant's]).
    std::vector<const Node *> group_nodes; // [g]: the capturing wrapper of group g
    std::map<const Node *, std::pair<int, int>> span_;
    int collect_groups(const Node &nd, int at)
    {
        const int mine = at++;
        if (nd.kind == Node::CAT && nd.cap && nd.group > 0) {
            if ((size_t)nd.group >= group_nodes.size()) group_nodes.resize((size_t)nd.group + 1, nullptr);
            if (!group_nodes[(size_t)nd.group]) group_nodes[(size_t)nd.group] = &nd;
        }
        for (const Node &k : nd.kids) at = collect_groups(k, at);
        if (nd.kind == Node::RECURSE || (nd.kind == Node::CAT && nd.cap && nd.group > 0)) span_[&nd] = {mine, at - 1};
        return at;
    }
    bool incomplete_call(const Node &call) const // RECURSE: the called group is not complete at the call
    {
        if (call.group == 0) return true;
        if (call.group < 0 || (size_t)call.group >= group_nodes.size() || !group_nodes[(size_t)call.group]) return false;
        const auto c = span_.find(&call), g = span_.find(group_nodes[(size_t)call.group]);
        if (c == span_.end() || g == span_.end()) return false;
        return g->second.first > c->second.first || g->second.second >= c->second.first; // opens behind the call, or has not closed in front of it
    }
    bool scan_left(const Node &nd, int g, bool open, bool &loops, int depth = 0) const
    {
        switch (nd.kind) {
        case Node::SET: return false;
        case Node::BACKREF: return true;
        case Node::RECURSE: {
            if (open && nd.group == g) loops = true;
            if (incomplete_call(nd)) return true; // (the CAT it stands in makes that the whole branch's answer)
            if (nd.group == g || nd.group <= 0 || depth > 8 || (size_t)nd.group >= group_nodes.size() || !group_nodes[(size_t)nd.group]) return false;
            bool ignore = false; // (what the called group calls in its turn is its own business)
            return scan_left(*group_nodes[(size_t)nd.group], -1, false, ignore, depth + 1);
        }
        case Node::CAT: {
            bool e = true, wild = false;
            for (const Node &k : nd.kids) {
                const bool ke = scan_left(k, g, open && e, loops, depth);
                if (wild) continue; // (the scan of this branch has ended with "may be empty")
                e = e && ke;
                if (e && k.kind == Node::RECURSE && incomplete_call(k)) wild = true; // (a quantified call sits in a bracket of its own: ((?2)?b(?1))(a) is fine)
            }
            return e;
        }
        case Node::ALT: {
            bool any = nd.kids.empty();
            for (const Node &k : nd.kids) any = scan_left(k, g, open, loops, depth) || any;
            return any;
        }
        case Node::REP: {
            if (nd.max == 0 || nd.kids.empty()) return true;
            const bool ke = scan_left(nd.kids[0], g, open, loops, depth);
            return ke || nd.min == 0;
        }
        case Node::ATOMIC: return nd.kids.empty() ? true : scan_left(nd.kids[0], g, open, loops, depth);
        case Node::LOOK:
            // (observed with libpcre 8.39 and 8.45: a call in the SECOND or a later alternative of an assertion is not diagnosed --
            // (?!(?R))a is an error, (?!a|(?R))b is not; could_be_empty_branch skips an assertion alternative by alternative and an
            // assertion that is still open has no end to skip to)
            for (const Node &k : nd.kids) {
                if (k.kind == Node::ALT) {
                    for (size_t a = 0; a < k.kids.size(); a++) (void)scan_left(k.kids[a], g, open && a == 0, loops, depth);
                } else {
                    (void)scan_left(k, g, open, loops, depth);
                }
            }
            return true;
        case Node::ASSERT: return true;
        case Node::COND: {
            const size_t first = nd.cond == Node::C_ASSERT ? 1 : 0;
            bool any = nd.kids.size() - first < 2; // no else branch: it is empty
            // (pcre_compile does not make the check for a call inside a conditional group, whose condition may be there to stop the
            // recursion -- (?(R)a+|(?R)b) --: cond_depth)
            for (size_t k = 0; k < nd.kids.size(); k++) {
                const bool ke = scan_left(nd.kids[k], g, false, loops, depth);
                if (k >= first) any = any || ke;
            }
            return any;
        }
        default: return false;
        }
    }
    bool left_recursion(const Node &nd) const
    {
        if (nd.kind == Node::COND) return false; // (nothing inside a conditional group is checked: scan_left)
        // ((?|..) and (?J): a call by number goes to the FIRST group of that number -- only that one can be open around its own call)
        if (nd.kind == Node::CAT && nd.cap && nd.group > 0 && !nd.kids.empty() && (size_t)nd.group < group_nodes.size() && group_nodes[(size_t)nd.group] == &nd) {
            bool loops = false;
            (void)scan_left(nd.kids[0], nd.group, true, loops); // (the wrapper holds the body as its one kid)
            if (loops) return true;
        }
        for (const Node &k : nd.kids)
            if (left_recursion(k)) return true;
        return false;
    }

    bool parse(Node &root)
    {
        if (!parse_alt(root)) return false;
        if (!eof()) return fail(-1, "unmatched parentheses"); // a ')' at depth 0
        if (saw_bare_S && saw_bare_hv) return fail(1, "\\S next to \\h or \\v (libpcre's auto-possessification treats them as disjoint; 0xa0 and 0x85 are in both)");
        if (!resolve_refs(root)) return false;
        bool loops = false;
        (void)collect_groups(root, 0);
        (void)scan_left(root, 0, true, loops);
        if (!pcre_checked && (loops || left_recursion(root))) return fail(-1, "recursive call could loop indefinitely");
        return true;
    }
};

// ---- unfolding the tree into priority-ordered alternatives ----
// window bytes a path may carry: the device window adds up to three positions (context before / after, the repeat byte
// of a gapped path) and has to stay within kMaxWindow
constexpr size_t kWinCap = (size_t)kMaxWindow - 3;

// does the subtree hold a capturing group?
bool has_cap_group(const Node &n)
{
    if (n.kind == Node::CAT && n.cap) return true;
    for (const Node &k : n.kids)
        if (has_cap_group(k)) return true;
    return false;
}

struct Unfold {
    std::string why;
    int rc = 0;
    int gap_ids = 0;
    const Node *root = nullptr; // the whole tree: where subroutine calls find their groups
    std::vector<int> calling;   // groups whose bodies are being unfolded on behalf of a call
    static const Node *find_group_in(const Node &n, int g)
    {
        if (n.kind == Node::CAT && n.cap && n.group == g) return &n;
        for (const Node &k : n.kids)
            if (const Node *f = find_group_in(k, g)) return f;
        return nullptr;
    }
    static bool contains_node(const Node &n, const Node *x)
    {
        if (&n == x) return true;
        for (const Node &k : n.kids)
            if (contains_node(k, x)) return true;
        return false;
    }
    bool fail(const char *msg)
    {
        if (rc == 0) {
            rc = 1;
            why = msg;
        }
        return false;
    }
    bool overflow = false; // the failure is "too many paths": compile_pattern tries again with a window cap
    bool room(size_t count)
    {
        if (count <= (size_t)kMaxAlts * 4) return true;
        overflow = true;
        return fail("pattern unfolds into too many alternatives");
    }

    size_t cap_len = SIZE_MAX; // paths are frozen once their window has this many bytes (compile_pattern's second attempts)

    // freeze(a) appended to `out`, unless an identical frozen path is there already (order does not matter between
    // paths that are necessary conditions only)
    bool push_frozen(std::vector<Seq> &out, const Seq &a)
    {
        Seq f = freeze(a);
        for (const Seq &o : out)
            if (o.frozen && o.needs_cap == f.needs_cap && o.cap == f.cap && o.gapped == f.gapped && o.win.size() == f.win.size() && o.pwin.size() == f.pwin.size() && o.asserts == f.asserts &&
                o.p_asserts == f.p_asserts && (!f.gapped || o.gap == f.gap) && std::equal(o.win.begin(), o.win.end(), f.win.begin()) &&
                std::equal(o.pwin.begin(), o.pwin.end(), f.pwin.begin()))
                return true;
        out.push_back(std::move(f));
        return room(out.size());
    }

    // The path as far as it got: its variable repeat (if any) and everything behind it are left to the matcher.
    static Seq freeze(const Seq &a)
    {
        Seq f = a;
        f.has_tail = false;
        f.settled = false;
        f.frozen = true;
        f.inexact = true;
        return f;
    }

    // An unbounded greedy repeat of class T followed by nothing but assertions that are TRUE wherever that repeat stops
    // (so no backtracking into it ever happens and the repeat is still the end of the match):
    //   \w+\b   T is exactly the word characters and the byte before the repeat is a word character too: the repeat stops
    //            in front of a non-word byte or the chunk end, which is a boundary
    //   .*$ under (?m)   T is everything but newline: the repeat stops in front of a newline or at the chunk end
    static bool tail_settles(const Seq &a, const std::vector<Seq> &B)
    {
        if (a.tail_extra != kInf || a.tail_mode == 1 || a.win.empty() || B.size() != 1) return false;
        const Seq &b = B[0];
        if (!b.win.empty() || b.has_tail || b.cap || b.asserts.empty()) return false;
        const ByteSet word = set_word();
        for (const auto &as : b.asserts) {
            if (as.second == A_WB) {
                if (!(a.tail == word) || !(set_and(a.win.back(), word) == a.win.back())) return false;
            } else if (as.second == A_MEOL) {
                if (!(a.tail == set_dot())) return false;
            } else {
                return false;
            }
        }
        return true;
    }

    // every path of `a` followed by every path of `b`.  PCRE backtracks the most recent choice
    // first, so the order is: a's choices major (a variable repeat at the end of `a` counts as one:
    // longest first when greedy, shortest first when lazy), b's choices minor.
    bool concat(const std::vector<Seq> &A, const std::vector<Seq> &B, std::vector<Seq> &out)
    {
        out.clear();
        if (B.size() == 1 && B[0].empty()) {
            out = A;
            return true;
        }
        for (const Seq &a : A) {
            std::vector<Seq> heads;
            if (a.frozen) { // whatever follows is the matcher's business
                if (!push_frozen(out, a)) return false;
                continue;
            }
            if (a.win.size() >= cap_len) { // second attempt at a pattern that unfolded into too many paths: short prefixes only
                if (!push_frozen(out, a)) return false;
                continue;
            }
            if (!a.has_tail) {
                heads.push_back(a);
            } else if (a.settled) {
                // assertion behind a variable repeat, with more pattern behind it: PCRE may backtrack into the repeat
                if (!push_frozen(out, a)) return false;
                continue;
            } else if (tail_settles(a, B)) {
                out.push_back(a); // \w+\b, (?m).*$: the assertion holds wherever the greedy repeat stops; nothing to add
                out.back().settled = true;
                if (!room(out.size())) return false;
                continue;
            } else if (a.tail_extra == kInf || a.tail_extra > kMaxMidRepeat) {
                // An UNBOUNDED repeat with more pattern behind it: the path becomes  P . C{1,} . R  ("gapped": one per path).
                // Zero repetitions are the plain path P . R, tried last by a greedy repeat and first by a lazy one.
                // a large bounded or possessive repeat, a second unbounded one, an assertion inside the part in front of the
                // repeat: the path is frozen in front of this repeat -- its windows so far are what the kernels look for
                bool stop = a.gapped;
                for (const auto &as : a.asserts) stop = stop || as.first != 0;
                if (stop) {
                    if (!push_frozen(out, a)) return false;
                    continue;
                }
                // (a large bounded repeat is looked for as an unbounded one, a possessive one as a greedy one: wider than
                // the pattern, so the matcher has the last word)
                const bool wider = a.tail_extra != kInf || a.tail_mode == 2;
                Seq g;
                g.gapped = true;
                g.pwin = a.win;
                for (const auto &as : a.asserts) g.p_asserts.push_back(as.second);
                g.gap = a.tail;
                g.gap_mode = a.tail_mode == 1 ? 1 : 0;
                g.gap_id = ++gap_ids;
                g.cap = a.cap;
                g.inexact = a.inexact || wider;
                Seq plain = a;
                plain.has_tail = false;
                plain.inexact = a.inexact || wider;
                if (a.tail_mode == 1) {
                    heads.push_back(std::move(plain));
                    heads.push_back(std::move(g));
                } else {
                    heads.push_back(std::move(g));
                    heads.push_back(std::move(plain));
                }
            } else { // the repeat is no longer at the end: unfold it into explicit counts
                // (a possessive one never gives bytes back: every count is still something a match may begin with)
                for (uint32_t k = 0; k <= a.tail_extra; k++) {
                    const uint32_t t = a.tail_mode == 1 ? k : a.tail_extra - k;
                    Seq h = a;
                    h.has_tail = false;
                    h.inexact = a.inexact || a.tail_mode == 2;
                    h.win.insert(h.win.end(), t, a.tail);
                    heads.push_back(std::move(h));
                }
            }
            for (const Seq &h : heads)
                for (const Seq &b : B) {
                    Seq s = h;
                    for (const auto &as : b.asserts) s.asserts.emplace_back(as.first + (uint32_t)h.win.size(), as.second);
                    s.win.insert(s.win.end(), b.win.begin(), b.win.end());
                    s.has_tail = b.has_tail;
                    s.tail = b.tail;
                    s.tail_extra = b.tail_extra;
                    s.tail_mode = b.tail_mode;
                    s.settled = b.settled;
                    s.frozen = b.frozen;
                    s.inexact = h.inexact || b.inexact;
                    // (still nothing in front of the back reference: see compile_pattern)
                    s.needs_cap = h.needs_cap || (b.needs_cap && h.win.empty() && !h.gapped && !h.cap && b.win.empty() && !b.gapped);
                    s.cap = a.cap || b.cap;
                    if (b.gapped) { // the unbounded repeat sits in b: everything of h goes in front of it
                        bool stop = h.gapped || (!b.p_asserts.empty() && !h.win.empty());
                        for (const auto &as : h.asserts) stop = stop || as.first != 0;
                        if (stop) { // a second unbounded repeat, or an assertion that would end up inside the front part
                            if (!push_frozen(out, h)) return false;
                            continue;
                        }
                        s = b;
                        s.pwin = h.win;
                        s.pwin.insert(s.pwin.end(), b.pwin.begin(), b.pwin.end());
                        for (const auto &as : h.asserts) s.p_asserts.insert(s.p_asserts.begin(), as.second);
                        s.cap = h.cap || b.cap;
                        s.inexact = h.inexact || b.inexact;
                        if (s.pwin.size() > (size_t)kMaxWindow) return fail("window longer than the engine supports");
                    }
                    if (s.win.size() > kWinCap) { // longer than the kernels' windows go: its first kWinCap bytes, the matcher for the rest
                        s.win.resize(kWinCap);
                        s.asserts.erase(std::remove_if(s.asserts.begin(), s.asserts.end(), [](const std::pair<uint32_t, int> &as) { return as.first > kWinCap; }),
                                        s.asserts.end());
                        s = freeze(s);
                    }
                    out.push_back(std::move(s));
                    if (!room(out.size())) return false;
                }
        }
        return true;
    }

    // (?:E){lo,hi}: after each iteration the choice is "one more" (tried first when greedy) or "stop"
    bool repeat_paths(const std::vector<Seq> &E, uint32_t lo, uint32_t hi, bool lazy, std::vector<Seq> &out)
    {
        out.clear();
        if (hi == 0) {
            out.push_back(Seq());
            return true;
        }
        std::vector<Seq> rest, more;
        if (!repeat_paths(E, lo > 0 ? lo - 1 : 0, hi - 1, lazy, rest)) return false;
        if (!concat(E, rest, more)) return false;
        if (lo > 0) {
            out = std::move(more);
            return true;
        }
        if (lazy) out.push_back(Seq());
        out.insert(out.end(), more.begin(), more.end());
        if (!lazy) out.push_back(Seq());
        return room(out.size());
    }

    bool run(const Node &nd, std::vector<Seq> &out)
    {
        out.clear();
        switch (nd.kind) {
        case Node::SET: {
            Seq s;
            s.win.push_back(nd.set);
            out.push_back(std::move(s));
            return true;
        }
        case Node::ASSERT: {
            Seq s;
            if (nd.acode == A_KEEP) s.inexact = true; // no condition on the text; where the match is REPORTED to start is the matcher's to say
            else s.asserts.emplace_back(0u, nd.acode);
            out.push_back(std::move(s));
            return true;
        }
        case Node::LOOK: { // consumes nothing; what it demands of the text is the matcher's business
            Seq s;
            s.inexact = true;
            // a positive assertion keeps what its body captured ((?=(x))\1x matches "xx"): a back reference behind it is
            // alive, so the path must not be dropped as "reference to a group that cannot be set" (needs_cap && !cap).
            // Whether the match really closed a group is the matcher's to say (the path is inexact).
            if (!nd.neg && has_cap_group(nd)) s.cap = true;
            out.push_back(std::move(s));
            return true;
        }
        case Node::BACKREF: { // what it repeats is known at match time only: the path stops here
            Seq s;
            s.frozen = true;
            s.inexact = true;
            s.needs_cap = true;
            out.push_back(std::move(s));
            return true;
        }
        case Node::ATOMIC: // the paths of the body, minus PCRE's "no way back into the group": necessary conditions
            if (!run(nd.kids[0], out)) return false;
            for (Seq &s : out) s.inexact = true;
            return true;
        case Node::RECURSE: {
            // What the called group matches is the matcher's to find out: the path stops here -- after what one call must
            // begin with, if the call is not into a group it stands in (no bytes are known then: the path stops at once).
            const Node *grp = root && nd.group > 0 ? find_group_in(*root, nd.group) : nullptr;
            bool inside = !grp;
            for (int g : calling) inside = inside || g == nd.group;
            if (!inside && grp && !contains_node(*grp, &nd) && calling.size() < 4) {
                calling.push_back(nd.group);
                std::vector<Seq> body;
                const bool ok = run(grp->kids[0], body);
                calling.pop_back();
                if (!ok) return false;
                bool empty_call = false;
                for (const Seq &b : body) {
                    if (b.win.empty() && !b.gapped && !b.frozen) { // the call may match "": what follows it decides
                        empty_call = true;
                        if (b.has_tail) { // (x* as the whole call: a call that does consume begins with a byte of the repeat)
                            Seq one = b;
                            one.win.push_back(b.tail);
                            one.cap = one.needs_cap = false;
                            if (!push_frozen(out, one)) return false;
                        }
                        continue;
                    }
                    Seq f = freeze(b);
                    f.cap = false; // (a called group does not capture, and what it captured inside is dropped)
                    f.needs_cap = false;
                    if (!push_frozen(out, f)) return false;
                }
                if (empty_call) {
                    Seq none;
                    none.inexact = true;
                    out.push_back(none);
                }
                return room(out.size());
            }
            Seq s;
            s.frozen = true;
            s.inexact = true;
            out.push_back(std::move(s));
            return true;
        }
        case Node::COND: { // either branch may be the one taken (which one is decided at match time): necessary conditions
            const size_t base = nd.cond == Node::C_ASSERT ? 1 : 0;
            const bool cond_caps = base && has_cap_group(nd.kids[0]); // (also a negative one: matcher.cc, COND)
            for (size_t b = base; b < nd.kids.size(); b++) {
                std::vector<Seq> kid;
                if (!run(nd.kids[b], kid)) return false;
                out.insert(out.end(), kid.begin(), kid.end());
                if (!room(out.size())) return false;
            }
            if (nd.kids.size() == base + 1) out.push_back(Seq()); // no "no" branch: the group matches "" when the condition fails
            for (Seq &s : out) {
                s.inexact = true;
                if (cond_caps) s.cap = true; // (a group the condition's assertion closed stays set, see LOOK)
            }
            return room(out.size());
        }
        case Node::CAT: {
            out.push_back(Seq());
            for (const Node &k : nd.kids) {
                std::vector<Seq> kid, joined;
                if (!run(k, kid)) return false;
                if (!concat(out, kid, joined)) return false;
                out.swap(joined);
            }
            if (nd.cap) // every path through a capturing group sets it, even an empty one (a frozen path never got to its end)
                for (Seq &s : out)
                    if (!s.frozen) s.cap = true;
            return true;
        }
        case Node::ALT:
            for (const Node &k : nd.kids) {
                std::vector<Seq> kid;
                if (!run(k, kid)) return false;
                out.insert(out.end(), kid.begin(), kid.end());
                if (!room(out.size())) return false;
            }
            return true;
        case Node::REP: {
            const Node &k = nd.kids[0];
            if (nd.max == 0) { // {0}: matches nothing, consumes nothing
                out.push_back(Seq());
                return true;
            }
            if (k.kind == Node::SET) {
                Seq s;
                if (nd.min > kWinCap) { // x{300}: the kernels look for its first kWinCap bytes
                    s.win.assign(kWinCap, k.set);
                    out.push_back(freeze(s));
                    return true;
                }
                s.win.assign(nd.min, k.set);
                if (nd.max > nd.min) {
                    s.has_tail = true;
                    s.tail = k.set;
                    s.tail_extra = nd.max == kInf ? kInf : nd.max - nd.min;
                    s.tail_mode = nd.mode;
                }
                out.push_back(std::move(s));
                return true;
            }
            std::vector<Seq> E;
            if (!run(k, E)) return false;
            bool plain = nd.mode != 2 && nd.max != kInf && nd.max <= 2 * kMaxMidRepeat;
            for (const Seq &e : E) {
                if (e.win.empty() && !e.gapped) plain = false;                                  // an iteration may match ""
                if ((!e.asserts.empty() || !e.p_asserts.empty()) && (nd.min != 1 || nd.max != 1)) plain = false; // an assertion inside
                if (e.frozen) plain = false;
            }
            if (plain) return repeat_paths(E, nd.min, nd.max, nd.mode == 1, out);
            // A repeat of a group that cannot be unfolded (unbounded, possessive, may match "", ...): every path of ONE
            // iteration, frozen -- the kernels look for the first iteration, the matcher does the rest -- plus, if zero
            // iterations are allowed, the path that skips the group.
            // An iteration that consumes nothing contributes no bytes: the first byte a match takes then comes from a later
            // iteration (same paths) or from what follows the group -- the path that skips the group stands for that.
            bool skip = nd.min == 0, skip_cap = false;
            // an iteration that consumes nothing can still close a group: behind it, an iteration that begins with a back
            // reference is no longer dead
            bool empty_iteration = false;
            for (const Seq &e : E) empty_iteration = empty_iteration || (e.win.empty() && !e.gapped && !e.needs_cap);
            if (empty_iteration && nd.max > 1)
                for (Seq &e : E) e.needs_cap = false;
            for (const Seq &e : E) {
                if (!e.win.empty() || e.gapped || (e.frozen && e.win.empty())) { // (an iteration that is a back reference: unknown bytes)
                    if (!push_frozen(out, e)) return false;
                    continue;
                }
                skip = true;
                skip_cap = skip_cap || e.cap;
                if (e.has_tail) { // (?:a*)+ : an iteration that does consume begins with a byte of the repeat
                    Seq one = e;
                    one.win.push_back(e.tail);
                    if (!push_frozen(out, one)) return false;
                }
            }
            if (skip) {
                Seq none;
                none.inexact = true;
                none.cap = skip_cap; // (iterations that consumed nothing may have closed a group: a back reference behind them is alive)
                out.push_back(none);
            }
            return room(out.size());
        }
        }
        return fail("internal: unknown node");
    }
};

// ---- assertions -> context conditions -------------------------------------------------------------
// Every assertion of an unfolded path is either decided on the spot (it stands between two window
// positions whose classes settle it, possibly after narrowing a class or splitting it into its word
// and non-word parts), or it stands at one end of the window and becomes a condition on the single
// byte before / after the match.  Paths that can never match are dropped.  rc: 0 ok, 1 unsupported.
struct Resolver {
    std::string why;
    std::vector<Seq> out;
    ByteSet word = set_word(), nonword = set_not(set_word()), nl;
    Resolver() { nl.set('\n'); }

    bool fail(const char *msg)
    {
        if (why.empty()) why = msg;
        return false;
    }
    // An assertion that cannot be turned into a condition on the path's bytes is left out: the path then only says where
    // a match MAY be, and the matcher decides (Database::exact == false).
    bool skip(Seq s, size_t k, bool eol, bool eos)
    {
        s.inexact = true;
        return step(std::move(s), k + 1, eol, eos);
    }

    // An assertion right behind the unbounded repeat of a gapped path: the byte in front of it is a repeat byte, which
    // cannot be narrowed -- the repeat's class has to settle that side on its own.  Behind it: the first byte of the
    // rest, or, when the rest is empty, the byte after the match (post context).
    bool step_after_gap(Seq s, size_t k, bool eol, bool eos)
    {
        const int code = s.asserts[k].second;
        const bool trail = s.win.empty();
        if (trail && s.has_tail) return skip(std::move(s), k, eol, eos); // assertion between two repeats
        const bool gap_word = set_and(s.gap, word) == s.gap, gap_nonword = set_and(s.gap, nonword) == s.gap;
        switch (code) {
        case A_BOS: return true; // something in front of the subject start: never
        case A_MBOL:
            if (s.gap == nl) return step(std::move(s), k + 1, eol, eos);
            if (!s.gap.test('\n')) return true;
            return skip(std::move(s), k, eol, eos); // (?m)^ behind a repeat that may or may not end in a newline
        case A_EOS:
            if (!trail) return true;
            return step(std::move(s), k + 1, eol, true);
        case A_EOL:
            if (!trail) return skip(std::move(s), k, eol, eos); // $ before the end of an alternative
            return step(std::move(s), k + 1, true, eos);
        case A_MEOL:
            if (trail) s.post = set_and(s.post, nl);
            else s.win[0] = set_and(s.win[0], nl);
            return step(std::move(s), k + 1, eol, eos);
        case A_WB:
        case A_NWB: {
            if (!gap_word && !gap_nonword) return skip(std::move(s), k, eol, eos); // \b behind a repeat of word and non-word characters
            const int lw = gap_word ? 1 : 0;
            const int rw = (code == A_WB) ? 1 - lw : lw; // the other side must (not) differ
            const ByteSet &rset = rw ? word : nonword;
            if (trail) {
                s.post = set_and(s.post, rset);
                if (rw) s.post_end = false;
            } else {
                s.win[0] = set_and(s.win[0], rset);
            }
            return step(std::move(s), k + 1, eol, eos);
        }
        }
        return fail("internal: unknown assertion");
    }

    // the assertions in front of a gapped path's first part: conditions on the byte before the match
    bool lead_of_gapped(Seq &s, bool &alive)
    {
        alive = true;
        for (int code : s.p_asserts) {
            ByteSet &first = s.pwin.empty() ? s.gap : s.pwin[0];
            switch (code) {
            case A_BOS: s.pre = ByteSet(); break;
            case A_MBOL: s.pre = set_and(s.pre, nl); break;
            case A_WB:
            case A_NWB: {
                const bool fw = set_and(first, word) == first, fn = set_and(first, nonword) == first;
                if (!fw && !fn) { // \b in front of a class of word and non-word characters followed by an unbounded repeat
                    s.inexact = true;
                    break;
                }
                const bool prev_word = (code == A_WB) ? !fw : fw;
                s.pre = set_and(s.pre, prev_word ? word : nonword);
                if (prev_word) s.pre_start = false;
                break;
            }
            case A_MEOL: // (?m)$ in front of something: that something starts with a newline
                if (!s.pwin.empty()) {
                    s.pwin[0] = set_and(s.pwin[0], nl);
                    if (s.pwin[0].count() == 0) {
                        alive = false;
                        return true;
                    }
                } else if (!(s.gap == nl)) {
                    if (!s.gap.test('\n')) {
                        alive = false;
                        return true;
                    }
                    s.inexact = true; // (?m)$ in front of a repeat that may or may not start with a newline
                }
                break;
            case A_EOL: s.inexact = true; break; // $ before the end of an alternative
            default: alive = false; return true; // \\z in front of something: never
            }
        }
        s.p_asserts.clear();
        return true;
    }

    // s.asserts[k..] still to do; eol/eos remember a trailing $ / \z, post_set collects the other trailing conditions
    bool step(Seq s, size_t k, bool eol, bool eos)
    {
        if (k == s.asserts.size()) {
            if (eol || eos) {
                s.post_final_nl = eol && !eos && s.post.test('\n');
                s.post = ByteSet();
            }
            for (const ByteSet &b : s.win)
                if (b.count() == 0) return true; // narrowed to nothing: this path never matches
            if (s.gapped) {
                bool alive;
                if (!lead_of_gapped(s, alive)) return false;
                if (!alive) return true;
            }
            s.asserts.clear();
            out.push_back(std::move(s));
            return true;
        }
        const uint32_t pos = s.asserts[k].first;
        const int code = s.asserts[k].second;
        const size_t L = s.win.size();
        if (s.gapped && pos == 0) return step_after_gap(std::move(s), k, eol, eos);
        if (L == 0) return step(std::move(s), k + 1, eol, eos); // empty window: the pattern can match "" and every file is skipped anyway (Q2)
        const bool lead = pos == 0, trail = pos == L;
        // (an assertion AFTER a variable repeat never gets here: the repeat was no longer at the end of its path and was
        // unfolded, or refused, in Unfold::concat; with has_tail set the assertion stands between the window and the tail)
        switch (code) {
        case A_BOS:
            if (!lead) return true; // something must come before the subject start: never
            s.pre = ByteSet();
            return step(std::move(s), k + 1, eol, eos);
        case A_MBOL:
            if (lead) {
                s.pre = set_and(s.pre, nl);
            } else {
                if (trail) return skip(std::move(s), k, eol, eos); // (?m)^ at the end of an alternative
                s.win[pos - 1] = set_and(s.win[pos - 1], nl);
            }
            return step(std::move(s), k + 1, eol, eos);
        case A_EOS:
            if (!trail) return true;
            return step(std::move(s), k + 1, eol, true);
        case A_EOL:
            if (!trail) return skip(std::move(s), k, eol, eos); // $ before the end of an alternative
            return step(std::move(s), k + 1, true, eos);
        case A_MEOL:
            if (trail) s.post = set_and(s.post, nl);
            else s.win[pos] = set_and(s.win[pos], nl);
            return step(std::move(s), k + 1, eol, eos);
        case A_WB:
        case A_NWB: {
            // the bytes on the two sides: a window class, or the context byte (lead: before, trail: after)
            // split both sides into their word / non-word parts and keep the combinations the assertion allows
            for (int lw = 0; lw < 2; lw++)
                for (int rw = 0; rw < 2; rw++) {
                    const bool boundary = lw != rw;
                    if (boundary != (code == A_WB)) continue;
                    Seq v = s;
                    const ByteSet &lset = lw ? word : nonword, &rset = rw ? word : nonword;
                    if (lead) {
                        v.pre = set_and(v.pre, lset);
                        if (lw) v.pre_start = false; // the subject start counts as a non-word character
                    } else {
                        v.win[pos - 1] = set_and(v.win[pos - 1], lset);
                    }
                    if (trail) {
                        v.post = set_and(v.post, rset);
                        if (rw) v.post_end = false; // so does the chunk end
                    } else {
                        v.win[pos] = set_and(v.win[pos], rset);
                    }
                    if (!step(std::move(v), k + 1, eol, eos)) return false;
                }
            return true;
        }
        }
        return fail("internal: unknown assertion");
    }
};

// Rough rank of how common a byte is in text/source/binary corpora (higher = more
// common).  Only used to pick the rarest 4-byte anchor of a literal; any table is
// correct, a better one just triggers the verify path less often.
int byte_rank(unsigned b)
{
    static const char common[] = " etaoinsrhldcumfpgwybvkxjqz"; // most -> least frequent
    const char *q = b ? strchr(common, (int)b) : nullptr;
    if (q) return 255 - (int)(q - common) * 4;
    if (b == 0) return 230;              // NUL runs in binaries
    if (b == '\n' || b == '\t') return 150;
    if (b >= '0' && b <= '9') return 110;
    if (b >= 'A' && b <= 'Z') return 100;
    if (b == '_' || b == '.' || b == ',' || b == '/' || b == '-' || b == '=' || b == '(' || b == ')' ||
        b == ';' || b == '"' || b == '\'' || b == ':' || b == '*' || b == '>' || b == '<')
        return 90;
    if (b == 0xff) return 80;
    if (b >= 33 && b < 127) return 60;
    return 20;
}

// How often a byte value turns up in text, roughly (source code, logs, prose): what a class costs as a filter position is
// the sum over its members -- \W has 193 members and takes one byte in four, [a-z] has 26 and takes six in ten.  (Pricing a
// class by its SIZE, as the K3 depth choice does, is fine for telling literals from classes and useless for telling \w from \W.)
double byte_prob(unsigned b)
{
    if (b >= 'a' && b <= 'z') return 0.58 / 26;
    if (b == ' ') return 0.14;
    if (b >= 'A' && b <= 'Z') return 0.05 / 26;
    if (b >= '0' && b <= '9') return 0.06 / 10;
    if (b == '\n') return 0.025;
    if (b == '_') return 0.01;
    if (b == '\t') return 0.01;
    if (b >= 33 && b < 127) return 0.115 / 31; // punctuation
    return 0.01 / 160;                        // control characters, high bytes
}
double class_prob(const ByteSet &c)
{
    double p = 0;
    for (unsigned b = 0; b < 256; b++)
        if (c.test(b)) p += byte_prob(b);
    return std::min(1.0, p);
}

std::atomic<uint64_t> g_next_id{1};

uint64_t node_minlen(const Node &n);
const Node *find_group(const Node &n, int g);

// libpcre quirk no. 2 (8.39 and 8.45, interpreter and JIT): a group whose first item is a positive look-ahead that begins
// with a literal byte -- (?:(?=x))x\B -- gives pcre_exec both a "first byte" and a "required byte" that are the same
// occurrence; the start-up optimisation then looks for a SECOND one further on and reports "no match" when there is none.
// Refused rather than imitated.
const Node *first_item(const Node &n)
{
    if (n.kind == Node::CAT) return n.kids.empty() ? nullptr : first_item(n.kids[0]);
    return &n;
}
// the first item of the node that is not a zero-width assertion (\b in front of a literal does not hide it from libpcre's
// first-byte analysis)
const Node *first_consuming_item(const Node &n)
{
    if (n.kind != Node::CAT) return &n;
    for (const Node &k : n.kids) {
        if (k.kind == Node::ASSERT) continue;
        return first_consuming_item(k);
    }
    return nullptr;
}
bool lookahead_first_in_group(const Node &n, bool top)
{
    if (!top && (n.kind == Node::CAT || n.kind == Node::ALT)) {
        const Node *branches = n.kind == Node::ALT ? n.kids.data() : &n;
        const size_t nb = n.kind == Node::ALT ? n.kids.size() : 1;
        for (size_t b = 0; b < nb; b++) {
            const Node *f = first_item(branches[b]);
            if (f && f->kind == Node::REP && f->min >= 1) f = &f->kids[0]; // ((?=x){2} is (?=x))
            if (f && f->kind == Node::LOOK && !f->neg && !f->behind) {
                const Node *g = first_consuming_item(f->kids[0]);
                if (g && g->kind == Node::REP && g->min >= 1) g = &g->kids[0];
                if (g && g->kind == Node::SET && g->set.count() <= 2) return true;
            }
        }
    }
    for (const Node &k : n.kids) {
        // a CAT directly under the top-level CAT/ALT is a branch of the pattern itself, not a group
        const bool kid_top = top && (n.kind == Node::ALT) && k.kind == Node::CAT && !k.cap;
        if (k.kind == Node::LOOK) {
            if (lookahead_first_in_group(k.kids[0], true)) return true;
            continue;
        }
        if (lookahead_first_in_group(k, kid_top)) return true;
    }
    return false;
}

// libpcre quirks around the two new constructs (8.39 JIT and 8.45 alike), refused rather than imitated:
//  * a ^ inside a conditional group -- in the condition's assertion, in a branch, in a DEFINE -- is read by pcre_compile's
//    is_anchored / is_startline as if it stood at the head of the pattern: (?(DEFINE)^)\w is only tried at line starts,
//    (?:(?(?=^))[.]) only at the subject start;
//  * an OPTIONAL subroutine call that a match can begin with -- (?1)* (\s)?, \g<1>?b(..){0} -- is left out of the start-up
//    optimisation's first byte / start bits: offsets whose first byte only the call can take are skipped.
bool cond_holds_circumflex(const Node &n, bool inside)
{
    if (inside && n.kind == Node::ASSERT && (n.acode == A_BOS || n.acode == A_MBOL)) return true;
    for (const Node &k : n.kids)
        if (cond_holds_circumflex(k, inside || n.kind == Node::COND)) return true;
    return false;
}
// can the first byte of a match be taken inside an optional (min 0) item that holds a subroutine call?
bool begins_with_optional_call(const Node &n, bool optional)
{
    switch (n.kind) {
    case Node::RECURSE: return optional;
    case Node::SET:
    case Node::ASSERT:
    case Node::BACKREF:
    case Node::LOOK: return false;
    case Node::ATOMIC: return begins_with_optional_call(n.kids[0], optional);
    case Node::REP: return n.max != 0 && begins_with_optional_call(n.kids[0], optional || n.min == 0);
    case Node::ALT:
        for (const Node &k : n.kids)
            if (begins_with_optional_call(k, optional)) return true;
        return false;
    case Node::COND:
        for (size_t b = n.cond == Node::C_ASSERT ? 1 : 0; b < n.kids.size(); b++)
            if (begins_with_optional_call(n.kids[b], optional)) return true;
        return false;
    case Node::CAT:
        for (const Node &k : n.kids) {
            if (begins_with_optional_call(k, optional)) return true;
            if (node_minlen(k) > 0) return false; // (find_minlength's count: zero for anything that may match "")
        }
        return false;
    }
    return false;
}

//  * a group that is called as a subroutine and whose pattern begins with an unbounded repeat of one
//    class -- b(?1)c|(A*)x, b(?1)c|([^x]*)x -- : after an attempt that made the call has failed, the JIT build does not
//    find matches that begin before the place the call had reached ("bAAx x": the interpreter reports 1, the JIT 5).
//    It takes the group to stand at the head of an alternative of the pattern proper: behind another item (y(A*)x), or
//    inside a (?(DEFINE)..) that only the calls reach, both builds agree.
bool group_heads_a_branch(const Node &n, int g)
{
    if (n.kind == Node::COND && n.cond == Node::C_DEFINE) return false;
    if (n.kind == Node::CAT && !n.cap) { // the first item that is not an assertion (\b(1*a) is hit like (1*a))
        for (const Node &item : n.kids) {
            if (item.kind == Node::ASSERT || item.kind == Node::LOOK) continue;
            const Node *f = &item;
            while (f->kind == Node::ATOMIC || f->kind == Node::REP) f = &f->kids[0]; // ((a++){2}|(?1) as well)
            if (f->kind == Node::CAT && f->cap && f->group == g) return true;
            break;
        }
    }
    for (const Node &k : n.kids)
        if (group_heads_a_branch(k, g)) return true;
    return false;
}
bool called_group_begins_with_repeat(const Node &n, const Node &root)
{
    if (n.kind == Node::RECURSE && n.group > 0) {
        if (const Node *g = find_group(root, n.group)) {
            const Node &body = g->kids[0];
            const Node *branches = body.kind == Node::ALT ? body.kids.data() : &body;
            const size_t nb = body.kind == Node::ALT ? body.kids.size() : 1;
            for (size_t b = 0; b < nb; b++) {
                const Node *f = first_item(branches[b]);
                while (f && (f->kind == Node::ATOMIC || (f->kind == Node::CAT && !f->kids.empty()))) f = first_item(f->kids[0]);
                // (a lazy repeat too: auto-possessification makes a*? in front of a byte it cannot match a possessive a*+)
                if (f && f->kind == Node::REP && f->kids[0].kind == Node::SET && f->max == kInf && group_heads_a_branch(root, n.group)) return true;
            }
        }
    }
    for (const Node &k : n.kids)
        if (called_group_begins_with_repeat(k, root)) return true;
    return false;
}

// libpcre quirk no. 2b (8.39 and 8.45, interpreter and JIT): pcre_compile takes a forward assertion's "required byte" over as
// the pattern's own ("useful for /(?=abcde).+/"); when the item behind the assertion is that same byte and becomes the
// FIRST byte, the start-up check asks for two occurrences: (?=1[a-c]| 1)1 never matches "1a".  Refused: a positive
// look-ahead at the head of a branch, directly followed by a literal byte that also stands in the assertion as a literal.
bool holds_literal(const Node &n, const ByteSet &lit)
{
    if (n.kind == Node::SET && n.set.count() <= 2 && set_and(n.set, lit).count() > 0) return true;
    for (const Node &k : n.kids)
        if (holds_literal(k, lit)) return true;
    return false;
}
bool lookahead_hands_on_required_byte(const Node &n)
{
    if (n.kind == Node::CAT && !n.cap) {
        const Node *look = nullptr;
        for (const Node &item : n.kids) {
            if (item.kind == Node::ASSERT) continue;
            const Node *li = &item;
            if (li->kind == Node::REP && li->min >= 1 && li->kids[0].kind == Node::LOOK) li = &li->kids[0];
            if (li->kind == Node::LOOK && !li->neg && !li->behind) {
                if (!look) look = li;
                continue;
            }
            if (li->kind == Node::LOOK) continue;
            const Node *f = &item;
            if (f->kind == Node::REP && f->min >= 1) f = &f->kids[0];
            if (look && f->kind == Node::SET && f->set.count() <= 2 && holds_literal(look->kids[0], f->set)) {
                // (a literal byte further on in the branch becomes the required byte instead: (?=(a))ab is fine)
                bool later_literal = false;
                for (const Node *q = &item + 1; q < n.kids.data() + n.kids.size(); q++) {
                    const Node *g = q;
                    if (g->kind == Node::REP && g->min >= 1) g = &g->kids[0];
                    later_literal = later_literal || (g->kind == Node::SET && g->set.count() <= 2);
                }
                if (!later_literal) return true;
            }
            break;
        }
    }
    for (const Node &k : n.kids)
        if (lookahead_hands_on_required_byte(k)) return true;
    return false;
}

// libpcre quirk no. 2c (both builds): a pattern of ONE top-level branch that begins with a positive look-ahead whose
// alternatives all begin with the literal byte X gets X as its first byte; when the pattern's own X comes behind something
// optional -- (?=1)c*1 -- it becomes the required byte as well, and the start-up check looks for it BEHIND the first byte:
// "1" alone never matches.  (Directly behind the assertion -- (?=1)1 -- it is the first byte itself and all is well.)
bool leading_lookahead_sets_first_byte(const Node &root)
{
    if (root.kind != Node::CAT || root.cap) return false;
    size_t i = 0;
    while (i < root.kids.size() && root.kids[i].kind == Node::ASSERT) i++;
    if (i >= root.kids.size()) return false;
    const Node *lk = &root.kids[i];
    if (lk->kind == Node::REP && lk->min >= 1) lk = &lk->kids[0];
    const Node &look = *lk;
    if (look.kind != Node::LOOK || look.neg || look.behind) return false;
    const Node &body = look.kids[0];
    const Node *branches = body.kind == Node::ALT ? body.kids.data() : &body;
    const size_t nb = body.kind == Node::ALT ? body.kids.size() : 1;
    ByteSet x;
    for (size_t b = 0; b < nb; b++) {
        const Node *f = first_consuming_item(branches[b]);
        if (f && f->kind == Node::REP && f->min >= 1) f = &f->kids[0];
        if (!f || f->kind != Node::SET || f->set.count() > 2) return false;
        if (b > 0 && !(f->set == x)) return false;
        x = f->set;
    }
    size_t j = i + 1;
    while (j < root.kids.size() && (root.kids[j].kind == Node::ASSERT || root.kids[j].kind == Node::LOOK)) j++;
    if (j >= root.kids.size()) return false;
    const Node *f = &root.kids[j];
    if (f->kind == Node::REP && f->min >= 1) f = &f->kids[0];
    if (f->kind == Node::SET && f->set.count() <= 2 && set_and(f->set, x).count() > 0) return false; // the first byte itself
    for (size_t k = j; k < root.kids.size(); k++)
        if (holds_literal(root.kids[k], x)) return true;
    return false;
}

// libpcre quirk no. 2d (both builds): is_startline / is_anchored look INTO a positive look-ahead at the head of the pattern: when
// it begins with .* (or .*? .{0,} \N*) the whole pattern is taken to begin with .* -- tried at line starts only (at the
// subject start only under (?s)): (?=.*)[a-c] never matches " b".  Refused when any top-level alternative begins that way
// (an assertion or an optional item in front of the look-ahead switches the analysis off, and so does a class: [^\n]*).
static bool begins_with_dot_star(const Node &n)
{
    switch (n.kind) {
    case Node::REP: return n.kids[0].kind == Node::SET && n.min == 0 && n.max == kInf && n.mode != 2 && n.kids[0].set.count() >= 255;
    case Node::CAT: return !n.kids.empty() && begins_with_dot_star(n.kids[0]);
    case Node::ATOMIC: return begins_with_dot_star(n.kids[0]);
    case Node::ALT:
        for (const Node &k : n.kids)
            if (begins_with_dot_star(k)) return true;
        return false;
    default: return false;
    }
}
static bool head_is_lookahead_for_dot_star(const Node &n)
{
    switch (n.kind) {
    case Node::LOOK: return !n.neg && !n.behind && begins_with_dot_star(n.kids[0]);
    case Node::REP: return n.min >= 1 && head_is_lookahead_for_dot_star(n.kids[0]);
    case Node::CAT: return !n.kids.empty() && head_is_lookahead_for_dot_star(n.kids[0]);
    case Node::ATOMIC: return head_is_lookahead_for_dot_star(n.kids[0]);
    case Node::ALT:
        for (const Node &k : n.kids)
            if (head_is_lookahead_for_dot_star(k)) return true;
        return false;
    default: return false;
    }
}

bool has_optional_group(const Node &n)
{
    if (n.kind == Node::REP && n.kids[0].kind != Node::SET && n.min == 0) return true;
    for (const Node &k : n.kids)
        if (has_optional_group(k)) return true;
    return false;
}

// libpcre quirk (8.39 and 8.45 alike): a greedy repeat of a single class that is followed -- directly, or across group
// brackets and items that may match "" -- by a possessive group repeat with a variable count ((?:0)?+, (..){1,2}+: compiled
// as an atomic group whose last item is optional) is made possessive by pcre_compile's auto-possessification whenever
// the group's first bytes cannot continue the repeat, WITHOUT looking at what follows the group:  b[x.]{0,2}(?:0)?+[x.] x
// never matches "b.x x".  Such patterns are refused rather than imitated.  Returns whether the node can END with a
// greedy variable repeat of a class (`prev`: whether what precedes it can); sets `quirk` when the construct is met.
// (g_lazy_counts: the second pass -- auto-possessification also turns LAZY repeats into possessive GREEDY ones when its tables
// say the next item cannot continue them, and for \R / \X the tables are wrong: \S??\R takes NEL for \S.  That pass looks at
// \R / \X only.)
static thread_local bool g_lazy_counts = false;
bool ends_in_greedy_repeat(const Node &n, bool prev, bool &quirk)
{
    switch (n.kind) {
    case Node::SET: return false;
    case Node::ASSERT: return n.acode == A_KEEP ? false : prev; // (\K is an opcode of its own: the look-ahead stops there)
    case Node::BACKREF: return false; // (not an opcode auto-possessification looks through)
    case Node::RECURSE: return false;
    case Node::COND: { // (OP_COND is not an opcode it looks through either; the branches are patterns of their own)
        for (const Node &k : n.kids) ends_in_greedy_repeat(k.kind == Node::LOOK ? k.kids[0] : k, false, quirk);
        return false;
    }
    case Node::LOOK: { // an assertion opcode stops auto-possessification's look-ahead; its body is a pattern of its own
        ends_in_greedy_repeat(n.kids[0], false, quirk);
        return false;
    }
    case Node::ATOMIC: { // (?>..) is what a possessive group repeat compiles to: the same quirk when a branch of it can match ""
        if (n.newline_seq) { // \R, \X: one opcode; a greedy class repeat in front of it is made possessive by a wrong table
            if (prev) quirk = true;
            return false;
        }
        if (!g_lazy_counts && prev && node_minlen(n.kids[0]) == 0) quirk = true;
        return ends_in_greedy_repeat(n.kids[0], prev, quirk);
    }
    case Node::CAT: {
        bool f = prev;
        for (const Node &k : n.kids) f = ends_in_greedy_repeat(k, f, quirk);
        return f;
    }
    case Node::ALT: {
        bool f = false;
        for (const Node &k : n.kids) f = ends_in_greedy_repeat(k, prev, quirk) || f;
        return f;
    }
    case Node::REP: {
        const Node &k = n.kids[0];
        if (n.max == 0) return prev;
        if (k.kind == Node::SET) {
            if (n.max > n.min && (n.mode == 0 || (g_lazy_counts && n.mode == 1))) return true;
            return n.min == 0 ? prev : false;
        }
        if (!g_lazy_counts && n.mode == 2 && n.max > n.min && prev) quirk = true;
        bool f = ends_in_greedy_repeat(k, prev, quirk);
        if (n.max > 1) f = ends_in_greedy_repeat(k, f || prev, quirk) || f; // the end of one iteration precedes the next
        return (n.min == 0 || node_minlen(k) == 0) ? (f || prev) : f;
    }
    }
    return false;
}

// The shortest subject a match needs, the way pcre_study's find_minlength() counts it: alternatives take the minimum,
// repeats multiply, assertions count nothing (and are not checked for consistency).
// A back reference counts what its group counts (find_minlength: OP_REF), nothing when it stands inside that group or the
// groups refer to each other in a circle.
struct MinCtx {
    const Node *root;
    std::vector<int> active;
    // A counted repeat is compiled into copies, and every reference or call finds the FIRST copy's bracket (find_bracket):
    // from the second copy on, a group's reference to "itself" is a reference to another bracket -- no recursion, and what it
    // counts is worked out afresh, under the chain in force at that point.  `later`: the groups of a repeated item whose
    // later copy is being counted; `first_eval`: the first copy is being counted on behalf of such a reference (inside it
    // the groups are themselves again).
    // (Counted, because repeats nest: inside the first copy of an outer repeat an inner repeat has later copies of its own.
    // A reference is "from a later copy" while later_d[g] > first_d[g].)
    std::vector<int> later_d, first_d;
    bool from_later_copy(int g) const { return (size_t)g < later_d.size() && later_d[(size_t)g] > ((size_t)g < first_d.size() ? first_d[(size_t)g] : 0); }
    bool plain = false; // the true lower bound instead of find_minlength's: a reference may repeat "", every branch counts
};
static bool is_in(const std::vector<int> &v, int g)
{
    for (int x : v)
        if (x == g) return true;
    return false;
}
const Node *find_group(const Node &n, int g)
{
    if (n.kind == Node::CAT && n.cap && n.group == g) return &n;
    for (const Node &k : n.kids)
        if (const Node *f = find_group(k, g)) return f;
    return nullptr;
}
bool contains(const Node &n, const Node *x)
{
    if (&n == x) return true;
    for (const Node &k : n.kids)
        if (contains(k, x)) return true;
    return false;
}
bool has_kind(const Node &n, Node::Kind kind)
{
    if (n.kind == kind) return true;
    for (const Node &k : n.kids)
        if (has_kind(k, kind)) return true;
    return false;
}
void collect_groups(const Node &n, std::vector<const Node *> &out)
{
    if (n.kind == Node::CAT && n.cap) out.push_back(&n);
    for (const Node &k : n.kids) collect_groups(k, out);
}
// is the item a back reference (possibly quantified) to a group it stands in, or to one whose length is being computed?
bool recursive_ref(const Node &item, const MinCtx &cx)
{
    const Node *r = &item;
    if (r->kind == Node::REP && r->kids[0].kind == Node::BACKREF) {
        if (r->mode == 2) return false; // \1?+ is compiled as (?>\1?): a group of its own, whose recursion flag stays inside it
        r = &r->kids[0];
    }
    if (item.kind == Node::RECURSE) { // (?R), or (?n) inside group n: find_minlength's had_recurse
        if (cx.plain) return false;
        if (item.group == 0) return true;
        if (is_in(cx.active, item.group)) return true;
        if (cx.from_later_copy(item.group)) return false;
        const Node *grp = find_group(*cx.root, item.group);
        return grp && contains(*grp, &item);
    }
    if (r->kind != Node::BACKREF) return false;
    if (cx.plain) return false;
    if (is_in(cx.active, r->group)) return true;
    if (cx.from_later_copy(r->group)) return false;
    const Node *grp = find_group(*cx.root, r->group);
    return grp && contains(*grp, r);
}
// what a back reference to / a call of group g at node n counts (find_minlength: OP_REF, OP_RECURSE)
uint64_t node_minlen(const Node &n, MinCtx &cx);
uint64_t reference_minlen(const Node &n, int g, MinCtx &cx)
{
    const Node *grp = find_group(*cx.root, g);
    if (!grp) return 0;
    // (from a later copy EVERY reference into the repeated item lands in the first copy, whether or not it stands inside the
    // group it names: see MinCtx)
    const bool via_copy = cx.from_later_copy(g);
    const bool inside = !via_copy && contains(*grp, &n);
    if (inside || is_in(cx.active, g)) return 0; // recursion, directly or round the chain (8.39's recurse_check: one list for references and calls)
    cx.active.push_back(g);
    const std::vector<int> saved = cx.first_d;
    if (via_copy) { // everything of the repeated items is "first copy" from here on
        cx.first_d.resize(cx.later_d.size(), 0);
        for (size_t h = 0; h < cx.later_d.size(); h++) cx.first_d[h] = std::max(cx.first_d[h], cx.later_d[h]);
    }
    const uint64_t d = node_minlen(*grp, cx);
    cx.first_d = saved;
    cx.active.pop_back();
    return d;
}
uint64_t node_minlen(const Node &n, MinCtx &cx)
{
    constexpr uint64_t cap = 1u << 30;
    switch (n.kind) {
    case Node::SET: return 1;
    case Node::ASSERT:
    case Node::LOOK: return 0;
    case Node::BACKREF: return cx.plain ? 0 : reference_minlen(n, n.group, cx);
    case Node::ATOMIC: return node_minlen(n.kids[0], cx);
    case Node::RECURSE: {
        // find_minlength, OP_RECURSE: a call from inside the called group, or into a group whose length is being worked
        // out further up (mutual recursion), counts nothing; any other call counts what the called group counts
        if (n.group == 0) return 0;
        if (cx.plain) { // (the true lower bound: what the called group needs, unless the call is recursive)
            const Node *grp = find_group(*cx.root, n.group);
            if (!grp || contains(*grp, &n) || is_in(cx.active, n.group)) return 0;
            cx.active.push_back(n.group);
            const uint64_t d = node_minlen(*grp, cx);
            cx.active.pop_back();
            return d;
        }
        return reference_minlen(n, n.group, cx);
    }
    case Node::COND: {
        // find_minlength, OP_COND: a condition with one branch has an implied empty second one and counts nothing (that
        // covers DEFINE); with two it is a bracket like any other -- the shorter branch
        const size_t base = n.cond == Node::C_ASSERT ? 1 : 0;
        if (n.kids.size() < base + 2) return 0;
        if (cx.plain) return std::min(node_minlen(n.kids[base], cx), node_minlen(n.kids[base + 1], cx));
        bool first = true;
        uint64_t t = cap;
        for (size_t b = base; b < n.kids.size(); b++) {
            const Node &k = n.kids[b];
            const uint64_t bl = node_minlen(k, cx);
            bool rec = recursive_ref(k, cx);
            if (k.kind == Node::CAT && !k.cap)
                for (const Node &item : k.kids) rec = rec || recursive_ref(item, cx);
            if (first || (!rec && bl < t)) t = bl;
            first = false;
        }
        return t;
    }
    case Node::CAT: {
        uint64_t t = 0;
        for (const Node &k : n.kids) t = std::min(cap, t + node_minlen(k, cx));
        return t;
    }
    case Node::ALT: {
        // find_minlength's rule for a branch that holds a recursive reference at its own level: it sets the length only
        // when it is the first branch -- ((ab|\1?)c) counts 3, (a|\1b) counts 1
        bool first = true;
        uint64_t t = cap;
        for (const Node &k : n.kids) {
            const uint64_t bl = node_minlen(k, cx);
            bool rec = recursive_ref(k, cx);
            if (k.kind == Node::CAT && !k.cap)
                for (const Node &item : k.kids) rec = rec || recursive_ref(item, cx);
            if (first || (!rec && bl < t)) t = bl;
            first = false;
        }
        return t;
    }
    case Node::REP: {
        const Node &k = n.kids[0];
        const uint64_t m1 = node_minlen(k, cx);
        // A counted repeat is compiled into copies; a group that refers to itself finds, from its second copy on, the
        // FIRST copy's bracket: there the reference counts what that copy counts (and is no recursion any more).
        // (every capturing group inside the repeated item, whether the reference stands inside that group or next to it)
        if (n.min >= 2 && !cx.plain && (has_kind(k, Node::BACKREF) || has_kind(k, Node::RECURSE))) {
            std::vector<const Node *> groups;
            collect_groups(k, groups);
            for (const Node *g : groups) {
                if ((size_t)g->group >= cx.later_d.size()) cx.later_d.resize((size_t)g->group + 1, 0);
                cx.later_d[(size_t)g->group]++;
            }
            const uint64_t m2 = node_minlen(k, cx);
            for (const Node *g : groups) cx.later_d[(size_t)g->group]--;
            return std::min(cap, m1 + (uint64_t)(n.min - 1) * m2);
        }
        return std::min(cap, (uint64_t)n.min * m1);
    }
    }
    return 0;
}
uint64_t node_minlen(const Node &n)
{
    MinCtx cx{&n, {}, {}, {}, false};
    return node_minlen(n, cx);
}
uint64_t node_true_minlen(const Node &n)
{
    MinCtx cx{&n, {}, {}, {}, true};
    return node_minlen(n, cx);
}

} // namespace


// How far in front of a match start p the pattern can look: the device's verdict AT p, reached with the chunk's real bytes in
// front of p, is what pcre_exec finds with the subject starting at s (src/grab.cc:178) for every s <= p - reach.
// \b \B (?m)^ look at one byte; ^ \A \G hold at s itself only (reach 1: the host owns p == s); a look-behind steps back by its
// length and may look further back from there.
static uint32_t look_reach(const Node &n)
{
    uint32_t r = 0;
    if (n.kind == Node::ASSERT && (n.acode == A_BOS || n.acode == A_MBOL || n.acode == A_WB || n.acode == A_NWB)) r = 1;
    for (const Node &k : n.kids) r = std::max(r, look_reach(k));
    if (n.kind == Node::LOOK && n.behind) {
        uint32_t len = 0;
        std::function<long(const Node &)> flen = [&](const Node &x) -> long { // the longest fixed length among its top-level alternatives
            switch (x.kind) {
            case Node::SET: return 1;
            case Node::ASSERT:
            case Node::LOOK: return 0;
            case Node::ATOMIC: return flen(x.kids[0]);
            case Node::CAT: {
                long t = 0;
                for (const Node &k : x.kids) t += std::max(0l, flen(k));
                return t;
            }
            case Node::ALT: {
                long t = 0;
                for (const Node &k : x.kids) t = std::max(t, flen(k));
                return t;
            }
            case Node::REP: return std::max(0l, flen(x.kids[0])) * (long)std::min<uint32_t>(x.max, 4096u);
            default: return 4096;
            }
        };
        len = (uint32_t)std::min<long>(flen(n.kids[0]), 1l << 20);
        r += len;
    }
    return r;
}

// reach == 1: the bytes b for which "b stands in front of p" and "nothing stands in front of p" are the same thing to every
// assertion of the pattern that looks there -- a non-word byte for \b and \B, a newline for (?m)^, a byte outside its class
// for a one-byte look-behind (positive: fails either way; negative: holds either way); none for ^ \A \G.  When the byte in
// front of a restart position is one of them, the device's verdict AT the restart position is pcre_exec's too.
static void start_like_bytes(const Node &n, ByteSet &e)
{
    if (n.kind == Node::ASSERT) {
        ByteSet keep;
        bool narrows = true;
        switch (n.acode) {
        case A_WB:
        case A_NWB: keep = set_not(set_word()); break;
        case A_MBOL: keep.set('\n'); break;
        case A_BOS: break; // (nothing is like the subject start)
        default: narrows = false; break;
        }
        if (narrows) e = set_and(e, keep);
    }
    if (n.kind == Node::LOOK && n.behind) {
        const Node *b = &n.kids[0];
        while ((b->kind == Node::CAT || b->kind == Node::ALT) && b->kids.size() == 1 && !b->cap) b = &b->kids[0];
        if (b->kind == Node::SET) e = set_and(e, set_not(b->set));
        else e = ByteSet();
        return; // (what stands inside the look-behind looks further back: such a pattern's reach is > 1 anyway)
    }
    for (const Node &k : n.kids) start_like_bytes(k, e);
}

// One-byte look-behinds at the very head of the pattern -- (?<=\$)\d+ , (?<![A-Za-z0-9_])[A-Z]{2,} , (?<=\()[^()\n]+(?=\)) -- say
// which bytes may stand in front of a match that has a byte in front of it at all: a condition for the START windows' leading
// context position (the unfolder leaves look-arounds to the matcher: without this such a pattern's windows begin with its first
// consuming item, and [^()\n]+ lists nineteen bytes in twenty).  The set of allowed preceding bytes; all of them if there is none.
static ByteSet leading_behind(const Node &root)
{
    ByteSet allow = set_all();
    const Node *seq = &root;
    while ((seq->kind == Node::CAT && seq->kids.size() == 1 && !seq->cap) || (seq->kind == Node::ALT && seq->kids.size() == 1)) seq = &seq->kids[0];
    if (seq->kind != Node::CAT) return allow;
    for (const Node &k : seq->kids) {
        if (k.kind == Node::ASSERT) continue; // zero-width: what follows still stands at the match start
        if (k.kind != Node::LOOK) break;
        if (!k.behind) continue;              // a look-ahead: zero-width too
        const Node *b = &k.kids[0];
        while ((b->kind == Node::CAT || b->kind == Node::ALT) && b->kids.size() == 1 && !b->cap) b = &b->kids[0];
        if (b->kind != Node::SET) continue;   // (longer or branching bodies: no condition taken from them)
        allow = set_and(allow, k.neg ? set_not(b->set) : b->set);
    }
    return allow;
}

static bool has_keep(const Node &n)
{
    if (n.kind == Node::ASSERT && n.acode == A_KEEP) return true;
    for (const Node &k : n.kids)
        if (has_keep(k)) return true;
    return false;
}

// the bytes a match can begin with (nullable: it may begin without consuming one -- then nothing is known)
struct vm_first_t {
    ByteSet set;
    bool nullable = false;
};
static vm_first_t vm_first_bytes(const Node &n)
{
    vm_first_t f;
    switch (n.kind) {
    case Node::SET: f.set = n.set; break;
    case Node::CAT:
        f.nullable = true;
        for (const Node &k : n.kids) {
            const vm_first_t g = vm_first_bytes(k);
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
            const vm_first_t g = vm_first_bytes(k);
            f.set.merge(g.set);
            f.nullable = f.nullable || g.nullable;
        }
        break;
    case Node::REP:
        if (n.max == 0) {
            f.nullable = true;
            break;
        }
        f = vm_first_bytes(n.kids[0]);
        if (n.min == 0) f.nullable = true;
        break;
    case Node::ATOMIC: f = vm_first_bytes(n.kids[0]); break;
    case Node::ASSERT:
    case Node::LOOK: // consume nothing: what stands behind them decides
        f.nullable = true;
        break;
    default: // what a reference repeats, a condition picks or a call matches is not known here
        f.set.negate();
        f.nullable = true;
        break;
    }
    return f;
}

// DevProgram::vm_pair: the host matcher's verdict on every two-byte prefix a match could begin with (first the 256 one-byte
// prefixes: a first byte no match can begin with spares its 256 pairs).  ~20 000 short matcher runs, a few milliseconds.
static void fill_vm_pairs(Database &db)
{
    DevProgram &pg = db.prog;
    memset(pg.vm_pair, 0, sizeof pg.vm_pair);
    pg.vm_pair_ok = 0;
    if (getenv("GSCAN_NO_VM_PAIRS")) return;
    uint8_t t[2];
    for (unsigned b0 = 0; b0 < 256; b0++) {
        t[0] = (uint8_t)b0;
        if (!tree_prefix_viable(db, t, 1)) continue;
        for (unsigned b1 = 0; b1 < 256; b1++) {
            t[1] = (uint8_t)b1;
            if (tree_prefix_viable(db, t, 2)) pg.vm_pair[(b0 << 8 | b1) >> 5] |= 1u << (b1 & 31);
        }
    }
    pg.vm_pair_ok = 1;
    bool gapped = false; // (2: no gapped alternative -- a hit stands for a match AT the hit and nothing else: the kernel may drop it
                         // on the table's word alone, without calling vm_keep_hit)
    for (uint32_t i = 0; i < pg.n_alts; i++) gapped = gapped || pg.alt_gap_cls[i] != 0xffu;
    if (!gapped) pg.vm_pair_ok = 2;
}

// what a subroutine call (?g) runs: the capturing groups of the finished tree by number
static void index_groups(Database &db, int n_groups)
{
    db.group_nodes.assign((size_t)n_groups + 1, nullptr);
    db.group_nodes[0] = db.tree.get();
    for (int g = 1; g <= n_groups; g++) db.group_nodes[(size_t)g] = find_group(*db.tree, g);
}

int compile_pattern(const char *pat, size_t len, unsigned flags, Database &db, std::string &why)
{
    std::vector<Seq> seqs;
    Seq literal_seq;
    Node root;
    int n_groups = 0;
    bool has_backref = false, has_accept = false;
    if (flags & GSCAN_LITERAL) {
        Seq &s = literal_seq;
        root.kind = Node::CAT;
        for (size_t k = 0; k < len; k++) {
            ByteSet b;
            b.set((unsigned char)pat[k]);
            s.win.push_back(b);
            Node leaf;
            leaf.set = b;
            root.kids.push_back(leaf);
        }
        if (s.win.size() > kWinCap) { // a long literal: the kernels look for its first kWinCap bytes, the matcher compares the rest
            s.win.resize(kWinCap);
            s.frozen = s.inexact = true;
        }
    } else {
        Parser ps{(const unsigned char *)pat, len};
        ps.pcre_checked = (flags & GSCAN_PCRE_CHECKED) != 0;
        if (!ps.parse(root)) {
            why = ps.why;
            return ps.rc;
        }
        n_groups = ps.ngroups;
        has_backref = ps.has_backref;
        has_accept = ps.has_accept;
        bool quirk = false;
        ends_in_greedy_repeat(root, false, quirk);
        if (!quirk && has_kind(root, Node::ATOMIC)) { // (a pattern with \R or \X: once more, lazy repeats counting as well)
            g_lazy_counts = true;
            ends_in_greedy_repeat(root, false, quirk);
            g_lazy_counts = false;
        }
        if (quirk) {
            why = "possessive group repeat, \\R or \\X behind a greedy repeat (libpcre's auto-possessification treats it inconsistently)";
            return 1;
        }
        if (cond_holds_circumflex(root, false)) {
            why = "^ inside a conditional group (libpcre's anchoring analysis reads it as the head of the pattern)";
            return 1;
        }
        if (ps.has_recursion && begins_with_optional_call(root, false)) {
            why = "a match can begin inside an optional subroutine call (libpcre's start-up optimisation leaves the call out)";
            return 1;
        }
        if (ps.has_recursion && called_group_begins_with_repeat(root, root)) {
            why = "a subroutine call to a group that begins with an unbounded repeat of one class (libpcre's JIT loses matches behind a failed attempt that made the call)";
            return 1;
        }
        if (head_is_lookahead_for_dot_star(root)) {
            why = "a look-ahead for .* at the head of the pattern (libpcre then tries the pattern at line starts only)";
            return 1;
        }
        if (leading_lookahead_sets_first_byte(root)) {
            why = "a leading look-ahead for a literal byte that the pattern itself holds behind an optional item (libpcre then asks for that byte twice)";
            return 1;
        }
        if (lookahead_hands_on_required_byte(root)) {
            why = "a look-ahead at the head of a branch followed by a literal byte it also holds (libpcre hands the assertion's required byte on to the pattern and then asks for it twice)";
            return 1;
        }
        if (lookahead_first_in_group(root, true)) {
            why = "a group that begins with a look-ahead for a literal byte (libpcre's first-byte / required-byte start-up check misfires on it)";
            return 1;
        }
    }

    // PCRE_INFO_MINLENGTH counts every branch, also those whose assertions can never hold (\\z followed by a byte ...):
    // it comes from the parse tree, not from the unfolded paths -- the reference's loop bound and file-skip rule use it
    // (grab.cc:133,175).
    const size_t pcre_min = (size_t)node_minlen(root);
    if (pcre_min > 0 && has_backref && node_true_minlen(root) == 0) {
        // find_minlength skips branches that hold a recursive back reference, so PCRE_INFO_MINLENGTH can be positive for a
        // pattern that does match "" -- (a|\1?)b*: files are not skipped (Q2 does not apply), and the reference's loop either
        // stops at its first pcre_exec (the empty match has set a group: rc == 0) or never advances
        why = "the pattern can match the empty string although PCRE_INFO_MINLENGTH is positive (a recursive back reference)";
        return 1;
    }
    if (pcre_min == 0 || has_accept) { // can match the empty string (or holds (*ACCEPT): no minimum length either): PCRE_INFO_MINLENGTH == -1 and every file is skipped
        db = Database();
        db.tree = std::make_shared<Node>(std::move(root));
        index_groups(db, n_groups);
    index_groups(db, n_groups);
        db.id = g_next_id.fetch_add(1);
        memset(&db.prog, 0, sizeof db.prog);
        db.tier = GSCAN_TIER_NULL;
        db.minlen = -1;
        return 0;
    }

    // Unfold into paths, resolve their assertions, drop duplicates.  A pattern that unfolds into too many paths gets
    // further attempts in which every path is cut off ("frozen") once its window has `cap` bytes: short prefixes of what
    // a match must begin with, fewer of them, and the matcher confirms (Database::exact == false).
    auto build = [&](size_t cap, std::vector<Seq> &seqs) -> int { // 0 ok, 1 refused, 2 too many paths
        seqs.clear();
        if (flags & GSCAN_LITERAL) {
            seqs.push_back(literal_seq);
        } else {
            Unfold uf;
            uf.cap_len = cap;
            uf.root = &root;
            if (!uf.run(root, seqs)) {
                why = uf.why;
                return uf.overflow ? 2 : uf.rc;
            }
        }
        // a lazy repeat at the very end takes its minimum, a possessive one behaves like a greedy one
        for (Seq &s : seqs)
            if (s.has_tail && s.tail_mode == 1) s.has_tail = false;
        // an atom that can match no byte at all makes its alternative unmatchable
        for (const Seq &s : seqs) {
            for (const ByteSet &b : s.win)
                if (b.count() == 0) {
                    why = "empty character class";
                    return 1;
                }
        }
        for (const Seq &s : seqs) {
            for (const ByteSet &b : s.pwin)
                if (b.count() == 0) {
                    why = "empty character class";
                    return 1;
                }
            if (s.gapped && s.gap.count() == 0) {
                why = "empty character class";
                return 1;
            }
        }
        // assertions -> one byte of context at each end (or decided / narrowed / split on the spot)
        {
            Resolver rs;
            bool any = false;
            for (Seq &s : seqs) {
                any = any || !s.asserts.empty() || !s.p_asserts.empty();
                std::stable_sort(s.asserts.begin(), s.asserts.end(), [](const std::pair<uint32_t, int> &x, const std::pair<uint32_t, int> &y) { return x.first < y.first; });
                if (!rs.step(s, 0, false, false)) {
                    why = rs.why;
                    return 1;
                }
            }
            if (any) seqs.swap(rs.out); // (possibly nothing is left: the pattern's assertions can never hold)
        }
        for (const Seq &s : seqs) {
            if (s.has_tail && s.tail.count() == 0) {
                why = "empty character class";
                return 1;
            }
        }
        // a later duplicate of an alternative can never be the first one to match
        {
            std::vector<Seq> uniq;
            for (Seq &s : seqs) {
                bool dup = false;
                for (const Seq &u : uniq)
                    if (u.win.size() == s.win.size() && u.has_tail == s.has_tail &&
                        // (a path that is about to be dropped as dead -- it begins with a reference to a group nothing has set --
                        // must not stand in for a live one with the same window: ( ?\\3?)\\2x lost its empty-group path that way)
                        (u.needs_cap && !u.cap) == (s.needs_cap && !s.cap) &&
                        (!u.has_tail || (u.tail == s.tail && u.tail_extra == s.tail_extra)) && u.pre == s.pre && u.post == s.post &&
                        u.pre_start == s.pre_start && u.post_end == s.post_end && u.post_final_nl == s.post_final_nl &&
                        u.gapped == s.gapped && !s.gapped && // gapped paths are kept as they are: their order inside a repeat instance matters
                        std::equal(u.win.begin(), u.win.end(), s.win.begin()))
                        dup = true;
                if (!dup) uniq.push_back(std::move(s));
            }
            seqs.swap(uniq);
        }

        size_t total = 0;
        for (const Seq &s : seqs) total += s.win.size() + s.pwin.size() + 2;
        if (seqs.size() > (size_t)kMaxAlts || total > (size_t)kAltWindowBytes) {
            why = "pattern unfolds into too many alternatives";
            return 2;
        }
        return 0;
    };
    {
        int rc = 2;
        for (size_t cap : {(size_t)SIZE_MAX, (size_t)12, (size_t)8, (size_t)5, (size_t)3, (size_t)2, (size_t)1}) {
            rc = build(cap, seqs);
            if (rc != 2 || (flags & GSCAN_LITERAL)) break;
        }
        if (rc != 0) return rc == 2 ? 1 : rc;
    }

    bool exact = true;
    for (const Seq &s : seqs) exact = exact && !s.inexact && !s.frozen;
    {
        // a path that starts with a back reference, no group closed before it: the reference fails (unset group), the path is dead
        std::vector<Seq> alive;
        for (Seq &s : seqs)
            if (!(s.needs_cap && !s.cap && s.win.empty() && !s.gapped)) alive.push_back(std::move(s));
        seqs.swap(alive);
    }
    for (const Seq &s : seqs)
        if (s.win.empty() && !s.gapped) {
            why = "nothing fixed to look for in front of a repeated group or a back reference";
            return 1;
        }

    db = Database();
    db.exact = exact;
    db.n_groups = n_groups;
    db.has_backref = has_backref;
    db.tree = std::make_shared<Node>(std::move(root));
    index_groups(db, n_groups);
    db.id = g_next_id.fetch_add(1);
    memset(&db.prog, 0, sizeof db.prog);
    db.vm_ok = vm_compile(*db.tree, db.n_groups, db.has_backref, db.prog.vm); // the tree as a program for the device's VM (vm.h)

    const size_t minm = pcre_min;
    if (seqs.empty()) { // no path can ever match (a\\Ab, x^y ...): pcre_exec finds nothing, and neither is there anything to scan for
        db.minlen = (int)minm;
        db.tier = GSCAN_TIER_ANCHORED;
        return 0;
    }

    // class table + alternatives
    auto intern = [&](const ByteSet &b) -> int {
        for (size_t c = 0; c < db.classes.size(); c++)
            if (db.classes[c] == b) return (int)c;
        if ((int)db.classes.size() >= kMaxClasses) return -1;
        db.classes.push_back(b);
        return (int)db.classes.size() - 1;
    };
    for (const Seq &s : seqs) {
        AltSeq a;
        for (const ByteSet &b : s.win) {
            const int id = intern(b);
            if (id < 0) {
                why = "too many distinct classes";
                return 1;
            }
            a.window.push_back((uint8_t)id);
        }
        for (const ByteSet &b : s.pwin) {
            const int id = intern(b);
            if (id < 0) {
                why = "too many distinct classes";
                return 1;
            }
            a.pwindow.push_back((uint8_t)id);
        }
        a.gapped = s.gapped;
        a.gap = s.gap;
        a.gap_mode = s.gap_mode;
        a.gap_id = s.gap_id;
        a.has_tail = s.has_tail;
        a.tail = s.tail;
        a.tail_extra = s.has_tail ? s.tail_extra : 0;
        a.captures = s.cap;
        a.pre = s.pre;
        a.pre_start = s.pre_start;
        a.post = s.post;
        a.post_end = s.post_end;
        a.post_final_nl = s.post_final_nl;
        db.alts.push_back(std::move(a));
    }
    db.minlen = (int)minm;

    // Device windows: the alternative's window, plus one context position in front (behind) when ANY alternative
    // looks at the byte before (after) its match -- its own condition there, "any byte" for the others.
    auto class_id = [&](const ByteSet &b) -> int {
        for (size_t c = 0; c < db.classes.size(); c++)
            if (db.classes[c] == b) return (int)c;
        if ((int)db.classes.size() >= kMaxClasses) return -1;
        db.classes.push_back(b);
        return (int)db.classes.size() - 1;
    };
    // Two ways to put the alternatives in front of the kernels:
    //  * HIT windows (rounds 1-5): a plain alternative's window with its context bytes; a gapped alternative  P . C{1,} . R  by
    //    one repeat byte + the rest, C . R -- the kernels list where the part BEHIND the repeat begins and the host walks the run
    //    back to the match start (matcher.cc, next_gapped);
    //  * START windows (round 6, Database::resolve): every alternative by what a match must BEGIN with -- the leading context
    //    byte, then the window, or for a gapped alternative P and one repeat byte, P . C -- so that every listed offset is a
    //    possible match START and nothing else; the device's resolve pass (k_resolve) runs the pattern's VM program there with
    //    the chunk's real bytes in front of it and leaves the list of MATCHES with their ends.  No trailing context byte: a
    //    window that ends with the chunk is listed like any other, the VM knows where the chunk ends.
    struct Windows {
        std::vector<std::vector<uint8_t>> w;
        bool pre = false, post = false, can_hit = false;
        size_t min_dev = SIZE_MAX, total = 0;
        double density = 0; // expected hits per text byte: every window's classes priced by how often their members turn up in text (class_prob)
    };
    const ByteSet lead = leading_behind(*db.tree);
    auto make_windows = [&](bool starts, Windows &out) -> int { // 0 ok, 1 refused (why is set)
        for (const AltSeq &a : db.alts) {
            out.pre = out.pre || (a.has_pre() && (starts || !a.gapped)); // (hit windows: a gapped path's start is not where its device window is)
            out.post = out.post || (!starts && a.has_post());
        }
        if (starts && lead.count() != 256) out.pre = true;
        for (const AltSeq &a : db.alts) {
            const ByteSet pre = starts ? set_and(a.pre, lead) : a.pre;
            if (starts && pre.count() == 0) continue; // (can only sit at the subject start: the host's own test there finds it)
            std::vector<uint8_t> w;
            if (out.pre) {
                const int id = class_id(!starts && a.gapped ? set_all() : pre);
                if (id < 0) return 1;
                w.push_back((uint8_t)id);
            }
            if (starts && a.gapped) w.insert(w.end(), a.pwindow.begin(), a.pwindow.end());
            if (a.gapped) { // one repeat byte: behind P (start windows), in front of the rest (hit windows)
                const int id = class_id(a.gap);
                if (id < 0) return 1;
                w.push_back((uint8_t)id);
            }
            if (!(starts && a.gapped)) w.insert(w.end(), a.window.begin(), a.window.end());
            if (out.post) {
                const int id = class_id(a.post);
                if (id < 0) return 1;
                w.push_back((uint8_t)id);
            }
            if (w.size() > (size_t)kMaxWindow) {
                if (!starts) return 2;
                w.resize((size_t)kMaxWindow); // (a start window is a necessary condition: any prefix of it is one too)
            }
            out.can_hit = out.can_hit || ((a.gapped || a.pre.count() > 0) && a.post.count() > 0);
            out.min_dev = std::min(out.min_dev, w.size());
            out.total += w.size();
            double prod = 1;
            for (uint8_t c : w) prod *= class_prob(db.classes[c]);
            out.density += prod;
            out.w.push_back(std::move(w));
        }
        return 0;
    };
    Windows hitw, startw;
    if (int rc = make_windows(false, hitw)) {
        why = rc == 2 ? "window longer than the engine supports" : "too many distinct classes";
        return 1;
    }
    if (hitw.total > (size_t)kAltWindowBytes) {
        why = "pattern unfolds into too many alternatives";
        return 1;
    }
    // The resolve pass: for every pattern the VM can run whole (no subroutine calls, within its program limits; no \K: where
    // a match is REPORTED to start is not something the VM tracks) -- unless it is the one-window kind the kernels and the
    // per-record passes of rounds 1-3 already settle on their own (literals, the identifier regex: k_ends, k_lines, the host's
    // two-array-read walk), or its start windows would list a large part of the text where its hit windows list next to
    // nothing (.*needle: every byte of a line can begin a match, but only the needle is worth looking for).
    db.reach = look_reach(*db.tree);
    {
        const bool simple = db.exact && db.alts.size() == 1 && !db.alts[0].gapped && !hitw.pre && !hitw.post;
        bool want = hitw.can_hit && db.vm_ok && !simple && !has_keep(*db.tree) && db.reach <= 255u && !getenv("GSCAN_NO_RESOLVE");
        if (want && make_windows(true, startw) != 0) want = false;
        if (want && (startw.w.empty() || startw.total > (size_t)kAltWindowBytes)) want = false;
        const double dmax = getenv("GSCAN_RESOLVE_MAX_DENSITY") ? atof(getenv("GSCAN_RESOLVE_MAX_DENSITY")) : 0.35;
        if (want && !(startw.density <= dmax || startw.density <= 4.0 * hitw.density)) want = false;
        // Start windows that list a large part of the text, of a pattern without a gapped alternative that K3 can confirm in its
        // own cold path (the two-byte table in front of the VM drops most hits before they become records: (\w)\1{3,}x|foobardoes(?=not),
        // three offsets in four are hits, 125 matches per 8 GiB): that path stays -- 0.58 s against 0.90 s for 8 GiB,
        // profiles/r06_e_density_ab.txt.  (With a gapped alternative the same path walks every run back from every hit and the
        // host matcher follows: \w+(?=\() took 15.6 s there against 1.8 s through the resolve pass.)
        if (want && startw.density > dmax && !db.exact && !hitw.pre && vm_independent_of_subject_start(*db.tree)) {
            bool gapped = false;
            for (const AltSeq &a : db.alts) gapped = gapped || a.gapped;
            if (!gapped) want = false;
        }
        db.resolve = want;
        // (Round 6, measured and taken out again: start windows that list a large part of the text -- \w+(?=\() , three bytes in
        // four -- put to the VM in K3's own cold path first, so that only match starts become records.  The survivor queue runs the
        // VM worse than k_resolve's work list does: 341 ms of kernels for 4 GiB against 125 ms, the same wall clock --
        // profiles/r06_v_dense_candidates.txt.)
    }
    const Windows &win = db.resolve ? startw : hitw;
    db.dev_pre = win.pre;
    db.dev_post = win.post;
    db.dev_windows = win.w;
    const size_t min_dev = win.min_dev;
    const bool can_hit = hitw.can_hit; // some alternative can match away from the subject start and the chunk end
    {
        const vm_first_t f = vm_first_bytes(*db.tree);
        db.first = f.set;
        db.first_ok = !f.nullable;
        db.start_like = ByteSet();
        if (db.reach == 1) {
            db.start_like = set_all();
            start_like_bytes(*db.tree, db.start_like);
        }
    }

    if (db.exact && db.alts.size() == 1 && !db.alts[0].gapped && !db.dev_pre && !db.dev_post && db.dev_windows[0] == db.alts[0].window) {
        const std::vector<uint8_t> &w = db.dev_windows[0];
        for (size_t i = 0; i + 1 < w.size() && !db.solitary; i++) {
            bool common = false;
            for (int k = 0; k < 8; k++) common = common || (db.classes[w[i]].w[k] & db.classes[w[i + 1]].w[k]) != 0;
            db.solitary = !common;
        }
    }

    DevProgram &pg = db.prog;
    const std::vector<uint8_t> &w0 = db.dev_windows[0];
    const size_t m = w0.size();
    pg.m = (uint32_t)min_dev;
    pg.report_shift = db.dev_pre ? 1u : 0u;
    pg.n_classes = (uint32_t)db.classes.size();
    for (size_t c = 0; c < db.classes.size(); c++) memcpy(pg.cls_bits[c], db.classes[c].w, 32);
    pg.n_alts = (uint32_t)db.dev_windows.size(); // (start windows: the alternatives that can only sit at the subject start have none)
    {
        size_t at = 0;
        for (size_t i = 0; i < db.dev_windows.size(); i++) {
            const std::vector<uint8_t> &w = db.dev_windows[i];
            pg.alt_off[i] = (uint16_t)at;
            pg.alt_len[i] = (uint16_t)w.size();
            pg.alt_bucket[i] = (uint8_t)(i % kK3Buckets);
            memcpy(pg.alt_window + at, w.data(), w.size());
            at += w.size();
            // (a gapped alternative's HIT window begins with its repeat byte, behind a context position if there is one; a start
            // window stands for a match AT the hit, whatever the alternative)
            pg.alt_gap_cls[i] = (uint8_t)0xff;
            pg.alt_plen[i] = 0;
            if (!db.resolve) {
                const AltSeq &alt = db.alts[i];
                pg.alt_gap_cls[i] = alt.gapped && !db.dev_pre && !w.empty() ? w[0] : (uint8_t)0xff;
                pg.alt_plen[i] = (uint16_t)alt.pwindow.size();
            }
        }
    }
    pg.resolve = db.resolve ? 1u : 0u;
    pg.reach = db.reach;
    pg.est_permille = (uint32_t)std::min(1000.0, std::ceil(1000.0 * (db.resolve ? startw.density : hitw.density)));
    pg.first_ok = db.first_ok ? 1u : 0u;
    memcpy(pg.first_bits, db.first.w, 32);
    memcpy(pg.start_like_bits, db.start_like.w, 32);
    if (!can_hit) { // ^foo, foo$ and the like: a match can only sit at the subject start / chunk end, which is the host's job
        db.tier = GSCAN_TIER_ANCHORED;
        return 0;
    }

    // K3 filter: the kK3Depth window positions from k3_off on, per bucket.  The offset is common to
    // all alternatives (a hit at text position q means a window start at q - k3_off); it is chosen
    // to make the filter as selective as possible.  Positions beyond an alternative's end accept
    // any byte.
    {
        const size_t max_off = min_dev > (size_t)kK3Depth ? min_dev - kK3Depth : 0;
        double best = -1;
        size_t best_off = 0;
        for (size_t off = 0; off <= max_off; off++) {
            double score = 0;
            for (const std::vector<uint8_t> &w : db.dev_windows) {
                double prod = 1;
                for (int k = 0; k < kK3Depth; k++)
                    prod *= off + k < w.size() ? db.classes[w[off + k]].count() / 256.0 : 1.0;
                score += prod;
            }
            if (best < 0 || score < best) {
                best = score;
                best_off = off;
            }
        }
        pg.k3_off = (uint32_t)best_off;
        for (int b = 0; b < 256; b++) {
            uint32_t e = 0;
            for (size_t i = 0; i < db.dev_windows.size(); i++) {
                const std::vector<uint8_t> &w = db.dev_windows[i];
                for (int k = 0; k < kK3Depth; k++) {
                    const size_t pos = best_off + k;
                    if (pos >= w.size() || db.classes[w[pos]].test((unsigned)b))
                        e |= 1u << (8 * k + pg.alt_bucket[i]);
                }
            }
            pg.k3_table[b] = e;
        }
        // confirm tables: the same bucket sets for the first kK3Confirm window positions
        for (int k = 0; k < kK3Confirm; k++)
            for (int b = 0; b < 256; b++) {
                uint32_t e = 0;
                for (size_t i = 0; i < db.dev_windows.size(); i++) {
                    const std::vector<uint8_t> &w = db.dev_windows[i];
                    if ((size_t)k >= w.size() || db.classes[w[(size_t)k]].test((unsigned)b)) e |= 1u << pg.alt_bucket[i];
                }
                pg.k3_pos[k][b] = (uint8_t)e;
            }
        pg.k3_confirm_exact = db.dev_windows.size() <= (size_t)kK3Buckets;
        for (size_t i = 0; i < db.dev_windows.size(); i++) {
            if (db.dev_windows[i].size() > (size_t)kK3Confirm) pg.k3_confirm_exact = 0;
            if (i < (size_t)kK3Buckets) pg.k3_blen[i] = (uint8_t)std::min<size_t>(db.dev_windows[i].size(), 255);
        }
    }

    // The line-extent pass: a match then lies inside one line, and "which matches get printed" is decided line by line.
    if (db.exact && db.alts.size() == 1 && !db.alts[0].gapped && !db.dev_pre && !db.dev_post) {
        const AltSeq &a0 = db.alts[0];
        bool ok = !(a0.has_tail && a0.tail.test('\n')) && !a0.captures; // (a match that sets a capturing group ends the chunk: grab.cc:171,179)
        for (uint8_t c : a0.window) ok = ok && !db.classes[c].test('\n');
        pg.lines_ok = ok;
        if (a0.has_tail) {
            memcpy(pg.tail_bits, a0.tail.w, 32);
            pg.tail_extra = a0.tail_extra;
        }
        // the match-end pass: the byte that stops the (unbounded) tail cannot begin a window
        if (a0.has_tail && a0.tail_extra == UINT32_MAX && !a0.captures && !a0.window.empty()) {
            bool sub = true;
            const ByteSet &w0 = db.classes[a0.window[0]];
            for (unsigned b = 0; b < 256 && sub; b++) sub = !w0.test(b) || a0.tail.test(b);
            pg.ends_ok = sub;
        }
    }

    // Inexact patterns: the device confirms its own candidates with the VM (vm.h) where one verdict per offset serves every
    // restart position, i.e. the pattern never looks behind the match start.  That is K3's cold path.
    const bool vm_dev = !db.resolve && !db.exact && db.vm_ok && !db.dev_pre && vm_independent_of_subject_start(*db.tree) && !getenv("GSCAN_NO_VM");

    // several alternatives: the bucket filter is the one kernel that takes them -- unless they all look the same to the device
    // ([0-9]+\.[0-9]+ unfolds into two alternatives over the one window [0-9]\.[0-9]; what tells them apart is the host's
    // business): one window is K1's or K2's
    bool one_window = true;
    for (const std::vector<uint8_t> &w : db.dev_windows) one_window = one_window && w == db.dev_windows[0];
    if (getenv("GSCAN_SAME_WINDOW_K3")) one_window = db.dev_windows.size() == 1; // (A/B switch: the round-2 choice)
    if (!one_window) {
        db.tier = GSCAN_TIER_BUCKET;
        pg.vm_filter = vm_dev;
        if (vm_dev) fill_vm_pairs(db);
        return 0;
    }

    bool literal = true;
    std::vector<int> lit(m, -1);
    for (size_t k = 0; k < m; k++) {
        lit[k] = db.classes[w0[k]].single();
        if (lit[k] < 0) literal = false;
    }
    pg.is_literal = literal;
    for (size_t k = 0; k < m; k++) pg.window[k] = literal ? (uint8_t)lit[k] : w0[k];

    // K1 anchor: longest (<=4) run of single-byte positions, rarest bytes first
    int best_len = 0, best_off = 0, best_score = 1 << 30;
    for (size_t k = 0; k < m; k++) {
        int run = 0, score = 0;
        while (run < 4 && k + run < m && lit[k + run] >= 0) {
            score += byte_rank((unsigned)lit[k + run]);
            run++;
            // every prefix length is a candidate; longer always wins
            if (run > best_len || (run == best_len && score < best_score)) {
                best_len = run;
                best_off = (int)k;
                best_score = score;
            }
        }
    }
    if (best_len > 0) {
        uint32_t v = 0;
        for (int k = 0; k < best_len; k++) v |= (uint32_t)lit[best_off + k] << (8 * k);
        pg.anchor = v;
        pg.anchor_len = (uint32_t)best_len;
        pg.anchor_off = (uint32_t)best_off;
        pg.anchor_mask = best_len == 4 ? 0xffffffffu : ((1u << (8 * best_len)) - 1u);
    }

    // K2 program: runs of equal class ids
    bool k2 = db.classes.size() <= (size_t)kK2MaxClasses && m <= (size_t)kK2MaxWindow;
    if (k2) {
        uint32_t nr = 0;
        for (size_t k = 0; k < m;) {
            size_t e = k;
            while (e < m && w0[e] == w0[k]) e++;
            if (nr >= (uint32_t)kK2MaxRuns) {
                k2 = false;
                break;
            }
            pg.run_cls[nr] = w0[k];
            pg.run_len[nr] = (uint8_t)(e - k);
            pg.run_off[nr] = (uint8_t)k;
            nr++;
            k = e;
        }
        pg.nruns = k2 ? nr : 0;
        if (k2)
            for (int b = 0; b < 256; b++) {
                uint32_t bits = 0;
                for (size_t c = 0; c < db.classes.size(); c++)
                    if (db.classes[c].test((unsigned)b)) bits |= 1u << (8 * c);
                pg.k2_table[b] = bits;
            }
    }

    // tier choice: a >=3-byte literal anchor makes K1 the cheapest kernel (no LDS lookups);
    // otherwise the class-run kernel if the window fits it; otherwise the bucket filter on the
    // window's most selective 4 positions (any class sequence fits it).
    if (literal || best_len >= 3)
        db.tier = GSCAN_TIER_LITERAL;
    else if (k2)
        db.tier = GSCAN_TIER_CLASSRUN;
    else
        db.tier = GSCAN_TIER_BUCKET;
    // (a single-alternative pattern that would have gone to K2 goes to K3 for the VM; a literal anchor of 3+ bytes hits
    // rarely enough for the host: K1 stays)
    if (vm_dev && db.tier != GSCAN_TIER_LITERAL) {
        pg.vm_filter = 1;
        db.tier = GSCAN_TIER_BUCKET;
        fill_vm_pairs(db);
    }
    return 0;
}

} // namespace gscan
