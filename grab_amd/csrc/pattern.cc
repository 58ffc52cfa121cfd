// pattern.cc -- see pattern.h.  Host only; no HIP.
#include "pattern.h"

#include <atomic>
#include <cctype>
#include <cstring>

#include "../../include/gscan.h"

namespace gscan {
namespace {

struct Atom {
    ByteSet set;
    uint32_t min = 1, max = 1; // max == UINT32_MAX: unbounded
};

constexpr uint32_t kInf = UINT32_MAX;

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

    // After a backslash (i points at the escape letter).  Yields a set.  in_class
    // changes the meaning of \b and of \1..\7.
    bool escape(ByteSet &out, bool in_class, bool &is_set_escape)
    {
        is_set_escape = false;
        if (eof()) return fail(-1, "\\ at end of pattern");
        int c = p[i++];
        ByteSet s;
        switch (c) {
        case 'd': out = set_digit(); is_set_escape = true; return true;
        case 'D': out = set_not(set_digit()); is_set_escape = true; return true;
        case 'w': out = set_word(); is_set_escape = true; return true;
        case 'W': out = set_not(set_word()); is_set_escape = true; return true;
        case 's': out = set_space(); is_set_escape = true; return true;
        case 'S': out = set_not(set_space()); is_set_escape = true; return true;
        case 'h': out = set_hspace(); is_set_escape = true; return true;
        case 'H': out = set_not(set_hspace()); is_set_escape = true; return true;
        case 'v': out = set_vspace(); is_set_escape = true; return true;
        case 'V': out = set_not(set_vspace()); is_set_escape = true; return true;
        case 'N':
            if (in_class) return fail(-1, "\\N in class");
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
                v = 0; // not a valid \x{..}: falls back to \x with 0 digits
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
            if (!in_class) return fail(1, "back reference / octal escape");
            if (c >= '8') return fail(1, "\\8 / \\9 in class");
            {
                unsigned v = (unsigned)(c - '0');
                int k = 1;
                while (k < 3 && !eof() && p[i] >= '0' && p[i] <= '7') {
                    v = v * 8 + (unsigned)(p[i] - '0');
                    i++;
                    k++;
                }
                if (v > 255) return fail(-1, "octal value too large");
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

    bool bracket(ByteSet &out) // i points just past '['
    {
        ByteSet s;
        bool neg = false;
        if (!eof() && p[i] == '^') {
            neg = true;
            i++;
        }
        bool first = true;
        for (;;) {
            if (eof()) return fail(-1, "missing terminating ] for character class");
            int c = p[i];
            if (c == ']' && !first) {
                i++;
                break;
            }
            first = false;
            ByteSet lo;
            bool lo_is_set = false;
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
                    if (pneg) ps.negate();
                    s.merge(ps);
                    i = j + 2;
                    continue;
                }
                // no terminator: '[' is a plain member
            }
            if (c == '\\') {
                i++;
                if (!eof() && (p[i] == 'Q' || p[i] == 'E')) return fail(1, "\\Q..\\E inside class");
                if (!escape(lo, true, lo_is_set)) return false;
            } else {
                lo.set((unsigned)c);
                i++;
            }
            // range?
            if (!lo_is_set && i + 1 < n && p[i] == '-' && p[i + 1] != ']') {
                size_t save = i;
                i++;
                ByteSet hi;
                bool hi_is_set = false;
                if (p[i] == '\\') {
                    i++;
                    if (!escape(hi, true, hi_is_set)) return false;
                } else if (p[i] == '[' && i + 1 < n && p[i + 1] == ':') {
                    hi_is_set = true; // "a-[:digit:]": '-' is literal, class handled next round
                    i = save;
                    s.merge(lo);
                    s.set('-');
                    i++;
                    continue;
                } else {
                    hi.set(p[i]);
                    i++;
                }
                if (hi_is_set) { // e.g. [a-\d]: '-' literal
                    s.merge(lo);
                    s.set('-');
                    s.merge(hi);
                    continue;
                }
                int l = lo.single(), h = hi.single();
                if (h < l) return fail(-1, "range out of order in character class");
                s.set_range((unsigned)l, (unsigned)h);
                continue;
            }
            s.merge(lo);
        }
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

    bool parse(std::vector<Atom> &atoms)
    {
        bool quoting = false;
        while (!eof()) {
            int c = p[i];
            Atom a;
            if (quoting) {
                if (c == '\\' && i + 1 < n && p[i + 1] == 'E') {
                    quoting = false;
                    i += 2;
                    continue;
                }
                a.set.set((unsigned)c);
                i++;
            } else {
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
                    i++;
                    {
                        bool is_set;
                        if (!escape(a.set, false, is_set)) return false;
                    }
                    break;
                case '.':
                    a.set = set_dot();
                    i++;
                    break;
                case '[':
                    i++;
                    if (!bracket(a.set)) return false;
                    break;
                case '^': return fail(1, "anchor ^");
                case '$': return fail(1, "anchor $");
                case '|': return fail(1, "alternation");
                case '(': return fail(1, "group");
                case ')': return fail(-1, "unmatched parentheses");
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
                    i++;
                }
            }
            // quantifier (a quoted \Q..\E char may be quantified once the quote ended; PCRE
            // applies a quantifier after \E to the last quoted char -- same thing here)
            if (!quoting && !eof()) {
                int q = p[i];
                bool have = false;
                if (q == '*') { a.min = 0; a.max = kInf; i++; have = true; }
                else if (q == '+') { a.min = 1; a.max = kInf; i++; have = true; }
                else if (q == '?') { a.min = 0; a.max = 1; i++; have = true; }
                else if (q == '{') {
                    uint32_t mn, mx;
                    size_t end;
                    if (counted(mn, mx, end)) {
                        if (mx != kInf && mx < mn) return fail(-1, "numbers out of order in {} quantifier");
                        if (mn > 65535 || (mx != kInf && mx > 65535)) return fail(-1, "number too big in {} quantifier");
                        a.min = mn;
                        a.max = mx;
                        i = end;
                        have = true;
                    }
                }
                if (have && !eof()) {
                    int r = p[i];
                    if (r == '?') return fail(1, "lazy quantifier");
                    if (r == '+') return fail(1, "possessive quantifier");
                    uint32_t mn, mx;
                    size_t end;
                    if (r == '*' || (r == '{' && counted(mn, mx, end))) return fail(1, "stacked quantifiers");
                }
            }
            atoms.push_back(a);
        }
        return true;
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

std::atomic<uint64_t> g_next_id{1};

} // namespace

int compile_pattern(const char *pat, size_t len, unsigned flags, Database &db, std::string &why)
{
    std::vector<Atom> atoms;
    if (flags & GSCAN_LITERAL) {
        for (size_t k = 0; k < len; k++) {
            Atom a;
            a.set.set((unsigned char)pat[k]);
            atoms.push_back(a);
        }
    } else {
        Parser ps{(const unsigned char *)pat, len};
        if (!ps.parse(atoms)) {
            why = ps.why;
            return ps.rc;
        }
    }

    // drop {0} atoms; an atom that can match no byte at all makes the pattern unmatchable
    std::vector<Atom> kept;
    for (auto &a : atoms) {
        if (a.max == 0) continue;
        if (a.set.count() == 0) {
            why = "empty character class";
            return 1;
        }
        kept.push_back(a);
    }
    atoms.swap(kept);

    for (size_t k = 0; k + 1 < atoms.size(); k++)
        if (atoms[k].min != atoms[k].max) {
            why = "variable repeat before the last atom";
            return 1;
        }

    db = Database();
    db.id = g_next_id.fetch_add(1);
    memset(&db.prog, 0, sizeof db.prog);

    uint64_t m = 0;
    for (auto &a : atoms) m += a.min;
    if (m == 0) { // can match the empty string: PCRE_INFO_MINLENGTH == -1
        db.tier = GSCAN_TIER_NULL;
        db.minlen = -1;
        return 0;
    }
    if (m > (uint64_t)kMaxWindow) {
        why = "window longer than the engine supports";
        return 1;
    }

    // window + class table
    for (auto &a : atoms) {
        int id = -1;
        for (size_t c = 0; c < db.classes.size(); c++)
            if (db.classes[c] == a.set) id = (int)c;
        if (id < 0 && a.min > 0) {
            if ((int)db.classes.size() >= kMaxClasses) {
                why = "too many distinct classes";
                return 1;
            }
            db.classes.push_back(a.set);
            id = (int)db.classes.size() - 1;
        }
        for (uint32_t r = 0; r < a.min; r++) db.window.push_back((uint8_t)id);
    }
    if (!atoms.empty() && atoms.back().max > atoms.back().min) {
        db.has_tail = true;
        db.tail = atoms.back().set;
        db.tail_extra = atoms.back().max == kInf ? kInf : atoms.back().max - atoms.back().min;
    }
    db.minlen = (int)m;

    DevProgram &pg = db.prog;
    pg.m = (uint32_t)m;
    pg.n_classes = (uint32_t)db.classes.size();
    for (size_t c = 0; c < db.classes.size(); c++) memcpy(pg.cls_bits[c], db.classes[c].w, 32);

    bool literal = true;
    std::vector<int> lit(m, -1);
    for (size_t k = 0; k < m; k++) {
        lit[k] = db.classes[db.window[k]].single();
        if (lit[k] < 0) literal = false;
    }
    pg.is_literal = literal;
    for (size_t k = 0; k < m; k++) pg.window[k] = literal ? (uint8_t)lit[k] : db.window[k];

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
    bool k2 = db.classes.size() <= (size_t)kK2MaxClasses && m <= (uint64_t)kK2MaxWindow;
    if (k2) {
        uint32_t nr = 0;
        for (size_t k = 0; k < m;) {
            size_t e = k;
            while (e < m && db.window[e] == db.window[k]) e++;
            if (nr >= (uint32_t)kK2MaxRuns) {
                k2 = false;
                break;
            }
            pg.run_cls[nr] = db.window[k];
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

    // tier choice: a >=3-byte literal anchor makes K1 the cheapest kernel (no LDS
    // lookups); otherwise the class-run kernel if the window fits it; a weak anchor
    // (1-2 bytes) still scans correctly, just with a busier verify path.
    if (literal || best_len >= 3)
        db.tier = GSCAN_TIER_LITERAL;
    else if (k2)
        db.tier = GSCAN_TIER_CLASSRUN;
    else if (best_len >= 1)
        db.tier = GSCAN_TIER_LITERAL;
    else {
        why = "class sequence with more than 4 classes and no literal byte to anchor on";
        return 1;
    }
    return 0;
}

} // namespace gscan
