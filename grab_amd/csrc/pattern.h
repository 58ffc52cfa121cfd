// pattern.h -- PCRE-subset pattern compiler for the gfx950 scan engine.
//
// Replaces the reference's pcre_compile/pcre_study/pcre_fullinfo(MINLENGTH) step
// (/root/reference/src/grab.cc:101-123) for the patterns the GPU engine can scan:
// a concatenation of single-byte atoms (literal, '.', escape class, [...] class),
// each with a FIXED repeat count, optionally ending in ONE greedy variable repeat
// (*, +, ?, {n,}, {n,m}).  For that shape "pcre_exec reports a match starting at p"
// is a pure function of the minlen-byte window at p, which is what makes the
// "GPU emits all candidate starts, host walks the restart orbit" split exact
// (SURVEY.md Appendix C).  Everything else is reported as GSCAN_UNSUPPORTED.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

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

constexpr int kMaxWindow = 256; // window positions a database may hold
constexpr int kMaxClasses = 64; // distinct byte classes per database
constexpr int kK2MaxClasses = 4;
constexpr int kK2MaxWindow = 49; // 16 own positions + 48 bits of look-ahead
constexpr int kK2MaxRuns = 16; // run descriptors live in the lanes of one VGPR for the whole kernel

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
    uint8_t window[kMaxWindow];          // class id per window position (or the literal byte)
};

struct Database {
    int tier = 0;
    int minlen = -1;
    bool has_tail = false;
    uint32_t tail_extra = 0;
    ByteSet tail;
    std::vector<ByteSet> classes;
    std::vector<uint8_t> window; // class id per position
    DevProgram prog;
    uint64_t id = 0; // unique per compile; contexts key their device copy on it
};

// rc: 0 ok, 1 unsupported, -1 malformed.  `why` gets a short reason.
int compile_pattern(const char *pat, size_t len, unsigned flags, Database &db, std::string &why);

} // namespace gscan
