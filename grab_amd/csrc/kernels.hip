// kernels.hip -- gfx950 (MI355X / CDNA4) scan kernels.  Hand-written HIP, wave64.
//
// What they replace: the byte-stream work of pcre_exec over one chunk
// (/root/reference/src/grab.cc:178), restated as "emit every offset p whose
// minlen-byte window satisfies the pattern" (SURVEY.md Appendix C).
//
// Both kernels are HBM-bound byte-stream scans (no MFMA).  Shared structure:
//   * work unit = TILE of WAVES*ITER KiB of one segment; a 256-thread workgroup
//     (4 waves) takes a tile, each wave a contiguous ITER-KiB sub-tile; the tile's
//     16-byte descriptor comes in with one scalar load;
//   * every lane issues all its ITER+1 16-byte loads up front (buffer_load_dwordx4,
//     lane i -> bytes [16i,16i+16) of each KiB: fully coalesced, each input byte is
//     fetched from HBM exactly once; the +1 is a <=64-byte halo for windows that
//     straddle the sub-tile end; the buffer descriptor's bounds check zero-fills past
//     the segment end, so there are no per-load branches);
//   * per KiB step each lane produces a 16-bit candidate mask for its 16 positions;
//     a candidate whose predecessor position is known to be a candidate too is dropped
//     (only the START of each group of consecutive candidates has to be reported: the
//     host re-tests the window at its restart position, see gscan.h); masks stay in
//     registers until the tile is done (2 steps per VGPR);
//   * compaction: per-lane popcount -> wave reduce -> one LDS slot per wave -> ONE
//     global atomicAdd per tile that has any candidate reserves a contiguous run in
//     the record buffer (8 counters, one per record-buffer shard, tile t uses shard
//     t&7, so dense outputs do not serialise on one address); lanes then expand their
//     masks in text order using a wave prefix sum (K2, whose outputs are the dense
//     ones, first transposes the masks through LDS so that one scan serves the whole
//     sub-tile: emit_wave_t), and the sub-tile's descriptor
//     {count, base} is written at desc[tile].  Tiles are in text order, so walking
//     desc[] yields ascending offsets with no sort and no second pass over the text.
//
// K1 (literal / anchored class sequence): SWAR compare of the 4-byte anchor at all
//   16 byte alignments of the lane's data (v_alignbyte + v_bitop3 + v_min3), ~2.6 VALU
//   ops/byte; the rare anchor hit is verified against the whole window in a cold
//   path.  No LDS in the hot loop.
// K2 (class runs, <=4 classes, window <=49): LDS-staged byte->class-bits table,
//   replicated once per LDS bank (32 KiB) so the 64 random lookups of a wave never
//   conflict; 16 lookups/lane/step accumulate into two registers; neighbouring
//   lanes' masks come in by DPP wave shifts; runs of n consecutive class bits are
//   found by shift-AND doubling (log2 n steps) on 32- or 64-bit words.  The run
//   program sits in the lanes of one VGPR (v_readlane), not in memory.
// K3 (several alternatives, or one class sequence with > 4 classes): the multi-pattern
//   filter.  The alternatives share 8 buckets; one LDS lookup per text byte (same
//   bank-replicated 32 KiB table layout as K2) returns, for each of 4 window positions,
//   the set of buckets that accept the byte there; position q is a hit when some bucket
//   accepts bytes q..q+3 at positions 0..3 (3 VALU ops per byte to AND the four bucket
//   sets: the byte selects ride on SDWA operand modifiers).  Hits are verified against the full windows of their buckets'
//   alternatives in a cold path, unless the filter is already exact.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_args.h"

#include "kcommon.h"

namespace gscan {

namespace {

// Epilogue of one WAVE's sub-tile: hits[] holds the per-step 16-bit masks of this lane (two per register), cnt its
// popcount.  `bias` is subtracted from every position (K1 records window starts, its masks mark anchor positions).
// Every wave reserves its own run in the record buffer and writes its own descriptor, desc[d] (d = tile * waves + wave):
// no LDS, no workgroup barrier -- a wave that is done with its sub-tile goes straight on to the next tile's loads while
// its neighbours are still busy.  (Round 1 reserved one run per TILE: wave counts through LDS, a barrier, one atomic by
// thread 0, its result through LDS, another barrier -- every wave of the workgroup sat through the atomic's round trip
// once per tile, and the persistent kernels lost the overlap between one tile's tail and the next one's loads.)
template <int ITER>
__device__ __forceinline__ void emit_wave(const ScanArgs &a, uint32_t d, const uint32_t (&hits)[(ITER + 1) / 2], uint32_t cnt, int sub_off,
                                          uint32_t bias, uint32_t lane)
{
    const uint32_t wtot = wave_sum(cnt);
    if (wtot == 0) {
        if (lane == 0) a.desc[d] = 0ull;
        return;
    }
    const uint32_t shard = d & (kShards - 1);
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(a.counter + shard * kCtrStride, wtot); // index inside the shard's region
    const uint32_t base = __builtin_amdgcn_readfirstlane(b);
    const bool over = (unsigned long long)base + wtot > (unsigned long long)a.cap_shard;
    if (lane == 0) {
        a.desc[d] = (unsigned long long)wtot | ((unsigned long long)(shard * a.cap_shard + base) << 32);
        if (over) atomicOr(a.counter + kShards * kCtrStride, 1u);
    }
    if (over) return; // overflow: the host re-runs with a bigger buffer
    uint32_t run = shard * a.cap_shard + base;
#pragma unroll
    for (int k = 0; k < ITER; k++) {
        uint32_t bits = (hits[k >> 1] >> (16 * (k & 1))) & 0xffffu;
        const unsigned long long any = __ballot(bits != 0);
        if (any == 0ull) continue; // wave-uniform
        const uint32_t pos0 = (uint32_t)(sub_off + k * 1024) + lane * 16u - bias;
        if (__ballot((bits & (bits - 1u)) != 0) == 0ull) {
            // common case: no lane holds two records in this step -> rank = lanes below me that hold one
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
            if (bits) a.recs[run + rank] = pos0 + (uint32_t)__ffs((int)bits) - 1u;
            run += (uint32_t)__popcll(any);
            continue;
        }
        uint32_t c = (uint32_t)__popc(bits);
        uint32_t inc = wave_scan(c);
        uint32_t idx = run + inc - c;
        while (bits) {
            uint32_t j = (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            a.recs[idx++] = pos0 + j;
        }
        run += __builtin_amdgcn_readlane(inc, 63);
    }
}

// Dense-output epilogue (K2): the same per-wave reservation, but the per-step masks are first transposed through a
// wave-private strip of LDS so that lane L owns ITER consecutive (step, lane) cells, i.e. a contiguous piece of text.  One
// wave scan then places every lane's records; emit_wave pays a ballot + rank + branch for each of the ITER steps instead,
// which is what limits the kernels when most steps carry records (identifier regex: 3.6 records per KiB).  The strip is
// the wave's own: the only ordering needed is between the wave's LDS writes and its reads (no workgroup barrier).
//
// How the reservation got here (identifier scan / [0-9]{16} without records / [0-9]+\.[0-9]+ on K3, TB/s, same sweep):
//   round 1, one atomic per TILE behind two workgroup barriers                           5.2 / 5.9 / 4.4
//   one atomic per WAVE, 8 adjacent counters                                             1.1 / 6.3 / 1.1   (r02_d_...)
//   ... 64 adjacent counters                                                             2.5 / 6.4 / 2.5   (r02_e_...)
//   ... 64 counters, one per 128-byte line                                               5.1 / 5.9-6.4 / 4.5 (r02_h_...)
// Returning atomics on one cache line are served one after the other by that line's L2 channel whichever dword they
// name: adjacent counters were no counters at all.
template <int ITER>
__device__ __forceinline__ void emit_wave_t(const ScanArgs &a, uint32_t d, const uint32_t (&hits)[(ITER + 1) / 2], uint32_t cnt, int sub_off,
                                            uint32_t bias, uint32_t lane, uint16_t *xp)
{
    const uint32_t wtot = wave_sum(cnt);
    if (wtot == 0) {
        if (lane == 0) a.desc[d] = 0ull;
        return;
    }
#pragma unroll
    for (int k = 0; k < ITER; k++) xp[k * 64 + lane] = (uint16_t)(hits[k >> 1] >> (16 * (k & 1)));
    const uint32_t shard = d & (kShards - 1);
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(a.counter + shard * kCtrStride, wtot);
    GS_LDS_FENCE(__ATOMIC_RELEASE);
    __builtin_amdgcn_wave_barrier();
    GS_LDS_FENCE(__ATOMIC_ACQUIRE);
    unsigned long long w[ITER / 4];
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(xp + lane * ITER);
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < ITER / 4; q++) {
        w[q] = src[q];
        c += (uint32_t)__popcll(w[q]);
    }
    const uint32_t base = __builtin_amdgcn_readfirstlane(b);
    const bool over = (unsigned long long)base + wtot > (unsigned long long)a.cap_shard;
    if (lane == 0) {
        a.desc[d] = (unsigned long long)wtot | ((unsigned long long)(shard * a.cap_shard + base) << 32);
        if (over) atomicOr(a.counter + kShards * kCtrStride, 1u);
    }
    __builtin_amdgcn_wave_barrier();
    if (over) return;
    const uint32_t inc = wave_scan(c);
    uint32_t idx = shard * a.cap_shard + base + inc - c;
#pragma unroll
    for (int q = 0; q < ITER / 4; q++) {
        unsigned long long bitsq = w[q];
        while (bitsq) {
            const uint32_t bb = (uint32_t)__ffsll((long long)bitsq) - 1u;
            bitsq &= bitsq - 1ull;
            const uint32_t cell = lane * ITER + q * 4 + (bb >> 4);
            a.recs[idx++] = (uint32_t)sub_off + (cell >> 6) * 1024u + (cell & 63u) * 16u + (bb & 15u) - bias;
        }
    }
}

// ------------------------------------------------------------------------------------
// K1: literal / anchored window.
// ------------------------------------------------------------------------------------
template <int ITER, bool NT>
__global__ __launch_bounds__(kWG) void k1_anchor_scan(ScanArgs a, const TileDesc *__restrict__ tiles)
{
    constexpr uint32_t kTile = kWaves * ITER * 1024;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t anchor = a.anchor, amask = a.anchor_mask;
    const uint32_t aoff = a.anchor_off, m = a.m;

    for (uint32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        TileCtx c = tile_ctx(a, tiles, t, kTile);
        const int sub_off = c.tile_off + (int)(wave * ITER * 1024);
        uint32_t hits[(ITER + 1) / 2];
#pragma unroll
        for (int i = 0; i < (ITER + 1) / 2; i++) hits[i] = 0;
        uint32_t cnt = 0;

        if (c.live && sub_off < c.slen) {
            u32x4 buf[ITER + 1];
            load_subtile<ITER, NT, 1>(buf, c, sub_off, lane);
            // anchor positions that correspond to an in-bounds window
            const int lo = (int)aoff, hi = c.slen - (int)m + (int)aoff;
#pragma unroll
            for (int k = 0; k < ITER; k++) {
                const uint32_t d0 = buf[k].x, d1 = buf[k].y, d2 = buf[k].z, d3 = buf[k].w;
                const uint32_t nx = __builtin_amdgcn_readfirstlane(buf[k + 1].x);
                const uint32_t d4 = down1(d0, nx);
                // x_j == 0  <=>  the anchor sits at byte j of this lane's 16
                uint32_t acc = 0xffffffffu;
#define GS_X(lo_, hi_, sh_) ((__builtin_amdgcn_alignbyte(hi_, lo_, sh_) ^ anchor) & amask)
#define GS_MIN3(p_, q_, r_) min(min(p_, q_), r_)
                acc = GS_MIN3(acc, (d0 ^ anchor) & amask, GS_X(d0, d1, 1));
                acc = GS_MIN3(acc, GS_X(d0, d1, 2), GS_X(d0, d1, 3));
                acc = GS_MIN3(acc, (d1 ^ anchor) & amask, GS_X(d1, d2, 1));
                acc = GS_MIN3(acc, GS_X(d1, d2, 2), GS_X(d1, d2, 3));
                acc = GS_MIN3(acc, (d2 ^ anchor) & amask, GS_X(d2, d3, 1));
                acc = GS_MIN3(acc, GS_X(d2, d3, 2), GS_X(d2, d3, 3));
                acc = GS_MIN3(acc, (d3 ^ anchor) & amask, GS_X(d3, d4, 1));
                acc = GS_MIN3(acc, GS_X(d3, d4, 2), GS_X(d3, d4, 3));
                uint32_t bits = 0;
                if (acc == 0) { // cold: find which alignments hit, bounds-check, verify the window
                    const int pos0 = sub_off + k * 1024 + (int)lane * 16;
                    const uint32_t vm = valid16(pos0, lo, hi);
                    const uint32_t dd[5] = {d0, d1, d2, d3, d4};
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        uint32_t u = (j & 3) ? __builtin_amdgcn_alignbyte(dd[j / 4 + 1], dd[j / 4], j & 3) : dd[j / 4];
                        if (((u ^ anchor) & amask) == 0) bits |= 1u << j;
                    }
                    bits &= vm;
                    if (m > a.anchor_len) {
                        uint32_t keep = 0, b2 = bits;
                        while (b2) {
                            uint32_t j = (uint32_t)__ffs((int)b2) - 1u;
                            b2 &= b2 - 1u;
                            if (verify_window(c.seg, a.prog, (uint32_t)pos0 + j - aoff)) keep |= 1u << j;
                        }
                        bits = keep;
                    }
                    bits &= ~((bits << 1) & ~a.keep_all); // keep group starts (within the lane; a superset of them is fine)
                }
                // outside the branch: one shift-or per step instead of moving the whole hit array through the join
                hits[k >> 1] |= bits << (16 * (k & 1));
                cnt += (uint32_t)__popc(bits);
#undef GS_X
#undef GS_MIN3
            }
        }
        emit_wave<ITER>(a, t * kWaves + wave, hits, cnt, sub_off, aoff - a.report_shift, lane);
    }
}

// ------------------------------------------------------------------------------------
// K2: class runs.
// ------------------------------------------------------------------------------------
// One lookup: byte -> dword of class bits (bit 8c = member of class c); the table copy
// in LDS bank (lane&31) is used, so lanes never collide (ds_read_b32 services lanes
// 0-31 and 32-63 separately).
#define GS_LUT(w_, sh_) tbl[((((w_) >> (sh_)) & 0xffu) << 5) | bank]

// cand(p) = AND over runs r of "class cls_r holds at p+off_r .. p+off_r+len_r-1".  A run of
// n set bits starting at bit p of x: double the verified length (x &= x >> len) while it
// fits, then one overlapping step for the remainder.  Everything but x is in SGPRs.
template <typename W>
__device__ __forceinline__ W run_and(uint32_t nruns, uint32_t vrd, const W (&cls)[4])
{
    W cand = ~(W)0;
    for (uint32_t r = 0; r < nruns; r++) {
        const uint32_t d = __builtin_amdgcn_readlane(vrd, r); // descriptor r: one v_readlane, no memory, no wait
        const uint32_t c = d & 0xffu, n = (d >> 8) & 0xffu, off = d >> 16;
        W x = c == 0 ? cls[0] : c == 1 ? cls[1] : c == 2 ? cls[2] : cls[3];
        uint32_t len = 1;
        while (2 * len <= n) {
            x &= x >> len;
            len *= 2;
        }
        if (len < n) x &= x >> (n - len);
        cand &= x >> off;
    }
    return cand;
}

// PAIR = false: the general form (<= 4 classes): 256-thread workgroup, 32 KiB bank-replicated
//   table of per-byte class bits, 16 lookups per lane per step.
// PAIR = true: <= 2 classes: 512-thread workgroup sharing a 64 KiB table indexed by TWO text
//   bytes at once (entry = class bits of both bytes: class 0 in bits 0-1, class 1 in bits 4-5).
//   8 lookups per lane per step and about a third of the VALU work of the general form, which is
//   what the kernel is bound by (integer VALU issues one wave64 instruction per 4 cycles: PMC
//   runs under profiles/).  Bank conflicts are possible here (random 16-bit indices) but the LDS
//   has the cycles to spare; the table is built once per workgroup, so this form runs as a
//   persistent grid.
// (a << SH) | b in one instruction
template <int SH>
__device__ __forceinline__ uint32_t lshl_or(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(b));
    return r;
}

// one run of the branch-free program (ScanArgs::run_flat): d is wave-uniform, so every field is an s_bfe
__device__ __forceinline__ uint32_t run_one_flat(uint32_t d, uint32_t w0, uint32_t w1)
{
    uint32_t x = (d & 1u) ? w1 : w0;
    if (d & 0xfffeu) { // (a run of one byte has no steps at all: a uniform branch instead of ten no-ops)
        x &= x >> ((d >> 1) & 1u);
        x &= x >> ((d >> 2) & 3u);
        x &= x >> ((d >> 4) & 7u);
        x &= x >> ((d >> 7) & 15u);
        x &= x >> ((d >> 11) & 31u);
    }
    return x >> ((d >> 16) & 63u);
}

// NR: the two-class form with windows of <= 17 bytes runs the branch-free run program (run_one_flat) -- NR = 1..3: the
// pattern has exactly that many runs and their descriptors are decoded once, before the tile loop; NR = 0: any number,
// one v_readlane + a handful of s_bfe per run and step; NR = -1 (the other forms): run_and's scalar loops.  Measured on
// the identifier scan: the scalar loops cost 203 M SALU instructions per 4 GiB against 31 M, and 8 % of the time
// (profiles/r01_w_k2_opt_sweep.txt, r01_w_k2_pmc.txt).
// NW (pair form): waves per workgroup -- 8 (512 threads, ITER 12: 4 waves per SIMD) or 12 (768 threads, ITER 8, two
// workgroups per CU = 6 waves per SIMD within 80 VGPRs; same 96 KiB tile).  Measured: more waves buy nothing with records
// (5.16 vs 5.22 TB/s on the identifier scan) and +1..8 % without (1024 threads, 8 waves per SIMD, plain epilogue: 6.4 vs
// 5.9 TB/s) -- profiles/r02_b_kernel_sweep_workgroup_shapes.txt; the default stays 512 threads.
// PF (persistent forms): the NEXT tile's text is requested while this one is computed, in two bursts -- its first half the
// moment this tile's first half has been turned into look-up addresses, its second half (and halo) behind this tile's
// epilogue -- so a wave's loads are in flight through its compute and its epilogue instead of starting behind them (every
// wave of a persistent workgroup paid one full HBM round trip per tile, hidden only by whatever its three SIMD neighbours
// happened to be doing).  Two bursts at fixed places, issued on every path (out of range when there is no next tile),
// first-tile loads in the same order: the compiler's vmcnt counts then come out exact -- with the refills trickling in step
// by step it merged the loop with its preheader conservatively and made the tail steps of every tile wait for loads issued
// two steps earlier.
// (A per-lane copy of a 256-entry table in this kernel -- round 2's "LT" form: one look-up per byte, never a conflict, sixteen
// v_lshl_or_b32 to merge -- was built and measured: [0-9]{16} 5.60 -> 5.94 TB/s, the identifier scan 5.18 -> 4.87,
// profiles/r02_k_kernel_sweep_lane_table.txt.  Round 3's lane-table kernel (k2lane.hip) is that idea done right; the form is gone.)
template <int ITER, bool NT, bool WIDE, bool PAIR, int NR = -1, int NW = (PAIR ? 8 : 4), bool PF = false>
__global__ __launch_bounds__(NW * 64, NW == 12 || (!PAIR && NW == 8) ? 6 : 1) void k2_classrun_scan(ScanArgs a, const TileDesc *__restrict__ tiles)
{
    static_assert(!PF || PAIR, "the prefetching form is the pair form's");
    static_assert(NR < 0 || (PAIR && !WIDE), "the flat run program is the two-class, 32-bit form's");
    constexpr int kNW = NW; // waves per workgroup
    __shared__ uint32_t tbl[PAIR ? 65536 / 4 : 256 * 32];
    __shared__ __attribute__((aligned(8))) uint16_t s_xp[kNW * ITER * 64]; // epilogue transposition strips, 2 bytes per (step, lane), one per wave
    constexpr uint32_t kTile = kNW * ITER * 1024;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t bank = lane & 31u;
    const uint32_t m = a.m, ncls = a.n_classes, nruns = a.nruns;
    // The run program lives in the lanes of one VGPR (lane r = descriptor r) and is read back
    // with v_readlane inside the run loop.  Fetching it from the kernel-argument segment there
    // would put a scalar load + s_waitcnt lgkmcnt(0) into every step, and that wait also drains
    // the LDS lookups already in flight for the next step.
    const uint32_t vrd = a.run_desc[lane & (kK2MaxRuns - 1)];
    const uint32_t vrf = a.run_flat[lane & (kK2MaxRuns - 1)];
    uint32_t rf[3] = {0, 0, 0}; // NR > 0: the run descriptors, wave-uniform
    if (NR > 0)
        for (int r = 0; r < NR; r++) rf[r] = __builtin_amdgcn_readlane(vrf, r);
    const uint8_t *tbl8 = reinterpret_cast<const uint8_t *>(tbl);

    if (!PAIR) { // stage the class table: entry b replicated into all 32 banks (dword q = copy q & 31 of entry q >> 5)
        for (uint32_t q = threadIdx.x; q < 256u * 32u; q += NW * 64) tbl[q] = a.prog->k2_table[q >> 5];
    } else { // build the pair table from the 256-entry one: index = first byte | second byte << 8
        const uint32_t *base = a.prog->k2_table;
        for (uint32_t q = threadIdx.x; q < 65536 / 4; q += NW * 64) { // one dword = 4 consecutive first bytes
            const uint32_t b1 = q >> 6, b0 = (q & 63u) << 2;
            const uint32_t t1 = base[b1];
            const uint32_t hi = (((t1 & 1u) << 1) | ((t1 >> 8 & 1u) << 5)) * 0x01010101u; // second byte's bits, all 4 entries
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t t0 = base[b0 + j];
                lo |= ((t0 & 1u) | ((t0 >> 8 & 1u) << 4)) << (8 * j);
            }
            tbl[q] = lo | hi;
        }
    }
    __syncthreads();

    // The tile loop.  PF: the first pass has no tile of its own (cur == false) and only requests the first one's text --
    // every load of the kernel is issued inside the loop body, by the same instructions, whatever the path: no preheader
    // whose pending loads the compiler would have to merge with the steady state's.
    u32x4 buf[ITER + 1], halo_n = {0, 0, 0, 0};
    if (blockIdx.x >= a.n_tiles) return;
    uint32_t t = 0, tn = blockIdx.x;
    TileCtx c = tile_ctx(a, tiles, tn, kTile);
    int sub_off = 0;
    bool have = false, cur = false;
    for (;;) {
        // the tile after this one (PF: its loads go out during this tile's steps; out of range -- zeros, no traffic -- if there is none)
        const bool next = tn < a.n_tiles;
        TileCtx cn = c;
        int sub_off_n = 0;
        bool have_n = false;
        if (next) {
            cn = tile_ctx(a, tiles, tn, kTile);
            sub_off_n = cn.tile_off + (int)(wave * ITER * 1024);
            have_n = cn.live && sub_off_n < cn.slen;
        }
        uint32_t hits[(ITER + 1) / 2];
#pragma unroll
        for (int i = 0; i < (ITER + 1) / 2; i++) hits[i] = 0;
        uint32_t cnt = 0;

        if (have) {
            if (!PF) load_subtile<ITER, NT, 4>(buf, c, sub_off, lane);
            const int hi = c.slen - (int)m;
            const bool interior = sub_off + ITER * 1024 + 64 <= hi;

            // class masks of one 16-byte piece: P01 = cls0 | cls1<<16, P23 = cls2 | cls3<<16
            uint32_t p01n, p23n = 0;
            auto masks = [&](const u32x4 &d, uint32_t &p01, uint32_t &p23) {
                if (PAIR) {
                    // 8 two-byte lookups; e_k has class 0 of its two positions in bits 0-1, class 1 in bits 4-5
                    const uint32_t e0 = tbl8[d.x & 0xffffu], e1 = tbl8[d.x >> 16], e2 = tbl8[d.y & 0xffffu], e3 = tbl8[d.y >> 16];
                    const uint32_t e4 = tbl8[d.z & 0xffffu], e5 = tbl8[d.z >> 16], e6 = tbl8[d.w & 0xffffu], e7 = tbl8[d.w >> 16];
                    // byte q of g: low nibble = class 0 of positions 4q..4q+3, high nibble = class 1
                    // (one v_lshl_or_b32 per entry, in the order the look-ups return: the compiler's own choice for this
                    // expression is 7 shifts + 4 three-way ORs)
                    const uint32_t g = lshl_or<26>(e7, lshl_or<24>(e6, lshl_or<18>(e5, lshl_or<16>(e4, lshl_or<10>(e3, lshl_or<8>(e2, lshl_or<2>(e1, e0)))))));
                    // nibbles of g: c0 c1 c0 c1 | c0 c1 c0 c1 -> c0 c0 c0 c0 | c1 c1 c1 c1: swap nibbles 1 and 2 of each half
                    // (xor trick), then bytes 1 and 2 (v_perm): 5 operations
                    const uint32_t sw = ((g >> 4) ^ g) & 0x00f000f0u;
                    const uint32_t h = g ^ sw ^ (sw << 4);
                    p01 = __builtin_amdgcn_perm(h, h, 0x03010200u);
                    p23 = 0;
                } else {
                    uint32_t lo = 0, hi8 = 0;
                    lo = (lo << 1) | GS_LUT(d.y, 24); lo = (lo << 1) | GS_LUT(d.y, 16);
                    lo = (lo << 1) | GS_LUT(d.y, 8);  lo = (lo << 1) | GS_LUT(d.y, 0);
                    lo = (lo << 1) | GS_LUT(d.x, 24); lo = (lo << 1) | GS_LUT(d.x, 16);
                    lo = (lo << 1) | GS_LUT(d.x, 8);  lo = (lo << 1) | GS_LUT(d.x, 0);
                    hi8 = (hi8 << 1) | GS_LUT(d.w, 24); hi8 = (hi8 << 1) | GS_LUT(d.w, 16);
                    hi8 = (hi8 << 1) | GS_LUT(d.w, 8);  hi8 = (hi8 << 1) | GS_LUT(d.w, 0);
                    hi8 = (hi8 << 1) | GS_LUT(d.z, 24); hi8 = (hi8 << 1) | GS_LUT(d.z, 16);
                    hi8 = (hi8 << 1) | GS_LUT(d.z, 8);  hi8 = (hi8 << 1) | GS_LUT(d.z, 0);
                    // byte c of lo = positions 0-7 of class c, byte c of hi8 = positions 8-15
                    p01 = (lo & 0xffu) | ((hi8 & 0xffu) << 8) | ((lo & 0xff00u) << 8) | ((hi8 & 0xff00u) << 16);
                    p23 = ((lo >> 16) & 0xffu) | (((hi8 >> 16) & 0xffu) << 8) | ((lo >> 8) & 0xff0000u) | (hi8 & 0xff000000u);
                }
            };
            // The pair form keeps its look-ups one step further ahead: those of step k + 2 are issued before step k is computed
            // and merged after it, so a wave does not sit out its own LDS latency (random 16-bit indices: bank conflicts make
            // it long) between issuing eight look-ups and using them.
            uint32_t ea[8];
#pragma unroll
            for (int i = 0; i < 8; i++) ea[i] = 0;
            auto lookups = [&](const u32x4 &d) {
                ea[0] = tbl8[d.x & 0xffffu], ea[1] = tbl8[d.x >> 16], ea[2] = tbl8[d.y & 0xffffu], ea[3] = tbl8[d.y >> 16];
                ea[4] = tbl8[d.z & 0xffffu], ea[5] = tbl8[d.z >> 16], ea[6] = tbl8[d.w & 0xffffu], ea[7] = tbl8[d.w >> 16];
            };
            auto merge = [&]() -> uint32_t { // (the same five operations as in masks())
                const uint32_t g = lshl_or<26>(ea[7], lshl_or<24>(ea[6], lshl_or<18>(ea[5], lshl_or<16>(ea[4], lshl_or<10>(ea[3], lshl_or<8>(ea[2], lshl_or<2>(ea[1], ea[0])))))));
                const uint32_t sw = ((g >> 4) ^ g) & 0x00f000f0u;
                const uint32_t h = g ^ sw ^ (sw << 4);
                return __builtin_amdgcn_perm(h, h, 0x03010200u);
            };
            masks(buf[0], p01n, p23n);
            if (PAIR) lookups(buf[1]);
            uint32_t last63 = 0; // candidate mask of lane 63 in the previous step (0: unknown at the sub-tile start)
#pragma unroll
            for (int k = 0; k < ITER; k++) {
                const uint32_t p01 = p01n, p23 = p23n;
                if (PAIR) {
                    p01n = merge();                          // step k + 1's masks: lanes 61-63 look into them
                    if (k + 2 <= ITER) lookups(buf[k + 2]); // in flight while step k is computed
                    if (PF) { // what has just become look-up addresses takes the next tile's text
                        if (k == 0) {
                            buf[0] = load_step<ITER, NT, 4>(cn, sub_off_n, lane, 0, have_n);
                            buf[1] = load_step<ITER, NT, 4>(cn, sub_off_n, lane, 1, have_n);
                            halo_n = load_step<ITER, NT, 4>(cn, sub_off_n, lane, ITER, have_n); // (this tile's halo is in use until its last step)
                        }
                        if (k + 2 < ITER) buf[k + 2] = load_step<ITER, NT, 4>(cn, sub_off_n, lane, k + 2, have_n);
                    }
                } else {
                    masks(buf[k + 1], p01n, p23n); // next step's masks: lanes 61-63 look into them
                }
                const bool more = !PAIR && ncls > 2;
                uint32_t bits;
                if (!WIDE) { // look-ahead <= 16 positions: one neighbour
                    const uint32_t a01 = down1(p01, __builtin_amdgcn_readfirstlane(p01n));
                    const uint32_t a23 = more ? down1(p23, __builtin_amdgcn_readfirstlane(p23n)) : 0u;
                    // class c's 32 positions: the low (high) halves of this lane's and the next lane's masks -- one v_perm each
                    const uint32_t W[4] = {__builtin_amdgcn_perm(a01, p01, 0x05040100u), __builtin_amdgcn_perm(a01, p01, 0x07060302u),
                                           more ? __builtin_amdgcn_perm(a23, p23, 0x05040100u) : 0u, more ? __builtin_amdgcn_perm(a23, p23, 0x07060302u) : 0u};
                    if (NR > 0) {
                        uint32_t cand = 0xffffu;
#pragma unroll
                        for (int r = 0; r < NR; r++) cand &= run_one_flat(rf[r], W[0], W[1]);
                        bits = cand;
                    } else if (NR == 0) {
                        uint32_t cand = 0xffffu;
                        for (uint32_t r = 0; r < nruns; r++) cand &= run_one_flat(__builtin_amdgcn_readlane(vrf, r), W[0], W[1]);
                        bits = cand;
                    } else {
                        bits = run_and<uint32_t>(nruns, vrd, W) & 0xffffu;
                    }
                } else { // look-ahead <= 48 positions: three neighbours
                    const uint32_t s01_0 = __builtin_amdgcn_readlane(p01n, 0), s01_1 = __builtin_amdgcn_readlane(p01n, 1),
                                   s01_2 = __builtin_amdgcn_readlane(p01n, 2);
                    const uint32_t a01 = down1(p01, s01_0);
                    const uint32_t b01 = down1(a01, s01_1);
                    const uint32_t c01 = down1(b01, s01_2);
                    uint32_t a23 = 0, b23 = 0, c23 = 0;
                    if (more) {
                        const uint32_t s0 = __builtin_amdgcn_readlane(p23n, 0), s1 = __builtin_amdgcn_readlane(p23n, 1),
                                       s2 = __builtin_amdgcn_readlane(p23n, 2);
                        a23 = down1(p23, s0);
                        b23 = down1(a23, s1);
                        c23 = down1(b23, s2);
                    }
                    const uint64_t W[4] = {
                        (uint64_t)((p01 & 0xffffu) | (a01 << 16)) | ((uint64_t)((b01 & 0xffffu) | (c01 << 16)) << 32),
                        (uint64_t)((p01 >> 16) | (a01 & 0xffff0000u)) | ((uint64_t)((b01 >> 16) | (c01 & 0xffff0000u)) << 32),
                        (uint64_t)((p23 & 0xffffu) | (a23 << 16)) | ((uint64_t)((b23 & 0xffffu) | (c23 << 16)) << 32),
                        (uint64_t)((p23 >> 16) | (a23 & 0xffff0000u)) | ((uint64_t)((b23 >> 16) | (c23 & 0xffff0000u)) << 32)};
                    bits = (uint32_t)run_and<uint64_t>(nruns, vrd, W) & 0xffffu;
                }
                if (!interior) bits &= valid16(sub_off + k * 1024 + (int)lane * 16, 0, hi);
                // drop candidates whose predecessor position is a candidate: bit j-1 of this lane, or bit 15
                // of the previous lane (previous step's lane 63 for lane 0)
                const uint32_t prev = up1(bits, last63);
                last63 = __builtin_amdgcn_readlane(bits, 63);
                bits &= ~(((bits << 1) | (prev >> 15)) & ~a.keep_all);
                hits[k >> 1] |= bits << (16 * (k & 1));
                cnt += (uint32_t)__popc(bits);
            }
        } else if (PF) { // nothing of this tile is this wave's (or there is no tile yet): the same loads all the same
            buf[0] = load_step<ITER, NT, 4>(cn, sub_off_n, lane, 0, have_n);
            buf[1] = load_step<ITER, NT, 4>(cn, sub_off_n, lane, 1, have_n);
            halo_n = load_step<ITER, NT, 4>(cn, sub_off_n, lane, ITER, have_n);
#pragma unroll
            for (int j = 2; j < ITER; j++) buf[j] = load_step<ITER, NT, 4>(cn, sub_off_n, lane, j, have_n);
        }
        // K2's outputs are the dense ones: transposed epilogue (+8 % on the identifier scan, neutral without matches)
        if (cur) emit_wave_t<ITER>(a, t * kNW + wave, hits, cnt, sub_off, 0u - a.report_shift, lane, s_xp + wave * (ITER * 64));
        if (!next) break;
        if (PF) buf[ITER] = halo_n;
        t = tn;
        tn += gridDim.x;
        c = cn;
        sub_off = sub_off_n;
        have = have_n;
        cur = true;
    }
}
#undef GS_LUT

// ------------------------------------------------------------------------------------
// K3: bucket filter over 4 window positions (alternations; class sequences with > 4 classes).
// ------------------------------------------------------------------------------------
// Table entry of byte b (DevProgram::k3_table): byte k = buckets that accept b at window position k3_off + k, P_k(b).
//   hit(j) = P_0(t_j) & P_1(t_{j+1}) & P_2(t_{j+2}) & P_3(t_{j+3}) != 0.
// In LDS the entry's middle bytes are swapped -- [P_0, P_2, P_1, P_3] -- so that with e_j the entry of text byte j
//   Y_j = e_j.WORD_0 & e_{j+1}.WORD_1 = [P_0(t_j) & P_1(t_{j+1}),  P_2(t_j) & P_3(t_{j+1})]     one v_and_b32_sdwa
//   h_j = Y_j.BYTE_0 & Y_{j+2}.BYTE_1                                                          one v_and_b32_sdwa
// : TWO VALU operations per text byte for four window positions (the word and byte selects are operand modifiers).  The
// round-1 form (A_j = e_j.b0 & e_{j+1}.b1, B_j = e_j.b2 & e_{j+1}.b3, h_j = A_j & B_{j+2}) took three, and K3 is bound by
// VALU issue (profiles/r02_*k3*).
__device__ __forceinline__ uint32_t and_w0_w1(uint32_t x, uint32_t y) // low half of x & high half of y
{
    uint32_t r;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ uint32_t and_b0_b1(uint32_t x, uint32_t y) // byte 0 of x & byte 1 of y
{
    uint32_t r;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// byte BYTE of dst = byte 0 of x & byte 1 of y; the other bytes of dst are kept (BYTE 0: zeroed) -- the SDWA destination
// select packs four one-byte results into one register at no cost, so "any hit in this step" is the OR of four registers
// instead of sixteen, and the hit mask is made of four dot products (below)
template <int BYTE>
__device__ __forceinline__ void and_b0_b1_into(uint32_t &dst, uint32_t x, uint32_t y)
{
    if constexpr (BYTE == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(dst) : "v"(x), "v"(y));
    else if constexpr (BYTE == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(dst) : "v"(x), "v"(y));
    else if constexpr (BYTE == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(dst) : "v"(x), "v"(y));
    else asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(dst) : "v"(x), "v"(y));
}
// 16-bit mask of the NONZERO bytes of H[0..3] (byte k of H[q] = position 4 q + k).  one: every byte is 0 or 1 already (a
// single bucket).  Bytes -> 0 / 1 by the exact SWAR test, then one v_dot4_u32_u8 per register with the weights 1 2 4 8 /
// 16 32 64 128 adds the four bits of a register into place.
__device__ __forceinline__ uint32_t nonzero_bytes16(const uint32_t (&H)[4], bool one)
{
    uint32_t x[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        x[q] = H[q];
        if (!one) x[q] = ((((H[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | H[q]) >> 7) & 0x01010101u;
    }
    const uint32_t lo = __builtin_amdgcn_udot4(x[1], 0x80402010u, __builtin_amdgcn_udot4(x[0], 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(x[3], 0x80402010u, __builtin_amdgcn_udot4(x[2], 0x08040201u, 0u, false), false);
    return lo | (hi << 8);
}
// DEPTH: window positions the filter looks at.  4 by default; 3 when the compiler expects three positions to be selective
// enough (ScanArgs::k3_depth: literal-like alternatives) -- two SDWA operations per byte instead of three and one look-up
// less per step, paid for with more trips into the confirm path.  (TWO positions -- one operation per byte, 16 instead of 32 per
// step, 94 VGPRs -- was built and measured in round 6: 15 - 45 % SLOWER for every pattern tried, the three-literal alternation
// included: with two positions a wave-step has a hit more often than not and the cold path runs every step.
// profiles/r06_aa_k3_depth_sweep.txt; the change itself: profiles/r06_aa_k3_two_position_filter.patch.)
// NW: waves per workgroup.  8 (512 threads, ITER 12: 4 waves per SIMD) or 12 (768 threads, ITER 8, two workgroups per CU
// = 6 waves per SIMD within 80 VGPRs: the same 96 KiB tile, fewer bytes in flight per wave, more waves to hide the LDS
// and HBM latency behind).
// VM: the pattern is inexact (its alternatives only say what a match must begin with) and never looks behind the match
// start: a filter hit goes through the two-byte viability table (DevProgram::vm_pair) at once and, if it survives, at the
// end of the sub-tile through the window tables and the pattern's backtracking VM (vm.h, program in LDS); a hit at which no
// match can start is dropped instead of being recorded for the host's matcher (DevProgram::vm_filter).  The text is read
// with the default cache policy in this form: the VM comes back to it.
constexpr int kK3Queue = 192;
__device__ __noinline__ bool vm_keep_hit_dev(const DevProgram *pg, const VmProg *vm, const uint8_t *seg, uint32_t slen, uint32_t q)
{
    return vm_keep_hit(pg, vm, seg, slen, q);
}

// PF: the next tile's text is requested while this one is computed (see k2_classrun_scan).
template <int ITER, bool NT, int DEPTH = 4, int NW = 8, bool VM = false, bool PF = false>
// (the VM form is bound by the latency of its cold path's dependent loads -- text, class bitmaps, the pair table, its stack in
// scratch memory: asked to fit four waves per SIMD, 120 VGPRs instead of 135, a second workgroup shares the CU)
#ifndef GSCAN_VM_WAVES
#define GSCAN_VM_WAVES 4
#endif
__global__ __launch_bounds__(NW * 64, NW == 12 ? 6 : VM ? GSCAN_VM_WAVES : 1) void k3_bucket_scan(ScanArgs a, const TileDesc *__restrict__ tiles)
{
    // The filter table, one copy PER LANE: entry b of lane l lives at byte address b << 8 | l << 2.  Both fields are whole
    // bytes, so ONE v_perm_b32 turns a text byte into its LDS address (with the 32-copy layout the address took a v_bfe
    // and a v_lshl_or: 19 look-ups per step made that the kernel's largest single VALU item), and lanes never collide.
    // 64 KiB per workgroup: 512 threads share it, two workgroups per CU, run as a persistent grid.
    __shared__ uint32_t tbl[256 * 64];
    __shared__ __attribute__((aligned(16))) uint8_t s_pos[kK3Confirm * 256];
    __shared__ uint8_t s_blen[kK3Buckets];
    __shared__ __attribute__((aligned(16))) uint32_t s_vm[VM ? sizeof(VmProg) / 4 : 1];
    __shared__ uint16_t s_queue[VM ? NW * kK3Queue : 1]; // (VM) the waves' survivor queues (below): 192 entries each, three rounds of 64 lanes
    constexpr uint32_t kTile = NW * ITER * 1024;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t lane4 = lane << 2;
    const uint32_t koff = a.k3_off, m = a.m;
    const bool exact = (DEPTH == 4 ? a.k3_exact : a.k3_exact3) != 0; // the filtered positions ARE the pattern (three positions: windows of <= 3 bytes)
    const bool confirm_exact = a.prog->k3_confirm_exact != 0; // wave-uniform (scalar load)
    const bool vm_quick = VM && a.prog->vm_pair_ok == 2;      // (DevProgram::vm_pair may drop a hit on its own)
    const bool one_bucket = a.k3_one_bucket != 0;             // every table byte is 0 or 1
    const uint8_t *tbl8 = reinterpret_cast<const uint8_t *>(tbl);
    {
        // dword q of the table = copy (q & 63) of entry (q >> 6), middle bytes swapped (see above); consecutive threads
        // write consecutive dwords
        for (uint32_t q = threadIdx.x; q < 256u * 64u; q += NW * 64) {
            const uint32_t v = a.prog->k3_table[q >> 6];
            tbl[q] = __builtin_amdgcn_perm(v, v, 0x03010200u);
        }
        // confirm tables (cold path only): kK3Confirm window positions x 256 bytes, one copy
        const u32x4 *src = reinterpret_cast<const u32x4 *>(&a.prog->k3_pos[0][0]);
        u32x4 *dst = reinterpret_cast<u32x4 *>(s_pos);
        for (uint32_t q = threadIdx.x; q < (uint32_t)(kK3Confirm * 256 / 16); q += NW * 64) dst[q] = src[q];
        if (threadIdx.x < (uint32_t)kK3Buckets) s_blen[threadIdx.x] = a.prog->k3_blen[threadIdx.x];
        if (VM) {
            const uint32_t *vsrc = reinterpret_cast<const uint32_t *>(&a.prog->vm);
            for (uint32_t q = threadIdx.x; q < (uint32_t)(sizeof(VmProg) / 4); q += NW * 64) s_vm[q] = vsrc[q];
        }
    }
    __syncthreads();

    // (the tile loop: see k2_classrun_scan)
    u32x4 buf[ITER + 1], halo_n = {0, 0, 0, 0};
    if (blockIdx.x >= a.n_tiles) return;
    uint32_t t = 0, tn = blockIdx.x;
    TileCtx c = tile_ctx(a, tiles, tn, kTile);
    int sub_off = 0;
    bool have = false, cur = false;
    for (;;) {
        const bool next = tn < a.n_tiles;
        TileCtx cn = c;
        int sub_off_n = 0;
        bool have_n = false;
        if (next) {
            cn = tile_ctx(a, tiles, tn, kTile);
            sub_off_n = cn.tile_off + (int)(wave * ITER * 1024);
            have_n = cn.live && sub_off_n < cn.slen;
        }
        uint32_t hits[(ITER + 1) / 2];
#pragma unroll
        for (int i = 0; i < (ITER + 1) / 2; i++) hits[i] = 0;
        uint32_t cnt = 0;

        // a filter hit at p against the first kK3Confirm (24) window positions of every alternative, then (VM) against
        // the pattern itself
        auto confirm_hit = [&](uint32_t p) -> bool {
            // aligned dword loads (each one bounds-checked on its own: an unaligned 16-byte load that
            // straddles the segment end comes back as zeros altogether), shifted into place
            const int pa = (int)(p & ~3u);
            uint32_t dw[kK3Confirm / 4 + 1];
#pragma unroll
            for (int q = 0; q <= kK3Confirm / 4; q++) dw[q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(c.rsrc, pa + 4 * q, 0, 0);
            uint32_t w[kK3Confirm / 4];
#pragma unroll
            for (int q = 0; q < kK3Confirm / 4; q++) w[q] = __builtin_amdgcn_alignbyte(dw[q + 1], dw[q], p & 3u);
            uint32_t mk = 0xffu;
#pragma unroll
            for (int q = 0; q < kK3Confirm; q++) mk &= s_pos[q * 256 + ((w[q >> 2] >> (8 * (q & 3))) & 0xffu)];
            bool ok = false;
            if (mk) {
                if (confirm_exact) { // the tables are the pattern: only the segment end is left to check
                    while (mk) {
                        const uint32_t b = (uint32_t)__ffs((int)mk) - 1u;
                        mk &= mk - 1u;
                        if (p + s_blen[b] <= (uint32_t)c.slen) ok = true;
                    }
                } else {
                    // shared buckets (> 8 alternatives) or windows longer than the tables: the tables may let a
                    // cross-product through.  Accept the hit here; k3_settle decides it, record by record.
                    ok = p + m <= (uint32_t)c.slen;
                }
            }
            // the device's own confirmation: can a match start here (or, for a gapped alternative, along the run
            // of repeat bytes that ends here) at all?
            if (VM && ok) ok = vm_keep_hit_dev(a.prog, reinterpret_cast<const VmProg *>(s_vm), c.seg, (uint32_t)c.slen, p);
            return ok;
        };
        if (have) {
            if (!PF) load_subtile<ITER, NT, 1>(buf, c, sub_off, lane);
            // filter positions q = p + koff of window starts p with 0 <= p and p + m <= slen
            const int lo = (int)koff, hi = c.slen - (int)m + (int)koff;
            const bool interior = sub_off >= lo && sub_off + ITER * 1024 - 1 <= hi; // every filter position of the sub-tile belongs to a window inside the segment
#pragma unroll
            for (int k = 0; k < ITER; k++) {
                const u32x4 d = buf[k];
                // the three bytes behind this lane's sixteen: the next lane's first dword (lane 63: the next step's lane 0)
                const uint32_t nd = down1(d.x, __builtin_amdgcn_readfirstlane(buf[k + 1].x));
                uint32_t e[19];
                // address = [lane << 2, text byte, 0, 0]: selector byte 0 = lane4.b0, byte 1 = w.b(i), 0x0c = constant zero
#define GS_E(i_, w_, by_) e[i_] = *reinterpret_cast<const uint32_t *>(tbl8 + __builtin_amdgcn_perm((w_), lane4, 0x0c0c0400u | ((uint32_t)(by_) << 8)))
                GS_E(0, d.x, 0);  GS_E(1, d.x, 1);  GS_E(2, d.x, 2);  GS_E(3, d.x, 3);
                GS_E(4, d.y, 0);  GS_E(5, d.y, 1);  GS_E(6, d.y, 2);  GS_E(7, d.y, 3);
                GS_E(8, d.z, 0);  GS_E(9, d.z, 1);  GS_E(10, d.z, 2); GS_E(11, d.z, 3);
                GS_E(12, d.w, 0); GS_E(13, d.w, 1); GS_E(14, d.w, 2); GS_E(15, d.w, 3);
                GS_E(16, nd, 0);  GS_E(17, nd, 1);
                if (DEPTH == 4) GS_E(18, nd, 2);
#undef GS_E
                if (PF) { // this step's text is look-up addresses now: its registers take the next tile's
                    buf[k] = load_step<ITER, NT, 1>(cn, sub_off_n, lane, k, have_n);
                    if (k == 0) halo_n = load_step<ITER, NT, 1>(cn, sub_off_n, lane, ITER, have_n); // (this tile's halo is in use until its last step)
                }
                // Y_j = [P_0(t_j) & P_1(t_{j+1}), P_2(t_j) & P_3(t_{j+1})]; h_j = Y_j.b0 & Y_{j+2}.b1 (four positions) or
                // Y_j.b0 & e_{j+2}.b1 = ... & P_2(t_{j+2}) (three): two VALU operations per byte either way
                // (plain shift + and in place of the selects: twice that -- profiles/r01_o_sweep_k3_sdwa.txt)
                uint32_t H[4]; // byte k of H[q] = h_{4q+k}: the buckets that accept the four filter positions from position 4 q + k on
                uint32_t Y[18];
#pragma unroll
                for (int j = 0; j < (DEPTH == 3 ? 16 : 18); j++) Y[j] = and_w0_w1(e[j], e[j + 1]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    and_b0_b1_into<0>(H[q], Y[4 * q], DEPTH == 3 ? e[4 * q + 2] : Y[4 * q + 2]);
                    and_b0_b1_into<1>(H[q], Y[4 * q + 1], DEPTH == 3 ? e[4 * q + 3] : Y[4 * q + 3]);
                    and_b0_b1_into<2>(H[q], Y[4 * q + 2], DEPTH == 3 ? e[4 * q + 4] : Y[4 * q + 4]);
                    and_b0_b1_into<3>(H[q], Y[4 * q + 3], DEPTH == 3 ? e[4 * q + 5] : Y[4 * q + 5]);
                }
                const uint32_t any = (H[0] | H[1] | H[2]) | H[3];
                if (any) { // cold: which positions, bounds, full windows
                    const int pos0 = sub_off + k * 1024 + (int)lane * 16;
                    const bool direct = !VM && exact && pos0 + 16 + kK3Depth <= c.slen; // filter == pattern, windows in bounds
                    uint32_t hm = nonzero_bytes16(H, one_bucket);
                    if (!interior) hm &= valid16(pos0, lo, hi);
                    uint32_t bits = hm;
                    if (VM) {
                        // Inexact patterns: here only the two-byte table (DevProgram::vm_pair) -- two byte loads and a bit per
                        // hit, and most hits end there.  What is left is confirmed and put to the VM after the sub-tile's last
                        // step (below), when the lanes' few survivors can be worked off side by side instead of one lane
                        // running the VM while 63 wait for it, hit after hit.
                        if (vm_quick) {
                            bits = 0;
                            while (hm) {
                                const uint32_t j = (uint32_t)__ffs((int)hm) - 1u;
                                hm &= hm - 1u;
                                const uint32_t p = (uint32_t)pos0 + j - koff;
                                bool ok = true;
                                if (p + 1 < (uint32_t)c.slen) {
                                    const uint32_t idx = (uint32_t)c.seg[p] << 8 | c.seg[p + 1];
                                    ok = (a.prog->vm_pair[idx >> 5] >> (idx & 31)) & 1u;
                                }
                                if (ok) bits |= 1u << j;
                            }
                        }
                    } else if (!exact) {
                        // The filter is not the pattern (class-heavy alternatives, windows longer than four positions): every
                        // hit has to be confirmed -- but not here.  Inside the step a hit costs the WAVE a trip through the
                        // confirm path (seven dependent loads, 24 LDS look-ups) with one or two lanes active; with a hit every
                        // KiB or two that was more than the scan itself ([0-9]+\.[0-9]+: 0.56 of the HBM roofline).  The hits
                        // are kept as they are and confirmed after the sub-tile's last step (below), each lane working off its
                        // own, side by side -- the structure the VM form has had since round 2.
                    } else if (!direct) {
                        // (the filter IS the pattern, only this window may reach past the segment end: rare, settled on the spot)
                        // confirm every hit against the first kK3Confirm (24) window positions: the window's bytes are
                        // re-read (beyond the segment the descriptor returns zeros), then one LDS byte per position
                        // gives the buckets that accept the byte there
                        bits = 0;
                        while (hm) {
                            const uint32_t j = (uint32_t)__ffs((int)hm) - 1u;
                            hm &= hm - 1u;
                            if (confirm_hit((uint32_t)pos0 + j - koff)) bits |= 1u << j;
                        }
                    }
                    // keep group starts (within the lane; a superset of them is fine) -- unless hits may still be struck
                    // out by k3_settle, or have been by the VM: a real hit must not be dropped for following a false one
                    if (!VM && exact && (direct || confirm_exact)) bits &= ~((bits << 1) & ~a.keep_all);
                    hits[k >> 1] |= bits << (16 * (k & 1));
                    cnt += (uint32_t)__popc(bits);
                }
            }
        } else if (PF) {
            buf[0] = load_step<ITER, NT, 1>(cn, sub_off_n, lane, 0, have_n);
            halo_n = load_step<ITER, NT, 1>(cn, sub_off_n, lane, ITER, have_n);
#pragma unroll
            for (int j = 1; j < ITER; j++) buf[j] = load_step<ITER, NT, 1>(cn, sub_off_n, lane, j, have_n);
        }
        if (!VM && !exact && cur) { // the filter is not the pattern: the sub-tile's hits confirmed word by word, each lane working off its own
            cnt = 0;
#pragma unroll
            for (int w = 0; w < (ITER + 1) / 2; w++) {
                uint32_t word = hits[w], keep = word;
                while (word) {
                    const uint32_t b = (uint32_t)__ffs((int)word) - 1u;
                    word &= word - 1u;
                    const uint32_t k = 2u * (uint32_t)w + (b >> 4), j = b & 15u;
                    const uint32_t p = (uint32_t)(sub_off + (int)k * 1024 + (int)lane * 16) + j - koff;
                    if (!confirm_hit(p)) keep &= ~(1u << b);
                }
                // keep group starts (within a lane's sixteen positions of one step) where the tables have the last word
                if (confirm_exact) keep &= ~((keep << 1) & 0xfffefffeu & ~a.keep_all);
                hits[w] = keep;
                cnt += (uint32_t)__popc(keep);
            }
        }
        if (VM && cur) {
            // The survivors of the sub-tile go through confirm (+ VM) from a WAVE-WIDE QUEUE (round 5): every lane writes the
            // positions of its own hits into a strip of LDS at the rank a wave scan of the lanes' counts gives them, then lane i
            // takes entry i (+ 64, + 128) -- 64 survivors run side by side whichever lanes they came from.  Until round 4 each
            // lane worked off its own hits, word by word: with a hit every 150 bytes ([a-z]+\([a-z0-9, ]*\);) a word's loop
            // ran as often as its fullest lane had hits with a fifth, then a fiftieth of the lanes active -- ~7 % of the lanes
            // busy in the one part of the kernel that costs hundreds of instructions per hit (31 GB/s).  Verdicts come back
            // through the same strip (a dropped entry is overwritten) and the owners clear their bits.
            uint32_t mine = 0;
#pragma unroll
            for (int w = 0; w < (ITER + 1) / 2; w++) mine += (uint32_t)__popc(hits[w]);
            const uint32_t incl = wave_scan(mine), excl = incl - mine;
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            volatile uint16_t *q = s_queue + wave * kK3Queue;
            uint32_t keep[(ITER + 1) / 2]; // the verdicts; `hits` stays as it is until the last round: a survivor's rank is its rank among ALL hits
#pragma unroll
            for (int w = 0; w < (ITER + 1) / 2; w++) keep[w] = hits[w];
            for (uint32_t base = 0; base < total; base += kK3Queue) { // (wave-uniform; one round unless a sub-tile has > kK3Queue survivors)
                uint32_t idx = excl;
#pragma unroll
                for (int w = 0; w < (ITER + 1) / 2; w++) {
                    uint32_t word = hits[w];
                    while (word) {
                        const uint32_t b = (uint32_t)__ffs((int)word) - 1u;
                        word &= word - 1u;
                        if (idx - base < (uint32_t)kK3Queue) q[idx - base] = (uint16_t)((2u * (uint32_t)w + (b >> 4)) * 1024u + lane * 16u + (b & 15u));
                        idx++;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t n = min(total - base, (uint32_t)kK3Queue);
                for (uint32_t i = lane; i < n; i += kWave) {
                    const uint32_t e = q[i];
                    if (!confirm_hit((uint32_t)sub_off + e - koff)) q[i] = 0xffffu;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                idx = excl;
#pragma unroll
                for (int w = 0; w < (ITER + 1) / 2; w++) {
                    uint32_t word = hits[w];
                    while (word) {
                        const uint32_t b = (uint32_t)__ffs((int)word) - 1u;
                        word &= word - 1u;
                        if (idx - base < (uint32_t)kK3Queue && q[idx - base] == 0xffffu) keep[w] &= ~(1u << b);
                        idx++;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            cnt = 0;
#pragma unroll
            for (int w = 0; w < (ITER + 1) / 2; w++) {
                hits[w] = keep[w];
                cnt += (uint32_t)__popc(hits[w]); // (no group-start compression: a real hit must not be dropped for following one the VM struck out)
            }
        }
        if (cur) emit_wave<ITER>(a, t * NW + wave, hits, cnt, sub_off, koff - a.report_shift, lane);
        if (!next) break;
        if (PF) buf[ITER] = halo_n;
        t = tn;
        tn += gridDim.x;
        c = cn;
        sub_off = sub_off_n;
        have = have_n;
        cur = true;
    }
}

// ------------------------------------------------------------------------------------
// K3, second pass for patterns whose filter tables are not the whole truth (more than 8 alternatives share the 8
// buckets, or a window is longer than the confirm tables): every record of the first pass is checked against the
// alternatives themselves.  One thread per tile -- almost all tiles have no record -- walking that tile's records; a
// record that is no match is overwritten with kStruck and counted, the readers of the record buffer skip it.
// Doing this inside the scan kernel costs its hot loop 18 % (registers): profiles/r01_o_sweep_k3_compare_word_confirm_rejected.txt.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_settle(ScanArgs a, const TileDesc *__restrict__ tiles, uint32_t nw)
{
    const uint32_t st = blockIdx.x * 256u + threadIdx.x; // one thread per descriptor = per wave sub-tile
    if (st >= a.n_tiles * nw) return;
    const uint32_t t = st / nw;
    const unsigned long long d = a.desc[st];
    const uint32_t cnt = (uint32_t)d;
    if (cnt == 0) return;
    if (a.counter[kShards * kCtrStride] != 0) return; // some shard overflowed: the host rescans with a bigger buffer (and settles then)
    const uint32_t base = (uint32_t)(d >> 32);
    uint64_t seg_off;
    uint32_t slen;
    if (tiles) {
        seg_off = tiles[t].seg_off;
        slen = tiles[t].seg_len;
    } else {
        seg_off = a.seg0_off;
        slen = a.seg0_len;
    }
    const uint8_t *seg = a.base + seg_off;
    uint32_t struck = 0;
    for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t p = a.recs[base + i] - a.report_shift; // device window start
        if (!verify_alts(seg, slen, a.prog, p, 0xffu)) {
            a.recs[base + i] = kStruck;
            struck++;
        }
    }
    if (struck) atomicAdd(a.counter + kShards * kCtrStride + 1, struck);
}

// ------------------------------------------------------------------------------------
// Line extents and orbit selection on the device, for the reference's line-printing modes
// (/root/reference/src/grab.cc:188-209: print the line around the match, then restart at the END OF THAT LINE).
// Applies to patterns with one plain alternative none of whose classes contains a newline (DevProgram::lines_ok): a
// match then lies inside one line, the restart position after a printed match is that line's newline, and the matches
// that get printed are exactly THE FIRST CANDIDATE OF EVERY LINE -- a per-line question.  (A candidate that is first in
// its line heads a group of consecutive candidates, so it is in the record list.)
// One wave per tile, one lane per record; written parallel to the records, ext[i] = {m1, lb, le}:
//   m1 == 0           the record is not printed (an earlier candidate sits in the same line)
//   lb == kLineAsk    "ask the host": something here needs the reference's loop itself -- the 511-byte caps of
//                     grab.cc:173 (a line that runs on past the printed context may print again), a line start or a
//                     tail further away than this pass looks.  The host falls back to its walk from there on.
//   else              print bytes [lb, m0) + match [m0, m1) + [m1, le); the loop restarts at le.
// ------------------------------------------------------------------------------------
constexpr uint32_t kLineAsk = 0xffffffffu;
constexpr uint32_t kLineBack = 4096; // how far a line start / a tail end is searched before the host is asked
static_assert(kLineBack % 32 == 0, "k_lines looks back / ahead 32 bytes per step: the ask-the-host bound is tested once per step");

// exact "which bytes of x are zero" (bit 7 of every zero byte), no borrow across bytes
__device__ __forceinline__ unsigned long long zero_bytes(unsigned long long x)
{
    const unsigned long long k7f = 0x7f7f7f7f7f7f7f7full;
    return ~(((x & k7f) + k7f) | x | k7f);
}
__device__ __forceinline__ unsigned long long load8(const uint8_t *q) // any alignment
{
    unsigned long long v;
    __builtin_memcpy(&v, q, 8);
    return v;
}

// One wave per tile, one lane per record (round-robin): whether a record is printed depends only on the record in front
// of it -- it is the first candidate of its line iff the previous record lies before the line's start -- so nothing is
// carried from record to record.  Newlines are searched 8 bytes at a time.
// (descriptors are per wave sub-tile: st = tile * nw + wave covers sub_bytes bytes; the descriptors of a segment are
// consecutive, those of a segment's last tile beyond its end are empty)
// (Round 5, measured and not adopted: FOUR waves per descriptor, wave w taking the batches of 64 records w, w + 4, ... -- 125
// against 119 us per 64 MiB window, profiles/r05_j_per_window_passes.txt: the pass is not paced by descriptors with two batches
// but by its longest chains of dependent loads, the records in lines of several hundred bytes.  Those chains are what the
// 32-byte steps below shorten.)
__global__ __launch_bounds__(64) void k_lines(ScanArgs a, const TileDesc *__restrict__ tiles, uint32_t nw, uint32_t sub_bytes, uint32_t *__restrict__ ext,
                                              uint8_t *__restrict__ gather, uint32_t gather_cap)
{
    const uint32_t st = blockIdx.x;
    const uint32_t t = st / nw;
    const uint32_t tile_bytes = sub_bytes; // (the walk back over earlier descriptors below steps by sub-tiles)
    const unsigned long long d = a.desc[st];
    const uint32_t cnt = (uint32_t)d;
    if (cnt == 0 || a.counter[kShards * kCtrStride] != 0) return; // (overflow: the host rescans with a bigger buffer and this pass runs again)
    __shared__ uint8_t s_tail[256]; // the tail class, one byte per byte value (one wave per workgroup: its own writes, then its reads)
    for (uint32_t b = threadIdx.x; b < 256u; b += 64u) s_tail[b] = (uint8_t)((a.prog->tail_bits[b >> 5] >> (b & 31u)) & 1u);
    __syncthreads();
    const uint32_t base = (uint32_t)(d >> 32);
    uint64_t seg_off;
    uint32_t slen, tile_off;
    if (tiles) {
        seg_off = tiles[t].seg_off;
        slen = tiles[t].seg_len;
        tile_off = tiles[t].tile_off + (st % nw) * sub_bytes;
    } else {
        seg_off = a.seg0_off;
        slen = a.seg0_len;
        tile_off = st * sub_bytes;
    }
    const uint8_t *seg = a.base + seg_off;
    const DevProgram *pg = a.prog;
    const uint32_t m = a.m, tail_extra = pg->tail_extra;
    const unsigned long long kNl = 0x0a0a0a0a0a0a0a0aull;
    const uint32_t lane = threadIdx.x;
    // 64 records at a time: every lane settles one record (printed? its match end, line begin, line end) and copies its
    // printed line's text into the gather buffer -- one reservation per 64 records, the lines back to back in record order
    for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t mylen = 0, mylb = 0; // the printed line [lb, le) of this lane's record (0: nothing to gather)
        uint32_t *e = ext + 4ull * (base + i);
        if (i < cnt) {
            const uint32_t p = a.recs[base + i];
            // start of p's line: the byte after the last newline before p
            uint32_t ls = p;
            bool found = false;
            // (32 bytes per step while they are there -- four loads in flight instead of one: a line of several hundred bytes is a
            // chain of that many dependent round trips otherwise, and the launch takes as long as its longest chain)
            while (!found && ls >= 32 && p - ls < kLineBack) {
                const unsigned long long z3 = zero_bytes(load8(seg + ls - 8) ^ kNl), z2 = zero_bytes(load8(seg + ls - 16) ^ kNl);
                const unsigned long long z1 = zero_bytes(load8(seg + ls - 24) ^ kNl), z0 = zero_bytes(load8(seg + ls - 32) ^ kNl);
                if (z3 | z2 | z1 | z0) { // the nearest newline: the highest zero byte of the highest piece that has one
                    const unsigned long long z = z3 ? z3 : z2 ? z2 : z1 ? z1 : z0;
                    const uint32_t at = z3 ? 8u : z2 ? 16u : z1 ? 24u : 32u;
                    ls = ls - at + (uint32_t)((63 - __clzll((long long)z)) >> 3) + 1;
                    found = true;
                } else {
                    ls -= 32;
                }
            }
            while (!found && ls >= 8 && p - ls < kLineBack) {
                const unsigned long long z = zero_bytes(load8(seg + ls - 8) ^ kNl);
                if (z) {
                    ls = ls - 8 + (uint32_t)((63 - __clzll((long long)z)) >> 3) + 1; // highest zero byte = the nearest newline
                    found = true;
                } else {
                    ls -= 8;
                }
            }
            while (!found && ls > 0 && p - ls < kLineBack + 8) {
                if (seg[ls - 1] == '\n') found = true;
                else ls--;
            }
            int verdict = 1; // 0 not printed, 1 printed, 2 ask the host
            if (!found && ls > 0) verdict = 2; // no line start within reach
            // an earlier candidate in [ls, p)?  In this tile: the previous record.  Before it: the last record of the nearest
            // earlier tile of this segment that has one (tiles of a segment are consecutive: tile t - k starts k tiles earlier).
            if (verdict == 1) {
                bool first = true;
                if (i > 0) {
                    first = a.recs[base + i - 1] < ls;
                } else if (ls < tile_off) {
                    uint32_t u = st, uoff = tile_off;
                    while (uoff > ls) { // descriptor u - 1 covers [uoff - tile_bytes, uoff)
                        u--;
                        uoff -= tile_bytes;
                        const unsigned long long du = a.desc[u];
                        if ((uint32_t)du) {
                            first = a.recs[(uint32_t)(du >> 32) + (uint32_t)du - 1u] < ls;
                            break;
                        }
                    }
                }
                if (!first) verdict = 0;
            }
            uint32_t m1 = p + m, le = 0;
            if (verdict == 1) {
                // the match: window + greedy tail -- eight text bytes per load, the class test a 256-byte table in LDS (as k_ends
                // does it; a byte load and a look-up in the program's bitmap in global memory per tail byte were two dependent
                // round trips each)
                uint32_t extra = 0;
                bool open = true; // the tail has not been seen to stop yet
                while (open && m1 + 8 <= slen && extra + 8 <= tail_extra && extra < kLineBack) {
                    const unsigned long long v = load8(seg + m1);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (open && !s_tail[(uint32_t)(v >> (8 * k)) & 0xffu]) open = false;
                        if (open) m1++, extra++;
                    }
                }
                while (open && m1 < slen && extra < tail_extra) {
                    if (!s_tail[seg[m1]]) break;
                    m1++;
                    extra++;
                    if (extra >= kLineBack && extra < tail_extra) {
                        verdict = 2;
                        break;
                    }
                }
                // the rest of the line, at most 511 bytes of it (grab.cc:173,194-196)
                le = m1;
                found = false;
                while (!found && le + 32 <= slen && le - m1 < 480u) { // (32 bytes per step: see the line start)
                    const unsigned long long z0 = zero_bytes(load8(seg + le) ^ kNl), z1 = zero_bytes(load8(seg + le + 8) ^ kNl);
                    const unsigned long long z2 = zero_bytes(load8(seg + le + 16) ^ kNl), z3 = zero_bytes(load8(seg + le + 24) ^ kNl);
                    if (z0 | z1 | z2 | z3) { // the next newline: the lowest zero byte of the lowest piece that has one
                        const unsigned long long z = z0 ? z0 : z1 ? z1 : z2 ? z2 : z3;
                        le += (z0 ? 0u : z1 ? 8u : z2 ? 16u : 24u) + ((uint32_t)(__ffsll((long long)z) - 1) >> 3);
                        found = true;
                    } else {
                        le += 32;
                    }
                }
                while (!found && le + 8 <= slen && le - m1 < 504u) {
                    const unsigned long long z = zero_bytes(load8(seg + le) ^ kNl);
                    if (z) {
                        le += (uint32_t)(__ffsll((long long)z) - 1) >> 3; // lowest zero byte = the next newline
                        found = true;
                    } else {
                        le += 8;
                    }
                }
                while (!found && le < slen && seg[le] != '\n' && le - m1 < 511u) le++;
                if (le < slen && seg[le] != '\n') verdict = 2; // the line runs on: what follows may print again
            }
            if (verdict == 0) {
                e[0] = 0;
            } else if (verdict == 2) {
                e[0] = 1;
                e[1] = kLineAsk;
            } else {
                e[0] = m1;
                e[1] = mylb = p - ls > 511u ? p - 511u : ls; // at most 511 bytes in front of the match (grab.cc:173,190-193)
                e[2] = le;
                mylen = le - mylb;
            }
        }
        // gather: the wave's printed lines, back to back; every record learns where its text went (kLineAsk: nowhere --
        // the buffer is full or there is none, the host copies from the file)
        const uint32_t incl = wave_scan(mylen);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        uint32_t gb = 0;
        if (gather && total && lane == 0) gb = atomicAdd(a.counter + kShards * kCtrStride + 2, total);
        gb = __builtin_amdgcn_readfirstlane(gb);
        const bool room = gather && total && (unsigned long long)gb + total <= (unsigned long long)gather_cap;
        if (i < cnt && mylen) e[3] = room ? gb + incl - mylen : kLineAsk;
        if (room && mylen) { // every lane copies its own line, eight bytes at a time (any alignment), the lanes side by side
            const uint8_t *src = seg + mylb;
            uint8_t *dst = gather + (gb + incl - mylen);
            uint32_t o = 0;
            for (; o + 32 <= mylen; o += 32) { // (four loads, then four stores: the text and the gather buffer do not overlap)
                const unsigned long long v0 = load8(src + o), v1 = load8(src + o + 8), v2 = load8(src + o + 16), v3 = load8(src + o + 24);
                __builtin_memcpy(dst + o, &v0, 8);
                __builtin_memcpy(dst + o + 8, &v1, 8);
                __builtin_memcpy(dst + o + 16, &v2, 8);
                __builtin_memcpy(dst + o + 24, &v3, 8);
            }
            for (; o + 8 <= mylen; o += 8) {
                const unsigned long long v = load8(src + o);
                __builtin_memcpy(dst + o, &v, 8);
            }
            for (; o < mylen; o++) dst[o] = src[o];
        }
    }
}

// Match ends for -O -l (grab.cc:175-213 with a == 0): one wave per descriptor, one lane per record.  The match that
// starts at a listed p ends where its greedy tail stops: the first byte from p + m on outside the tail class, or the
// segment end (DevProgram::ends_ok: one plain alternative, unbounded tail).  Eight text bytes per load, the class test a
// 256-byte LDS table.  A tail that runs on for more than kLineBack bytes is left to the host (0).
__global__ __launch_bounds__(64) void k_ends(ScanArgs a, const TileDesc *__restrict__ tiles, uint32_t nw, uint32_t *__restrict__ ends)
{
    __shared__ uint8_t s_tail[256];
    const DevProgram *pg = a.prog;
    for (uint32_t b = threadIdx.x; b < 256u; b += 64u) s_tail[b] = (uint8_t)((pg->tail_bits[b >> 5] >> (b & 31u)) & 1u);
    __syncthreads();
    const uint32_t st = blockIdx.x;
    const unsigned long long d = a.desc[st];
    const uint32_t cnt = (uint32_t)d;
    if (cnt == 0 || a.counter[kShards * kCtrStride] != 0) return; // (overflow: the host rescans with a bigger buffer and this pass runs again)
    const uint32_t base = (uint32_t)(d >> 32);
    const uint32_t t = st / nw;
    const uint8_t *seg = a.base + (tiles ? tiles[t].seg_off : a.seg0_off);
    const uint32_t slen = tiles ? tiles[t].seg_len : a.seg0_len;
    const uint32_t m = a.m;
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) {
        const uint32_t p = a.recs[base + i];
        if (p == kStruck) {
            ends[base + i] = 0;
            continue;
        }
        uint32_t e = p + m;
        bool open = true; // the tail has not been seen to stop yet
        while (open && e + 8 <= slen && e - (p + m) < kLineBack) {
            const unsigned long long v = load8(seg + e);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (open && !s_tail[(uint32_t)(v >> (8 * k)) & 0xffu]) open = false;
                if (open) e++;
            }
        }
        while (open && e < slen && e - (p + m) < kLineBack + 8) {
            if (!s_tail[seg[e]]) open = false;
            else e++;
        }
        ends[base + i] = (open && e < slen) ? 0u : e; // still open short of the segment end: too long, the host's
    }
}

// ------------------------------------------------------------------------------------
// The resolve pass (round 6; DevProgram::resolve): the pattern itself, run AT every record.
// What it replaces: the reference's pcre_exec (/root/reference/src/grab.cc:178) as far as "is this offset a match start, and
// where does that match end" goes -- rounds 1-5 left that question, for every pattern that is not one plain window, to the
// host's backtracking matcher inside the report loop, candidate by candidate (35-320 ns each).  Here the scan
// kernels have listed every offset where one of the pattern's START windows fits (ScanArgs::keep_all: no group-start
// compression); this pass runs the pattern's VM program (vm.h -- TreeMatch's order of exploration construct by construct,
// i.e. PCRE's) at each of them with the segment's real bytes in front (s0 = 0), and
//   * drops the records at which no match starts: the survivors are moved to the front of the descriptor's run, the
//     descriptor's count shrinks, counter[kShards * kCtrStride + 1] counts the dropped ones;
//   * writes, parallel to the surviving records, ends[i] = the match's end -- or GSCAN_END_CAPTURES when its path closed a
//     capturing group (the reference's one-pair ovector: pcre_exec returns 0 and the chunk loop ends, grab.cc:171,179), or
//     GSCAN_END_ASK when the VM gave up (step / stack limit) or the answer looks odd: the host's matcher decides that one.
// One workgroup of four waves per descriptor, one lane per record: first every record's verdict, side by side, then the
// survivors are moved to the front of the run.  A verdict reached this way is pcre_exec's for every restart
// position s <= p - reach (Database::reach): the host asks its own matcher about the offsets closer to s than that
// (gscan_next_resolved).
// ------------------------------------------------------------------------------------
constexpr uint32_t kResolveWG = 256;   // threads per descriptor: four waves run the VM side by side, then compact the run together
constexpr uint32_t kResolveChunkMax = 1024;  // descriptors one workgroup looks after, at most
constexpr uint32_t kResolveDrop = 0xffffffffu; // (in ends[], between the two phases: no match starts at this record)
// chunk: descriptors per workgroup (<= kResolveChunkMax).  1 for a window of the host-chunk path (a few thousand descriptors,
// many of them with hundreds of records: one workgroup each); hundreds for an arena of millions of descriptors nearly all of
// which are empty (the device-resident path over 64 GiB: 5.6 M).  The workgroup fetches its descriptors' counts with coalesced
// loads and leaves at once if there is no record; otherwise the records of ALL its descriptors form one work list (a prefix sum
// over the counts; a record finds its descriptor by bisection) that the four waves work off side by side -- a launch with
// 65 536 records scattered over 5.6 M descriptors costs what its records cost, not a pass per descriptor.
__global__ __launch_bounds__(kResolveWG) void k_resolve(ScanArgs a, const TileDesc *__restrict__ tiles, uint32_t nw, uint32_t *__restrict__ ends, uint32_t chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_vm[sizeof(VmProg) / 4];
    __shared__ uint32_t s_off[kResolveChunkMax + 1]; // first the descriptors' record counts, then their exclusive prefix sums; [kResolveChunkMax] = the total
    __shared__ uint32_t s_wave[kResolveWG / 64];
    const uint32_t n_desc = a.n_tiles * nw, st0 = blockIdx.x * chunk;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool overflow = a.counter[kShards * kCtrStride] != 0; // (the host rescans with a bigger buffer and this pass runs again)
    for (uint32_t k = threadIdx.x; k < kResolveChunkMax; k += kResolveWG) {
        const uint32_t stq = st0 + k;
        s_off[k] = !overflow && k < chunk && stq < n_desc ? (uint32_t)a.desc[stq] : 0u;
    }
    __syncthreads();
    constexpr uint32_t kPer = kResolveChunkMax / kResolveWG; // consecutive entries per thread
    uint32_t c[kPer], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) sum += c[q] = s_off[kPer * threadIdx.x + q];
    const uint32_t incl = wave_scan(sum);
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kResolveWG / 64; w++) {
        const uint32_t v = s_wave[w];
        before += w < wave ? v : 0u;
        total += v;
    }
    if (total == 0u) return;
    {
        uint32_t at = before + incl - sum;
#pragma unroll
        for (uint32_t q = 0; q < kPer; q++) {
            s_off[kPer * threadIdx.x + q] = at;
            at += c[q];
        }
        if (threadIdx.x == 0) s_off[kResolveChunkMax] = total;
        const uint32_t *vsrc = reinterpret_cast<const uint32_t *>(&a.prog->vm);
        for (uint32_t q = threadIdx.x; q < (uint32_t)(sizeof(VmProg) / 4); q += kResolveWG) s_vm[q] = vsrc[q];
    }
    __syncthreads();
    const VmProg *vm = reinterpret_cast<const VmProg *>(s_vm);
    // phase 1: every record's verdict, in place (ends[i]); the records stay where they are
    for (uint32_t r = threadIdx.x; r < total; r += kResolveWG) {
        uint32_t j = 0; // the descriptor record r belongs to: the last one whose prefix sum is <= r
#pragma unroll
        for (uint32_t step = kResolveChunkMax / 2; step; step >>= 1)
            if (s_off[j + step] <= r) j += step;
        const uint32_t st = st0 + j, i = r - s_off[j];
        const uint32_t base = (uint32_t)(a.desc[st] >> 32);
        const uint32_t t = st / nw;
        const uint8_t *seg = a.base + (tiles ? tiles[t].seg_off : a.seg0_off);
        const uint32_t slen = tiles ? tiles[t].seg_len : a.seg0_len;
        const uint32_t p = a.recs[base + i];
        uint32_t code = kResolveDrop;
        if (p != kStruck && p < slen) {
            VmOut o{0u, 0u};
            const int v = vm_run(vm, seg, slen, p, 0u, o);
            if (v != 0) code = resolve_code(a.prog, seg, slen, p, v, o); // (2 = gave up: GSCAN_END_ASK, the host's matcher decides)
        }
        ends[base + i] = code;
    }
    __syncthreads();
    // phase 2: the survivors to the front of every descriptor's run, a wave per descriptor, 64 records per round.  (Survivor k of a
    // round goes to index out + k <= the index its own lane read from: a round's loads are done before its first store, and
    // later rounds read further on.)
    for (uint32_t j = wave; j < chunk; j += kResolveWG / 64) {
        const uint32_t cnt = s_off[j + 1] - s_off[j];
        if (cnt == 0u) continue; // (wave-uniform)
        const uint32_t st = st0 + j;
        const unsigned long long d = a.desc[st];
        const uint32_t base = (uint32_t)(d >> 32);
        uint32_t out = 0;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64u) {
            const uint32_t i = i0 + lane;
            uint32_t p = 0, code = kResolveDrop;
            if (i < cnt) {
                p = a.recs[base + i];
                code = ends[base + i];
            }
            const bool keep = code != kResolveDrop;
            const unsigned long long m = __ballot(keep);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (keep && out + rank != i) {
                a.recs[base + out + rank] = p;
                ends[base + out + rank] = code;
            }
            out += (uint32_t)__popcll(m);
        }
        if (lane == 0 && out != cnt) {
            a.desc[st] = (d & 0xffffffff00000000ull) | out;
            atomicAdd(a.counter + kShards * kCtrStride + 1, cnt - out);
        }
    }
}

// ------------------------------------------------------------------------------------
// Ordered compaction of a chunk's result.  The scan kernels leave the records in 64 shard regions, every wave's run where
// its reservation fell; the descriptors say where, in text order.  Fetching that took the host one strided 64-row copy per
// array (2.7 ms per window for 2 MB: the rows go one by one) and a merge over ~5 000 runs.  These two kernels write the
// records -- and their per-record extras -- once more, in TEXT ORDER and back to back: the host fetches ONE linear range
// and hands it out as it is.
//   k_order_prefix (one workgroup): dpos[d] = number of records of the descriptors before d; counter[.. + 3] = their total
//   k_order_copy   (one wave per descriptor): out[dpos[d] + i] = recs[base_d + i], out_ext likewise (ew words per record)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_order_prefix(const unsigned long long *__restrict__ desc, uint32_t n_desc, uint32_t *__restrict__ dpos,
                                                       uint32_t *__restrict__ counter)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = (n_desc + 1023u) / 1024u;
    const uint32_t lo = min(tid * chunk, n_desc), hi = min(lo + chunk, n_desc);
    uint32_t sum = 0;
    for (uint32_t d = lo; d < hi; d++) sum += (uint32_t)desc[d];
    const uint32_t incl = wave_scan(sum);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        const uint32_t v = s_wave[w];
        before += w < wave ? v : 0u;
        total += v;
    }
    uint32_t at = before + incl - sum;
    for (uint32_t d = lo; d < hi; d++) {
        dpos[d] = at;
        at += (uint32_t)desc[d];
    }
    if (tid == 0) counter[kShards * kCtrStride + 3] = total;
}

__global__ __launch_bounds__(256) void k_order_copy(const uint32_t *__restrict__ recs, const uint32_t *__restrict__ ext, uint32_t ew,
                                                    const unsigned long long *__restrict__ desc, const uint32_t *__restrict__ dpos, uint32_t n_desc,
                                                    uint32_t *__restrict__ out, uint32_t *__restrict__ out_ext, const uint32_t *__restrict__ counter)
{
    const uint32_t d = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (d >= n_desc || counter[kShards * kCtrStride] != 0) return; // (overflow: the host rescans with a bigger buffer and this runs again)
    const unsigned long long dd = desc[d];
    const uint32_t cnt = (uint32_t)dd, base = (uint32_t)(dd >> 32), p = dpos[d];
    for (uint32_t i = lane; i < cnt; i += 64u) {
        out[p + i] = recs[base + i];
        for (uint32_t w = 0; w < ew; w++) out_ext[(size_t)(p + i) * ew + w] = ext[(size_t)(base + i) * ew + w];
    }
}

} // namespace

// ---- host-callable launchers (engine.hip) ----
static const int kIters[4] = {16, 8, 12, 16};

// K2 runs its pair-table form whenever the pattern has at most two classes
static bool k2_pair(const ScanArgs &a) { return a.n_classes <= 2; }

static int variant_wg(int tier, int variant, uint32_t n_classes);

uint32_t scan_tile_bytes_vm() { return 8u * 8u * 1024u; } // K3 with the VM: 8 waves x 8 KiB whatever the variant

// The shape of the launch launch_scan() picks: bytes per workgroup tile and waves per workgroup.  Every wave writes one
// descriptor per tile (desc[tile * waves + wave]) for its sub-tile of tile_bytes / waves bytes.
bool k2_lane_form(int tier, int variant, uint32_t m, uint32_t nruns)
{
    return tier == GSCAN_TIER_CLASSRUN && (variant & kVariantLane) && m <= 17u && nruns >= 1u;
}

void scan_geometry(int tier, int variant, const DevProgram &pg, uint32_t *tile_bytes, uint32_t *waves)
{
    if (k2_lane_form(tier, variant, pg.m, pg.nruns)) {
        *tile_bytes = k2_lane_tile_bytes();
        *waves = k2_lane_waves();
        return;
    }
    if (tier == GSCAN_TIER_BUCKET && pg.vm_filter) {
        *tile_bytes = scan_tile_bytes_vm();
        *waves = 8;
        return;
    }
    *tile_bytes = scan_tile_bytes(tier, variant, pg.n_classes);
    *waves = *tile_bytes / ((uint32_t)kIters[variant & 3] * 1024u);
}

uint32_t scan_tile_bytes(int tier, int variant, uint32_t n_classes)
{
    const int wg = variant_wg(tier, variant, n_classes);
    const int waves = wg ? wg : ((tier == GSCAN_TIER_CLASSRUN && n_classes <= 2) || tier == GSCAN_TIER_BUCKET) ? 8 : kWaves;
    return (uint32_t)(waves * kIters[variant & 3] * 1024);
}

uint32_t scan_min_tile_bytes() { return (uint32_t)(kWaves * 8 * 1024); }

// resident workgroups per CU the kernel is designed for when it runs as a persistent grid (0 = no preference)
uint32_t scan_persistent_blocks(int tier, int variant, const DevProgram &pg)
{
    const uint32_t n_classes = pg.n_classes;
    if (k2_lane_form(tier, variant, pg.m, pg.nruns)) return 2u; // (64 KiB of table per workgroup)
    if ((tier == GSCAN_TIER_CLASSRUN && n_classes <= 2) || tier == GSCAN_TIER_BUCKET) return 2u; // (64 KiB of table to stage per workgroup)
    if (tier == GSCAN_TIER_CLASSRUN && variant_wg(tier, variant, n_classes) == 8) return 3u;    // (32 KiB, 512 threads)
    return 0u;
}

template <int ITER, bool NT>
static void launch_k2(bool wide, bool pair, int wg, const ScanArgs &a, dim3 g, hipStream_t st)
{
    const TileDesc *tiles = a.tiles;
    const bool w12 = wg == 12;
    if (pair && w12 && ITER == 8 && NT) { // 768-thread workgroups, two per CU
        constexpr int I = 8;
        if (wide) hipLaunchKernelGGL((k2_classrun_scan<I, true, true, true, -1, 12>), g, dim3(768), 0, st, a, tiles);
        else if (a.nruns == 1) hipLaunchKernelGGL((k2_classrun_scan<I, true, false, true, 1, 12>), g, dim3(768), 0, st, a, tiles);
        else if (a.nruns == 2) hipLaunchKernelGGL((k2_classrun_scan<I, true, false, true, 2, 12>), g, dim3(768), 0, st, a, tiles);
        else if (a.nruns == 3) hipLaunchKernelGGL((k2_classrun_scan<I, true, false, true, 3, 12>), g, dim3(768), 0, st, a, tiles);
        else hipLaunchKernelGGL((k2_classrun_scan<I, true, false, true, 0, 12>), g, dim3(768), 0, st, a, tiles);
    } else if (pair && wg == -1 && ITER == 12 && NT) { // the prefetching form
        if (wide) hipLaunchKernelGGL((k2_classrun_scan<12, true, true, true, -1, 8, true>), g, dim3(512), 0, st, a, tiles);
        else if (a.nruns == 1) hipLaunchKernelGGL((k2_classrun_scan<12, true, false, true, 1, 8, true>), g, dim3(512), 0, st, a, tiles);
        else if (a.nruns == 2) hipLaunchKernelGGL((k2_classrun_scan<12, true, false, true, 2, 8, true>), g, dim3(512), 0, st, a, tiles);
        else if (a.nruns == 3) hipLaunchKernelGGL((k2_classrun_scan<12, true, false, true, 3, 8, true>), g, dim3(512), 0, st, a, tiles);
        else hipLaunchKernelGGL((k2_classrun_scan<12, true, false, true, 0, 8, true>), g, dim3(512), 0, st, a, tiles);
    } else if (pair) {
        if (wide) hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, true, true>), g, dim3(512), 0, st, a, tiles);
        else if (a.nruns == 1) hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, false, true, 1>), g, dim3(512), 0, st, a, tiles);
        else if (a.nruns == 2) hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, false, true, 2>), g, dim3(512), 0, st, a, tiles);
        else if (a.nruns == 3) hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, false, true, 3>), g, dim3(512), 0, st, a, tiles);
        else hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, false, true, 0>), g, dim3(512), 0, st, a, tiles);
    } else if (wg == 8 && ITER == 8 && NT) { // general form, 512-thread workgroups
        if (wide) hipLaunchKernelGGL((k2_classrun_scan<8, true, true, false, -1, 8>), g, dim3(512), 0, st, a, tiles);
        else hipLaunchKernelGGL((k2_classrun_scan<8, true, false, false, -1, 8>), g, dim3(512), 0, st, a, tiles);
    } else {
        if (wide) hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, true, false>), g, dim3(kWG), 0, st, a, tiles);
        else hipLaunchKernelGGL((k2_classrun_scan<ITER, NT, false, false>), g, dim3(kWG), 0, st, a, tiles);
    }
}

template <int ITER>
static hipError_t launch_iter(int tier, bool nt, bool wide, int wg, const ScanArgs &a, uint32_t grid, hipStream_t st)
{
    dim3 g(grid);
    const TileDesc *tiles = a.tiles;
    if (tier == GSCAN_TIER_BUCKET) {
        if (a.vm_filter) { // candidates confirmed on the device: one form, whatever the variant
            hipLaunchKernelGGL((k3_bucket_scan<8, false, 4, 8, true>), g, dim3(kK3WG), 0, st, a, tiles);
        } else if (wg == 12 && ITER == 8 && nt) { // 768-thread workgroups, two per CU
            if (a.k3_depth == 3) hipLaunchKernelGGL((k3_bucket_scan<ITER == 8 ? 8 : 12, true, 3, 12>), g, dim3(768), 0, st, a, tiles);
            else hipLaunchKernelGGL((k3_bucket_scan<ITER == 8 ? 8 : 12, true, 4, 12>), g, dim3(768), 0, st, a, tiles);
        } else if (wg == -1 && ITER == 12 && nt) { // the prefetching form
            if (a.k3_depth == 3) hipLaunchKernelGGL((k3_bucket_scan<12, true, 3, 8, false, true>), g, dim3(kK3WG), 0, st, a, tiles);
            else hipLaunchKernelGGL((k3_bucket_scan<12, true, 4, 8, false, true>), g, dim3(kK3WG), 0, st, a, tiles);
        } else if (a.k3_depth == 3 && ITER == 12 && nt) hipLaunchKernelGGL((k3_bucket_scan<ITER, true, ITER == 12 ? 3 : 4>), g, dim3(kK3WG), 0, st, a, tiles);
        else if (nt) hipLaunchKernelGGL((k3_bucket_scan<ITER, true>), g, dim3(kK3WG), 0, st, a, tiles);
        else hipLaunchKernelGGL((k3_bucket_scan<ITER, false>), g, dim3(kK3WG), 0, st, a, tiles);
    } else if (tier == GSCAN_TIER_LITERAL) {
        if (nt) hipLaunchKernelGGL((k1_anchor_scan<ITER, true>), g, dim3(kWG), 0, st, a, tiles);
        else hipLaunchKernelGGL((k1_anchor_scan<ITER, false>), g, dim3(kWG), 0, st, a, tiles);
    } else {
        if (nt) launch_k2<ITER, true>(wide, k2_pair(a), wg, a, g, st);
        else launch_k2<ITER, false>(wide, k2_pair(a), 0, a, g, st);
    }
    return hipGetLastError();
}

// Fills the pattern-program part of the argument block from a compiled database.
void fill_program(ScanArgs &a, const DevProgram &pg)
{
    a.m = pg.m;
    a.anchor = pg.anchor;
    a.anchor_mask = pg.anchor_mask;
    a.anchor_off = pg.anchor_off;
    a.anchor_len = pg.anchor_len;
    a.n_classes = pg.n_classes;
    a.nruns = pg.nruns;
    a.k3_off = pg.k3_off;
    a.vm_filter = pg.vm_filter;
    a.report_shift = pg.report_shift;
    a.keep_all = pg.resolve ? 0xffffffffu : 0u;
    // the filter IS the pattern when every alternative has its own bucket and lies inside the filtered positions
    a.k3_one_bucket = 1;
    for (int b = 0; b < 256; b++)
        if (pg.k3_table[b] & 0xfefefefeu) a.k3_one_bucket = 0;
    a.k3_exact = pg.n_alts <= (uint32_t)kK3Buckets && pg.k3_off == 0;
    a.k3_exact3 = a.k3_exact;
    for (uint32_t i = 0; i < pg.n_alts; i++) {
        if (pg.alt_len[i] > (uint32_t)kK3Depth) a.k3_exact = 0;
        if (pg.alt_len[i] > 3u) a.k3_exact3 = 0;
    }
    // Three filter positions are enough when a hit of theirs is rare: expected hits per KiB step of a wave, pricing a
    // class by its size over the ~64 byte values text is made of, below 2 %.
    {
        double p3 = 0;
        for (uint32_t i = 0; i < pg.n_alts; i++) {
            double prod = 1;
            for (uint32_t k = 0; k < 3; k++) {
                const uint32_t pos = pg.k3_off + k;
                if (pos >= pg.alt_len[i]) continue;
                int members = 0;
                const uint32_t *bits = pg.cls_bits[pg.alt_window[pg.alt_off[i] + pos]];
                for (int w = 0; w < 8; w++) members += __builtin_popcount(bits[w]);
                prod *= std::min(1.0, members / 64.0);
            }
            p3 += prod;
        }
        a.k3_depth = p3 * 1024.0 < 0.02 || a.k3_exact3 ? 3u : 4u; // (windows of <= 3 bytes: three positions are the whole pattern)
    }
    for (int r = 0; r < kK2MaxRuns; r++) {
        a.run_desc[r] = (uint32_t)pg.run_cls[r] | ((uint32_t)pg.run_len[r] << 8) | ((uint32_t)pg.run_off[r] << 16);
        const uint32_t n = pg.run_len[r];
        uint32_t have = 1, f = pg.run_cls[r] & 1u;
        if (n >= 2) f |= 1u << 1, have = 2;
        if (n >= 4) f |= 2u << 2, have = 4;
        if (n >= 8) f |= 4u << 4, have = 8;
        if (n >= 16) f |= 8u << 7, have = 16;
        f |= ((n > have ? n - have : 0u) & 31u) << 11;
        f |= ((uint32_t)pg.run_off[r] & 63u) << 16;
        a.run_flat[r] = f;
    }
    // the lane-table form's program (k2lane.hip): up to four runs are put in the order of their step counts
    {
        uint32_t order[kK2MaxRuns], steps[kK2MaxRuns], shifts[kK2MaxRuns][5];
        for (int r = 0; r < kK2MaxRuns; r++) {
            order[r] = (uint32_t)r;
            steps[r] = k2_lane_steps(pg.run_len[r] ? pg.run_len[r] : 1u, shifts[r]);
        }
        const uint32_t n = std::min<uint32_t>(pg.nruns, (uint32_t)kK2MaxRuns);
        if (n >= 2 && n <= 4) std::stable_sort(order, order + n, [&](uint32_t x, uint32_t y) { return steps[x] < steps[y]; });
        for (int r = 0; r < kK2MaxRuns; r++) {
            const uint32_t q = order[r];
            uint32_t f = ((uint32_t)pg.run_cls[q] & 3u) | (((uint32_t)pg.run_off[q] & 31u) << 2);
            for (int i = 0; i < 5; i++) f |= (shifts[q][i] & 31u) << (7 + 5 * i);
            a.run_lane[r] = f;
        }
        // what the launcher specialises on: one run -> its steps; 2..4 runs -> the most steps among all but the last, the last one's
        a.lane_steps[0] = a.lane_steps[1] = 0;
        if (n == 1) a.lane_steps[0] = steps[order[0]];
        else if (n >= 2 && n <= 4) {
            for (uint32_t r = 0; r + 1 < n; r++) a.lane_steps[0] = std::max(a.lane_steps[0], steps[order[r]]);
            a.lane_steps[1] = steps[order[n - 1]];
        }
        a.lane_smax = 0;
        for (uint32_t r = 0; r < n; r++) a.lane_smax = std::max(a.lane_smax, steps[r]);
    }
}

// does the pattern need the second pass?  (the same condition the scan kernel reads as !k3_confirm_exact)
bool scan_needs_settle(int tier, const DevProgram &pg)
{
    return tier == GSCAN_TIER_BUCKET && !pg.k3_confirm_exact && !pg.vm_filter && !pg.resolve; // (the VM has already decided every hit / k_resolve will)
}

hipError_t launch_settle(const ScanArgs &a, uint32_t nw, hipStream_t st)
{
    if (a.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k3_settle, dim3((a.n_tiles * nw + 255u) / 256u), dim3(256), 0, st, a, a.tiles, nw);
    return hipGetLastError();
}

hipError_t launch_order(const ScanArgs &a, uint32_t nw, const uint32_t *ext, uint32_t ew, uint32_t *dpos, uint32_t *out, uint32_t *out_ext, hipStream_t st)
{
    const uint32_t n_desc = a.n_tiles * nw;
    if (n_desc == 0) return hipSuccess;
    hipLaunchKernelGGL(k_order_prefix, dim3(1), dim3(1024), 0, st, a.desc, n_desc, dpos, a.counter);
    hipLaunchKernelGGL(k_order_copy, dim3((n_desc + 3u) / 4u), dim3(256), 0, st, a.recs, ext, ew, a.desc, dpos, n_desc, out, out_ext, a.counter);
    return hipGetLastError();
}

hipError_t launch_resolve(const ScanArgs &a, uint32_t nw, uint32_t *ends, hipStream_t st)
{
    if (a.n_tiles == 0) return hipSuccess;
    const uint32_t n_desc = a.n_tiles * nw;
    const uint32_t chunk = std::max(1u, std::min(kResolveChunkMax, n_desc / 8192u));
    hipLaunchKernelGGL(k_resolve, dim3((n_desc + chunk - 1) / chunk), dim3(kResolveWG), 0, st, a, a.tiles, nw, ends, chunk);
    return hipGetLastError();
}

hipError_t launch_ends(const ScanArgs &a, uint32_t nw, uint32_t sub_bytes, uint32_t *ends, hipStream_t st)
{
    (void)sub_bytes;
    if (a.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ends, dim3(a.n_tiles * nw), dim3(64), 0, st, a, a.tiles, nw, ends);
    return hipGetLastError();
}

hipError_t launch_lines(const ScanArgs &a, uint32_t nw, uint32_t sub_bytes, uint32_t *ext, uint8_t *gather, uint32_t gather_cap, hipStream_t st)
{
    if (a.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lines, dim3(a.n_tiles * nw), dim3(64), 0, st, a, a.tiles, nw, sub_bytes, ext, gather, gather_cap);
    return hipGetLastError();
}

// variant: bits 0-1 KiB per wave {0: 16, 1: 8, 2: 12}; bit 2 nontemporal loads; with 8 KiB per wave + nontemporal, bit 3
// (variant 13): the table kernels (K3, K2's pair form) run 768-thread workgroups.  K2's general form (3-4 classes): variant 13 = 512-thread workgroups sharing one 32 KiB table,
// three per CU, as a persistent grid.  Returns the waves per workgroup asked for, 0 = the kernel's own.
static int variant_wg(int tier, int variant, uint32_t n_classes)
{
    const bool pair = tier == GSCAN_TIER_CLASSRUN && n_classes <= 2;
    if (variant == 13 && (pair || tier == GSCAN_TIER_BUCKET)) return 12;
    if (variant == 13 && tier == GSCAN_TIER_CLASSRUN) return 8;
    return 0;
}

hipError_t launch_scan(int tier, int variant, const ScanArgs &a, uint32_t grid, hipStream_t st)
{
    if (k2_lane_form(tier, variant, a.m, a.nruns)) return launch_k2_lane(a, grid, st);
    variant &= ~kVariantLane;
    const bool nt = (variant >> 2) & 1;
    const bool wide = a.m > 17; // K2: look-ahead beyond one neighbouring lane
    switch (variant & 3) {
    case 1: return launch_iter<8>(tier, nt, wide, variant_wg(tier, variant, a.n_classes), a, grid, st);
    case 2: return launch_iter<12>(tier, nt, wide, variant == 14 ? -1 : 0, a, grid, st); // 14: the table kernels prefetch the next tile
    default: return launch_iter<16>(tier, nt, wide, 0, a, grid, st);
    }
}

} // namespace gscan
