// k2lane.hip -- K2, lane-table form (round 3): class runs of <= 4 classes with windows of <= 17 bytes.
//
// What it replaces: pcre_exec's scan for patterns like [A-Za-z_][A-Za-z0-9_]{15,} (/root/reference/src/grab.cc:178;
// BASELINE configs[2]).  Same contract as every scan kernel here (kernels.hip): per wave sub-tile one descriptor
// {count, base} and a run of ascending candidate offsets, at least the start of every group of consecutive candidates.
//
// Why a new form.  The pair-table form (k2_classrun_scan<.., PAIR = true>) looks the text up two bytes at a time in one
// 64 KiB table shared by the workgroup: 8 ds_read_u8 per lane and KiB step whose bank is bits 2-6 of a TEXT byte.  Text
// crowds those into a handful of banks: SQ_LDS_BANK_CONFLICT 254 M of 431 M SQ_LDS_IDX_ACTIVE cycles per 4 GiB, i.e. the
// LDS was busy 82 % of the kernel's time (profiles/r02_final_sq_counters.txt) -- that, not HBM, bounded cfg3 at 0.63.
// A hash of the index cannot fix it (32 random lanes on 32 banks still collide 3-4 deep); only a layout in which the bank
// is the LANE can.  This form:
//   * table = one 16-bit entry per (byte value, position of the byte inside its dword t = 0..3, lane mod 32) at byte
//     address  b << 8 | (t >> 1) << 7 | (lane & 31) << 2 | (t & 1) << 1  -- 64 KiB.  The address is ONE v_perm_b32 of the
//     text dword and a per-lane constant; the bank is the lane: no conflict, ever (ds_read_u16: 2 LDS cycles);
//   * the entry is pre-shifted by t (class c of byte t -> bit 8c + t for two classes, 4c + t for four), so the four
//     entries of a dword are simply OR-ed (v_or3_b32) instead of shifted into place one by one: 16 look-ups are merged
//     into the lane's class masks by 11 operations (two classes) or 16 (four) -- the per-lane table tried in round 2
//     (LT) spent 16 v_lshl_or_b32 on that and lost what the conflicts had cost;
//   * no control flow inside a sub-tile: the run program is a template parameter (number of runs, doubling steps per
//     run), validity (the last window start of the segment) and group-start suppression are applied once per sub-tile in
//     the epilogue, on the transposed masks, 32 positions per operation;
//   * the per-step candidate mask goes straight into the wave's LDS strip (one ds_write_b16, no VALU): the epilogue
//     reads it back transposed -- lane L owns 192 consecutive text positions -- masks, suppresses, counts, reserves its
//     run with one atomic per wave and writes its records (scalar base + one 32-bit offset per lane).
#include "kcommon.h"

namespace gscan {

namespace {

constexpr int kLIter = 12; // KiB per wave sub-tile
#ifndef GSCAN_LANE_WAVES
#define GSCAN_LANE_WAVES 8
#endif
constexpr int kLNW = GSCAN_LANE_WAVES;    // waves per workgroup (64 KiB table + 12 KiB of strips: two workgroups per CU)
                           // (10 -- five waves per SIMD, the kernel fits 96 VGPRs and 2 x 79 KiB of LDS -- measured: identifier scan
                           // 4.9 against 5.15 TB/s, [0-9]{16} the same, [a-z][0-9][A-Z]{3} 5.6 against 5.2, [a-z]{2,5} 2.85 against 3.0;
                           // profiles/r03_aa_lane_ten_waves_per_workgroup.txt)
constexpr int kLanePF = 0; // where the next tile's loads go (k2_lane_scan's PF).  Measured, same box, 16 GiB, identifier scan / [0-9]{16}
                           // (profiles/r03_c_lane_prefetch_and_subtile_sweep.txt): 0 (none) 5.70 / 6.41 TB/s, 1 (before the epilogue)
                           // 5.13 / 6.46, 2 (behind the atomic) 5.45 / 6.23; 16 KiB per wave: 5.17 / 6.29, 5.45 / 6.40, 4.40 / 4.74.
                           // Behind in-flight loads the epilogue's atomic can only be waited for with vmcnt(0) -- the wave then
                           // sits out the whole HBM round trip before it may write a record; without records there is nothing to lose.

// 16-bit table entry of a byte whose k2_table word is v (bit 8c = member of class c), at dword position t
template <int NCLS>
__device__ __forceinline__ uint32_t lane_entry(uint32_t v, uint32_t t)
{
    if (NCLS == 2) return ((v & 1u) << t) | (((v >> 8) & 1u) << (8u + t));
    return ((v & 1u) << t) | (((v >> 8) & 1u) << (4u + t)) | (((v >> 16) & 1u) << (8u + t)) | (((v >> 24) & 1u) << (12u + t));
}

// two classes: e[4k + t] = entry of byte t of dword k: class 0 at bit t, class 1 at bit 8 + t.
// -> class 0's 16 positions in the low half, class 1's in the high half
__device__ __forceinline__ uint32_t lane_merge2(const uint32_t (&e)[16])
{
    const uint32_t a0 = e[0] | e[1] | e[2], a1 = e[4] | e[5] | e[6];
    const uint32_t g = ((a1 << 4) | a0) | ((e[7] << 4) | e[3]); // byte 0: class 0 of positions 0-7, byte 1: class 1
    const uint32_t a2 = e[8] | e[9] | e[10], a3 = e[12] | e[13] | e[14];
    const uint32_t h = ((a3 << 4) | a2) | ((e[15] << 4) | e[11]); // the same for positions 8-15
    return __builtin_amdgcn_perm(h, g, 0x05010400u);             // [g.b0, h.b0, g.b1, h.b1]
}

// four classes: entry = one nibble per class (class c of byte t at bit 4c + t).  u_k = the four nibbles of dword k;
// a 4 x 4 nibble transposition gives each class its 16 positions: p01 = class 0 | class 1 << 16, p23 = class 2 | class 3 << 16
__device__ __forceinline__ void lane_merge4(const uint32_t (&e)[16], uint32_t &p01, uint32_t &p23)
{
    const uint32_t u0 = (e[0] | e[1] | e[2]) | e[3], u1 = (e[4] | e[5] | e[6]) | e[7];
    const uint32_t u2 = (e[8] | e[9] | e[10]) | e[11], u3 = (e[12] | e[13] | e[14]) | e[15];
    const uint32_t w02 = u0 | (u2 << 16), w13 = u1 | (u3 << 16);
    const uint32_t M = 0x0f0f0f0fu;
    const uint32_t t = (w02 & M) | ((w13 << 4) & ~M);  // bytes: [cls0 pos 0-7, cls2 pos 0-7, cls0 pos 8-15, cls2 pos 8-15]
    const uint32_t s = ((w02 >> 4) & M) | (w13 & ~M);  // bytes: [cls1 pos 0-7, cls3 pos 0-7, cls1 pos 8-15, cls3 pos 8-15]
    p01 = __builtin_amdgcn_perm(s, t, 0x06040200u);    // [t.b0, t.b2, s.b0, s.b2]
    p23 = __builtin_amdgcn_perm(s, t, 0x07050301u);    // [t.b1, t.b3, s.b1, s.b3]
}

// One run of the program (ScanArgs::run_lane, wave-uniform): cls @0 (2 bits), window offset @2 (5 bits), the shift
// amounts of up to five doubling steps @7, 12, 17, 22, 27 (5 bits each).  S = steps actually taken (a template
// parameter where the launcher could specialise; 5 with zero shifts for the rest otherwise).
// "class holds at n consecutive positions from p" by doubling: x &= x >> len verifies 2 len, one overlapping step closes
// the remainder.
template <int S>
__device__ __forceinline__ uint32_t lane_run(uint32_t x, uint32_t d)
{
    if (S >= 1) x &= x >> ((d >> 7) & 31u);
    if (S >= 2) x &= x >> ((d >> 12) & 31u);
    if (S >= 3) x &= x >> ((d >> 17) & 31u);
    if (S >= 4) x &= x >> ((d >> 22) & 31u);
    if (S >= 5) x &= x >> ((d >> 27) & 31u);
    return x >> ((d >> 2) & 31u);
}

// A tile's segment as four scalars (the software-pipelined tile loop carries the next tile's from one pass to the next; a
// TileCtx with its opaque buffer descriptor went through scratch memory there).  The 16-byte descriptor comes in with ONE
// scalar load -- the table is written by the host before the launch and never by a kernel, so the scalar cache is safe; left
// to itself the compiler used two dependent vector loads here, each behind an s_waitcnt vmcnt(0), once per tile.
struct TileS {
    uint32_t alo, ahi; // segment base address
    uint32_t len;      // segment length
    int toff;          // first byte of the tile inside the segment
};

__device__ __forceinline__ TileS tile_scalars(const ScanArgs &a, const TileDesc *__restrict__ tiles, uint32_t t, uint32_t tile_bytes)
{
    uint64_t seg_off;
    TileS r;
    if (tiles) {
        u32x4 d; // {seg_off lo, seg_off hi, seg_len, tile_off}
        const TileDesc *p = tiles + t;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(d) : "s"(p));
        seg_off = (uint64_t)d.x | ((uint64_t)d.y << 32);
        r.len = d.z;
        r.toff = (int)d.w;
    } else {
        seg_off = a.seg0_off;
        r.len = a.seg0_len;
        r.toff = (int)(t * tile_bytes);
    }
    const uint64_t addr = (uint64_t)a.base + seg_off;
    r.alo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
    r.ahi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    r.len = __builtin_amdgcn_readfirstlane(r.len);
    r.toff = (int)__builtin_amdgcn_readfirstlane((uint32_t)r.toff);
    return r;
}

// the wave's loads of one sub-tile: the halo first (the 16 bytes behind the sub-tile, ONE byte each to lanes 0-15 -- see
// lane_halo), then ITER 16-byte pieces per lane; valid == false: the same loads, far out of range -- zeros, no memory traffic
template <int ITER>
__device__ __forceinline__ void lane_loads(u32x4 (&buf)[ITER], uint32_t &halo, const TileS &s, int sub_off, uint32_t lane, bool valid)
{
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)s.ahi << 32) | s.alo), 0, (int)((s.len + 15u) & ~15u), 0x00020000);
    halo = __builtin_amdgcn_raw_buffer_load_b8(rsrc, valid && lane < 16 ? sub_off + ITER * 1024 + (int)lane : 0x7ffffff0, 0, 0);
    const int v0 = valid ? sub_off + (int)lane * 16 : 0x7ffffff0;
#pragma unroll
    for (int k = 0; k < ITER; k++) buf[k] = load16<true>(rsrc, valid ? v0 + k * 1024 : 0x7ffffff0);
}
// The class masks of the 16 halo bytes (what lane 0 of a step past the sub-tile would hold).  Lane j < 16 holds byte j and
// its table entry at dword position 0 (class c at bit 8c for two classes, 4c for four): one look-up, and a ballot per class
// is that class's 16 positions -- instead of 16 look-ups and a merge whose other 63 lanes nobody reads.
template <int NCLS>
__device__ __forceinline__ void lane_halo(uint32_t eh, uint32_t &p01, uint32_t &p23)
{
    constexpr uint32_t sh = NCLS == 2 ? 8u : 4u;
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_ballot_w64((eh & 1u) != 0) & 0xffffu;
    const uint32_t c1 = (uint32_t)__builtin_amdgcn_ballot_w64((eh & (1u << sh)) != 0) & 0xffffu;
    p01 = c0 | (c1 << 16);
    p23 = 0u;
    if (NCLS == 4) {
        const uint32_t c2 = (uint32_t)__builtin_amdgcn_ballot_w64((eh & (1u << 8)) != 0) & 0xffffu;
        const uint32_t c3 = (uint32_t)__builtin_amdgcn_ballot_w64((eh & (1u << 12)) != 0) & 0xffffu;
        p23 = c2 | (c3 << 16);
    }
}

// Epilogue of one wave's sub-tile: xp = the wave's strip, xp[k * 64 + lane] = step k's 16-bit candidate mask of `lane`,
// i.e. a bitmap of the sub-tile in text order.  Lane L takes bits [192 L, 192 L + 192) of it.
// The epilogue of a wave's sub-tile comes in two halves.  lane_count(): the strip read back transposed, masked, reduced to
// group starts, counted, scanned.  lane_write(): descriptor and records, at a base handed in.  Between the two sits ONE
// returning atomic for TWO sub-tiles (lane_flush): the wave keeps the first tile's 192 bits per lane in registers while it scans
// its next tile and reserves for both at once -- the atomic's round trip (1-2 us, sat out behind s_waitcnt vmcnt(0): most of
// the difference between the identifier scan and the same kernel without records) is paid once per 24 KiB instead of once
// per 12.  (Tried and measured slower, profiles/r03_p_lane_reserve_ahead_sweep.txt: reserving by a guess AHEAD of a tile's
// loads -- the answer then queues in front of the loads -- and, r03_c: the next tile's loads behind the atomic.)
// What the record path costs, measured by taking it away (identifier scan, 16 GiB, same box, interleaved runs; profiles/
// r03_y_lane_neither_reservation_nor_stores.txt, r03_w_*, r03_z_*): as here 5.2 - 5.7 TB/s; no reservation (fixed slices) but the
// stores 5.3 - 5.8; the reservation but no stores 5.7 - 5.8; NEITHER 6.0 - 6.1 -- which is what the same arithmetic runs at on a
// pattern without a match ([A-Za-z_][0-9]{15,}: 6.2 - 6.3; [0-9]{16}: 6.3).  So records cost the identifier scan ~10 %, the
// returning atomic and the store loops each about half of it, and not additively.  Two cheaper-looking writers were slower:
// records staged in the strip and stored 64 at a time (5.65 - 5.68 against 5.68 - 5.80), and a loop-free one for runs of <= 64
// records (rank -> source lane by a running maximum over marks in the strip, the source's 192 bits by ds_bpermute, one store:
// 65 instructions against ~130, but 5.1 - 5.3 against 5.6) -- with four waves per SIMD it is the length of a wave's dependent
// chain (LDS round trips) that counts, not its instruction count.  For the same reason rolling refill (piece k of the next tile
// requested the moment piece k of this one is looked up: a sub-tile in flight all the time) was SLOWER, 5.1 and 6.1 TB/s
// (r03_x_*): the kernel is not waiting for memory latency that more loads in flight would hide.
template <int NWORD>
struct LaneCounted {
    uint32_t y[NWORD]; // group starts of this lane's 32 * NWORD positions
    uint32_t first;    // records of the lanes in front of this one
    uint32_t wtot;     // records of the wave
    uint32_t d;        // descriptor index
    uint32_t pos;      // reported offset of this lane's first position
};

template <int ITER>
__device__ __forceinline__ void lane_count(const ScanArgs &a, uint32_t d, int sub_off, int hi, uint32_t lane, const uint16_t *xp, LaneCounted<ITER / 2> &o)
{
    constexpr int NWORD = ITER / 2;
    static_assert(ITER % 4 == 0, "the strip is read back in 8-byte pieces");
    // the wave's own LDS writes, then its reads: LDS operations of one wave execute in order; the fences only keep the
    // compiler from moving them across each other -- fences on the LDS address space alone: a plain workgroup fence also
    // waits for every global store in flight (s_waitcnt vmcnt(0)), i.e. for the records the wave has just written
    GS_LDS_FENCE(__ATOMIC_RELEASE);
    __builtin_amdgcn_wave_barrier();
    GS_LDS_FENCE(__ATOMIC_ACQUIRE);
    uint32_t x[NWORD];
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(xp + lane * ITER);
#pragma unroll
    for (int q = 0; q < NWORD / 2; q++) {
        const unsigned long long v = src[q];
        x[2 * q] = (uint32_t)v;
        x[2 * q + 1] = (uint32_t)(v >> 32);
    }
    const int p0 = sub_off + (int)lane * (ITER * 16);
    if (sub_off + ITER * 1024 - 1 > hi) { // the segment's last window start lies inside this sub-tile (wave-uniform)
#pragma unroll
        for (int i = 0; i < NWORD; i++) {
            const int nv = hi - (p0 + 32 * i) + 1; // valid positions of this word
            x[i] &= nv >= 32 ? 0xffffffffu : nv <= 0 ? 0u : ((1u << nv) - 1u);
        }
    }
    // keep the START of every group of consecutive candidates: drop a candidate whose predecessor is one (the
    // predecessor of the sub-tile's first position is unknown: kept -- reporting more of a group is allowed)
    const uint32_t prev = up1(x[NWORD - 1], 0u);
    const uint32_t sup = ~a.keep_all; // (ScanArgs::keep_all: every candidate is a record of its own -- the resolve pass's patterns)
    o.y[0] = x[0] & ~(((x[0] << 1) | (prev >> 31)) & sup);
    uint32_t c = (uint32_t)__popc(o.y[0]);
#pragma unroll
    for (int i = 1; i < NWORD; i++) {
        o.y[i] = x[i] & ~(__builtin_amdgcn_alignbit(x[i], x[i - 1], 31) & sup); // (x[i] << 1) | (x[i - 1] >> 31)
        c += (uint32_t)__popc(o.y[i]);
    }
    const uint32_t inc = wave_scan(c);
    o.wtot = __builtin_amdgcn_readlane(inc, 63);
    o.first = inc - c;
    o.d = d;
    o.pos = (uint32_t)p0 + a.report_shift;
}

// descriptor + records of one counted sub-tile; `base` = absolute record index of its run (wave-uniform), over = it does not fit
template <int ITER>
__device__ __forceinline__ void lane_write(const ScanArgs &a, uint32_t lane, const LaneCounted<ITER / 2> &o, uint32_t base, bool over)
{
    constexpr int NWORD = ITER / 2;
    if (o.wtot == 0) {
        if (lane == 0) a.desc[o.d] = 0ull;
        return;
    }
    if (lane == 0) a.desc[o.d] = (unsigned long long)o.wtot | ((unsigned long long)base << 32);
    if (over) return; // the host re-runs with a bigger buffer
    // the wave's run of records: a scalar base + a 32-bit byte offset per lane
    char *out = reinterpret_cast<char *>(a.recs + base);
    uint32_t boff = o.first * 4u;
#pragma unroll
    for (int i = 0; i < NWORD; i++) {
        uint32_t bits = o.y[i];
        while (bits) {
            const uint32_t j = (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            *reinterpret_cast<uint32_t *>(out + boff) = o.pos + 32u * (uint32_t)i + j;
            boff += 4u;
        }
    }
}

constexpr int kLaneBatchMax = 2; // sub-tiles a wave holds counted but unwritten before it reserves for all of them (10 VGPRs each; 1..3)

// one reservation for the sub-tiles in p[0 .. n): their runs lie back to back in the shard of the first one's descriptor
template <int ITER>
__device__ __forceinline__ void lane_flush(const ScanArgs &a, uint32_t lane, const LaneCounted<ITER / 2> (&p)[kLaneBatchMax], int n)
{
    const uint32_t total = p[0].wtot + (kLaneBatchMax > 1 && n > 1 ? p[kLaneBatchMax > 1 ? 1 : 0].wtot : 0u) + (kLaneBatchMax > 2 && n > 2 ? p[kLaneBatchMax > 2 ? 2 : 0].wtot : 0u);
    uint32_t base = 0;
    bool over = false;
    if (total) {
        const uint32_t shard = p[0].d & (kShards - 1);
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(a.counter + shard * kCtrStride, total); // index inside the shard's region
        const uint32_t at = __builtin_amdgcn_readfirstlane(b);
        over = (unsigned long long)at + total > (unsigned long long)a.cap_shard;
        if (over && lane == 0) atomicOr(a.counter + kShards * kCtrStride, 1u);
        base = shard * a.cap_shard + at;
    }
    lane_write<ITER>(a, lane, p[0], base, over);
    if (kLaneBatchMax > 1 && n > 1) lane_write<ITER>(a, lane, p[kLaneBatchMax > 1 ? 1 : 0], base + p[0].wtot, over);
    if (kLaneBatchMax > 2 && n > 2) lane_write<ITER>(a, lane, p[kLaneBatchMax > 2 ? 2 : 0], base + p[0].wtot + p[kLaneBatchMax > 1 ? 1 : 0].wtot, over);
}

// (Round 4 built the same in two halves with a whole tile between them -- the returning atomic issued, the next sub-tile
// requested and scanned, only then the answer taken and the records written; the disassembly showed no s_waitcnt vmcnt(0)
// left between scan and stores -- and measured NOTHING on the identifier scan, -5 % on three classes and 1.68 against 3.0
// TB/s on dense output (one reservation per sub-tile instead of per two): the atomic's latency is not what the kernel waits
// for, profiles/r04_e_lane_deferred_reservation_vs_batched.txt.  The code is gone; so is a scheduling barrier that kept all
// sixteen look-ups of a step above the step's arithmetic: 5.36 against 5.41 TB/s, r04_o_*.)

// NCLS: 2 or 4 (table entry layout).  NR: runs of the program, sorted by their doubling steps -- 1 or 2 (two classes):
// exactly that many, S0 / S1 steps; 3 or 4: exactly that many, the last one S1 steps, the others S0 (the most any of them
// needs; zero shifts where one needs fewer) -- everything about them wave-uniform and decoded before the tile loop; 0: any
// number, five steps each, one v_readlane per run and step.
// The tile loop is software-pipelined by one tile: the NEXT tile's text is requested between this tile's last step and
// its epilogue, so the loads are in flight while the wave counts, reserves (one returning atomic) and writes its records
// -- the epilogue's latency and the next tile's HBM latency overlap instead of adding up.  The loads are issued on every
// path (far out of range -- zeros, no traffic -- when there is no next tile or none of it is this wave's).
// PF: where the next tile's loads are issued -- 0: at the top of its own pass (no prefetch), 1: between this tile's last step
// and its epilogue, 2: inside the epilogue, right behind the reserving atomic.
template <int NCLS, int NR, int S0, int S1, int PF = 0, int ITER = kLIter>
__global__ __launch_bounds__(kLNW * 64, kLNW / 2) void k2_lane_scan(ScanArgs a, const TileDesc *__restrict__ tiles)
{
    constexpr uint32_t kTile = kLNW * ITER * 1024;
    static_assert(NCLS == 2 || NCLS == 4, "two entry layouts");
    static_assert(NR == 0 || NR >= 3 || NCLS == 2, "one- and two-run programs have at most two classes");
    __shared__ uint32_t tbl[65536 / 4];
    __shared__ __attribute__((aligned(8))) uint16_t s_xp[kLNW * ITER * 64];
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave); // (scalar: the descriptor index, the shard and the record base follow)
    const uint8_t *tbl8 = reinterpret_cast<const uint8_t *>(tbl);
    const int m = (int)a.m;

    { // stage the table: dword q sits at byte address 4 q = b << 8 | (t >> 1) << 7 | l << 2 and holds t = 2 (t >> 1) (low half) and t + 1
        const uint32_t *k2 = a.prog->k2_table;
        for (uint32_t q = threadIdx.x; q < 65536u / 4u; q += kLNW * 64) {
            const uint32_t v = k2[q >> 6], th = (q >> 5) & 1u;
            tbl[q] = lane_entry<NCLS>(v, 2u * th) | (lane_entry<NCLS>(v, 2u * th + 1u) << 16);
        }
    }
    // the low address byte of this lane's entries for t = 0..3, one per byte of lc
    const uint32_t l4 = (lane & 31u) << 2;
    const uint32_t lc = l4 | ((l4 | 2u) << 8) | ((l4 | 128u) << 16) | ((l4 | 130u) << 24);
    // the run program: lane r of one VGPR holds descriptor r
    const uint32_t vrl = a.run_lane[lane & (kK2MaxRuns - 1)];
    uint32_t rd[4], sel[4]; // the first four descriptors and the byte selectors that pick their class's half of the masks
#pragma unroll
    for (int r = 0; r < 4; r++) {
        rd[r] = __builtin_amdgcn_readlane(vrl, r);
        sel[r] = (rd[r] & 1u) ? 0x07060302u : 0x05040100u;
    }
    const uint32_t rd0 = rd[0], rd1 = rd[1], sel0 = sel[0], sel1 = sel[1];
    const uint32_t nruns = a.nruns;
    __syncthreads();

    // entry of byte t_ of dword d_: address = [lc.byte t, d.byte t, 0, 0]
#define GL_LUT(d_, t_) ((uint32_t)*reinterpret_cast<const uint16_t *>(tbl8 + __builtin_amdgcn_perm((d_), lc, 0x0c0c0000u | ((4u + (t_)) << 8) | (t_))))
#define GL_LOOKUPS(v_)                                                                                                  \
    do {                                                                                                                \
        e[0] = GL_LUT((v_).x, 0u), e[1] = GL_LUT((v_).x, 1u), e[2] = GL_LUT((v_).x, 2u), e[3] = GL_LUT((v_).x, 3u);     \
        e[4] = GL_LUT((v_).y, 0u), e[5] = GL_LUT((v_).y, 1u), e[6] = GL_LUT((v_).y, 2u), e[7] = GL_LUT((v_).y, 3u);     \
        e[8] = GL_LUT((v_).z, 0u), e[9] = GL_LUT((v_).z, 1u), e[10] = GL_LUT((v_).z, 2u), e[11] = GL_LUT((v_).z, 3u);   \
        e[12] = GL_LUT((v_).w, 0u), e[13] = GL_LUT((v_).w, 1u), e[14] = GL_LUT((v_).w, 2u), e[15] = GL_LUT((v_).w, 3u); \
    } while (0)
#define GL_MERGE(p01_, p23_)                                    \
    do {                                                        \
        if (NCLS == 2) p01_ = lane_merge2(e), p23_ = 0u;        \
        else lane_merge4(e, p01_, p23_);                        \
    } while (0)

    uint32_t t = blockIdx.x;
    if (t >= a.n_tiles) return;
    TileS c = tile_scalars(a, tiles, t, kTile);
    int sub_off = c.toff + (int)(wave * ITER * 1024);
    bool have = c.len >= a.m && sub_off < (int)c.len; // some of this tile is this wave's (wave-uniform)
    u32x4 buf[ITER];
    uint32_t halo = 0;
    if (PF) lane_loads<ITER>(buf, halo, c, sub_off, lane, have);
    uint16_t *xp = s_xp + wave * (ITER * 64);
    LaneCounted<ITER / 2> held[kLaneBatchMax]; // counted sub-tiles whose records are not written yet
    int n_held = 0;
    for (;;) {
        const uint32_t tn = t + gridDim.x;
        const bool next = tn < a.n_tiles;
        TileS cn = c;
        int sub_off_n = 0;
        bool have_n = false;
        if (next) {
            cn = tile_scalars(a, tiles, tn, kTile);
            sub_off_n = cn.toff + (int)(wave * ITER * 1024);
            have_n = cn.len >= a.m && sub_off_n < (int)cn.len;
        }
        const uint32_t d = t * kLNW + wave;
        if (!PF && have) lane_loads<ITER>(buf, halo, c, sub_off, lane, true);
        if (have) {
            uint32_t e[16];
            uint32_t pa, qa, pb, qb; // class masks of step k (pa, qa) and of step k + 1 (pb, qb)
            uint32_t hp, hq; // the halo's masks (wave-uniform)
            lane_halo<NCLS>(GL_LUT(halo, 0u), hp, hq);
            GL_LOOKUPS(buf[0]);
            GL_MERGE(pa, qa);
            GL_LOOKUPS(buf[1]);
#pragma unroll
            for (int k = 0; k < ITER; k++) {
                if (k + 1 < ITER) GL_MERGE(pb, qb);       // step k + 1
                else pb = hp, qb = hq;                    // ... of the last step: the halo (only lane 0 of it is looked at)
                if (k + 2 < ITER) GL_LOOKUPS(buf[k + 2]); // in flight while step k is computed
                // the next lane's masks (lane 63: lane 0 of the next step)
                const uint32_t a01 = down1(pa, __builtin_amdgcn_readfirstlane(pb));
                uint32_t cand;
                if (NR == 1) {
                    cand = lane_run<S0>(__builtin_amdgcn_perm(a01, pa, sel0), rd0);
                } else if (NR == 2) {
                    cand = lane_run<S0>(__builtin_amdgcn_perm(a01, pa, sel0), rd0) & lane_run<S1>(__builtin_amdgcn_perm(a01, pa, sel1), rd1);
                } else if (NR > 2) {
                    const uint32_t a23 = NCLS == 4 ? down1(qa, __builtin_amdgcn_readfirstlane(qb)) : 0u;
                    cand = 0xffffu;
#pragma unroll
                    for (int r = 0; r < NR; r++) { // (sorted by their step counts: the last run takes S1 steps, the others at most S0)
                        const bool up = NCLS == 4 && (rd[r] & 2u);
                        const uint32_t x = __builtin_amdgcn_perm(up ? a23 : a01, up ? qa : pa, sel[r]);
                        cand &= r == NR - 1 ? lane_run<S1>(x, rd[r]) : lane_run<S0>(x, rd[r]);
                    }
                } else {
                    const uint32_t a23 = NCLS == 4 ? down1(qa, __builtin_amdgcn_readfirstlane(qb)) : 0u;
                    cand = 0xffffu;
                    for (uint32_t r = 0; r < nruns; r++) {
                        const uint32_t dd = __builtin_amdgcn_readlane(vrl, r);
                        const bool up = NCLS == 4 && (dd & 2u);
                        const uint32_t own = up ? qa : pa, nb = up ? a23 : a01;
                        cand &= lane_run<5>(__builtin_amdgcn_perm(nb, own, (dd & 1u) ? 0x07060302u : 0x05040100u), dd);
                    }
                }
                xp[k * 64 + lane] = (uint16_t)cand; // bit j: a window of the pattern starts at position 16 lane + j of this step
                pa = pb;
                qa = qb;
            }
        }
        // the next tile's text, requested before (PF 1) or inside (PF 2) this tile's epilogue
        if (PF == 1) lane_loads<ITER>(buf, halo, cn, sub_off_n, lane, have_n);
        if (have) {
            if (n_held == 0 || kLaneBatchMax == 1) lane_count<ITER>(a, d, sub_off, (int)c.len - m, lane, xp, held[0]);
            else if (n_held == 1 || kLaneBatchMax == 2) lane_count<ITER>(a, d, sub_off, (int)c.len - m, lane, xp, held[kLaneBatchMax > 1 ? 1 : 0]);
            else lane_count<ITER>(a, d, sub_off, (int)c.len - m, lane, xp, held[kLaneBatchMax > 2 ? 2 : 0]);
            n_held++;
        } else if (lane == 0) {
            a.desc[d] = 0ull; // nothing of this tile is this wave's
        }
        if (PF == 2) lane_loads<ITER>(buf, halo, cn, sub_off_n, lane, have_n);
        if (n_held >= kLaneBatchMax || (!next && n_held)) {
            lane_flush<ITER>(a, lane, held, n_held);
            n_held = 0;
        }
        if (!next) break;
        t = tn;
        c = cn;
        sub_off = sub_off_n;
        have = have_n;
    }
#undef GL_LUT
#undef GL_LOOKUPS
#undef GL_MERGE
}

template <int NCLS, int NR, int S0, int S1>
void launch_one(const ScanArgs &a, dim3 g, hipStream_t st)
{
    hipLaunchKernelGGL((k2_lane_scan<NCLS, NR, S0, S1, kLanePF>), g, dim3(kLNW * 64), 0, st, a, a.tiles);
}

} // namespace

uint32_t k2_lane_tile_bytes() { return (uint32_t)(kLNW * kLIter * 1024); }
uint32_t k2_lane_waves() { return (uint32_t)kLNW; }

// Doubling steps a run of n positions takes: 1 -> 2 -> 4 ... while it fits, one overlapping step for the rest.
uint32_t k2_lane_steps(uint32_t n, uint32_t *shifts /* [5] */)
{
    uint32_t have = 1, s = 0;
    for (int i = 0; i < 5; i++) shifts[i] = 0;
    while (2 * have <= n && s < 5) shifts[s++] = have, have *= 2;
    if (have < n && s < 5) shifts[s++] = n - have, have = n;
    return have == n ? s : 99u; // (99: does not fit five steps -- windows of <= 17 bytes always do)
}

// The launcher: picks the instantiation for the program fill_program() put into a.run_lane / a.lane_steps.
hipError_t launch_k2_lane(const ScanArgs &a, uint32_t grid, hipStream_t st)
{
    const dim3 g(grid);
    const uint32_t s0 = a.lane_steps[0], s1 = a.lane_steps[1];
    const bool four = a.n_classes > 2;
#define GL_CASE(n_, r_, a_, b_) case (a_) * 8 + (b_): launch_one<n_, r_, a_, b_>(a, g, st); break;
#define GL_SORTED(n_, r_)                                                                                                      \
    switch (s0 * 8 + s1) { /* fill_program() sorts the runs by their step counts: s0 <= s1 */                                  \
        GL_CASE(n_, r_, 0, 0) GL_CASE(n_, r_, 0, 1) GL_CASE(n_, r_, 0, 2) GL_CASE(n_, r_, 0, 3) GL_CASE(n_, r_, 0, 4) GL_CASE(n_, r_, 0, 5) \
        GL_CASE(n_, r_, 1, 1) GL_CASE(n_, r_, 1, 2) GL_CASE(n_, r_, 1, 3) GL_CASE(n_, r_, 1, 4) GL_CASE(n_, r_, 1, 5)          \
        GL_CASE(n_, r_, 2, 2) GL_CASE(n_, r_, 2, 3) GL_CASE(n_, r_, 2, 4) GL_CASE(n_, r_, 2, 5)                                \
        GL_CASE(n_, r_, 3, 3) GL_CASE(n_, r_, 3, 4) GL_CASE(n_, r_, 3, 5)                                                      \
        GL_CASE(n_, r_, 4, 4) GL_CASE(n_, r_, 4, 5)                                                                            \
    default: launch_one<n_, r_, 5, 5>(a, g, st); break;                                                                        \
    }
    if (a.nruns == 3) {
        if (four) { GL_SORTED(4, 3) } else { GL_SORTED(2, 3) }
    } else if (a.nruns == 4) {
        if (four) { GL_SORTED(4, 4) } else { GL_SORTED(2, 4) }
    } else if (four) {
        launch_one<4, 0, 5, 5>(a, g, st);
    } else if (a.nruns == 1) {
        switch (s0) {
        case 0: launch_one<2, 1, 0, 0>(a, g, st); break;
        case 1: launch_one<2, 1, 1, 0>(a, g, st); break;
        case 2: launch_one<2, 1, 2, 0>(a, g, st); break;
        case 3: launch_one<2, 1, 3, 0>(a, g, st); break;
        case 4: launch_one<2, 1, 4, 0>(a, g, st); break;
        default: launch_one<2, 1, 5, 0>(a, g, st); break;
        }
    } else if (a.nruns == 2) {
        GL_SORTED(2, 2)
    } else {
        launch_one<2, 0, 5, 5>(a, g, st);
    }
#undef GL_SORTED
#undef GL_CASE
    return hipGetLastError();
}

} // namespace gscan
