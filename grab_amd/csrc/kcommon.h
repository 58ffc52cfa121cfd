// kcommon.h -- device helpers shared by the scan kernels (kernels.hip, k2lane.hip): the text loads through a buffer
// descriptor, wave-level DPP primitives, tile bookkeeping.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy, nothing is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_args.h"

namespace gscan {

namespace {

// A fence on the LDS address space alone (the wave's own strip: written, then read back transposed).  The three-argument
// form -- order, scope, address space -- exists from clang 19 (ROCm 6.3) on; the plain workgroup fence of older toolchains
// does the job too, at a price: it is an s_waitcnt vmcnt(0) as well, i.e. it waits for the record stores the wave issued a
// moment ago (-4 % on the identifier scan).  The Makefile's toolchain is ROCm 7.2 (clang 22).
#if defined(__clang_major__) && __clang_major__ >= 19
#define GS_LDS_FENCE(order_) __builtin_amdgcn_fence((order_), "workgroup", "local")
#else
#define GS_LDS_FENCE(order_) __builtin_amdgcn_fence((order_), "workgroup")
#endif

constexpr int kWave = 64;
constexpr int kWG = 256;
constexpr int kWaves = kWG / kWave;
constexpr int kK3WG = 512; // K3 and the two-class form of K2: 8 waves share one 64 KiB table

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: the hardware bounds check (offset >=
// num_records -> zeros) replaces per-load exec-mask branches, so the ITER+1 loads of a
// sub-tile issue back to back and the compiler can wait on them one by one (vmcnt(N)).
// NT sets the nontemporal bit for the read-once text stream.
template <bool NT>
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t rsrc, int voff)
{
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, NT ? 2 : 0));
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// x of lane+1; lane 63 gets `fill` (a wave-uniform value).  One v_mov_b32_dpp
// wave_shl:1 (gfx9-generation wavefront shift): lane 63 has no source lane and, with
// bound_ctrl off, keeps the `old` operand.
__device__ __forceinline__ uint32_t down1(uint32_t x, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x130, 0xf, 0xf, false);
}

// Inclusive prefix sum over the 64 lanes, all in the VALU: four row_shr DPP adds scan each row
// of 16, row_bcast:15 / row_bcast:31 carry the row totals across (the sequence the gfx9 backend
// itself uses for wave scans).  No LDS round trips: a ds_bpermute chain here costs ~6 x 100+
// cycles of dependent latency per step that has candidates.
#define GS_DPP_ADD(v_, ctrl_, rows_) v_ += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v_, ctrl_, rows_, 0xf, false)
// x of lane-1; lane 0 gets `fill` (wave_shr:1).
__device__ __forceinline__ uint32_t up1(uint32_t x, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
    GS_DPP_ADD(v, 0x111, 0xf); // row_shr:1
    GS_DPP_ADD(v, 0x112, 0xf); // row_shr:2
    GS_DPP_ADD(v, 0x114, 0xf); // row_shr:4
    GS_DPP_ADD(v, 0x118, 0xf); // row_shr:8
    GS_DPP_ADD(v, 0x142, 0xa); // row_bcast:15 -> rows 1,3
    GS_DPP_ADD(v, 0x143, 0xc); // row_bcast:31 -> rows 2,3
    return v;
}
#undef GS_DPP_ADD

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return __builtin_amdgcn_readlane(wave_scan(v), 63); }

// 16-bit mask of the positions pos0..pos0+15 that lie in [lo, hi] (signed).
__device__ __forceinline__ uint32_t valid16(int pos0, int lo, int hi)
{
    int jlo = lo - pos0, jhi = hi - pos0;
    if (jlo < 0) jlo = 0;
    if (jhi > 15) jhi = 15;
    if (jlo > 15 || jhi < 0 || jlo > jhi) return 0u;
    return (0xffffu << jlo) & (0xffffu >> (15 - jhi)) & 0xffffu;
}

// Cold path of K1: does the whole window match at p?  (segment-relative p, in bounds)
__device__ __noinline__ bool verify_window(const uint8_t *seg, const DevProgram *pg, uint32_t p)
{
    const uint32_t m = pg->m;
    if (pg->is_literal) {
        for (uint32_t i = 0; i < m; i++)
            if (seg[p + i] != pg->window[i]) return false;
        return true;
    }
    for (uint32_t i = 0; i < m; i++) {
        uint32_t b = seg[p + i];
        uint32_t c = pg->window[i];
        if (!((pg->cls_bits[c][b >> 5] >> (b & 31)) & 1u)) return false;
    }
    return true;
}

// Cold path of K3: does some alternative of the buckets in `buckets` match at p, inside the segment?
__device__ __forceinline__ bool verify_alts(const uint8_t *seg, uint32_t slen, const DevProgram *pg, uint32_t p, uint32_t buckets)
{
    const uint32_t n = pg->n_alts;
    for (uint32_t i = 0; i < n; i++) {
        if (!((buckets >> pg->alt_bucket[i]) & 1u)) continue;
        const uint32_t m = pg->alt_len[i];
        if (p + m > slen) continue;
        const uint8_t *w = pg->alt_window + pg->alt_off[i];
        uint32_t k = 0;
        for (; k < m; k++) {
            const uint32_t b = seg[p + k];
            if (!((pg->cls_bits[w[k]][b >> 5] >> (b & 31)) & 1u)) break;
        }
        if (k == m) return true;
    }
    return false;
}

struct TileCtx {
    const uint8_t *seg;          // segment base
    __amdgpu_buffer_rsrc_t rsrc; // descriptor over [seg, seg + slen rounded up to 16)
    int slen;                    // segment length
    int tile_off;                // first byte of the tile inside the segment
    bool live;                   // segment long enough to hold a window at all
};

// t is blockIdx-derived, so everything here is wave-uniform; the descriptor is read with
// a scalar load and readfirstlane makes the uniformity provable, which keeps the buffer
// descriptor in SGPRs (otherwise hipcc wraps every buffer load in a waterfall loop).
__device__ __forceinline__ TileCtx tile_ctx(const ScanArgs &a, const TileDesc *__restrict__ tiles, uint32_t t,
                                            uint32_t tile_bytes)
{
    uint64_t seg_off;
    uint32_t len, toff;
    if (tiles) {
        const TileDesc d = tiles[t];
        seg_off = d.seg_off;
        len = d.seg_len;
        toff = d.tile_off;
    } else {
        seg_off = a.seg0_off;
        len = a.seg0_len;
        toff = t * tile_bytes;
    }
    const uint64_t addr = (uint64_t)a.base + seg_off;
    const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
    const uint32_t ahi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    len = __builtin_amdgcn_readfirstlane(len);
    TileCtx c;
    c.seg = (const uint8_t *)(((uint64_t)ahi << 32) | alo);
    c.slen = (int)len;
    c.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)c.seg, 0, (int)((len + 15u) & ~15u), 0x00020000);
    c.tile_off = (int)__builtin_amdgcn_readfirstlane(toff);
    c.live = len >= a.m;
    return c;
}

// Load the wave's sub-tile: ITER full steps + the halo step (only the first
// HALO_LANES lanes carry data there).  16-byte blocks that start at or beyond the
// segment end come back as zeros (descriptor bounds check); the last partial block is
// read whole (segment bases are 16-byte aligned, so this stays inside the allocation)
// and its garbage bytes can only reach windows that valid16() removes.
// one step of a sub-tile: k < ITER a full KiB, k == ITER the halo (only the first HALO_LANES lanes carry data)
// (valid == false: the load is issued all the same, far out of range -- zeros, no memory traffic: the prefetching kernels
// keep the SAME sequence of loads on every path so that the compiler's vmcnt bookkeeping is exact)
template <int ITER, bool NT, int HALO_LANES>
__device__ __forceinline__ u32x4 load_step(const TileCtx &c, int sub_off, uint32_t lane, int k, bool valid = true)
{
    const int v0 = sub_off + (int)lane * 16;
    if (k < ITER) return load16<NT>(c.rsrc, valid ? v0 + k * 1024 : 0x7ffffff0);
    return load16<false>(c.rsrc, valid && lane < (uint32_t)HALO_LANES ? v0 + ITER * 1024 : 0x7ffffff0);
}

template <int ITER, bool NT, int HALO_LANES>
__device__ __forceinline__ void load_subtile(u32x4 (&buf)[ITER + 1], const TileCtx &c, int sub_off, uint32_t lane)
{
    const int v0 = sub_off + (int)lane * 16;
#pragma unroll
    for (int k = 0; k < ITER; k++) buf[k] = load16<NT>(c.rsrc, v0 + k * 1024);
    // lanes >= HALO_LANES point far out of range -> zeros, no branch
    buf[ITER] = load16<false>(c.rsrc, lane < (uint32_t)HALO_LANES ? v0 + ITER * 1024 : 0x7ffffff0);
}

} // namespace

} // namespace gscan
