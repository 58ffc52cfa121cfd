// scan_args.h -- kernel argument block + launcher shared by kernels.hip and engine.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gscan.h"
#include "pattern.h"

namespace gscan {

#ifndef GSCAN_SHARDS
#define GSCAN_SHARDS 64
#endif
constexpr int kShards = GSCAN_SHARDS; // record-buffer regions, each with its own reservation counter.  Every WAVE reserves its run with one atomic
                            // (descriptor d uses shard d & 63): with 8 counters the dense patterns queued up on them -- same-address
                            // atomics are served one at a time, ~90 ns each (profiles/r02_d_kernel_sweep_per_wave_8_shards.txt)
// The counters sit one per 128-byte line: returning atomics on the SAME cache line are served one after the other by that
// line's L2 channel, whichever dword they name -- 64 adjacent counters were two lines.
constexpr int kCtrStride = 32; // words between two shard counters
constexpr uint32_t kStruck = 0xffffffffu; // a record the second pass (k3_settle) found to be no match: readers skip it

// One scan unit of the launch: a tile of one segment.  16 bytes so a workgroup fetches it
// with a single scalar load.
struct TileDesc {
    uint64_t seg_off;  // segment start, bytes from ScanArgs::base (16-byte aligned)
    uint32_t seg_len;  // segment length
    uint32_t tile_off; // first byte of this tile inside the segment
};

// Everything a kernel needs, passed by value (kernarg segment -> SGPRs; nothing the hot
// loop uses is fetched from global memory).
struct ScanArgs {
    const uint8_t *base;     // arena
    const TileDesc *tiles;   // [n_tiles]; nullptr = one segment {seg0_off, seg0_len} tiled in order
    uint64_t seg0_off;
    uint32_t seg0_len;
    uint32_t n_tiles;
    uint32_t cap_shard;      // record capacity of ONE shard region (regions are back to back)
    uint32_t *recs;          // candidate starts, segment-relative
    unsigned long long *desc; // [n_tiles * waves per workgroup] one per wave sub-tile: count | base<<32 (base = absolute record index)
    uint32_t *counter;       // [k * kCtrStride] records reserved in shard k, [kShards * kCtrStride] overflow flag, [.. + 1] records struck out by k3_settle, [.. + 2] bytes of line text gathered by k_lines, [.. + 3] records in the ordered copy (k_order_prefix)
    const DevProgram *prog;  // cold paths only (K1 verify, K2 table staging)
    // pattern program, hot-loop copy
    uint32_t m;              // window length
    uint32_t anchor, anchor_mask, anchor_off, anchor_len; // K1
    uint32_t n_classes, nruns;                            // K2
    uint32_t k3_off, k3_exact, k3_depth;                  // K3 (k3_depth: 3 or 4 filter positions)
    uint32_t k3_exact3;                                   // K3: ... and within THREE positions (the three-position filter is exact too)
    uint32_t k3_one_bucket;                               // K3: the filter table uses bucket 0 only (its bytes are 0 / 1)
    uint32_t vm_filter;                                   // K3: every filter hit is put to the VM (DevProgram::vm_filter)
    uint32_t report_shift;   // reported offset = device window start + this (1 when the windows carry a leading context position)
    uint32_t keep_all;       // 0xffffffff: list EVERY candidate, not only the start of every group of consecutive ones (DevProgram::resolve:
                             // each record is a match start of its own for k_resolve); 0: group starts (a mask: x & ~(prev & ~keep_all))
    uint32_t run_desc[kK2MaxRuns];                        // K2: cls | len<<8 | off<<16
    // K2, windows of <= 17 bytes: the run's shift program, decoded on the host -- cls @0, then the shift amounts of the
    // doubling steps (0 = step not taken) 1 @1, 2 @2 (2 bits), 4 @4 (3 bits), 8 @7 (4 bits), the remainder @11 (5 bits),
    // the run's window offset @16 (6 bits)
    uint32_t run_flat[kK2MaxRuns];
    // K2, lane-table form (k2lane.hip; windows of <= 17 bytes): cls @0 (2 bits), the run's window offset @2 (5 bits), the
    // shift amounts of its doubling steps @7, 12, 17, 22, 27 (5 bits each; 0 = no step).  lane_steps = the steps the first two
    // runs take (two runs are sorted by them): the launcher picks the kernel instantiation with exactly that many.
    uint32_t run_lane[kK2MaxRuns];
    uint32_t lane_steps[2];
    uint32_t lane_smax;      // the most steps any run takes (programs of three and four runs give every run that many)
};

// variant: bits 0-1 select KiB per wave {0:16, 1:8, 2:12}; bit 2 = nontemporal loads; 13: bigger workgroups for the table kernels (kernels.hip, variant_wg);
// bit 5 (38 = the default): K2 patterns with windows of <= 17 bytes run the lane-table form (k2lane.hip)
uint32_t scan_tile_bytes(int tier, int variant, uint32_t n_classes);
uint32_t scan_tile_bytes_vm();
void scan_geometry(int tier, int variant, const DevProgram &pg, uint32_t *tile_bytes, uint32_t *waves);
constexpr uint32_t kMaxWavesPerTile = 16;  // the largest workgroup any variant launches
constexpr uint32_t kMinSubTileBytes = 8192; // the smallest sub-tile (8 KiB per wave)
uint32_t scan_min_tile_bytes();
uint32_t scan_persistent_blocks(int tier, int variant, const DevProgram &pg);
void fill_program(ScanArgs &a, const DevProgram &pg);
hipError_t launch_scan(int tier, int variant, const ScanArgs &a, uint32_t grid, hipStream_t st);
// k2lane.hip
constexpr int kVariantLane = 32;
bool k2_lane_form(int tier, int variant, uint32_t m, uint32_t nruns);
uint32_t k2_lane_tile_bytes();
uint32_t k2_lane_waves();
uint32_t k2_lane_steps(uint32_t n, uint32_t *shifts);
hipError_t launch_k2_lane(const ScanArgs &a, uint32_t grid, hipStream_t st);
bool scan_needs_settle(int tier, const DevProgram &pg);
hipError_t launch_settle(const ScanArgs &a, uint32_t waves, hipStream_t st);
// line extents + orbit selection + line gather for the line-printing modes: ext[4 * record index] = {m1, lb, le, goff}, the
// printed lines' text [lb, le) copied to gather[goff ..] (kernels.hip, k_lines); counter[kShards * kCtrStride + 2] = bytes gathered
hipError_t launch_lines(const ScanArgs &a, uint32_t waves, uint32_t sub_bytes, uint32_t *ext, uint8_t *gather, uint32_t gather_cap, hipStream_t st);
constexpr uint32_t kLineAskHost = 0xffffffffu;
// the chunk's records (and their ew extra words each) once more, in text order and back to back: out[0 .. total), out_ext[0 ..
// total * ew); counter[kShards * kCtrStride + 3] = total (kernels.hip, k_order_prefix / k_order_copy); dpos: n_tiles * waves words
hipError_t launch_order(const ScanArgs &a, uint32_t waves, const uint32_t *ext, uint32_t ew, uint32_t *dpos, uint32_t *out, uint32_t *out_ext, hipStream_t st);
// the resolve pass (DevProgram::resolve): the pattern's VM program run at every record; records at which no match starts are
// dropped (the descriptors' counts shrink, counter[kShards * kCtrStride + 1] counts them), ends[record index] = the match's end,
// GSCAN_END_ASK or GSCAN_END_CAPTURES for the others (kernels.hip, k_resolve)
hipError_t launch_resolve(const ScanArgs &a, uint32_t waves, uint32_t *ends, hipStream_t st);
// match ends for -O -l: ends[record index] = end of the match that starts at the record (0: ask the host) (kernels.hip, k_ends)
hipError_t launch_ends(const ScanArgs &a, uint32_t waves, uint32_t sub_bytes, uint32_t *ends, hipStream_t st);

} // namespace gscan
