// scan_args.h -- kernel argument block + launcher shared by kernels.hip and engine.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gscan.h"
#include "pattern.h"

namespace gscan {

struct ScanArgs {
    const uint8_t *base;        // arena
    const gscan_seg *segs;      // [nseg]
    const uint32_t *tile_first; // [nseg+1] first tile of each segment
    const uint32_t *tile_seg;   // [n_tiles] segment of each tile
    uint32_t n_tiles;
    uint32_t cap;               // record capacity
    uint32_t *recs;             // candidate starts, segment-relative
    unsigned long long *desc;   // [n_tiles] count | base<<32
    uint32_t *counter;          // [0] records reserved, [1] overflow flag
    const DevProgram *prog;
};

// variant: bits 0-1 select KiB per wave {0:16, 1:8, 2:4}; bit 2 = nontemporal loads
uint32_t scan_tile_bytes(int variant);
hipError_t launch_scan(int tier, int variant, uint32_t window, const ScanArgs &a, uint32_t grid, hipStream_t st);

} // namespace gscan
