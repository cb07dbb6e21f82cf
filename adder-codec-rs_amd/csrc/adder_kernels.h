// adder_kernels.h -- launch interface between the C-ABI (adder_hip_api.cpp) and the
// gfx950 kernels (adder_kernels.hip).  Internal; the public boundary is include/adder_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kBlockThreads = 256;
constexpr uint32_t kUnitsPerLane = 4;                              // pixel-channels per lane
constexpr uint32_t kTileUnits = kBlockThreads * kUnitsPerLane;     // 1024 units per tile
constexpr uint32_t kGroupTiles = 32;                               // tiles per prefix group
constexpr uint32_t kSlotsPerLane = 12;                             // 4 px x 3 fast-path events

// bits of the device status word
constexpr uint32_t kStatusCapacity = 1u;  // an event did not fit into the output buffer
constexpr uint32_t kStatusDepth = 2u;     // a pixel needed more than max_depth stored levels
constexpr uint32_t kStatusTimeout = 4u;   // a bounded wait expired

struct AdderEventPod {  // same layout as AdderEvent (include/adder_hip.h)
    uint16_t x, y;
    uint8_t c, d;
    uint16_t pad;
    uint32_t t;
};

struct FrameArgs {
    // structure-of-arrays pixel state, resident in HBM across frames
    uint32_t *hdr;      // [n_pad]
    float *lastf;       // [n_pad] last_fired_t (AbsoluteT)
    float *lv_integ;    // [max_depth][n_pad]
    float *lv_dt;       // [max_depth][n_pad]
    float *lv_bdt;      // [max_depth][n_pad]
    uint8_t *lv_bd;     // [max_depth][n_pad]  best_d (d = fired_d(best_d))
    uint8_t *running;   // optional running_intensities side plane, or nullptr
    size_t plane_stride;  // n_pad
    // this frame
    const uint8_t *frame;  // n_units bytes, packed [rows][width][channels]
    AdderEventPod *out;
    uint64_t out_cap;
    uint64_t *frame_offsets;  // [frame_idx] is read, [frame_idx+1] is written
    uint32_t frame_idx;
    // ordered compaction: per-tile and per-group event counts of this frame
    uint64_t *agg_cur;    // [num_tiles]  {1<<32 | count}, zeroed beforehand
    uint64_t *agg_next;   // zeroed by this launch for the next frame
    uint64_t *gsum_cur;   // [num_groups] {1<<32 | sum over the group's tiles}
    uint64_t *gsum_next;
    // pixels that need the generic step this frame: {unit, frame-relative output position}
    uint2 *worklist;
    uint32_t *wl_count_cur;
    uint32_t *wl_count_next;
    uint32_t *status;
    uint32_t *census;     // non-null: residency census only
    uint32_t n_units;
    uint32_t num_tiles;
    uint32_t width, channels, rowlen, row_begin;
    uint32_t spin_limit;
    uint32_t ablate;      // experiments only (ADDER_HIP_ABLATE)
    uint32_t generic;     // 1: pixels deeper than one fired level are possible (worklist + generic kernel)
    StepConsts sc;
};

}  // namespace adder

extern "C" {
hipError_t adder_launch_frame(const adder::FrameArgs *args, uint32_t grid, hipStream_t stream);
hipError_t adder_launch_generic(const adder::FrameArgs *args, uint32_t grid, hipStream_t stream);
hipError_t adder_frame_kernel_occupancy(const adder::FrameArgs *args, int *blocks_per_cu);
hipError_t adder_launch_reset_c_thresh(uint32_t *hdr, size_t n, uint32_t baseline, hipStream_t stream);
hipError_t adder_launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t stream);
hipError_t adder_launch_chunk_offsets(const adder::AdderEventPod *ev, uint32_t n, uint32_t row_begin,
                                      uint32_t chunk_rows, uint32_t num_chunks, uint32_t *offsets,
                                      hipStream_t stream);
hipError_t adder_launch_synth(uint8_t *dst, int content, uint64_t seed, uint32_t W, uint32_t H, uint32_t C,
                              uint32_t y0, uint32_t rows, uint32_t k0, uint32_t nframes, hipStream_t stream);
}
