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
    float *tinteg;      // [n_pad] tail integration
    float *tdt;         // [n_pad] tail delta_t
    uint8_t *td;        // [n_pad] tail d
    float *lastf;       // [n_pad] last_fired_t (AbsoluteT)
    float *lv_integ;    // [max_depth][n_pad]
    float *lv_dt;       // [max_depth][n_pad]
    float *lv_bdt;      // [max_depth][n_pad]
    uint16_t *lv_dbd;   // [max_depth][n_pad]  d | best_d << 8
    uint8_t *running;   // optional running_intensities side plane, or nullptr
    size_t plane_stride;  // n_pad
    // this frame
    const uint8_t *frame;  // n_units bytes, packed [rows][width][channels]
    AdderEventPod *out;
    uint64_t out_cap;
    uint64_t *frame_offsets;  // [frame_idx] is read, [frame_idx+1] is written
    uint32_t frame_idx;
    uint64_t *desc_cur;   // [num_tiles] look-back descriptors of this frame (zeroed beforehand)
    uint64_t *desc_next;  // [num_tiles] zeroed by this launch for the next frame
    uint32_t *status;
    uint32_t *census;     // non-null: residency census only
    uint32_t n_units;
    uint32_t num_tiles;
    uint32_t width, channels, rowlen, row_begin;
    uint32_t spin_limit;
    StepConsts sc;
};

}  // namespace adder

extern "C" {
hipError_t adder_launch_frame(const adder::FrameArgs *args, uint32_t grid, hipStream_t stream);
hipError_t adder_frame_kernel_occupancy(int *blocks_per_cu);
hipError_t adder_launch_reset_c_thresh(uint32_t *hdr, size_t n, uint32_t baseline, hipStream_t stream);
hipError_t adder_launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t stream);
hipError_t adder_launch_chunk_offsets(const adder::AdderEventPod *ev, uint32_t n, uint32_t row_begin,
                                      uint32_t chunk_rows, uint32_t num_chunks, uint32_t *offsets,
                                      hipStream_t stream);
hipError_t adder_launch_synth(uint8_t *dst, int content, uint64_t seed, uint32_t W, uint32_t H, uint32_t C,
                              uint32_t y0, uint32_t rows, uint32_t k0, uint32_t nframes, hipStream_t stream);
}
