// adder_framer_kernels.h -- shared between the framer kernels and the framer C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adder_framer.hpp"

namespace adder {

constexpr uint32_t kFramerStatusRing = 1u;       // an event reached past the frame ring
constexpr uint32_t kFramerStatusMalformed = 2u;  // coordinates outside the plane / band
constexpr uint32_t kFramerStatusRange = 4u;      // frame index out of range
constexpr uint32_t kFramerRowsMaxFrames = 256;  // frames per adder_framer_tiles_kernel launch

struct FramerArgs {
    FramerPx *px;        // [n_units] {ts u64, last_filled i32, last_intensity u32}: one 16-byte record per unit
    uint8_t *ring;       // [ring_frames][n_units] elements of 1 << k.value_type bytes
    uint32_t *status;
    uint32_t n_units, width, channels, row_begin, rows;
    uint32_t ring_frames;
    int32_t frames_written;
    FastDivU32 by_ring;  // frame -> ring slot without a division
    FramerConsts k;
};

}  // namespace adder

extern "C" {
hipError_t adder_framer_launch_segment(const void *ev, uint64_t e0, uint64_t e1, const adder::FramerArgs *args,
                                       hipStream_t s);
uint32_t adder_framer_num_tiles(uint32_t n_units);
hipError_t adder_framer_launch_tiles(const void *ev, const uint64_t *d_seg_offsets, uint32_t T, uint64_t *tile_off,
                                     uint32_t window_rows, const adder::FramerArgs *args, hipStream_t s);
hipError_t adder_framer_launch_minmax(const adder::FramerPx *px, uint32_t n, int32_t *out, hipStream_t s);
hipError_t adder_framer_launch_pop(const uint8_t *ring, const adder::FramerPx *px, uint32_t n_units, uint32_t ring_frames,
                                   int32_t f0, uint32_t nf, uint32_t masked, uint8_t *out, uint32_t value_type, hipStream_t s);
hipError_t adder_framer_launch_flush(uint8_t *ring, adder::FramerPx *px, uint32_t n_units, uint32_t ring_frames, int32_t f0,
                                     uint32_t value_type, hipStream_t s);
hipError_t adder_framer_launch_init(adder::FramerPx *px, uint32_t n, hipStream_t s);
}
