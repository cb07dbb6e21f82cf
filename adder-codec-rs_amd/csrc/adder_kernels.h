// adder_kernels.h -- launch interface between the C-ABI (adder_hip_api.cpp) and the
// gfx950 kernels (adder_kernels.hip).  Internal; the public boundary is include/adder_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "adder_pixel.hpp"

namespace adder {

constexpr uint32_t kBlockThreads = 256;
#ifndef ADDER_UNITS_PER_LANE
#define ADDER_UNITS_PER_LANE 2
#endif
#ifndef ADDER_EXPAND_SEGS
#define ADDER_EXPAND_SEGS 8
#endif
#ifndef ADDER_FRAME_WAVES_PER_SIMD
#define ADDER_FRAME_WAVES_PER_SIMD 8
#endif
constexpr uint32_t kUnitsPerLane = ADDER_UNITS_PER_LANE;           // pixel-channels per lane (2 or 4)
constexpr uint32_t kWaveUnits = 64 * kUnitsPerLane;                // units per wave segment
constexpr uint32_t kTileUnits = kBlockThreads * kUnitsPerLane;     // units per K1 block
constexpr uint32_t kSlotsPerLane = 3 * kUnitsPerLane;              // <= 3 fast-path events per pixel
constexpr uint32_t kFrameKernelWavesPerSimd = ADDER_FRAME_WAVES_PER_SIMD;  // register budget of K1
constexpr uint32_t kParkPerWave = 64 * kSlotsPerLane;              // parked-event capacity of a segment
constexpr uint32_t kMaxChunk = 16;                                 // frames per scan/expand launch
constexpr uint32_t kMaxFramesPerLaunch = 16;                       // temporal blocking depth of K1

// bits of the device status word
constexpr uint32_t kStatusCapacity = 1u;  // an event did not fit into the output buffer
constexpr uint32_t kStatusWire = 4u;      // wire serialisation met c = None on a multi-channel plane
constexpr uint32_t kStatusDepth = 2u;     // a pixel needed more than max_depth stored levels

struct AdderEventPod {  // same layout as AdderEvent (include/adder_hip.h)
    uint16_t x, y;
    uint8_t c, d;
    uint16_t pad;
    uint32_t t;
};

// Everything the per-frame kernels need.  `f` = frame index inside the batch.
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
    uint64_t *frame_offsets;  // [frame_idx] = first event of the frame, [frame_idx+1] = end
    uint32_t frame_idx;
    // ordered compaction, stage 1 (frame kernel): per wave segment of 256 units
    uint2 *park;          // [num_waves][park_stride] {t, d | unit_in_wave<<8 | final_offset_in_wave<<16}
    uint32_t *wtot;       // [num_waves] events of the segment (low 16) | parked events (high 16)
    // stage 2 (scan kernel): exclusive prefix of the low halves of wtot, frame total
    uint32_t *wpref;      // [num_waves]
    uint32_t *ftot;       // [1] events of the frame
    uint32_t *status;
    uint32_t n_units;
    uint32_t num_waves;
    uint32_t width, channels, rowlen, row_begin;
    uint32_t generic;     // 1: pixels deeper than one fired level are possible (GENERIC kernel variants)
    uint32_t park4;       // 1: the frame kernel parks compact 4-byte records (non-generic DeltaT variants)
    StepConsts sc;
};

// Device-resident description of one batch of frames.  The kernels take (BatchArgs*, f), so
// a captured hipGraph of T frames can be replayed for ANY batch of T frames: the host only
// rewrites this struct (and the running_t table) before launching the graph.
struct BatchArgs {
    FrameArgs base;           // per-frame fields (frame, frame_idx, park, wtot, wpref, ftot, sc.running_t) are derived
    const uint8_t *frames;    // packed [T][n_units]
    const float *running_t;   // [T] PixelArena::running_t before each frame's integrate
    // compaction scratch: a ring of `slots` frames
    uint2 *park_ring;         // [slots][num_waves][park_stride]
    uint32_t park_stride;     // parked-event capacity of one segment (kParkPerWave, or more for generic batches)
    uint32_t *wtot_ring;      // [slots][num_waves]
    uint32_t *wpref_ring;     // [slots][num_waves]
    uint32_t *ftot_ring;      // [slots]
    uint32_t slots;
};

__device__ __forceinline__ FrameArgs frame_args(const BatchArgs *b, uint32_t f) {
    FrameArgs a = b->base;
    const uint32_t slot = f % b->slots;
    a.frame = b->frames + (size_t)f * a.n_units;
    a.frame_idx = f;
    a.sc.running_t = b->running_t[f];
    a.sc.running_t_u32 = f32_as_u32(a.sc.running_t);
    a.park = b->park_ring + (size_t)slot * a.num_waves * b->park_stride;
    a.wtot = b->wtot_ring + (size_t)slot * a.num_waves;
    a.wpref = b->wpref_ring + (size_t)slot * a.num_waves;
    a.ftot = b->ftot_ring + slot;
    return a;
}

}  // namespace adder

extern "C" {
// variant = collapse | abs_t << 1 | generic << 2 (host copy of what BatchArgs holds)
// K1: frames [f, f + nb) in one launch (nb > 1 = temporal blocking); the same grid also
// expands frames [exp_f0, exp_f0 + exp_nf) of the previous, already scanned chunk (exp_nf may be 0)
hipError_t adder_launch_frame(const adder::BatchArgs *b, uint32_t f, uint32_t nb, uint32_t variant,
                              uint32_t num_waves, uint32_t exp_f0, uint32_t exp_nf, hipStream_t stream);
// frames [f0, f0 + nf): per-frame scan, frame_offsets chain, expansion of the parked events
hipError_t adder_launch_divtest(unsigned long long *d_bad, hipStream_t stream);
hipError_t adder_launch_wire(const adder::AdderEventPod *ev, uint64_t n, uint32_t rec, uint8_t *out, uint32_t *status,
                             hipStream_t stream);
hipError_t adder_launch_scan(const adder::BatchArgs *b, uint32_t f0, uint32_t nf, hipStream_t stream);
hipError_t adder_launch_offsets(const adder::BatchArgs *b, uint32_t f0, uint32_t nf, hipStream_t stream);
hipError_t adder_launch_expand(const adder::BatchArgs *b, uint32_t f0, uint32_t nf, uint32_t num_waves,
                               hipStream_t stream);
hipError_t adder_launch_reset_c_thresh(uint32_t *hdr, size_t n, uint32_t baseline, hipStream_t stream);
hipError_t adder_launch_fill_u32(uint32_t *p, size_t n, uint32_t v, hipStream_t stream);
hipError_t adder_launch_chunk_offsets(const adder::AdderEventPod *ev, uint32_t n, uint32_t row_begin,
                                      uint32_t chunk_rows, uint32_t num_chunks, uint32_t *offsets,
                                      hipStream_t stream);
hipError_t adder_launch_synth(uint8_t *dst, int content, uint64_t seed, uint32_t W, uint32_t H, uint32_t C,
                              uint32_t y0, uint32_t rows, uint32_t k0, uint32_t nframes, hipStream_t stream);
}
