/* examples/transcode_raw.c -- the C-ABI from plain C, no Python, no C++:
 * a synthetic 8-bit clip -> ADDER events on the MI355X -> a raw `.adder` file (header + 9-byte
 * events serialised on the device + EOF), then the same events -> reconstructed frames with the
 * GPU framer.  The counterpart of the reference's examples/framed_video_to_adder.rs.
 *
 *   make -C adder-codec-rs_amd && gcc -O2 -Iinclude examples/transcode_raw.c \
 *       -Ladder-codec-rs_amd -ladder_hip -Wl,-rpath,$PWD/adder-codec-rs_amd -o /tmp/transcode_raw
 *   /tmp/transcode_raw out.adder out_frames.gray
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adder_framer.h"
#include "adder_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != ADDER_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, adder_hip_last_error(ctx)); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    const char *out_events = argc > 1 ? argv[1] : "out.adder";
    const char *out_frames = argc > 2 ? argv[2] : "out_frames.gray";
    const uint16_t W = 320, H = 180;
    const uint32_t T = 48, REF = 255, DTM = 255 * 8, FPS = 30;
    AdderHipCtx *ctx = NULL;

    /* a moving gradient with a blinking square */
    uint8_t *frames = (uint8_t *)adder_hip_alloc_pinned((size_t)T * W * H);
    if (!frames) return 1;
    for (uint32_t k = 0; k < T; ++k)
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                uint32_t v = ((x + 2 * k) * 255 / W + y) & 255;
                if (x > 100 && x < 140 && y > 60 && y < 100) v = (k / 6) % 2 ? 250 : 5;
                frames[((size_t)k * H + y) * W + x] = (uint8_t)v;
            }

    AdderHipParams p;
    adder_hip_default_params(&p, W, H, 1);
    p.time_mode = ADDER_TIME_ABSOLUTE_T; /* the reference's defaults: AbsoluteT, Collapse */
    p.multi_mode = ADDER_MULTI_COLLAPSE;
    p.ref_time = REF;
    p.delta_t_max = DTM;
    p.c_thresh_max = 7;        /* crf 3: CRF[3] = (baseline 2, max 7, velocity 7) (rate_controller.rs:12) */
    p.c_increase_velocity = 7;
    p.c_thresh_start = 2;
    p.c_counter_start = 0;
    if (adder_hip_create(&p, &ctx) != ADDER_OK) {
        fprintf(stderr, "adder_hip_create: %s\n", adder_hip_last_error(NULL));
        return 1;
    }

    /* transcode: wire bytes come back ready to be written */
    const size_t cap_bytes = (size_t)W * H * T * 2 * 9;
    uint8_t *wire = (uint8_t *)adder_hip_alloc_pinned(cap_bytes);
    uint64_t *frame_offsets = (uint64_t *)malloc((T + 1) * sizeof(uint64_t));
    size_t n_bytes = 0, n_events = 0;
    CHECK(adder_hip_integrate_batch_raw(ctx, frames, T, 0, 0, (float)REF, wire, cap_bytes, &n_bytes, &n_events,
                                        frame_offsets));

    uint8_t hdr[64], eof[16];
    const size_t nh = adder_raw_header(hdr, 3, W, H, 1, REF * FPS, REF, DTM, 0 /* FramedU8 */, ADDER_TIME_ABSOLUTE_T,
                                       DTM / REF);
    const size_t ne = adder_raw_eof(eof);
    FILE *f = fopen(out_events, "wb");
    if (!f) return 1;
    fwrite(hdr, 1, nh, f);
    fwrite(wire, 1, n_bytes, f);
    fwrite(eof, 1, ne, f);
    fclose(f);
    printf("%zu events (%.3f per pixel per frame) -> %s (%zu bytes)\n", n_events,
           (double)n_events / ((double)W * H * T), out_events, nh + n_bytes + ne);

    /* reconstruct frames from the events (host copy of the 12-byte events this time) */
    CHECK(adder_hip_reset(ctx));
    AdderEvent *events = (AdderEvent *)adder_hip_alloc_pinned(n_events * sizeof(AdderEvent));
    size_t n2 = 0;
    CHECK(adder_hip_integrate_batch(ctx, frames, T, 0, 0, (float)REF, events, n_events, &n2, frame_offsets));
    if (n2 != n_events) {
        fprintf(stderr, "the two passes disagree: %zu vs %zu events\n", n2, n_events);
        return 1;
    }
    AdderFramerParams fp;
    adder_framer_default_params(&fp, W, H, 1);
    fp.codec_version = 3;
    fp.time_mode = ADDER_TIME_ABSOLUTE_T;
    fp.tps = REF * FPS;
    fp.ref_interval = REF;
    fp.delta_t_max = DTM;
    fp.output_fps = (float)FPS;
    AdderFramer *fr = NULL;
    if (adder_framer_create(&fp, &fr) != ADDER_OK) {
        fprintf(stderr, "adder_framer_create: %s\n", adder_framer_last_error(NULL));
        return 1;
    }
    uint8_t *rec = (uint8_t *)malloc((size_t)T * W * H);
    uint32_t popped = 0, total = 0;
    if (adder_framer_ingest(fr, events, frame_offsets, T) != ADDER_OK ||
        adder_framer_pop(fr, rec, T, &popped) != ADDER_OK) {
        fprintf(stderr, "framer: %s\n", adder_framer_last_error(fr));
        return 1;
    }
    total = popped;
    /* end of stream: flush the frames the slowest pixels still hold back */
    for (int ready = 1; ready && total < T;) {
        if (adder_framer_flush(fr, &ready) != ADDER_OK) return 1;
        if (ready) {
            if (adder_framer_write_frame(fr, rec + (size_t)total * W * H) != ADDER_OK) return 1;
            total += 1;
        }
    }
    f = fopen(out_frames, "wb");
    if (!f) return 1;
    fwrite(rec, 1, (size_t)total * W * H, f);
    fclose(f);
    double err = 0;
    for (size_t i = 0; i < (size_t)total * W * H; ++i) err += abs((int)rec[i] - (int)frames[i]);
    printf("%u frames complete + %u flushed -> %s; mean |reconstruction - source| = %.2f (crf 3 is lossy)\n", popped,
           total - popped, out_frames, err / ((double)total * W * H));

    adder_framer_destroy(fr);
    adder_hip_destroy(ctx);
    adder_hip_free_pinned(events);
    adder_hip_free_pinned(wire);
    adder_hip_free_pinned(frames);
    free(rec);
    free(frame_offsets);
    return 0;
}
