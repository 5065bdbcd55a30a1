/* hrfuser_hip_debug.h - measurement and tuning entry points of libhrfuser_hip.so.
 *
 * NOT part of the drop-in boundary (include/hrfuser_hip.h): nothing on the product path (hrfuser_amd.backbone / runtime /
 * trainer in normal operation) calls these; bench.py's per-stage report, tools/ (A/B knobs, lane stamps, critical-lane probe)
 * and the tests do.  Same conventions: plain C, integer status codes, a hipStream_t as void*.
 */
#ifndef HRFUSER_HIP_DEBUG_H
#define HRFUSER_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs for micro-benchmarks and same-box A/B runs.  key 0 = pixel-split cap of hrf_conv_bwd_weight (0 = default),
 * 1 = its atomics replaced by plain stores (wrong results; 2 = the ci*9+tap variant instead of the tap-blocked one), 2 = the
 * generic weight-gradient kernel, 3 = split cap of the pixel-major kernel, 4 / 6 = the generic implicit-GEMM engine instead of
 * the row-GEMM / 3x3 halo engines, 5 = minimum width of the halo engine's data gradient, 7 = time the grouped weight-gradient
 * launches (hrf_wgrad_group_report), 8 = the LDS-tiled weight gradient (1 = off, 2 = for every stride-1 1x1 problem), 9 = 1: no split over K in the generic forward convolution; 16..19 pointwise.hip, 24..27 conv3w_engine.hip, 28..31 lin2_engine.hip. */
int hrf_debug_knob(int key, int value);

/* hrf_debug_knob(7, 1): HIP-event durations of the grouped weight-gradient launches of eager steps, aggregated per kernel
 * variant into rows of 12 doubles (key, launches, problems, total us, algorithmic bytes, flops, heaviest problem's Cin, Cout,
 * H, W, stride, KH); returns the number of rows written and clears the log. */
long hrf_wgrad_group_report(double* out, long cap_rows);

/* When `stream` reaches this point one thread stores the GPU's constant-rate 100 MHz timestamp counter (wall_clock64) to
 * *dst.  Unlike HIP events this can be timed INSIDE a replayed hipGraph: bench.py brackets the stages of the captured training
 * step with it (stems, transitions, fusion_a/b/c, stage2-4 with the modality stages beside them;
 * hrfuser_hrformer_based.py:535-607) for the per-stage roofline report. */
int hrf_stamp(long long* dst, void* stream);

/* Critical-lane probe: one idle workgroup that lasts `ticks` of the same clock.  Padding ONE lane of a multi-lane schedule
 * with it and watching the step time tells whether that lane is on the critical path (HRF_DEBUG_PAD in
 * hrfuser_amd/backbone.py; tools/race_check.py perturbs lane timing with it). */
int hrf_debug_spin(long ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
