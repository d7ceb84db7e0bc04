/* rmx.h -- C ABI of the MI355X (gfx950) TDoA cross-correlation engine.
 *
 * What this boundary replaces in the reference (physiii/radio-mapper):
 *   The reference has no FFI for this path: the path sits behind plain
 *   Python methods of tdoa_processor.py.  The seam is the body of the pair loop of
 *   TDoACalculator.calculate_tdoa_measurements (tdoa_processor.py:156-193), whose time difference
 *   `time_diff_ns = det2.gps_timestamp_ns - det1.gps_timestamp_ns` (tdoa_processor.py:166) is
 *   replaced by a lag measured from IQ with the cross-correlation primitive the module imports
 *   (`from scipy.signal import correlate`, tdoa_processor.py:20).  rmx_xcorr_batch() is that
 *   primitive, batched over capture windows x buoy pairs:
 *
 *     for every window w, for every pair (i, j), i < j, in the reference's nested-loop order
 *     (tdoa_processor.py:156-157):
 *         c   = correlate(x[w][j], x[w][i], mode='full', method='fft')   (complex64, 2N-1 lags)
 *         m   = |c|                                                     (float32)
 *         k   = argmax m   (ties -> lowest k)        lag_int  = k - (N-1)
 *               (Parity statement: lag_int is BIT-EXACT against scipy's float32 path wherever that path's own two
 *               largest magnitudes differ by more than 1e-5 relative -- every committed fixture, every BASELINE shape --
 *               and the lowest index on exact ties.  Where two candidates are closer than that, two correct float32
 *               FFTs may order them differently; the tests accept only the oracle's own second candidate there.)
 *               (The kernels search the maximum of |c|^2 and take the square root of the three taps only.  sqrt is
 *               monotonic, so the two orders agree except in one class of inputs: two lags whose |c|^2 differ in the
 *               last bit but whose float32 |c| round to the SAME value.  numpy then sees a tie and takes the lower
 *               index; the kernels take the lag with the larger square.  Such pairs are one float32 ulp apart in
 *               magnitude -- below what two float32 FFTs agree on -- and cannot occur when the two values are
 *               bit-identical, which is the case the tie rule is tested on.)
 *         d   = 3-point parabolic vertex offset      lag_frac = d  (0 at the edges / flat top)
 *               (d = (a - c) / (2 (a - 2b + c)) in double precision from the float32 taps a, b, c = m[k-1], m[k], m[k+1].
 *               Against another float32 implementation of the same definition it agrees to 1e-5 * max(|lag|, 1)
 *               wherever that formula is well conditioned -- 8.7 M random pair-windows, N = 16 ... 2^18, worst 7e-6 --
 *               and to the formula's own conditioning where it is not: two pair-windows of those, e.g. a flat peak on a
 *               32-sample window at 0 dB (a - 2b + c = 4e-3 b), where one float32 ulp on each tap moves d by 2.3e-5,
 *               differed by 3.3e-5.  The rule -- within 1e-5, or within four such
 *               one-ulp bounds, nothing else -- is pinned in tests/test_gpu_parity.py::test_flat_peak_rule_on_short_noisy_windows
 *               and used by tests/soak_parity.py.)
 *         peak = m[k]
 *     lag = lag_int + lag_frac samples = delay(buoy j) - delay(buoy i)   (sign of
 *     TDoAMeasurement.time_difference_ns: buoy2 - buoy1, tdoa_processor.py:51).
 *
 *   The ctypes binding a maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, no torch / numpy types.  The caller owns every buffer it
 * passes; the library owns only ctx-internal device scratch.  Functions return 0 (RMX_OK) or a
 * negative code and never abort; rmx_last_error() gives the text.  A ctx is single-owner (not
 * thread safe); use one ctx per device (one process per GPU, or one host thread per ctx).
 */
#ifndef RMX_H
#define RMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmx_ctx rmx_ctx;

enum {
    RMX_OK = 0,
    RMX_E_INVAL = -1,       /* bad argument */
    RMX_E_NODEV = -2,       /* no usable HIP device */
    RMX_E_HIP = -3,         /* a HIP runtime call failed (text in rmx_last_error) */
    RMX_E_NOMEM = -4,       /* device or host allocation failed */
    RMX_E_UNSUPPORTED = -5  /* shape not supported by this build */
};

/* flags of rmx_xcorr_batch() */
enum {
    RMX_IN_DEVICE = 1u,   /* `iq` is a device pointer (already resident in HBM) */
    RMX_OUT_DEVICE = 2u,  /* lag_int / lag_frac / peak are device pointers; the call is then
                             asynchronous on the ctx stream (rmx_synchronize to wait) */
    RMX_IN_U8 = 4u        /* `iq` is raw rtl_sdr uint8 I,Q interleaved [W][B][N][2]; decoded in
                             the first kernel as (float)u8 - 127.5f (buoy_node.py:392-398) */
};

#define RMX_VERSION 1

int rmx_version(void);

/* Number of HIP devices visible, 0 if none / no driver.  Never fails. */
int rmx_device_count(void);

/* Create an engine for windows of `n_samples` complex64 samples from `n_buoys` buoys.
 * n_samples: power of two, 16 .. 4194304.  max_windows: largest n_windows a later call may pass.
 * flags: reserved, pass 0. */
int rmx_create(rmx_ctx** out, int device_id, int n_buoys, int n_samples, int max_windows,
               unsigned flags);
void rmx_destroy(rmx_ctx* ctx);

/* Text of the last error on this ctx (or of the last failed rmx_create when ctx == NULL). */
const char* rmx_last_error(const rmx_ctx* ctx);

/* Use an existing hipStream_t (e.g. torch's current stream) for all work of this ctx. */
int rmx_set_stream(rmx_ctx* ctx, void* hip_stream);

/* Per-ctx options (all optional): "chunk_windows" (windows per launch), "timing" (1: bracket every
 * launch with HIP events, read back with rmx_last_timing), "fused" (default 1; 0 forces the separate
 * forward + pair kernels that custom pair lists use), "resident" and "pairs_per_block" (variants of
 * that unfused pair kernel; without an explicit "pairs_per_block" the library sizes the blocks to the batch),
 * "small4096" (default 1: batches of N = 4096 too small to fill the chip with one workgroup per window -- fewer than
 * about 0.25 ... 0.5 windows per CU, depending on the buoy count -- run through those per-transform kernels, which spread
 * one window's spectra and pairs over the CUs: 13 us instead of 92 us for a single window of 8 buoys; 0: always the
 * fused kernel), "stag" (0..5: which waves of the fused N = 4096 kernel run the two halves
 * between barriers in the opposite order; default 1), "win8" / "pk" (two other builds of that kernel,
 * tools/experiments/: present only in a -DRMX_EXPERIMENTS build, RMX_E_UNSUPPORTED otherwise), "dbg"
 * (ablation masks of the -DRMX_ABLATE timing build; every other build rejects the key with
 * RMX_E_UNSUPPORTED and compiles the masks out of all kernels).
 * Returns RMX_E_INVAL for an unknown key. */
int rmx_set_option(rmx_ctx* ctx, const char* key, long value);

/* Kernel-selection defaults for engines created AFTERWARDS (process-wide; an existing ctx keeps the values it was
 * created with, so its kernels, block sizes and LDS sizes stay consistent).  For tests and A/B measurements: the
 * library never reads the environment.  Keys (radio-mapper_amd/csrc/host_plan.hpp lists ranges and meanings):
 * "stag", "ncus", "chunk_windows", "small4096", "generic4096", "small_maxl", "logl1", "wfused", "wscr", "wscr14",
 * "wscr_per_cu", "rows_anchor", "fused", "fused_def", "gen_chunk", "rows_tpr", "cols_threads", "col_logt", "kwin8k"
 * (N = 8192, batches that fill the chip: 1 = k_win8kl, the default; 0 = g_win_scr14), "kwin16k" (N = 16384: 1 = k16_fwd +
 * k16_pairs from "k16_min_windows" windows on -- default 100 / (buoys + pairs) -- unless "wscr" = 2; 2 = for every batch;
 * 0 = g_win_eo15 / the four-step kernels).
 * Returns RMX_E_INVAL (text through rmx_last_error(NULL)) for an unknown key or a value out of range;
 * value == LONG_MIN removes a key; rmx_clear_default_options() removes all. */
int rmx_set_default_option(const char* key, long value);
void rmx_clear_default_options(void);

/* The hot path.
 *   iq        complex64 interleaved I,Q  [n_windows][n_buoys][n_samples][2] float32, row-major
 *             (or uint8 with RMX_IN_U8); host pointer unless RMX_IN_DEVICE.
 *   pairs     int32 [n_pairs][2] = (i, j) buoy indices, or NULL for all i<j in nested-loop order
 *             (then n_pairs must be n_buoys*(n_buoys-1)/2 or 0).  Host pointer always.
 *   lag_int   int32   [n_windows][n_pairs]
 *   lag_frac  float32 [n_windows][n_pairs]   sub-sample offset in [-0.5, 0.5]
 *   peak      float32 [n_windows][n_pairs]   |c| at the integer peak (scipy scaling)
 * Range: everything is float32 with exact power-of-two scaling inside; at N = 4096 the squared
 * magnitudes stay finite for fully coherent inputs up to |sample| ~ 4e6 (rtl_sdr data is +-127.5).
 */
int rmx_xcorr_batch(rmx_ctx* ctx, const void* iq, int n_windows, const int32_t* pairs, int n_pairs,
                    int32_t* lag_int, float* lag_frac, float* peak, unsigned flags);

/* Cross-ambiguity variant of the hot path (BASELINE configs[4]): for
 * every window and pair (i, j) the later buoy's window is de-rotated by each Doppler hypothesis,
 *     c_d = correlate(x[w][j] * exp(-2*pi*i*doppler_cps[d]*n), x[w][i], 'full', 'fft'),
 * and the peak is searched over (d, lag) in d-major order (ties -> lowest d, then lowest lag index);
 * the lag is interpolated along the lag axis of the winning row exactly as in rmx_xcorr_batch.
 *   doppler_cps  host array [n_dopplers], cycles per sample (f_d / fs)
 *   dop_idx      int32 [n_windows][n_pairs]  index of the winning hypothesis
 * Same buffer/flag conventions as rmx_xcorr_batch.  The reference has no counterpart (its TDoA is a
 * timestamp subtraction, tdoa_processor.py:166); the oracle is oracle/xcorr_ref.py:caf_pair. */
int rmx_caf_batch(rmx_ctx* ctx, const void* iq, int n_windows, const int32_t* pairs, int n_pairs,
                  const double* doppler_cps, int n_dopplers, int32_t* dop_idx, int32_t* lag_int,
                  float* lag_frac, float* peak, unsigned flags);

/* Batched hyperbolic position solve: the consumer of the lags.  One
 * independent 3-unknown least-squares problem per window with the reference's objective
 * (HyperbolicPositioning.triangulate_position, tdoa_processor.py:249-273),
 *     f(p) = sum_q weight[w][q] * ( |p - b_j| - |p - b_i| - d[w][q] )^2 ,  (i, j) = pairs[q],
 *     d = (lag_int + lag_frac) / sample_rate_hz * 299792458  (tdoa_processor.py:141,169-170),
 * started from the centroid of the buoys (tdoa_processor.py:275-280).  The reference minimises f
 * with scipy BFGS one window at a time; this entry uses a fixed Levenberg-Marquardt rule in float64
 * (lambda_0 = 1e-3 on diag(J^T J); accept on decrease: lambda /= 3, floor 1e-12; reject: lambda *= 4;
 * stop at an accepted step < 1e-4 m, lambda > 1e12, or max_iter iterations), restated on the CPU in
 * oracle/solve_ref.py.
 *   buoy_xyz   host double [n_buoys][3], ECEF metres (GeodeticCalculator.lat_lng_to_xyz)
 *   pairs      host int32 [n_pairs][2] or NULL for all i<j in nested-loop order
 *   lag_int / lag_frac   [n_windows][n_pairs], exactly what rmx_xcorr_batch wrote (device pointers
 *              with RMX_IN_DEVICE: the two calls chain on the ctx stream without a host round trip)
 *   weight     float [n_windows][n_pairs] = 1/(confidence+0.1) (tdoa_processor.py:267), or NULL = 1
 *   pos        double [n_windows][3]  ECEF metres        cost    double [n_windows]  f at pos
 *   iters      int32  [n_windows]     iterations used    (device pointers with RMX_OUT_DEVICE)
 * accuracy_meters of the reference = sqrt(cost / n_pairs) (tdoa_processor.py:300). */
int rmx_solve_batch(rmx_ctx* ctx, const double* buoy_xyz, int n_buoys, const int32_t* pairs, int n_pairs,
                    const int32_t* lag_int, const float* lag_frac, const float* weight,
                    double sample_rate_hz, int n_windows, int max_iter, double* pos, double* cost,
                    int32_t* iters, unsigned flags);

/* Spectral detection: the reference's per-capture detector
 * (buoy_node.py:401-433, iq_stream_client.py:186-217) batched over windows.  Per window of n_samples
 * (power of two, 16..16384; independent of the ctx's n_samples) complex samples:
 *     P[k] = 20 log10(|FFT_N(iq)[k]| + 1e-12)                            (float32, unpadded, unwindowed)
 *     peaks = scipy.signal.find_peaks(P, height=threshold_db, distance=distance)   (buoy_node.py:411-415)
 *     floor = median(P); snr = P[peak] - floor; confidence = min(max(snr/20, 0), 1) (buoy_node.py:425-427)
 *     a peak is reported unless |signed FFT bin| < dc_exclude_bins (the +-10 kHz of buoy_node.py:419,
 *     i.e. 10e3 * n_samples / sample_rate) or confidence < min_confidence (0.3, buoy_node.py:430)
 *   iq          complex64 [n_windows][n_samples] (or uint8 I,Q pairs with RMX_IN_U8); device with RMX_IN_DEVICE
 *   count       int32 [n_windows]   peaks found (may exceed max_peaks; the arrays hold the first max_peaks,
 *               in ascending bin order as find_peaks returns them)
 *   bin / power_db / snr_db / confidence   [n_windows][max_peaks];  noise_floor_db float [n_windows]
 *   (device pointers with RMX_OUT_DEVICE).  Frequency of a bin: fftfreq, f_centre + (bin < N/2 ? bin : bin - N) * fs / N. */
int rmx_detect_batch(rmx_ctx* ctx, const void* iq, int n_windows, int n_samples, float threshold_db, int distance,
                     double dc_exclude_bins, float min_confidence, int max_peaks, int32_t* count, int32_t* bin,
                     float* power_db, float* snr_db, float* confidence, float* noise_floor_db, unsigned flags);

/* Wait for all work queued on the ctx stream. */
int rmx_synchronize(rmx_ctx* ctx);

/* With option "timing"=1: accumulated HIP-event times of the last rmx_xcorr_batch call, per kernel
 * family (forward-spectrum kernels, pair kernels), and the number of launches of each.  Waits for
 * the stream.  Any pointer may be NULL. */
int rmx_last_timing(rmx_ctx* ctx, float* fwd_ms, int* fwd_launches, float* pair_ms,
                    int* pair_launches);

/* The same per kernel family of EVERY path (N = 4096, whole-window, four-step, CAF): kind = 0, 1, 2, ... until the call
 * returns RMX_E_INVAL; *name is a static string ("k_win|k_pair", "g_cols_fwd", "g_rows_fused", "g_rows_anchor",
 * "g_cols_inv", "g_final", "g_win_*", ...), *ms the summed HIP-event time of that family's launches in the last
 * rmx_xcorr_batch / rmx_caf_batch call, *launches their number.  Diagnostic only: no reference counterpart (the
 * reference times nothing on this path); bench.py reports it per BASELINE shape.  Any pointer may be NULL. */
int rmx_last_timing_kind(rmx_ctx* ctx, int kind, const char** name, float* ms, int* launches);

/* Static text naming what this binary was built from: "RMX_BUILD_INFO source_digest=<16 hex> arch=gfx950", the digest
 * being sha256 over the kernel sources, this header and the compiler flags as __graft_entry__.source_digest() computes
 * it.  bench.py prints it beside the digest of the sources it sees and refuses to run on a mismatch; never fails. */
const char* rmx_build_info(void);

/* Bytes of device scratch the ctx holds (spectra, tables, staging). */
size_t rmx_scratch_bytes(const rmx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RMX_H */
