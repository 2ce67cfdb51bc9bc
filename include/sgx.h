/*
 * sgx.h -- C ABI of the MI355X (gfx950) streaming-STFT spectrogram engine.
 *
 * This is the drop-in boundary for ONE path of JacksonCampolattaro/spectrogram-rs: PCM ->
 * Hann + 2x zero-pad -> c2c FFT -> stereo magnitudes -> log-frequency resample -> dB -> colour
 * -> RGBA pixel columns.  Every entry point names the reference interface it replaces
 * (file:line relative to the reference repository).  Plain pointers and sizes only; no C++ or
 * framework types cross this boundary.  INTEGRATION.md shows the Rust `extern "C"` binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - All `d_*` pointers are DEVICE pointers (hipMalloc'ed, or a torch tensor's data_ptr()).
 *     All `h_*` pointers are host pointers.  The caller owns every I/O buffer.
 *   - Work is enqueued on the context's stream (sgx_set_stream; default: the NULL stream) and is
 *     asynchronous; sgx_sync() waits for it.  The batch calls (sgx_stft_batch*, sgx_render_*, sgx_magnitude_in,
 *     sgx_image_write_columns / _read, sgx_view_write_rows / _draw, sgx_synth_white_noise, sgx_checksum_add) only enqueue;
 *     the calls that hand host data back or free host slots (sgx_process_one, sgx_live_tick*, sgx_spectrum_levels,
 *     sgx_checksum) and the calls that replace tables (sgx_set_gradient*, sgx_set_builtin_*) wait for the context's OWN
 *     stream; no call waits for another context's stream (tests/test_gpu_streams.py).
 *   - Every function returns SGX_OK (0) or a negative sgx_status; the text of the last error is
 *     available from sgx_last_error().  The library never aborts the host process (the
 *     reference unwrap()s: fft.rs:24,77).
 *   - A context is not thread-safe: one context per thread / stream / GPU (the reference holds
 *     its transform in a RefCell on the GTK main thread: gpu_spectrogram.rs:58).
 *   - Device buffers are at least 8-byte aligned ((l, r) pairs are moved as 8-byte words); hipMalloc and torch
 *     allocations are 256-byte aligned, which is also what the store pattern of the kernels is tuned for.
 *   - There is NO CPU fallback: without a HIP device sgx_create() fails with SGX_ERR_NO_DEVICE.
 */
#ifndef SGX_H
#define SGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SGX_API __attribute__((visibility("default")))
#else
#define SGX_API
#endif

typedef enum sgx_status {
    SGX_OK = 0,
    SGX_ERR_INVALID_ARG = -1,
    SGX_ERR_UNSUPPORTED = -2, /* e.g. a transform length this build has no kernel for */
    SGX_ERR_HIP = -3,         /* a HIP runtime call failed; see sgx_last_error */
    SGX_ERR_NOMEM = -4,
    SGX_ERR_NO_DEVICE = -5
} sgx_status;

/* interpolated_frequency_sample.rs:46-48 calls cubic_interpolate (:88-105); cosine_interpolate
 * (:78-86) is dead code there but is what README.md:19-20 and BASELINE config 3 name. */
#define SGX_INTERP_CUBIC 0u
#define SGX_INTERP_COSINE 1u

/* colorous Gradient::eval_continuous index rule for 256-entry ramps (un-vendored crate; the
 * rule is an input so it can be corrected without touching kernels) */
#define SGX_LUT_FLOOR_N 0u   /* idx = clamp(floor(t * n), 0, n-1) */
#define SGX_LUT_ROUND_NM1 1u /* idx = clamp(round(t * (n-1)), 0, n-1) */

typedef struct sgx_ctx sgx_ctx;

/* One plain struct replaces the reference's compile-time constants:
 *   FastFourierTransform::new(sample_rate, period)                      fft.rs:18
 *   AudioStreamTransform::new(stream, transform, stride)                audio_transform.rs:22-32
 *   TEXTURE_HEIGHT = 1024, y range 32..22030 Hz                         simple_spectrogram.rs:34-35,107
 *   MIN_DB = -70, MAX_DB = -10                                          colorscheme.rs:16-17        */
typedef struct sgx_config {
    uint32_t struct_size;    /* = sizeof(sgx_config); set by sgx_config_init */
    float sample_rate;       /* Hz */
    float period;            /* s; W = (period * sample_rate) as usize            (fft.rs:19)  */
    float stride;            /* s; H = (stride * sample_rate) as usize  (audio_transform.rs:35) */
    uint32_t window_samples; /* if non-zero, W directly (period ignored)                        */
    uint32_t hop_samples;    /* if non-zero, H directly (stride ignored)                        */
    uint32_t channels;       /* interleaved channels in the PCM stream: 1 = mono, expanded to
                                (s, s) as audio_input_list_model.rs:67-69 does; 2 = (l, r);
                                2k = k independent (l, r) pairs (extension, BASELINE config 4)  */
    uint32_t rows;           /* R, pixel rows per column (default 1024)                         */
    double f_min, f_max;     /* log axis range in Hz (default 32, 22030)                        */
    float min_db, max_db;    /* default -70, -10                                                */
    uint32_t interp;         /* SGX_INTERP_*                                                    */
    uint32_t lut_index_mode; /* SGX_LUT_*                                                       */
    int32_t device;          /* HIP device ordinal, or -1 for the current device                */
    uint32_t flags;          /* SGX_FLAG_*                                                      */
} sgx_config;

#define SGX_FLAG_FORCE_GENERIC 1u /* use the generic power-of-two kernel even where a tuned one exists, the chirp-z kernel even where the mixed-radix one applies (testing) */
#define SGX_FLAG_NO_FUSED_RENDER 4u /* sgx_render_batch: run STFT and pixel stage as two kernels even where the fused one applies (A/B) */
/* Mono streams.  The reference duplicates a mono sample into (s, s) and transforms every frame on its own
 * (audio_input_list_model.rs:67-69, fft.rs:47-99).  DEFAULT here, at every window and hop: exactly that dataflow, one transform per
 * frame -- every frame within north_star's tolerance of its OWN peak on any input.  At W 2048 (any hop, any alignment of the stream)
 * it is computed as the 4096-point spectrum of a REAL frame, a 2048-point complex transform + one butterfly per bin
 * (stft4096_real.hip; at H 256, the BASELINE shape, with the window sliding in registers): the cost of half a transform.  The
 * mixed-radix kernel does the same at every window it serves (stft_mixed.hip, real-input mode: the application's 2400 and 2205, every
 * 2-3-5-7-smooth length, W 512 / 1024 / 4096; any hop and alignment), the chirp-z kernel too from W 86 on (1102 at 22.05 kHz).
 * Elsewhere (W 8192, the small windows of the generic kernels) it is the (s, s) transform itself.
 * SGX_FLAG_PAIRED_FRAMES (opt-in): two frames (2j, 2j+1) per transform in its real and imaginary part -- half the work of the (s, s)
 * transform, but the quieter frame of a pair carries the louder one's float32 rounding floor: the tolerance then holds against the
 * PAIR's peak only (measured: up to 4.7 x the own-peak tolerance across a 60 dB step inside one hop, unbounded next to digital
 * silence; invisible on stationary signals.  DESIGN.md section 4). */
#define SGX_FLAG_PAIRED_FRAMES 1024u /* mono: two frames per transform (the default of rounds 1-3) */
#define SGX_FLAG_COMPLEX_MONO 512u /* mono: the literal (s, s) 2W-point complex transform per frame -- fft.rs:47-57 -- wherever a
                                      real-input kernel would run (A/B; implies no pairing) */
#define SGX_FLAG_LUT_WALK 64u      /* fused pixel kernel: walk the dB thresholds from the log2 seed even where the host has shown that one compare pair settles the LUT index (A/B, and the test of the fallback) */
#define SGX_FLAG_MIXED_GENERIC 256u /* W = 2400: the composite-radix kernel (any 2-3-5-7-smooth length) instead of the tuned 4800-point one (A/B) */

typedef struct sgx_info {
    uint32_t struct_size;
    uint32_t window_samples;  /* W  = num_input_samples()            fft.rs:41 */
    uint32_t fft_length;      /* P  = 2 W                            fft.rs:44 */
    uint32_t num_frequencies; /* M  = num_output_frequencies() = W-1 fft.rs:33 */
    uint32_t hop_samples;     /* H                                              */
    uint32_t channels;
    uint32_t pairs;           /* (l, r) pairs per hop position: 1 for mono/stereo, channels/2 otherwise */
    uint32_t rows;            /* R */
    uint32_t sample_rate_u32; /* SampleRate(sample_rate as u32)      simple_spectrogram.rs:138 */
    uint32_t total_samples_per_column; /* sum over rows of magnitude_in's sample count */
    uint32_t stft_kernel;     /* 0 = generic power-of-two, 2 = 4096-point workgroup-per-transform (default at W 2048), 4 = Bluestein chirp-z (any 2W), 6 = mixed radix (2W = 2^a 3^b 5^c 7^d <= 20480, e.g. the application's 4800, 4410 and 19200), 9 = 4800-point workgroup-per-transform, 16 x 20 x 15 (default for W = 2400, the application's window at 48 kHz; streams of more than two channels run 6), 10 = 16384-point as 32 x 32 x 16 in one 512-thread workgroup, 32 points per thread (W = 8192; the designs 5, 7 and 8 of rounds 1-5 are gone: profiles/r06_k16.txt) */
                              /* (magnitudes: the tuned kernels -- 2, 6, 9, 10 and the chirp-z plans of 4 -- take |re + i im| through the hardware
                                 square root of fma(re, re, im * im): 1 ulp; the generic kernel (0) and the radix-4 Bluestein ladder use the
                                 correctly rounded sqrtf.  Rows of different kernels for the same input agree within the tolerance, not bit for bit) */
    uint32_t render_path;     /* sgx_render_batch as configured NOW (palette included): bit 0 = one fused PCM-to-pixel kernel (the
                                 4096-point kernel, or a compile-time plan of the mixed-radix kernel whose LDS image holds the column),
                                 bit 1 = its LUT index is seed + one compare pair (else seed + walk); neither = two kernels;
                                 bit 2 = the transform runs a compile-time plan of the composite-radix stages: stft_kernel 6 at the
                                 0.05 s windows of the usual sample rates (8 kHz to 192 kHz) and the powers of two from 512 on;
                                 stft_kernel 4 (chirp-z) for W = 86 .. 5461, e.g. 1102 at 22.05 kHz;
                                 bit 3 = a mono stream runs a real-input kernel, the W-point transform of sample pairs (unless
                                 SGX_FLAG_PAIRED_FRAMES / SGX_FLAG_COMPLEX_MONO): at W 2048, and at every window the mixed-radix kernel or the chirp-z
                                 stages serve (stft_kernel 6, 9 and 4 with bit 2; any hop and alignment) */
    uint64_t mags_bytes_per_frame; /* pairs * M * 2 * 4 */
    uint64_t rgba_bytes_per_frame; /* pairs * R * 4     */
} sgx_info;

/* ---- lifetime ---------------------------------------------------------------------------- */

SGX_API const char *sgx_version(void);

/* Fill `cfg` with the reference's defaults for BASELINE config A:
 * 48 kHz, W 2048 / H 256 (given as window_samples / hop_samples), 1 channel, 1024 rows,
 * 32..22030 Hz, -70..-10 dB, cubic interpolation, SGX_LUT_FLOOR_N, current device. */
SGX_API int sgx_config_init(sgx_config *cfg);

/* Replaces FastFourierTransform::new (fft.rs:18-31: FFTW planning) + AudioStreamTransform::new
 * (audio_transform.rs:22-32) + SimpleSpectrogram's axis/palette construction
 * (simple_spectrogram.rs:88-113).  Builds the Hann table, twiddles, per-row resampling tables,
 * dB thresholds and the default gradient (Magma, simple_spectrogram.rs:95) on the device. */
SGX_API int sgx_create(const sgx_config *cfg, sgx_ctx **out_ctx);
SGX_API void sgx_destroy(sgx_ctx *ctx);

/* Text of the most recent error on this context (ctx == NULL: most recent sgx_create failure on
 * the calling thread).  Never NULL. */
SGX_API const char *sgx_last_error(const sgx_ctx *ctx);

SGX_API int sgx_query(const sgx_ctx *ctx, sgx_info *out);

/* Number of frames the hop loop of AudioStreamTransform::process (audio_transform.rs:34-42)
 * yields from n_samples samples per channel: max(0, (n - W) / H + 1).  Frame t covers samples
 * [t*H, t*H + W).  (The reference also skips H samples on its terminating short read -- a
 * live-capture artefact that a batch engine must not reproduce.) */
SGX_API size_t sgx_num_frames(const sgx_ctx *ctx, size_t n_samples);

/* `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default. */
SGX_API int sgx_set_stream(sgx_ctx *ctx, void *stream);
SGX_API int sgx_sync(sgx_ctx *ctx);

/* ---- the transform: replaces AudioTransform::process / AudioStreamTransform::process ----- */

/* Batched FastFourierTransform::process (fft.rs:43-99) driven by the hop loop
 * (audio_transform.rs:34-42).
 *   d_pcm   [n_samples][channels] float, interleaved
 *   d_mags  [n_out][pairs][M][2] float: (left, right) magnitudes of bins k = 1..W-1, scaled 2/W
 * Frames [first_frame, first_frame + max_frames) that exist are produced; *n_out (may be NULL)
 * receives how many.  Too few samples is not an error: *n_out = 0 (the reference returns None,
 * fft.rs:72). */
SGX_API int sgx_stft_batch(sgx_ctx *ctx, const float *d_pcm, size_t n_samples, size_t first_frame,
                           size_t max_frames, float *d_mags, size_t *n_out);

/* The same transform with the magnitudes stored as IEEE half (l, r) pairs, 4 bytes per bin
 * (round to nearest even): d_mags_f16 [n_out][pairs][M][2] half.  This is the texel format of the
 * F16F16 ring texture the default widget uploads its frames into (gpu_spectrogram.rs:218-226,
 * 268-274: rows of M texels), so a GL-interop consumer can take the rows as they are; it also
 * halves the bytes the transform writes. */
SGX_API int sgx_stft_batch_f16(sgx_ctx *ctx, const float *d_pcm, size_t n_samples, size_t first_frame,
                               size_t max_frames, void *d_mags_f16, size_t *n_out);

/* One call of AudioTransform::process (audio_transform.rs:10, fft.rs:43) on HOST buffers, for a
 * per-frame shim: h_lr [n_avail][2] (l, r) pairs, h_out [M][2].  Returns 1 (Some), 0 (None:
 * n_avail < W) or a negative sgx_status.  Synchronous. */
SGX_API int sgx_process_one(sgx_ctx *ctx, const float *h_lr, size_t n_avail, float *h_out);

/* ---- the pixel path: replaces the loop of SimpleSpectrogram::snapshot --------------------- */

/* PCM -> RGBA pixel columns in one call (simple_spectrogram.rs:136-165 for every frame):
 * InterpolatedFrequencySample::magnitude_in over the log-spaced row edges of
 * LogCoordf64::unmap (log_scaling.rs:114-119), ColorScheme::color_for (colorscheme.rs:55-71),
 * Pixbuf::put_pixel (simple_spectrogram.rs:153-160).
 *   d_rgba [n_out][pairs][R][4] uint8; index [y] is the IMAGE row y = R-1-py, i.e. row 0 is the
 *   highest frequency, exactly the column that put_pixel writes (simple_spectrogram.rs:150). */
SGX_API int sgx_render_batch(sgx_ctx *ctx, const float *d_pcm, size_t n_samples, size_t first_frame,
                             size_t max_frames, uint8_t *d_rgba, size_t *n_out);

/* The pixel stage alone, from magnitudes already on the device (stage-wise parity):
 *   d_mags [n_columns][M][2] -> d_rgba [n_columns][R][4]. */
SGX_API int sgx_render_mags(sgx_ctx *ctx, const float *d_mags, size_t n_columns, uint8_t *d_rgba);

/* FrequencySample::magnitude_in(f0..f1) (src/fourier/mod.rs:17-21; interpolated_frequency_sample.rs:60-75)
 * for n_ranges arbitrary frequency ranges and every column: the mean of the interpolated (l, r) samples,
 * no colour.  h_ranges [n_ranges][2] = (f0, f1) in Hz on the host; d_out [n_columns][n_ranges][2] float.
 * This is what SpectrumAnalyzer::push_frequencies consumes (spectrum_analyzer.rs:48-61: 128 log-spaced
 * bands).  The per-range tables are built on the host and cached until the range set changes. */
SGX_API int sgx_magnitude_in(sgx_ctx *ctx, const float *d_mags, size_t n_columns, const float *h_ranges,
                             uint32_t n_ranges, float *d_out);

/* SpectrumAnalyzer::push_frequencies (spectrum_analyzer.rs:46-68) for ONE column of magnitudes
 * d_column [M][2] on the device: the n_bars (the widget has 128) adjacent bands of
 * log_space(32, max(sample_rate / 2, 22050), n_bars + 1, 10) (:20-36,52-58; f32 logf / powf on the host),
 * magnitude_in of each band on the device, then on the host level = (10 log10f(hypotf(l, r) + 1e-7) + 70) / 60
 * as f64 and h_levels[i] = max(level, h_levels[i] * 0.99) (:61-66: the bars decay by 1 % per pushed
 * column; a fresh widget starts every bar at 0.3, :92).  h_levels [n_bars] double, in/out.  Synchronous. */
SGX_API int sgx_spectrum_levels(sgx_ctx *ctx, const float *d_column, uint32_t n_bars, double *h_levels);

/* ---- live capture: the ring between the audio callback and the GUI tick -------------------- */

/* The reference's producer is the cpal input callback pushing (l, r) pairs into a 4096-entry
 * ringbuf::HeapRb (audio_input_list_model.rs:30,63-72); its consumer is the GTK tick running the
 * hop loop over that ring (audio_transform.rs:34-42 from gpu_spectrogram.rs:255-262 /
 * simple_spectrogram.rs:136-139).  sgx_live keeps the CONSUMED side of that ring on the device:
 * a push lands in a pinned host ring; a tick sends only the samples that arrived since the last
 * tick (one async copy, two when the ring wraps), runs every complete frame through one launch
 * of the context's transform (context channels must be 2: the ring holds (l, r) pairs), returns
 * the results to the host and keeps the W - H overlap resident.  One producer thread and one
 * consumer thread may run concurrently (lock-free single-producer / single-consumer, as HeapRb). */
typedef struct sgx_live sgx_live;

#define SGX_LIVE_MAGS 0     /* float [frames][M][2]: what AudioStreamTransform::process yields        */
#define SGX_LIVE_MAGS_F16 1 /* half  [frames][M][2]: rows of the F16F16 ring (gpu_spectrogram.rs:268) */
#define SGX_LIVE_RGBA 2     /* uint8 [frames][R][4]: columns of SimpleSpectrogram::snapshot           */

#define SGX_LIVE_REFERENCE_SKIP 1u /* also skip H samples on the terminating short read, exactly as
                                      audio_transform.rs:37-41 does (it drops up to H samples per tick);
                                      default: frames are t*H exact */

/* capacity_pairs: ring size in (l, r) pairs (the reference: 4096); must hold at least one window.
 * A ring belongs to its context: destroy it before sgx_destroy(ctx). */
SGX_API int sgx_live_create(sgx_ctx *ctx, size_t capacity_pairs, uint32_t flags, sgx_live **out);
SGX_API void sgx_live_destroy(sgx_live *live);
/* The cpal callback (audio_input_list_model.rs:63-75): h_samples holds n_values floats, interleaved by
 * `channels`.  1 channel -> (s, s) pairs, 2 -> (l, r) pairs, anything else -> SGX_ERR_UNSUPPORTED (the
 * reference prints "N-channel input not supported!").  Pairs that do not fit are dropped (push_iter).
 * Returns the number of pairs accepted, or a negative sgx_status.  Producer thread. */
SGX_API long long sgx_live_push(sgx_live *live, const float *h_samples, size_t n_values, uint32_t channels);
/* HeapRb::occupied_len: pairs pushed and not yet skipped */
SGX_API size_t sgx_live_occupied(const sgx_live *live);
/* One GUI tick (audio_transform.rs:34-42): every complete frame of the ring, at most max_frames, in the
 * format `what`, into the HOST buffer h_out; the ring advances by H per frame.  *n_frames may be 0 (not
 * an error).  Synchronous: returns when h_out is filled.  Consumer thread. */
SGX_API int sgx_live_tick(sgx_live *live, int what, void *h_out, size_t max_frames, size_t *n_frames);

/* ---- ColorScheme ---------------------------------------------------------------------------- */

/* ColorScheme::new_mono(gradient, name) / new_stereo(gradient, background, name)
 * (colorscheme.rs:24-39).  h_rgb: [n][3] table standing in for colorous' Gradient; stereo != 0
 * selects the diverging branch of color_for (colorscheme.rs:63-66: colour from the left/right
 * balance, alpha from the bounded dB magnitude). */
SGX_API int sgx_set_gradient(sgx_ctx *ctx, const uint8_t *h_rgb, uint32_t n, int stereo);
/* The same two constructors for ANY continuous gradient: `eval` stands in for colorous'
 * Gradient::eval_continuous(t) (colorscheme.rs:43-69) -- the integrator passes a thunk around colorous
 * itself, so spline (ColorBrewer) and closed-form (Turbo, Cividis, Cubehelix ...) gradients are rendered
 * with exactly the bytes colorous returns and no table of this library is involved.  `eval` is called
 * on the host, only inside this function (and sgx_lookup_table), never during a launch: the colour
 * as a function of the bounded dB value (or of the left/right balance) is a step function of bytes; its
 * switch points are located by bisection over float (double) bit patterns and uploaded as thresholds. */
typedef void (*sgx_gradient_fn)(double t, uint8_t rgb_out[3], void *user);
SGX_API int sgx_set_gradient_fn(sgx_ctx *ctx, sgx_gradient_fn eval, void *user, int stereo);
/* "viridis" | "magma" | "inferno" | "plasma" (colorscheme.rs:131-139) */
SGX_API int sgx_set_builtin_gradient(sgx_ctx *ctx, const char *name);
/* ColorScheme::new_mono / new_stereo (colorscheme.rs:24-39) with one of the gradients this library can evaluate itself:
 * "viridis" "magma" "inferno" "plasma" (256-entry ramps) and colorous' ColorBrewer B-spline gradients "red_yellow_blue"
 * "red_blue" "spectral" "red_yellow_green" "pink_green" "purple_orange" "purple_green" "brown_green" "red_grey" "reds"
 * "blues" "greens" "greys" "oranges" "purples" (continuous: anchors from ColorBrewer, d3's interpolateRgbBasis),
 * "turbo" "cividis" (d3's quintics) and "cubehelix" "cool" "warm" (d3's interpolateCubehelixLong, bytes by truncation as
 * the reference's screenshots/colorscheme-cool.png shows colorous doing) -- every one of the 19 entries of
 * default_color_schemes (colorscheme.rs:125-151).  [third-party]: colorous is not vendored; what the reference's four
 * screenshots pin (the values of the Viridis / Magma / Plasma tables, Cool's curve and byte rule) is tests/test_host_logic.py.
 * stereo != 0: the diverging rule of colorscheme.rs:63-66 (colour from l / (|l| + |r|), alpha from the level). */
SGX_API int sgx_set_builtin_scheme(sgx_ctx *ctx, const char *name, int stereo);
/* eval_continuous(t) of a built-in gradient on the host (ColorScheme::background / foreground, colorscheme.rs:41-53) */
SGX_API int sgx_builtin_gradient_eval(const char *name, double t, uint8_t rgb_out[3]);
SGX_API int sgx_builtin_gradient(const char *name, uint8_t *h_rgb_out /* [256][3] */);

/* ColorScheme::lookup_table(resolution) (colorscheme.rs:73-92): h_out [res][res][4] float,
 * the palette texture of the GLSL path (gpu_spectrogram.rs:233-238). */
SGX_API int sgx_lookup_table(sgx_ctx *ctx, uint32_t resolution, float *h_out);

/* ---- introspection (tests, tooling) --------------------------------------------------------- */

/* R+1 row edges in Hz as f32: LogCoordf64::unmap(p, (0, R)) for p = 0..R
 * (log_scaling.rs:114-119; simple_spectrogram.rs:142-145). */
SGX_API int sgx_bin_edges(const sgx_ctx *ctx, float *h_out);
/* per-row sample counts of magnitude_in (interpolated_frequency_sample.rs:63-64): h_out [R] */
SGX_API int sgx_row_sample_counts(const sgx_ctx *ctx, uint32_t *h_out);
/* the Hann table (fft.rs:61): h_out [W] */
SGX_API int sgx_window(const sgx_ctx *ctx, float *h_out);

/* ---- the default widget's variant: F16F16 ring texture + fragment program ---------------------
 * GPUSpectrogram keeps its frames as rows of a VIEWPORT_FRAMES x (W - 1) F16F16 texture used as a ring
 * (gpu_spectrogram.rs:21,67,218-226) and draws it with a fragment program (:150-186: log-frequency lookup, dB, pan,
 * 32 x 32 palette texture).  sgx_view is that texture on the device and that program as a kernel.  Not a parity target:
 * OpenGL leaves the filtering arithmetic to the implementation; the program text and the sampler state are restated
 * (GL_LINEAR = two nearest texels per axis weighted by the fractional part, REPEAT for the ring, CLAMP for the
 * palette; float32; the shader's hard-coded 32 / 22030 Hz, quirk Q9).  Destroy views before sgx_destroy(ctx). */
typedef struct sgx_view sgx_view;
SGX_API int sgx_view_create(sgx_ctx *ctx, uint32_t viewport_frames /* 2048 */, sgx_view **out);
SGX_API void sgx_view_destroy(sgx_view *view);
/* The upload loop of render() (gpu_spectrogram.rs:255-275): append n_rows rows of [M][2] half -- a DEVICE pointer,
 * e.g. what sgx_stft_batch_f16 wrote -- at the ring's offset, wrapping at its height; the new offset is returned. */
SGX_API int sgx_view_write_rows(sgx_view *view, const void *d_rows_f16, size_t n_rows, uint32_t *offset_out);
SGX_API uint32_t sgx_view_offset(const sgx_view *view);
/* One GUI tick of the default widget (gpu_spectrogram.rs:255-275: `for frame in fft.process()` -> fft_texture.write): every
 * complete frame of the live ring, at most max_frames, is transformed and its half-pair row appended to the view's ring texture,
 * device to device -- the only host traffic of the tick is the new samples going up.  `view` must belong to the ring's context: a view of another
 * (or of a destroyed) context is refused with SGX_ERR_INVALID_ARG before anything is uploaded, transformed or skipped. */
SGX_API int sgx_live_tick_view(sgx_live *live, sgx_view *view, size_t max_frames, size_t *n_frames);
/* The fragment program over a width x height viewport: d_rgba_f32 [height][width][4] float = f_color per fragment, row 0
 * at the BOTTOM (GL).  The palette texture is ColorScheme::lookup_table(32) of the context's current colour scheme,
 * min_db / max_db the context's. */
SGX_API int sgx_view_draw(sgx_view *view, uint32_t width, uint32_t height, float *d_rgba_f32);

/* ---- the CPU pixel path's image ring --------------------------------------------------------------
 * SimpleSpectrogram keeps a row-major width x height RGBA Pixbuf (simple_spectrogram.rs:89-94; 1024 x 1024), writes every new
 * frame as one pixel column at x = offset, image row height - 1 - py (:140-161), advances offset = (px + 1) % width (:164) and
 * composes the scrolling picture from the sub-images [offset, width) and [0, offset) (:181-209).  sgx_image is that Pixbuf on
 * the device: [height][width][4] bytes, rowstride 4 * width, height = the context's rows.  Destroy images before sgx_destroy(ctx)
 * (an image that outlives its context answers SGX_ERR_INVALID_ARG). */
typedef struct sgx_image sgx_image;
SGX_API int sgx_image_create(sgx_ctx *ctx, uint32_t width /* 1024 */, sgx_image **out);
SGX_API void sgx_image_destroy(sgx_image *image);
/* The pixel loop's put_pixel + offset update for n columns at once (:140-164): d_rgba [n][rows][4] -- a DEVICE pointer, e.g. what
 * sgx_render_batch wrote -- column i lands at x = (offset + i) % width, offset advances by n (mod width); with n > width the ring
 * laps itself and the later columns win, as written one by one.  The new offset is returned. */
SGX_API int sgx_image_write_columns(sgx_image *image, const uint8_t *d_rgba, size_t n_columns, uint32_t *offset_out);
SGX_API uint32_t sgx_image_offset(const sgx_image *image);
/* the Pixbuf's get_width() / get_height() (:89-94): width as created, height = the context's rows -- what a caller sizes the
 * buffer of sgx_image_read with (0 for a null image) */
SGX_API uint32_t sgx_image_width(const sgx_image *image);
SGX_API uint32_t sgx_image_height(const sgx_image *image);
/* One GUI tick of SimpleSpectrogram::snapshot (:136-165: `for frequency_sample in fft.process()` -> put_pixel): every complete frame
 * of the live ring, at most max_frames, becomes a pixel column of the image, device to device. */
SGX_API int sgx_live_tick_image(sgx_live *live, sgx_image *image, size_t max_frames, size_t *n_frames);
/* d_out [height][width][4]: the buffer as it lies (scrolled = 0), or the composed picture -- columns [offset, width) followed by
 * [0, offset) (:181-209; scrolled != 0). */
SGX_API int sgx_image_read(sgx_image *image, int scrolled, uint8_t *d_out);
/* the device buffer itself ([height][width][4], valid until sgx_image_destroy; ordered on the context's stream) */
SGX_API const uint8_t *sgx_image_pixels(const sgx_image *image);

/* ---- synthetic input + verification helpers (bench / multi-GPU harness, not reference API) -- */

/* Counter-based white noise, identical on CPU and GPU:
 * x[n] = ((lowbias32((seed + hi32(n) * 0x9E3779B9) ^ lo32(n)) >> 8) * 2^-23) - 1.
 * Writes d_out[i * channels + c] for sample index first + i and channel c (seed + c). */
SGX_API int sgx_synth_white_noise(sgx_ctx *ctx, float *d_out, uint64_t first, size_t n_samples,
                                  uint32_t channels, uint32_t seed);
/* 64-bit order-independent checksum (sum of a 64-bit mix of each (index, 32-bit word)) of a device buffer of
 * n_bytes (multiple of 4), with word indices starting at base_word: equal for any sharding of
 * the same bytes.  Synchronous. */
SGX_API int sgx_checksum(sgx_ctx *ctx, const void *d_buf, size_t n_bytes, uint64_t base_word, uint64_t *h_out);
/* The same sum, ADDED (mod 2^64) into a caller-owned device accumulator *d_acc and asynchronous on the context's
 * stream: the root of a sharded run consumes every gathered piece of pixel columns this way without a host
 * round trip per piece (1e8 columns are 410 GB: the image is never materialised).  Zero *d_acc first. */
SGX_API int sgx_checksum_add(sgx_ctx *ctx, const void *d_buf, size_t n_bytes, uint64_t base_word, uint64_t *d_acc);

#ifdef __cplusplus
}
#endif
#endif /* SGX_H */
