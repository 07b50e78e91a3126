/*
 * lbm_abi.h -- C ABI of liblbm_hip.so: the MI355X (gfx950) kernels of the batched EEG epoch
 * pipeline (band-pass FIR -> ICA apply -> epoch post-processing -> Welch PSD -> band powers ->
 * VQ-VAE encoder -> nearest-codebook tokenisation).
 *
 * The reference (beetleberries/LBM) is pure Python and has no FFI; each entry point below names the
 * reference call site (file:line under /root/reference) whose arithmetic it replaces.  The
 * reference-side binding is a ctypes stub.
 *
 * Conventions
 *   - every pointer is CALLER-OWNED DEVICE memory (hipMalloc / torch CUDA tensor data_ptr());
 *     the library never allocates, frees or synchronises; scratch is passed in explicitly;
 *   - tensors are row-major with the innermost dimension contiguous, float32 unless stated;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: LBM_OK (0) or a negative LBM_E* code; lbm_last_error() gives a thread-local
 *     human-readable message for the last failing call on this thread.  Nothing ever exit()s or
 *     throws across the ABI (contrast the reference scripts' exit() calls);
 *   - empty inputs (rows == 0, E == 0, N == 0) are valid no-ops.
 */
#ifndef LBM_ABI_H
#define LBM_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBM_ABI_VERSION 1

#define LBM_OK 0
#define LBM_EINVAL (-1)       /* bad argument (NULL pointer, non-positive size, misaligned, ...) */
#define LBM_EUNSUPPORTED (-2) /* argument combination outside what the kernels implement */
#define LBM_ELAUNCH (-3)      /* HIP reported an error when launching */

typedef void* lbm_stream_t;

int lbm_version(void);
const char* lbm_last_error(void);

/* ---- a1 / a2: zero-phase FIR ------------------------------------------------------------------
 * Replaces mne raw.filter(l_freq=4, h_freq=30, fir_design='firwin') at training/preprocess.py:77,
 * multi-train/withpsd.py:118, preprocessing_unsupervised.py:36, and raw.notch_filter(...) at
 * training/preprocess.py:82, multi-train/withpsd.py:119 (same machinery, different taps).
 * y[r, n] = sum_k taps[k] * xpad[r, n + (ntaps-1) - k], xpad = x padded by (ntaps-1)/2 per side.
 * x, y: [rows, T]; taps: [ntaps] device floats, ntaps odd (type-I, delay (ntaps-1)/2 removed).
 * pad_mode: LBM_PAD_REFLECT_LIMITED = MNE 'reflect_limited' (odd reflection 2*x[0]-x[m], zero where
 * reflection runs out of samples); LBM_PAD_ZERO = zero padding.  x and y must not overlap. */
#define LBM_PAD_REFLECT_LIMITED 0
#define LBM_PAD_ZERO 1
int lbm_fir_f32(const float* x, float* y, int64_t rows, int32_t T, const float* taps, int32_t ntaps,
                int32_t pad_mode, lbm_stream_t stream);

/* Same filter by fast convolution (MNE applies raw.filter by FFT overlap-add, mne/filter.py
 * _overlap_add_filter; call sites as above): identical arguments and result as lbm_fir_f32 (to fp32
 * rounding), about 10x fewer flops for the 825-tap band-pass.  The caller owns a device table buffer:
 * lbm_fir_fft_tables_bytes(T, ntaps) gives its size (0 = this (T, ntaps) is not supported: ntaps > 4096;
 * use lbm_fir_f32), lbm_fir_fft_tables_init fills it from the device taps (DFT of the taps and twiddles,
 * evaluated in float64 on the device, rounded once); it then serves every lbm_fir_fft_f32 call with the
 * same (T, ntaps).  tables must be 16-byte aligned; x and y must not overlap. */
size_t lbm_fir_fft_tables_bytes(int32_t T, int32_t ntaps);
int lbm_fir_fft_tables_init(void* tables, const float* taps, int32_t T, int32_t ntaps, lbm_stream_t stream);
int lbm_fir_fft_f32(const float* x, float* y, int64_t rows, int32_t T, const void* tables, int32_t ntaps,
                    int32_t pad_mode, lbm_stream_t stream);

/* ---- a3: per-sample channel affine map (ICA apply / unmix / average reference) -------------------
 * Replaces ica.apply(raw, exclude=...) at training/preprocess.py:144, multi-train/withpsd.py:152,
 * preprocessing_unsupervised.py:41 (and get_sources, set_eeg_reference('average') :35), folded to
 * y[e, o, t] = sum_c A[o, c] * x[e, c, t] + b[o].
 * x: [E, Cin, T]; y: [E, Cout, T]; A: [Cout, Cin]; b: [Cout] or NULL.  1 <= Cin, Cout <= 256.
 * x and y must not overlap. */
int lbm_affine_ch_f32(const float* x, float* y, int64_t E, int32_t Cin, int32_t Cout, int32_t T,
                      const float* A, const float* b, lbm_stream_t stream);

/* ---- 8(f)-3: ICA fit -- FastICA fixed-point statistics -------------------------------------------------
 * Replaces the per-iteration pass over the data inside ICA(method='fastica').fit(raw) at
 * training/preprocess.py:91-98, multi-train/withpsd.py:127-130, preprocessing_unsupervised.py:39-40 (MNE
 * hands the PCA-whitened recording to sklearn FastICA(whiten=False); its `_ica_par` loop body is
 *   gwtx, g_wtx = g(W X);  W1 = sym_decorrelation(gwtx X^T / T - g_wtx[:, None] * W)):
 *   y = W z,  gxt[i, j] = sum_t fun(y_i(t)) z_j(t),  gsum[i] = sum_t fun'(y_i(t))
 * z: [n, T] float32 with row pitch ldz (the whitened recording); W: [n, n] float32 row-major;
 * gxt: [n, n] and gsum: [n] float64 outputs (sums, not means), reduced in a fixed order (deterministic).
 * fun: LBM_ICA_FUN_LOGCOSH (tanh(alpha y); sklearn default), _EXP, _CUBE, or _IDENTITY (fun(y) = y and
 * gsum = sum_t y: with W = I this is the second-moment + sum pass of the PCA whitening step).
 * n <= 64.  workspace: lbm_ica_stats_workspace_bytes(T, n) bytes of device scratch.  The small n x n
 * algebra between passes (symmetric decorrelation, convergence test) stays on the host (lbm_amd/ica.py). */
#define LBM_ICA_FUN_LOGCOSH 0
#define LBM_ICA_FUN_EXP 1
#define LBM_ICA_FUN_CUBE 2
#define LBM_ICA_FUN_IDENTITY 3
size_t lbm_ica_stats_workspace_bytes(int64_t T, int32_t n);
int lbm_ica_stats_f32(const float* z, int64_t ldz, int64_t T, int32_t n, const float* W, int32_t fun, float alpha,
                      double* gxt, double* gsum, void* workspace, lbm_stream_t stream);

/* ---- a4 (slicing): epochs from a continuous recording ------------------------------------------------
 * Replaces the window extraction of mne.Epochs(raw, events, tmin, tmax) at training/preprocess.py:262-271,
 * multi-train/withpsd.py:213-216: out[e, c, t] = raw[c, starts[e] + t], t < Tw.  raw: [C, Trec] with row
 * pitch ldr; starts: [E] int64 device array (event sample + round(tmin*sfreq)); out: [E, C, Tw].
 * Samples outside [0, Trec) read as 0 (MNE drops such events; lbm_amd.make_epochs does too). */
int lbm_epochs_gather_f32(const float* raw, int64_t ldr, const int64_t* starts, float* out, int64_t E, int32_t C,
                          int32_t Tw, int64_t Trec, lbm_stream_t stream);

/* ---- 8(f)-4: FFT resampling of a continuous recording ----------------------------------------------------
 * Replaces raw.resample(250) at preprocessing_unsupervised.py:37 (mne.filter.resample(x, up, down, npad='auto',
 * window='boxcar', pad='reflect_limited')): each row is padded to the next power of two N by odd reflection,
 * transformed, cut (or extended) to round(ratio * N) bins with MNE's Nyquist-bin rule, transformed back and
 * trimmed to lbm_resample_length(T, up, down) = round(T * up / down) samples (round-half-even).
 * x: [rows, T] with pitch ldx; y: [rows, length] with pitch ldy; workspace: lbm_resample_workspace_bytes bytes.
 * Any ratio: when the resampled padded length round(ratio * N) is a power of two (up/down = 2^s) the inverse transform is a
 * plain FFT; for every other length (250/600, 250/512, 256/500 ...) it is computed as a chirp-z (Bluestein) convolution of the
 * next power of two >= 2 * length - 1, chirp phases from n^2 mod 2L in exact integer arithmetic.  rows <= 65535 per call;
 * rows shorter than 16 padded samples return LBM_EUNSUPPORTED (workspace_bytes / length return 0). */
int64_t lbm_resample_length(int64_t T, int32_t up, int32_t down);
size_t lbm_resample_workspace_bytes(int64_t rows, int64_t T, int32_t up, int32_t down);
int lbm_resample_fft_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t T, int32_t up,
                         int32_t down, void* workspace, lbm_stream_t stream);

/* ---- a4 / a5 / a6: epoch post-processing ----------------------------------------------------------
 * Replaces mne.Epochs(..., baseline=(None, 0)) baseline correction (training/preprocess.py:262-271),
 * reject=dict(eeg=150e-6)+drop_bad (multi-train/withpsd.py:44-46,213-219) and the per-epoch
 * per-channel z-score (training/preprocess.py:279-287).
 * For every (epoch, channel) row of x[E, C, T]:
 *   base  = mean(x[:n_baseline])            (n_baseline == 0: no baseline correction)
 *   y_base = x - base                        (written when y_base != NULL; may alias x)
 *   keep[e] = 0 if any channel has max-min > ptp_thresh, else 1 (written when keep != NULL)
 *   y_std = (y_base - mean_t) / (std_t + eps), population std (written when y_std != NULL and
 *            standardize_mode == LBM_STD_EPOCH_CHANNEL; y_std may alias x)
 * T <= 40960. */
#define LBM_STD_NONE 0
#define LBM_STD_EPOCH_CHANNEL 1
int lbm_epoch_post_f32(const float* x, float* y_base, float* y_std, uint8_t* keep, int64_t E, int32_t C,
                       int32_t T, int32_t n_baseline, float ptp_thresh, int32_t standardize_mode, float eps,
                       lbm_stream_t stream);

/* a3 + a4 / a5 / a6 fused: the ICA channel map and the epoch post-processing in ONE pass over the epoch (one read, one
 * write per requested output instead of affine r+w then post r+w).  Replaces ica.apply (training/preprocess.py:144,
 * multi-train/withpsd.py:152) followed by mne.Epochs(..., baseline=(None, 0)) (:262-271), reject=dict(eeg=150e-6)
 * (multi-train/withpsd.py:213-219) and the per-epoch per-channel z-score (training/preprocess.py:279-287):
 *   y = A x + b;  base = mean(y[:n_baseline]);  y_base = y - base;  keep[e] = 0 if any channel's max - min > ptp_thresh;
 *   y_std = (y_base - mean_t) / (std_t + eps);  row_scale[e, o] = std_t + eps.
 * x: [E, Cin, T]; A: [Cout, Cin]; b: [Cout] or NULL; y_base, y_std: [E, Cout, T] or NULL (not written); row_scale: [E, Cout]
 * or NULL; keep: [E] or NULL.  y_std / row_scale need standardize_mode == LBM_STD_EPOCH_CHANNEL.  row_scale is what
 * lbm_welch_scaled_f32 takes to compute the PSD of y_base from y_std, so y_base need not exist at all.
 * fp32 MFMA (exact ascending-channel fmaf chain starting from b).  Supported: Cin <= 64, Cout <= 256, T <= 1536
 * (lbm_affine_post_supported returns 1); otherwise LBM_EUNSUPPORTED -- use lbm_affine_ch_f32 + lbm_epoch_post_f32.
 * Outputs must not overlap x. */
int lbm_affine_post_supported(int32_t Cin, int32_t Cout, int32_t T);
int lbm_affine_post_f32(const float* x, const float* A, const float* b, float* y_base, float* y_std, float* row_scale,
                        uint8_t* keep, int64_t E, int32_t Cin, int32_t Cout, int32_t T, int32_t n_baseline,
                        float ptp_thresh, int32_t standardize_mode, float eps, lbm_stream_t stream);

/* mne.decoding.Scaler(scalings='mean') (multi-train/withpsd.py:267-270) = fit + transform:
 * lbm_channel_moments_f32: moments[c] = {sum, sum of squares, sample count} in float64 over all kept
 *   epochs x time (keep == NULL: all epochs), reduced in a fixed order (deterministic).
 *   row_ws: scratch, E*C*2 doubles.  moments: [C, 3] doubles.
 * lbm_channel_scale_f32:  y[e,c,t] = (x[e,c,t] - mean[c]) / scale[c]  (may run in place). */
int lbm_channel_moments_f32(const float* x, const uint8_t* keep, double* row_ws, double* moments, int64_t E,
                            int32_t C, int32_t T, lbm_stream_t stream);
int lbm_channel_scale_f32(const float* x, float* y, const float* mean, const float* scale, int64_t E,
                          int32_t C, int32_t T, lbm_stream_t stream);

/* lbm_channel_moments_f32 without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it
 * exists for callers that cannot allocate device memory around the call -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  Same arguments (minus row_ws) and bit-identical moments; the E*C*2 doubles of
 * per-row partial sums live in a static device array inside the library (16 MiB; a file-scope __device__ array is part of
 * the code object, so EVERY GPU that loads the library holds its copy until the process ends, whether or not this call was
 * ever made there), whose address is resolved once per device at the FIRST call on that device: that first call takes a
 * mutex and calls hipGetSymbolAddress (which may load the module) and is therefore NOT capture-safe -- make one warm-up
 * call per device (E = 0 is enough) before capturing a graph; later calls only enqueue kernels.
 * Limit: E * C <= LBM_STATIC_MOMENTS_MAX_ROWS, otherwise LBM_EUNSUPPORTED (use lbm_channel_moments_f32).
 * All calls on one GPU share that array: calls that may overlap on one GPU (different streams, different threads) race --
 * serialise them.  The call itself makes no hipMalloc / hipFree and allocates nothing on the host. */
#define LBM_STATIC_MOMENTS_MAX_ROWS (1 << 20)
int lbm_channel_moments_static_f32(const float* x, const uint8_t* keep, double* moments, int64_t E, int32_t C,
                                   int32_t T, lbm_stream_t stream);

/* ---- a7: Welch PSD ----------------------------------------------------------------------------------
 * Replaces epochs.compute_psd(method='welch', fmin, fmax, n_fft, window, n_overlap, average) +
 * spectrum.get_data() at multi-train/withpsd.py:236-250 and raw.compute_psd(fmin, fmax, n_fft=512) at
 * preprocessing_unsupervised.py:43 (MNE psd_array_welch -> scipy.signal.spectrogram semantics:
 * per-segment mean removal, periodic window, rFFT of length nfft, |X|^2 / (fs * sum(w^2)), x2 for all
 * bins except DC and (even nfft) Nyquist, then the average over segments).
 *
 * Tables: lbm_welch_tables_bytes() gives the size of a caller-owned device buffer that
 * lbm_welch_tables_init() fills (window, twiddles and the folded DFT basis for bins bin_lo..bin_hi,
 * evaluated in float64 on the device and rounded once).  The same buffer then serves every
 * lbm_welch_f32 call with the same (nfft, nperseg, window_id, bin_lo, bin_hi).
 * x: [rows, T]; psd: [rows, F] with F = bin_hi - bin_lo + 1 (bins are rFFT indices, inclusive);
 * seg_or_null: [rows, F, nseg] per-segment spectra (MNE average=None), nseg = (T-noverlap)/(nperseg-noverlap);
 * average_mode: LBM_AVG_MEAN | LBM_AVG_MEDIAN (scipy median-bias corrected) | LBM_AVG_NONE (psd may be NULL).
 * Requires nperseg <= nfft <= T, 0 <= noverlap < nperseg, 0 <= bin_lo <= bin_hi <= nfft/2. */
#define LBM_WIN_BOXCAR 0
#define LBM_WIN_HANN 1
#define LBM_WIN_HAMMING 2
#define LBM_AVG_MEAN 0
#define LBM_AVG_MEDIAN 1
#define LBM_AVG_NONE 2
size_t lbm_welch_tables_bytes(int32_t nfft, int32_t nperseg, int32_t bin_lo, int32_t bin_hi);
int lbm_welch_tables_init(void* tables, int32_t nfft, int32_t nperseg, int32_t window_id, int32_t bin_lo,
                          int32_t bin_hi, lbm_stream_t stream);
int lbm_welch_f32(const float* x, float* psd, float* seg_or_null, int64_t rows, int32_t T, float fs,
                  int32_t nfft, int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo,
                  int32_t bin_hi, int32_t average_mode, lbm_stream_t stream);

/* lbm_welch_f32 of rows that were rescaled per row: psd[r] = row_scale[r]^2 * Welch(x[r]).  With x = the z-scored epochs and
 * row_scale = std_t + eps (lbm_affine_post_f32) this is the PSD of the cleaned, un-standardised epochs of
 * multi-train/withpsd.py:236-250 (Welch removes each segment's mean, so the z-score's offset never matters) without those
 * epochs ever being written.  Per-segment outputs (seg_or_null) are scaled the same way. */
int lbm_welch_scaled_f32(const float* x, const float* row_scale, float* psd, float* seg_or_null, int64_t rows, int32_t T,
                         float fs, int32_t nfft, int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo,
                         int32_t bin_hi, int32_t average_mode, lbm_stream_t stream);

/* lbm_welch_f32 (mean over segments) with the peak-to-peak reject test of the epoch stage riding along: keep[e] = 0 where any
 * channel of epoch e has max - min > ptp_thresh (reject=dict(eeg=150e-6), multi-train/withpsd.py:213-219), 1 otherwise -- the
 * features-only flow (PSD + band powers of unfiltered epochs, BASELINE config 2) then reads every epoch once instead of twice.
 * x: [E, C, T]; psd: [E, C, F]; keep: [E].  Where the launch is not the team kernel the test runs as its own pass; results are
 * those of lbm_epoch_post_f32's mask and lbm_welch_f32 either way. */
int lbm_welch_keep_f32(const float* x, float* psd, uint8_t* keep, int64_t E, int32_t C, int32_t T, float fs, int32_t nfft,
                       int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo, int32_t bin_hi, float ptp_thresh,
                       lbm_stream_t stream);

/* ---- a8: theta/alpha/beta band powers and ratios ----------------------------------------------------
 * Replaces preprocessing_unsupervised.py:45-61 (also :102-110, failed/label.py:13-23):
 * log = 10*log10(psd) (skipped when already_db), theta = mean over channels x bins with 4<=f<8,
 * alpha 8<=f<13, beta 13<=f<30; out = {theta, alpha, beta, beta/theta, alpha/beta, theta/beta}
 * (ratios of the dB means, as the reference computes them).
 * psd: [E, C, F]; freqs: [F] device floats; out: [E, 6] (per_channel == 0) or [E, C, 6]. */
int lbm_bandpower_f32(const float* psd, const float* freqs, float* out, int64_t E, int32_t C, int32_t F,
                      int32_t per_channel, int32_t already_db, lbm_stream_t stream);

/* ---- a10: VQ-VAE encoder convolution ----------------------------------------------------------------
 * Replaces one nn.Conv1d(+ReLU) of VQVAE.encoder, training/model.py:169-177 (used by encode :208).
 * x: [B, Cin, Lin]; w: [Cout, Cin, ksize] (torch layout); bias: [Cout] or NULL;
 * y: [B, Cout, Lout], Lout = (Lin + 2*pad - ksize)/stride + 1.  fp32 MFMA, exact fp32 fmaf chains.
 * lbm_conv1d_pitched_f32: the same with pitched rows: channel row (b, c) of x starts at x + (b*Cin + c)*ldx,
 * of y at y + (b*Cout + c)*ldy (ldx >= Lin, ldy >= Lout, in floats).  With 16-byte aligned x, y and pitches
 * that are multiples of 4 every global access is a 16-byte one (the layout of the pipeline's activation
 * buffers); the up-to-3 floats of padding at the end of a y row may then be overwritten with unspecified
 * values.  lbm_conv1d_f32 is the dense case ldx = Lin, ldy = Lout.
 * Round 3: the encoder's two 4-tap stride-2 padding-1 layers, (Cin, Cout) = (30, 64) and (64, 128), on such 16-byte rows with
 * 16-byte aligned weights run as weights-stationary persistent kernels (csrc/conv_k4s2_ws.hip: grid = one workgroup per CU, no limit
 * on B) -- same products in the same order, same results bit for bit.  Their first call on a device makes two host-side runtime
 * queries (not capture-safe: warm up once before capturing a graph). */
int lbm_conv1d_f32(const float* x, const float* w, const float* bias, float* y, int64_t B, int32_t Cin,
                   int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu,
                   lbm_stream_t stream);
int lbm_conv1d_pitched_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                           int64_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride,
                           int32_t pad, int32_t relu, lbm_stream_t stream);

/* The same layer on the bf16 matrix cores with fp32-grade products (three bf16 pieces per operand, six piece products,
 * fp32 accumulate; see lbm_vq_argmin_bf16x6_f32 for the arithmetic and its measured accuracy): results within fp32
 * rounding of lbm_conv1d_pitched_f32, not bit-identical to it.  The weights are passed pre-split:
 * lbm_conv1d_split_tables_bytes(Cin, Cout, ksize) sizes a caller-owned, 16-byte aligned device table that
 * lbm_conv1d_split_tables_init fills from the fp32 weights [Cout, Cin, ksize] (0 bytes = unsupported shape).
 * Output channels are processed 256 at a time (Cout is padded up: meant for the wide layers).  Cin * ldx < 2^31. */
size_t lbm_conv1d_split_tables_bytes(int32_t Cin, int32_t Cout, int32_t ksize);
int lbm_conv1d_split_tables_init(void* tables, const float* w, int32_t Cin, int32_t Cout, int32_t ksize, lbm_stream_t stream);
int lbm_conv1d_bf16x6_f32(const float* x, int64_t ldx, const void* tables, const float* bias, float* y, int64_t ldy, int64_t B,
                          int32_t Cin, int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu,
                          lbm_stream_t stream);

/* The 3-tap, stride-1, padding-1 layer (the encoder's third convolution, training/model.py:174) with its Winograd-transformed weights
 * taken from a caller-owned table instead of being rebuilt from w by every workgroup: same arithmetic, same results bit for bit as
 * lbm_conv1d_pitched_f32 on that shape (the per-chunk weight loads become four coalesced 16-byte loads per thread: 3-4 % faster on
 * dense random inputs, within noise on the encoder's own post-ReLU activations).
 * lbm_conv1d_wino3_tables_bytes(Cin, Cout): size of the 16-byte aligned device table (0 = unsupported: Cin % 8 != 0 or Cout < 64);
 * lbm_conv1d_wino3_tables_init fills it from w [Cout, Cin, 3].  x, y, ldx, ldy, bias, relu as for lbm_conv1d_pitched_f32.
 * Round 3: for Cin = 128, Cout a multiple of 64 and 16-byte rows (the encoder's layer) the call runs a weights-stationary persistent
 * kernel (csrc/conv_wino_ws.hip: the table's U values as MFMA operands held in registers, 137 KB of LDS per workgroup, one workgroup per
 * CU): 8-10 % faster, the same results bit for bit.  First call on a device as above (warm up before graph capture). */
size_t lbm_conv1d_wino3_tables_bytes(int32_t Cin, int32_t Cout);
int lbm_conv1d_wino3_tables_init(void* tables, const float* w, int32_t Cin, int32_t Cout, lbm_stream_t stream);
int lbm_conv1d_wino3_f32(const float* x, int64_t ldx, const void* tables, const float* bias, float* y, int64_t ldy, int64_t B,
                         int32_t Cin, int32_t Lin, int32_t Cout, int32_t relu, lbm_stream_t stream);

/* The 4-tap, stride-2, padding-1 layers (the encoder's first two convolutions, training/model.py:169-172) in a form with 2/3 of the
 * products (round 3): y[p] = (w1 x[2p] + w3 x[2p+2]) + (w0 x[2p-1] + w2 x[2p+1]) is two 2-tap stride-1 correlations, one over the even
 * and one over the odd input samples, and each is computed three outputs at a time by the minimal-filtering (Winograd) form F(3,2):
 * 4 products per 3 outputs and phase instead of 6, the transforms exact or one rounding each (csrc/conv_k4s2_wino_ws.hip:
 * weights-stationary persistent kernel, 117 KB of LDS per workgroup, one workgroup per CU).  The results differ from
 * lbm_conv1d_pitched_f32's by rounding only (both within ~1e-6 of a float64 convolution at the encoder's scales) and are NOT
 * bit-identical to it.  Supported (lbm_conv1d_k4s2_wino_supported != 0): 1 <= Cin <= 64, Cout a multiple of 64; rows must be 16-byte
 * pitched (ldx, ldy multiples of 4) and x, w, y 16-byte aligned, otherwise LBM_EUNSUPPORTED (callers then use lbm_conv1d_pitched_f32).
 * w is [Cout, Cin, 4]; x, y, ldx, ldy, bias, relu as for lbm_conv1d_pitched_f32; Lout = (Lin - 2) / 2 + 1.  The first call on a device
 * sets the kernels' dynamic-LDS attribute (warm up before graph capture, as for lbm_conv1d_wino3_f32). */
int lbm_conv1d_k4s2_wino_supported(int32_t Cin, int32_t Cout);
int lbm_conv1d_k4s2_wino_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t B,
                             int32_t Cin, int32_t Lin, int32_t Cout, int32_t relu, lbm_stream_t stream);

/* ---- a12 / a13: codebook distance + argmin ----------------------------------------------------------
 * Replaces VectorQuantizerEMA.get_code_indices, training/model.py:152-158 (same math in forward :90-95):
 * d[n,k] = (sum(x_n^2) + sum(e_k^2)) - 2 * (x_n . e_k) in fp32, idx[n] = first k attaining the minimum.
 * The [N,K] distance matrix is never written to memory.
 * layout LBM_VQ_ROWS: x is [N, D] with row stride ldx (floats);
 * layout LBM_VQ_CHANNEL_MAJOR: x is the encoder output z[B, D, S] (N = B*S, row n = (b = n / S, s = n % S)),
 *   which removes the permute(0,2,1).contiguous() of training/model.py:209; ldx >= S is the pitch of the
 *   [b][d] rows of z (element (n, d) is x[(b*D + d)*ldx + s]).  S is ignored for LBM_VQ_ROWS.
 * codebook: [K, D]; e_sqnorm: [K] = lbm_rownorm2_f32(codebook); idx_out: [N] int64;
 * mind_or_null: [N] minimum distance; hist_or_null: [K] usage counts, ADDED to (caller zeroes).
 * D must be a multiple of 8 and <= 256; K >= 1.
 * layout | LBM_VQ_NO_XNORM: the per-row constant sum(x_n^2) is left out (score[n,k] = e_sqnorm[k] - 2 x_n . e_k;
 *   same argmin in exact arithmetic, mind holds the score).  For a codebook folded through the encoder's last
 *   1x1 convolution: with z = W h + b, d[n,k] - |z_n|^2 = (|e_k|^2 - 2 b.e_k) - 2 h_n.(W^T e_k),
 *   so tokens follow from h directly with codebook' = E W and e_sqnorm' = |e|^2 - 2 E b
 *   (lbm_amd.quantizer.fold_projection); tokens then differ from the unfolded path only at fp32 near-ties. */
#define LBM_VQ_ROWS 0
#define LBM_VQ_CHANNEL_MAJOR 1
#define LBM_VQ_NO_XNORM 2
int lbm_rownorm2_f32(const float* m, float* out, int64_t rows, int32_t D, lbm_stream_t stream);
int lbm_vq_argmin_f32(const float* x, int64_t ldx, int32_t layout, int64_t S, const float* codebook,
                      const float* e_sqnorm, int64_t* idx_out, float* mind_or_null,
                      unsigned long long* hist_or_null, int64_t N, int32_t D, int32_t K, lbm_stream_t stream);

/* Same contract on the bf16 matrix cores: every fp32 operand is cut into three bf16 pieces (exact residuals) and
 * x.e keeps the six piece products down to 2^-18 relative, accumulated in fp32 -- as accurate against float64 as
 * the fp32 fmaf chain above (tools/probe/bf16x_probe.hip) at 16/6 of the fp32 matrix rate.  Indices equal those of
 * lbm_vq_argmin_f32 except at fp32 near-ties (SURVEY.md 8(a) rule).  The codebook is passed pre-split:
 * lbm_vq_split_tables_bytes(K, D) sizes a caller-owned, 16-byte aligned device table that
 * lbm_vq_split_tables_init fills from the fp32 codebook [K, D] (0 bytes = unsupported shape; the table ends with an upper
 * bound of every code's norm, which only the screened search below reads).  1 <= D <= 256, any K >= 1.
 * e_sqnorm as above (lbm_rownorm2_f32 of the fp32 codebook, or the folded constants with LBM_VQ_NO_XNORM). */
size_t lbm_vq_split_tables_bytes(int32_t K, int32_t D);
int lbm_vq_split_tables_init(void* tables, const float* codebook, int32_t K, int32_t D, lbm_stream_t stream);
int lbm_vq_argmin_bf16x6_f32(const float* x, int64_t ldx, int32_t layout, int64_t S, const void* tables,
                             const float* e_sqnorm, int64_t* idx_out, float* mind_or_null,
                             unsigned long long* hist_or_null, int64_t N, int32_t D, int32_t K, lbm_stream_t stream);

/* lbm_vq_argmin_f32's result -- the SAME indices, bit for bit, ties and near-ties included -- at a fraction of its matrix
 * time (VectorQuantizerEMA.get_code_indices, training/model.py:152-158).  Pass 1 screens every (row, code) score with three
 * bf16 piece products on the bf16 matrix cores and a rigorous per-pair bound of its distance from the fp32 fmaf chain's score
 * (csrc/vq_screen.hip derives it); a row whose best code's upper bound lies below every other code's lower bound is settled.
 * Pass 2 re-searches the remaining rows (1-2 % on the shipped codebooks; exact duplicates of a code row always) with
 * lbm_vq_argmin_f32's own kernel code.  tables: lbm_vq_split_tables_init(codebook) (above); codebook / e_sqnorm / layout /
 * hist_or_null as for lbm_vq_argmin_f32 (no minimum-distance output).  workspace: lbm_vq_screen_workspace_bytes(N, K) bytes of
 * caller-owned, 8-byte aligned device memory; after the call its first int32 holds the number of rows pass 2 searched.
 * D a multiple of 8 and <= 256, N < 2^31; with a histogram K <= 32768 (checked before anything is enqueued).
 * The bound's one assumption beyond IEEE arithmetic: one bf16 MFMA accumulate errs by at most 16 x 2^-24 (|C| + sum|a b|), which is
 * gfx950 behaviour AS MEASURED (tools/probe/mfma_acc_probe.hip: <= 7.8 on operands built against the adder), not a specified
 * property -- on other silicon re-run the probe (and bench.py's screen_check compares every token of a step in every run). */
size_t lbm_vq_screen_workspace_bytes(int64_t N, int32_t K);
int lbm_vq_argmin_screened_f32(const float* x, int64_t ldx, int32_t layout, int64_t S, const float* codebook, const void* tables,
                               const float* e_sqnorm, int64_t* idx_out, unsigned long long* hist_or_null, void* workspace,
                               int64_t N, int32_t D, int32_t K, lbm_stream_t stream);

/* VectorQuantizerEMA.forward tail, training/model.py:100,130-139: quantized = E[idx] (returned with the
 * straight-through arithmetic x + (E[idx] - x)), sq_err_sum[0] += sum((E[idx] - x)^2) in float64
 * (caller zeroes; mse = sq_err_sum / (N*D); loss = (1 + commitment_cost) * mse in the forward pass). */
int lbm_vq_gather_loss_f32(const float* x, const float* codebook, const int64_t* idx, float* quantized,
                           double* sq_err_sum, int64_t N, int32_t D, int32_t K, lbm_stream_t stream);

/* ---- EMA codebook training step -------------------------------------------------------
 * Replaces the one-hot matmuls of VectorQuantizerEMA.forward in training mode, training/model.py:103-126.
 * lbm_vq_ema_sums_f32: counts[k] = number of rows with idx == k (= sum(encodings, 0), :109-110) and
 *   dw[k, :] = sum of those rows of x (= encodings^T @ flat_input, :122), as a deterministic segmented sum
 *   (stable counting sort + fixed-order piecewise reduction; no float atomics, bitwise reproducible).
 *   x: [N, D] rows with stride ldx; idx: [N] int64 (entries outside [0, K) are ignored); dw: [K, D];
 *   counts: [K] uint64.  workspace: lbm_vq_ema_workspace_bytes(N, D, K) bytes of 16-byte aligned device
 *   memory (0 = unsupported sizes: K > 8192).  D <= 1024, N < 2^32.  (counts, dw) are what data-parallel
 *   ranks all-reduce before the apply step.
 * lbm_vq_ema_apply_f32 (in place, fp32, the reference's operation order): cluster_size = cluster_size*decay +
 *   (1-decay)*counts; n = sum(cluster_size); cluster_size = (cluster_size + eps)/(n + K*eps)*n;
 *   ema_w = ema_w*decay + (1-decay)*dw; weight = ema_w / cluster_size[:, None]. */
size_t lbm_vq_ema_workspace_bytes(int64_t N, int32_t D, int32_t K);
int lbm_vq_ema_sums_f32(const float* x, int64_t ldx, const int64_t* idx, float* dw, unsigned long long* counts,
                        void* workspace, int64_t N, int32_t D, int32_t K, lbm_stream_t stream);
int lbm_vq_ema_apply_f32(float* ema_cluster_size, float* ema_w, float* weight, const float* dw,
                         const unsigned long long* counts, int32_t K, int32_t D, float decay, float one_minus_decay,
                         float eps, lbm_stream_t stream);

/* Searching a sub-codebook (round 3).  VectorQuantizerEMA.get_code_indices (training/model.py:152-158) takes the FIRST minimum, so a code
 * row that repeats an earlier row bit for bit (together with its e_sqnorm) can never be returned: callers may search the codebook with such
 * rows removed -- the shipped checkpoint has 100 all-zero rows among its 512 -- and translate the indices back.  code_map [Kc] maps a
 * searched row to the caller's code number (ascending, so ties keep resolving to the lowest original index).  In place: idx[i] <-
 * code_map[idx[i]] for 0 <= idx[i] < Kc; with the two histograms, hist[code_map[k]] += hist_c[k] and hist_c[k] <- 0
 * (hist_c: the counts the search accumulated over searched rows); with hist alone (hist_c NULL, Kc <= 12288) the tokens of THIS call are
 * counted into hist while they are remapped -- the search is then called without a histogram and the tokens are read once less.  No counterpart in the reference (it searches all rows);
 * the host side is lbm_amd.quantizer.dedup_codebook. */
int lbm_vq_remap_codes_i64(int64_t* idx, int64_t N, const int64_t* code_map, int32_t Kc, unsigned long long* hist_c_or_null,
                           unsigned long long* hist_or_null, lbm_stream_t stream);

/* training/model.py:142-144: perplexity = exp(-sum(p * log(p + 1e-10))), p = hist / sum(hist). out: [1]. */
int lbm_perplexity_u64(const unsigned long long* hist, float* out, int32_t K, lbm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LBM_ABI_H */
