/*
 * libartspeech_hip.so -- C ABI of the MI355X-native ArtSpeech acoustic-model inference path.
 *
 * The reference (Zhongxu-Wang/ArtSpeech) is pure Python/PyTorch and has no FFI; its "operator API"
 * for this path is the Python call surface listed in SURVEY.md section 8(b).  Each entry point below
 * names the reference interface it replaces (file:line in the reference tree).  INTEGRATION.md
 * shows the ctypes binding a maintainer would add on the reference side.
 *
 * Conventions (all entry points)
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer owned by the caller
 *     unless the parameter name ends in _host.
 *   - returns 0 on success, <0 for an invalid argument (AS_EINVAL = -1), >0 = a hipError_t.
 *   - never allocates, never synchronises, never throws: work is enqueued on `stream`
 *     (a hipStream_t passed as void*), scratch comes from a caller-provided workspace whose size
 *     the matching *_workspace_bytes() query returns.  Safe to capture in a hipGraph.
 *   - thread-safe for concurrent calls that use distinct streams and workspaces.
 */
#ifndef ARTSPEECH_HIP_H
#define ARTSPEECH_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* as_stream_t; /* hipStream_t */

/* library/ABI version; bumped on any change of a signature or of a struct's layout (never held back for anything outside this header:
 * bench.py's source id of the kernel sources leaves version.hip out).  as_abi_version() returns the AS_ABI_VERSION the library was built
 * from: a caller compiled against another header must not go on (artspeech_amd/_lib.py refuses to). */
#define AS_ABI_VERSION 8
int as_abi_version(void);

/* Device-side status.  The reference's operators cannot return silently stale results: nn.Embedding raises on an id >= n_token
 * (RelTransformerEnc.py:11-16), cuDNN's LSTM (models.py:555-561) and the loops of S_monotonic_align.py:5-95 have no
 * cross-workgroup protocol that could time out.  Kernels of this library that can fail at run time raise a sticky bit instead of
 * carrying on unnoticed:
 *   bit AS_STATUS_LSTM_TIMEOUT  a member of a clustered H = 256 recurrence gave up waiting for its peers (as_bilstm_cluster_f32)
 *   bit AS_STATUS_MAS_TIMEOUT   a band of the alignment search gave up waiting for the band above it (as_mas_f32)
 *   bit AS_STATUS_BAD_TOKEN     a token id outside [0, n_token) reached the embedding (it was clamped)
 *   bit AS_STATUS_F16_RANGE     a conv GEMM produced a non-finite accumulator: an operand beyond fp16's range, |x| > 65504, or a
 *                               non-finite input (every launch under the debug probe as_set_range_probe; ALWAYS the last conv of
 *                               as_forward_test / as_decoder_forward, `to_out`: inf / NaN anywhere upstream reaches it -- every
 *                               ReLU of the library keeps a NaN, as torch.relu does -- so a non-finite mel is never handed back
 *                               silently)
 *   bit AS_STATUS_BAD_LAYOUT    an utterance wider than AS_META_MAX_W columns reached as_make_meta (its descriptors are void), an
 *                               utterance wider than the post_max_w its caller named -- or, with one utterance, another column range
 *                               than [0, N) -- reached as_conv_gemm_multi_post_f32's reduction, or (as_lanes_set_debug) the device
 *                               buffers of a submission changed while it was waiting for its group
 *   bit AS_STATUS_CAPACITY      the predicted durations of a batch (or of one submission of a merged call) add up to more frames than
 *                               the capacity its caller named (as_forward_io.frame_cap): what was computed was cut at the capacity --
 *                               nothing was written out of bounds, the mel is void; run it again with more room (the counterpart of
 *                               AS_ENOSPC where no host read-back tells the host in time)
 * as_device_status returns the bits raised on the current HIP device since the last clear (0 = healthy) without synchronising; it is
 * final for work whose stream has been synchronised.  The module-level entry points (as_*_forward, as_forward_test*) return
 * AS_EDEVICE while any bit is set: results computed since it was raised are invalid; clear it to go on. */
#define AS_EDEVICE (-3)
enum { AS_STATUS_LSTM_TIMEOUT = 0, AS_STATUS_MAS_TIMEOUT = 1, AS_STATUS_BAD_TOKEN = 2, AS_STATUS_F16_RANGE = 3, AS_STATUS_BAD_LAYOUT = 4,
       AS_STATUS_CAPACITY = 5, AS_STATUS_KINDS = 6 };
int as_device_status(int clear);
/* test hook: raise `kind` from a kernel on `stream`, exactly as a failing kernel would */
int as_device_status_raise_for_test(int kind, as_stream_t stream);
/* debug switch: activations are not range-scaled (only the weights are, as_prep_weight_f16x2_host), so a value beyond fp16's range
 * (|x| > 65504) enters its operand image as h = inf, l = -inf and turns every product it takes part in into NaN -- in the accumulators
 * of the conv GEMM that consumes the image.  on = 1 makes every as_conv_gemm_f32 launch test its accumulators and raise
 * AS_STATUS_F16_RANGE for a non-finite one (AS_DEBUG=1 in the environment also prints the launch's shape).  Off by default: the
 * test costs 16 compares per accumulator tile. */
int as_set_range_probe(int on);
#define AS_PROBE_THIS 2

/* Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg; no
 * reference counterpart).  Classes: 0 conv-GEMM, 1 AdaIN, 2 LayerNorm, 3 attention, 4 LSTM, 5 MAS, 6 other.
 * as_prof_collect blocks until the recorded events have completed and returns, per class, the summed
 * kernel time (ms), algorithmic flop, algorithmic bytes and launch count since as_prof_enable(1). */
int as_prof_enable(int on);
int as_prof_collect(double* ms, double* flops, double* bytes, int32_t* launches, int n_classes);
/* algorithmic flop / bytes of the NEXT launch of this host thread, for entry points whose arguments do not determine them
 * (their geometry tables are device arrays); ignored while profiling is off */
int as_prof_hint(double flops, double bytes);
/* what one event bracket adds to the kernel inside it (ms): the command processor's work between the two markers, which a kernel
 * trace does not count.  Measured with brackets of 1, 2, 4, 8 empty kernels (the intercept); blocks. */
int as_prof_bracket_overhead(as_stream_t stream, double* overhead_ms);
/* what this device's matrix cores SUSTAIN on random fp16 operands (TFLOP/s of v_mfma_f32_32x32x16_f16): a bare loop of the conv GEMM's
 * own MFMA pattern with the operands in registers, and with them re-read from LDS at the GEMM's ratio (8 ds_read_b128 per 12 MFMAs).
 * The chip lowers its clock under matrix-core load on non-trivial data, so this -- not the data sheet's 2516.6 -- is what any kernel
 * can be compared with on this device.  Allocates and frees ~1.5 MB, blocks ~50 ms. */
int as_prof_mfma_sustained(as_stream_t stream, double* tflops_registers, double* tflops_lds_fed);

/* ---------------------------------------------------------------------------------------------
 * Monotonic alignment search (K1).
 * Replaces: maximum_path1  S_monotonic_align.py:5-47    (tie_mode = 1, "move")
 *           maximum_path2  S_monotonic_align.py:50-95   (tie_mode = 0, "stay")
 *           Triton maximum_path  S_monotonic_align_Triton.py:7-71 (tie_mode = 0)
 *           Cython wrapper  utils.py:11-24 (call surface; arithmetic source not in the tree)
 * value [B][Tx][Ty] fp32 (already masked or not: only the [t_x[b]) x [t_y[b]) corner is read; it is
 * never written).  t_x, t_y int32 [B] = valid text / mel lengths (what the reference recovers from
 * the mask at S_monotonic_align.py:15-16).
 * Outputs (each may be NULL): path fp32 [B][Tx][Ty] dense 0/1 (zero-filled here);
 * dur int32 [B][Tx] = path.sum(-1) (train_second.py:185); rows int32 [B][Ty] = text row of every mel
 * column, -1 past t_y[b].
 * ------------------------------------------------------------------------------------------- */
size_t as_mas_workspace_bytes(int B, int Tx, int Ty);
int as_mas_f32(const float* value, const int32_t* t_x, const int32_t* t_y, int B, int Tx, int Ty,
               int tie_mode, float* path, int32_t* dur, int32_t* rows,
               void* workspace, size_t workspace_bytes, as_stream_t stream);
/* The training scripts' producer of K1 in one call (train_second.py:181-184; train_first.py:171-177 without its masked_fill):
 * attn = softmax(feat, dim = softmax_dim) over the whole [Tx][Ty] slab (2 = the last axis, 1 = the Tx axis), then as_mas_f32 on
 * attn with the LENGTHS (what mask_from_lens + maximum_path do through a dense mask); dur = d_gt (train_second.py:185).
 * attn fp32 [B][Tx][Ty] is an output (s2s_attn is used downstream) and must not alias feat. */
int as_softmax_mas_f32(const float* feat, const int32_t* t_x, const int32_t* t_y, int B, int Tx, int Ty, int softmax_dim,
                       int tie_mode, float* attn, float* path, int32_t* dur, int32_t* rows, void* workspace,
                       size_t workspace_bytes, as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Packed-frames layout (DESIGN.md "Data layout").  An activation is fp32 [C][N] (row stride ld >= N)
 * with every utterance of the batch concatenated along the contiguous column axis, no padding.
 * Column j carries a descriptor  h | H<<10 | w<<20 | W<<42  (AS_META_PACK: 10-bit rows, 22-bit columns): its (row, column)
 * inside its own utterance's H x W image (H = 1, w = frame index for 1-D sequences); H <= AS_META_MAX_H (mel bins),
 * W <= AS_META_MAX_W (the vocoder's last stage has 300 columns per mel frame: 174 s of audio per utterance).
 * as_make_meta builds the descriptors from per-utterance widths (int32 [B], device) for a common
 * height H; col_off int32 [B+1] are the utterances' first columns (exclusive prefix sums of H*W_b).  A width beyond
 * AS_META_MAX_W raises AS_STATUS_BAD_LAYOUT on the device (the widths are device data).
 * ------------------------------------------------------------------------------------------- */
#define AS_META_MAX_H 1023
#define AS_META_MAX_W 4194303
#define AS_META_PACK(h, w, H, W) \
    ((uint64_t)(h) | ((uint64_t)(H) << 10) | ((uint64_t)(w) << 20) | ((uint64_t)(W) << 42))
#define AS_META_h(md) ((int)((md) & 1023u))
#define AS_META_H(md) ((int)(((md) >> 10) & 1023u))
#define AS_META_w(md) ((int)(((md) >> 20) & 4194303u))
#define AS_META_W(md) ((int)((md) >> 42))
int as_make_meta(const int32_t* widths, const int32_t* col_off, int B, int H, int n_cols_max,
                 uint64_t* meta, as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense convolution / linear layer as implicit GEMM on the matrix cores (K3, K4, K8, K10).
 *   Y[m][j] = epi( sum_t sum_k Wt[t][k][m] * X[k][j + dh[t]*W_j + dw[t]] ),   tap valid iff inside
 *   the column's own H x W image (zero padding otherwise).
 * Replaces nn.Conv1d (RelTransformerEnc.py:110-118,257-258,306-314; models.py:176-181,480-495,592-594),
 * nn.Conv2d 3x3/1x1 stride 1 (models.py:71-77,385-399,530-535) and nn.Linear over rows.
 * epi: *acc_scale, +bias[m], +res[m][j], /sqrt(2) (models.py:201, :100, :156), then act.
 *
 * Arithmetic ("f16x3"): every fp32 operand is split into two fp16 numbers, x = h + l (round-to-nearest-even twice: 22
 * significand bits), and a product is accumulated in fp32 from h*l + l*h + h*h on v_mfma_f32_32x32x16_f16 -- the dropped
 * l*l term is below 2^-22 |x w|.  Weights are scaled by a power of two before the split (as_prep_weight_f16x2 chooses it,
 * acc_scale undoes it) so that their l parts stay out of the fp16 subnormal range.  Operands must be finite and below
 * 65504 in magnitude.  n_prod = 1 keeps h*h only (plain fp16 operands: the "16-bit operand" mode of BASELINE.md C2).
 *
 * Operand images ("split images"), both staged global -> LDS by LDS-DMA in exactly this order:
 *   activations Xh [KBx][4][N+1][8] fp16: k-block kb = k/16, plane q = p*2 + kh holds part p (0 = h, 1 = l) of
 *       k = 16 kb + 8 kh + 0..7 for every column; column N is all zeros (where a tap outside the utterance reads);
 *       KBx = ceil(K/16) rounded up to a multiple of 4 (zero blocks).  4 bytes per element: the size of the fp32 tensor.
 *       Written by as_split_f16x2_f32, by the producers that feed only convolutions (as_adain_split_f32,
 *       as_channel_layernorm_split_f32, ...) or by the previous GEMM's epilogue (ConvGemmArgs.Yh).
 *   weights Wh [G][T][KBx][4][M][8] fp16 (as_prep_weight_f16x2), G = weight sets of a grouped launch.
 * ------------------------------------------------------------------------------------------- */
#define AS_MAX_TAPS 25
typedef struct ConvGemmArgs {
    const uint16_t* Wh;    /* split weights [G][T][KBx][4][M][8] fp16 */
    const float* W;        /* optional fp32 [T][Kp][M] image: only the Cin = 1 direct kernel reads it (K == 1 launches) */
    const float* X;        /* fp32 [K][ldx], or NULL when Xh is given */
    const uint16_t* Xh;    /* split activations [KBx][4][N+1][8] fp16 (in_act already applied by its writer), or NULL:
                              the library splits X into ws first (as_conv_gemm_workspace_bytes asks for the room) */
    float* Y;              /* fp32 [M][ldy] output, or NULL when only Yh is wanted */
    uint16_t* Yh;          /* optional: the output ALSO (or only) as the split image [KBy][4][N+1][8] of the convolution
                              that consumes it (KBy from M as KBx from K; rows >= M and column N zero) */
    const float* bias;     /* [G][M] or NULL */
    const float* res;      /* [M][ldr] or NULL (may alias Y) */
    const uint64_t* meta;  /* [N] column descriptors, or NULL = every tap valid */
    void* ws;              /* scratch: split-K partial slabs, then the split activations; NULL = neither */
    size_t ws_bytes;
    int32_t M, N, K, T;
    int32_t Kp;            /* K rounded up to a multiple of 16 (rows of zeros) */
    int32_t ldx, ldy, ldr;
    int32_t act;           /* epilogue activation: 0 none, 1 ReLU, 2 LeakyReLU(act_slope), 3 tanh, 4 |x| (Utils/JDC/model.py:137),
                              5 swish x*sigmoid(x) (Utils/EMA/conformer/conformer/activation.py:29) */
    int32_t div_sqrt2;     /* epilogue: divide by sqrt(2) after bias and residual */
    int32_t in_act;        /* 2 = LeakyReLU(in_slope) applied to X while it is split (models.py:89,142; Vocoder/vocoder.py:38,102);
                              ignored when Xh is given */
    int32_t transpose_out; /* 1 = write Y[j][m] (time-major, row stride ldy >= M) */
    int32_t yh_lrelu;      /* 1 = the image Yh holds LeakyReLU(in_slope)(y) (the consumer's in_act) while Y stays plain */
    int32_t n_prod;        /* 0 or 3 = f16x3 (fp32-accurate); 1 = h*h only */
    int32_t dh[AS_MAX_TAPS];   /* tap row offsets */
    int32_t dw[AS_MAX_TAPS];   /* tap column offsets */
    float in_slope, act_slope; /* LeakyReLU slopes of in_act / yh_lrelu and of act == 2, used exactly as given (the acoustic path's is
                                * AS_SLOPE_PATH = 0.2, models.py:163; nothing is substituted for 0: a slope of 0 IS ReLU); must be finite */
    float acc_scale;           /* multiplies the accumulator: 1 / (weight scale of as_prep_weight_f16x2); 0 = 1 */
    /* Grouped launch: G layers of the same shape side by side along the column axis (the text and articulatory encoders,
     * RelTransformerEnc.py; the F0 / energy / TV branches of ArtsPredictor, models.py:606-618): columns
     * [g * group_cols, (g+1) * group_cols) use weight set g.  n_groups <= 1: off. */
    int32_t n_groups, group_cols;
    int32_t range_probe;       /* AS_PROBE_THIS (2) from the caller: THIS launch tests its accumulators whatever as_set_range_probe says
                                * (as_forward_test's last conv: a non-finite mel always raises AS_STATUS_F16_RANGE); any other value is
                                * overwritten with as_set_range_probe's switch ... */
    uint32_t* status;          /* ... and the device's status words (as_device_status) */
    /* A 1x1 convolution of a SECOND operand summed into the same accumulators before the epilogue -- a residual block's learned
     * shortcut (models.py:79-84,185-186: conv1x1, no bias) evaluated by the launch of the block's last conv, as K2 more channels
     * of the reduction: Y = epi(sum_t W_t X(t) + W2 X2).  Xh2: split image [KBx2][4][N+1][8] of X2 (same N and column order as
     * Xh); its weights follow the T taps inside every weight set (as_prep_weight_f16x2_sc_host).  K2 = 0: none.  Needs Xh. */
    const uint16_t* Xh2;
    int32_t K2;
    /* Strided / valid convolutions (the 5x5 valid convs that close the 2-D towers, models.py:391,399,535): the OUTPUT columns are their
     * own layout (N of them) and output column j reads the input image at column src_col[j] + dh * W_in + dw, where meta[j] now
     * describes that INPUT position (h, w of tap (0, 0), H and W of the input image: taps outside it read zero) and Xh has N_in columns
     * (+ its zero column).  src_col NULL: off (the input is laid out like the output).  Needs Xh and meta; not with Xh2. */
    const int32_t* src_col;
    int32_t N_in;
    /* ileave_u >= 2: the M = ileave_u * C output rows are (phase r, channel m) -- row r C + m -- and Y is [C][ldy] with
     * Y[m][ileave_u * j + r] = result[r C + m][j]: a ConvTranspose1d(k = 2u, stride u) written as ONE 3-tap conv leaves in time order
     * (the interleave pass of as_interleave_phases_f32 folded into the store; the bias is given per ROW: the channel's, u times).
     * C a multiple of 32, ldy >= ileave_u * N; Y only (no Yh, res, transpose_out, weight groups); never K-sliced. */
    int32_t ileave_u;
    int32_t slab_tr;           /* library-owned (overwritten): the K slices' partial sums are stored time-major for a reduction that also
                                * computes the channel LayerNorm behind the conv (as_conv_gemm_multi_post_f32) */
    /* Capacity layouts (as_forward_io.frame_cap: the column count of a launch is a capacity, how many of the columns hold utterances is
     * known on the device only): *n_valid (a DEVICE int) = the leading columns that are valid -- of every weight group's column range when
     * the launch is grouped.  Columns behind them are filler: they read the zero column, nothing is stored for them, and a tile that lies
     * wholly in the filler ends at once, so a launch costs what its valid columns cost.  NULL: every column is valid. */
    const int32_t* n_valid;
} ConvGemmArgs;
#define AS_SLOPE_PATH 0.2f
int as_conv_gemm_f32(const ConvGemmArgs* args_host, as_stream_t stream);
/* Several INDEPENDENT convolutions (no problem reads what another writes) as ONE launch: the grid walks the tiles of all of them, longest
 * tiles first, so a problem that would own the chip alone at a fraction of its width -- the 40-tile convs of the duration predictor beside
 * the encoders' 240-tile ones, the small towers beside the mel tower: branches of ArtsSpeech.forward with no edge between them,
 * models.py:356-360, 417-424, 540-546 -- costs what it adds to the busiest CU, not a launch of its own.  Every problem as in
 * as_conv_gemm_f32 (own epilogue, groups, second operand, source positions), with these limits: operand images only (Xh given) and one
 * n_prod -- or a set made ONLY of Cin = 1 convs of <= 9 taps (the towers' stems: the direct kernel, one launch for the set); 1 <= n <=
 * AS_MAX_MULTI.  One tile shape serves the set (the cost model of the single launch on the
 * summed tile count: as_conv_gemm_multi_tile says which), so a problem's result can differ in the last bits from its single launch where
 * that one would have split K inside the workgroup -- same arithmetic, other order of the partial sums.  n == 1 is as_conv_gemm_f32. */
#define AS_MAX_MULTI 6
int as_conv_gemm_multi_f32(const ConvGemmArgs* list_host, int n, as_stream_t stream);
int as_conv_gemm_multi_tile(const ConvGemmArgs* list_host, int n);
/* workspace (ConvGemmArgs.ws) a problem wants when it may go into a multi-problem launch: a long reduction on a few tiles beside
 * shorter ones is cut into K slices there (as many as ws_bytes holds fp32 [M][N] slabs for) so that it does not outlast the launch;
 * >= as_conv_gemm_workspace_bytes */
size_t as_conv_gemm_multi_workspace_bytes(const ConvGemmArgs* args_host);
/* which kernel as_conv_gemm_f32 runs for these arguments (tests, tuning): *kind 0 = the direct Cin = 1 kernel, 1 = the tiled kernel
 * (*tile = 22 / 21 / 12 / 11 / 14 / 2: 128x128, 128x64, 64x128, 64x64, 64x256, 32x128); *slices = K slices */
int as_conv_gemm_plan(const ConvGemmArgs* args_host, int32_t* kind, int32_t* tile, int32_t* slices);
/* Bytes of workspace this shape wants (0 = none): split-K slabs for shapes whose tile grid cannot fill the 256 CUs (a second
 * kernel sums the slabs in a fixed order: deterministic), then the split image of X when Xh is NULL. */
size_t as_conv_gemm_workspace_bytes(const ConvGemmArgs* args_host);
/* X fp32 [K][ldx] (N columns) -> Xh (layout above).  in_act 2 applies LeakyReLU(in_slope) first (the slope as given).  One image can
 * feed every conv reading the same activations.  xh: 16-byte aligned, as_split_f16x2_bytes(K, N) bytes. */
size_t as_split_f16x2_bytes(int K, int N);
int as_split_f16x2_f32(const float* x, int ldx, int K, int N, int in_act, float in_slope, uint16_t* xh, as_stream_t stream);
/* Host-side weight preparation (no GPU work): w fp32 [G][Cout][Cin][T] (a folded nn.Conv / nn.Linear weight; T = product of
 * the kernel dims) -> wh [G][T][KBx][4][Cout][8] fp16 in HOST memory (as_prep_weight_f16x2_bytes(G, Cout, Cin, T) bytes),
 * scaled by *scale_out = the power of two that puts max |w| in [2^13, 2^14).  Pass ConvGemmArgs.acc_scale = 1 / *scale_out. */
size_t as_prep_weight_f16x2_bytes(int G, int Cout, int Cin, int T);
int as_prep_weight_f16x2_host(const float* w_host, int G, int Cout, int Cin, int T, uint16_t* wh_host, float* scale_out);
/* The same with the weights w2 fp32 [G][Cout][Cin2] of a 1x1 convolution on a second operand (ConvGemmArgs.Xh2 / K2) appended to every
 * weight set: wh [G][T * KBx + KBx2][4][Cout][8], one common scale.  Cin2 = 0 (w2 NULL): exactly the functions above. */
size_t as_prep_weight_f16x2_sc_bytes(int G, int Cout, int Cin, int T, int Cin2);
int as_prep_weight_f16x2_sc_host(const float* w_host, const float* w2_host, int G, int Cout, int Cin, int T, int Cin2, uint16_t* wh_host,
                                 float* scale_out);

/* ---------------------------------------------------------------------------------------------
 * Bandwidth-bound kernels on packed frames.  col_off int32 [B+1] = first column of each utterance.
 * ------------------------------------------------------------------------------------------- */
/* emb(x)*sqrt(C), transposed to [C][N]      RelTransformerEnc.py:373-374 */
int as_embed_f32(const int32_t* tokens, const float* emb, int C, int N, int V, float scale, float* y, int ldy,
                 as_stream_t stream);
/* the *_groups_* variants: column group g = column / n_split (utterance group g = utterance / b_split) takes parameter set
 * first + g * (second - first): two separately stored sets for two groups, or any number of equally spaced ones (a stack) -- encoders of the same
 * shape run as one double-width launch (ConvGemmArgs.n_groups is the GEMM's counterpart); NULL second set = the plain call */
/* as_embed_groups_f32: tokens holds n_tok ids.  n_tok == N: one id per column.  n_tok < N (the same tokens through every table):
 * column g * n_split + j reads tokens[j]; columns past n_tok inside a group are filler (id 0). */
int as_embed_groups_f32(const int32_t* tokens, int n_tok, const float* emb, const float* emb2, int n_split, int C, int N, int V,
                        float scale, float* y, int ldy, as_stream_t stream);
int as_channel_layernorm_groups_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta, const float* gamma2,
                                    const float* beta2, int n_split, float eps, int relu, float* y, int ldy, as_stream_t stream);
/* Channel LayerNorm (+ReLU) written as the split operand image of the conv that follows (as_split_f16x2_f32's layout; pass it as
 * ConvGemmArgs.Xh): in the encoders a LayerNorm's output feeds nothing but that conv.  C <= 1024; second affine pair as above. */
int as_channel_layernorm_split_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta, const float* gamma2,
                                   const float* beta2, int n_split, float eps, int relu, uint16_t* xs, as_stream_t stream);
int as_relpos_attention_groups_f32(const float* qkv, int ld, int C, int heads, int window, const float* emb_rel_k,
                                   const float* emb_rel_v, const float* emb_rel_k2, const float* emb_rel_v2, int b_split,
                                   const int32_t* col_off, int B, int max_len, float* out, int ldo, as_stream_t stream);
/* The same attention fed by the q/k/v projection's operand image (as_conv_gemm_f32 with Y and Yh: qkv fp32 [3C][ld] AND qkv_h, its image
 * over n_total columns): Q / K fragments come straight from the image, only V is read as fp32.  128-channel heads.  Writes out fp32
 * [C][ldo] and / or out_h, the operand image of the o-projection (n_total columns). */
int as_relpos_attention_image_f32(const float* qkv, int ld, const uint16_t* qkv_h, int n_total, int C, int heads, int window,
                                  const float* emb_rel_k, const float* emb_rel_v, const float* emb_rel_k2, const float* emb_rel_v2,
                                  int b_split, const int32_t* col_off, int B, int max_len, float* out, int ldo, uint16_t* out_h,
                                  as_stream_t stream);
/* channel LayerNorm (eps 1e-4) (+ReLU)       RelTransformerEnc.py:272-290, :322-323 */
int as_channel_layernorm_f32(const float* x, int ldx, int C, int N, const float* gamma, const float* beta, float eps,
                             int relu, float* y, int ldy, as_stream_t stream);
/* AdaIN1d + LeakyReLU, per-utterance instance-norm statistics (biased var, eps 1e-5); gamma_beta [B][ldgb]
 * = fc(style) (gamma first).  With pool_w/pool_b [C][3]/[C] the depthwise ConvTranspose1d(k3,s2,p1,op1)
 * is fused: y gets 2x the frames (utterance b starts at 2*col_off[b]) and x_up (optional) the nearest-x2
 * copy of x.                                  models.py:189-197, 230-240, 172, 184, 261-270 */
int as_adain_f32(const float* x, int ldx, int C, const float* gamma_beta, int ldgb, const int32_t* col_off, int B,
                 float* y, int ldy, int lrelu, const float* pool_w, const float* pool_b, float* x_up, int ld_up,
                 as_stream_t stream);
/* The same AdaIN1d + LeakyReLU(0.2) written as the split operand image of the conv that follows (as_split_f16x2_f32's
 * layout, N = total columns; pass it as ConvGemmArgs.Xh): the fp32 activations are never stored. */
int as_adain_split_f32(const float* x, int ldx, int C, const float* gamma_beta, int ldgb, const int32_t* col_off, int B, int N,
                       int lrelu, uint16_t* xs, as_stream_t stream);
/* General form of the above (what the module-level entry points launch): per-utterance addressing of gamma / beta and of the
 * input columns, so that several layers of the same shape run as ONE launch on a grouped layout (the F0 / energy / TV branches of
 * ArtsPredictor, models.py:606-618) and read gamma / beta straight from a GEMM's [rows][utterances] output; optional fused
 * depthwise ConvTranspose1d x2 up-sampler (models.py:172,195) with the nearest-x2 copy of x as second output (models.py:184). */
typedef struct AsAdainArgs {
    const float* x;          /* [C][ldx] */
    int32_t ldx, C;
    const float* gb;         /* gamma(u, c) = gb[gb_off[u] + c * gb_sc], beta(u, c) = gb[gb_off[u] + (C + c) * gb_sc] */
    const int32_t* gb_off;   /* [U] float offsets; NULL = u * ldgb */
    int32_t ldgb, gb_sc;
    const int32_t* col_off;  /* [U + 1] OUTPUT columns of utterance u (times 2 with the up-sampler) */
    const int32_t* src_off;  /* [U] first INPUT column of utterance u; NULL = col_off[u] */
    int32_t U, N;            /* utterances; total output columns of the image (its zero column) */
    int32_t lrelu;
    uint16_t* yh;            /* the split image [KBx(C)][4][N+1][8] */
    const float* pool_w;     /* [C][3] depthwise ConvTranspose1d weights, or NULL = no up-sampling */
    const float* pool_b;     /* [C] */
    float* x_up;             /* optional [C][ld_up]: nearest x2 copy of x */
    int32_t ld_up;
    const int32_t* col_w;    /* optional [U]: utterance u has col_w[u] INPUT columns (times 2 out with the up-sampler) instead of col_off[u + 1] -
                              * col_off[u] -- layouts whose utterances do not lie back to back (capacity layouts: filler between groups) */
} AsAdainArgs;
int as_adain_image_f32(const AsAdainArgs* args_host, as_stream_t stream);
/* as_conv_gemm_multi_f32 for convolutions whose RESULT is (also) read through an AdaIN1d + LeakyReLU -- conv1 -> norm2 -> actv -> conv2
 * inside an AdainResBlk1d, conv2 -> the next block's norm1 (models.py:189-202) -- : problem i with post_host[i].yh != NULL also leaves
 * the operand image post[i].yh = split(LeakyReLU?(AdaIN(y_i))) over the utterances post[i].col_off [post[i].U + 1] of its N columns,
 * y_i = the conv's result after its own epilogue (post[i].x / ldx / C / N / src_off / pool_* are ignored: the input IS the conv's result,
 * list[i].Y must be given, C = M; gb / gb_off / ldgb / gb_sc / lrelu as in as_adain_image_f32).  What it buys: a launch that is cut into
 * K slices (few columns: batch 1, BASELINE config C2) ends in a reduction kernel that already holds a channel's whole time axis when no
 * utterance is wider than 256 columns (post_max_w[i] = the widest utterance of problem i), so that kernel computes the statistics and
 * writes the image itself -- conv -> reduce+AdaIN -> conv instead of conv -> reduce -> AdaIN -> conv, bit-identical to the separate
 * launches.  Everywhere else the call is exactly as_conv_gemm_multi_f32 followed by as_adain_image_f32 on list[i].Y.
 * post_host NULL: as_conv_gemm_multi_f32.  Not with transpose_out / ileave_u.  With ONE utterance (post[i].U == 1) the fused reduction
 * takes the utterance to be columns [0, N) without waiting for col_off; a col_off that says otherwise raises AS_STATUS_BAD_LAYOUT (the
 * two-launch form honours it: the two never differ silently).
 * The same for a conv whose result is read through a channel LayerNorm (+ ReLU) -- the encoders' conv -> residual add -> LayerNorm -> conv
 * (RelTransformerEnc.py:72-87, 318-325): problem i with post_ln_host[i].yh != NULL also leaves the operand image
 * yh = split(ReLU?(LayerNorm(y_i))) (as_channel_layernorm_split_f32's arithmetic and parameter addressing: column group g = column /
 * n_split takes gamma + g (gamma2 - gamma), NULL second set = one set).  K-sliced launches of <= 512 output channels (a multiple of 8)
 * store their partial sums time-major and the reduction kernel -- a wave per column -- writes the image itself; everywhere else the call
 * is the conv followed by as_channel_layernorm_split_f32 on list[i].Y.  post_ln_host NULL: none.  A problem has at most one of the two.
 * So that the fused form is the rule at batch 1, a conv with either normalisation behind it whose output is at most 1 MB is cut into two
 * K slices even where the slicing rules would leave it whole (as_conv_gemm_workspace_bytes / _multi_workspace_bytes leave room for the two
 * slabs of any such output): another order of two partial sums, the same arithmetic. */
typedef struct AsLnArgs {
    const float *gamma, *beta, *gamma2, *beta2;   /* [C] per set */
    int32_t n_split;
    float eps;
    int32_t relu;
    uint16_t* yh;                                  /* the split image [KBx(M)][4][N+1][8], or NULL: no LayerNorm behind this conv */
} AsLnArgs;
int as_conv_gemm_multi_post_f32(const ConvGemmArgs* list_host, const AsAdainArgs* post_host, const int32_t* post_max_w,
                                const AsLnArgs* post_ln_host, int n, as_stream_t stream);
/* x [B][ldx] (one K-vector per utterance) -> the split image of its transpose [K][B] (columns = utterances): the operand of
 * the GEMM that evaluates every AdaIN fc layer of the model at once (models.py:237). */
int as_rows_image_f32(const float* x, int ldx, int K, int B, uint16_t* xh, as_stream_t stream);
/* Y[m][j] = bias[m] + sum_k W[m][k] X[k][j] for a handful of output rows (M <= 16): the F0 / energy / TV projections behind the
 * BiLSTMs and duration_proj (models.py:565,619-621) -- bandwidth-bound, one column per thread. */
int as_project_cols_f32(const float* x, int ldx, int K, int N, const float* w, const float* bias, int M, float* y, int ldy,
                        as_stream_t stream);
/* Y[m][j] = bias[m] + sum_{k < K} W[m][k] X[k][j] for K <= 16 input channels (decoder.F0_conv / N_conv / EMA_conv as one block-diagonal
 * 12 -> 128 1x1 conv, models.py:480-482,503-505), written to up to two destinations, each as fp32 rows (y1 / y2, or NULL) and / or as the
 * rows of an operand image over N columns (yh1 / yh2: where the image's -- or a k-block-aligned part's -- first row group lives; rows
 * past M up to the image's 64-row block and the zero column N are written as zeros).  w fp32 [M][K]. */
int as_pointwise_small_f32(const float* x, int ldx, int K, int N, const float* w, const float* bias, int M, float* y1, int ld1,
                           uint16_t* yh1, float* y2, int ld2, uint16_t* yh2, as_stream_t stream);
/* y[b][m] = bias[m] + W[m][:] . x[b][:]       nn.Linear on per-utterance vectors (models.py:237,412-415,538) */
int as_linear_rows_f32(const float* x, int ldx, const float* w, const float* bias, int B, int M, int K, float* y,
                       int ldy, as_stream_t stream);
/* round-half-even + clamp(min=1) (or forced integer durations), per-utterance frame offsets [B+1] and the
 * frame -> token map; then the gather that replaces `T_en @ pred_aln_trg`.    models.py:361-368, :500 */
int as_durations_f32(const float* dur_f32, const int32_t* forced_dur, const int32_t* tok_off, int B, int32_t* dur_i32,
                     int32_t* frame_off, int32_t* tok_of_frame, int max_frames, as_stream_t stream);
int as_expand_f32(const float* x, int ldx, int C, const int32_t* tok_of_frame, int n_frames, int repeat, float* y,
                  int ldy, as_stream_t stream);
/* log_norm + stats.json normalisation; feat rows 0 = energy, 1 = f0, 2..11 = EMA.   models.py:431,447-449,655-660
 * stats24 = {energy_mean, energy_std, pitch_mean, pitch_std, EMA_mean[10], EMA_std[10]} */
int as_ref_features_f32(const float* mel, int ldm, int n_mels, const float* f0_raw, const float* ema_raw, int lde,
                        int N, const float* stats24, float* feat, int ldf, as_stream_t stream);
/* per-utterance column window copy (the T-1 crop, models.py:459-471) */
int as_crop_f32(const float* src, int lds, const int32_t* src_off, int start, float* dst, int ldd,
                const int32_t* dst_off, int B, int C, int max_len, as_stream_t stream);
/* H channel rows -> per-utterance H x W images packed [1][sum H*W_b] (img_off = the image layout's column offsets):
 * dst[img_off[b] + h*W_b + i] = src[h][src_off[b] + start + i]       the 2-D towers' inputs, models.py:419-421 */
int as_rows_to_images_f32(const float* src, int lds, const int32_t* src_off, int start, int H, float* dst,
                          const int32_t* img_off, int B, int max_w, as_stream_t stream);
/* style-tower helpers: LearnedDownSample / ResBlk1d.pool (models.py:27-31,116), DownSample (+ residual merge,
 * models.py:43-57,99-100,127-130), im2col for the valid KxK convs (models.py:391,399,535), LeakyReLU+avg-pool */
int as_dwconv_down_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* y, int ldy,
                       const int32_t* out_off, const int32_t* out_w, int Hout, const float* w, const float* bias,
                       int kh, int B, int C, int max_out, int lrelu, as_stream_t stream);
/* JDCNet's BatchNorm2d (eval: y = x*scale[c] + shift[c]) -> LeakyReLU(slope) -> MaxPool2d over the mel axis by k, floor
 * mode (Utils/JDC/model.py:29-34,166-170).  Images are [H rows = mel bins][W = frames of the utterance]; tok_off int32
 * [B+1] = first frame of each utterance in a 1-D layout, image b starts at column H*tok_off[b].  Output: images of
 * Hout = H / k rows (to_channels = 0), or the 1-D layout with C*Hout rows, row c*Hout + h (to_channels = 1: the
 * permute(0,2,1,3).view(.., 512) of model.py:126). */
int as_bn_lrelu_maxpool_rows_f32(const float* x, int ldx, const int32_t* tok_off, int B, int C, int H, int k, const float* scale,
                                 const float* shift, float slope, float* y, int ldy, int to_channels, int total_frames,
                                 as_stream_t stream);
int as_avgpool_down_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* y, int ldy,
                        const int32_t* out_off, const int32_t* out_w, int Hout, int pool_h, const float* res, int ldr,
                        int B, int C, int max_out, as_stream_t stream);
int as_im2col_valid_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* col,
                        int ldc, const int32_t* out_off, const int32_t* out_w, int Hout, int K, int stride, int lrelu,
                        int B, int C, int max_out, as_stream_t stream);
/* The producers above writing the split operand image of the conv that consumes them instead of (or, avgpool: beside) the fp32
 * tensor: yh [KBx(C)][4][n_out + 1][8] over the n_out packed output columns (as_split_f16x2_f32's layout).  yh_lrelu: the image
 * holds LeakyReLU(0.2)(y) -- the consumer's input activation (models.py:89,142) -- while y (optional) stays plain. */
int as_dwconv_down_image_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, const int32_t* out_off,
                             const int32_t* out_w, int Hout, const float* w, const float* bias, int kh, int B, int C, int max_out,
                             int lrelu, uint16_t* yh, int n_out, as_stream_t stream);
int as_avgpool_down_image_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, int Hin, float* y, int ldy,
                              const int32_t* out_off, const int32_t* out_w, int Hout, int pool_h, const float* res, int ldr, int B,
                              int C, int max_out, uint16_t* yh, int n_out, int yh_lrelu, as_stream_t stream);
/* avgpool_down(stem(x)) for a tower's Cin = 1 stem conv (models.py:385,393 followed by the first block's shortcut, :79-84), from the
 * ONE-channel input x [sum H*W_b]: w = the stem's fp32 weight image [T][Kp][C] (row k = 0; T = 3 kh taps in taps_2d(3, 3) / taps_1d(3)
 * order), bias [C] or NULL, zero padding; (pool_h x 2) average with the last column replicated for odd widths; result as the operand
 * image yh over the n_out pooled columns.  The stem's own fp32 output is then never needed. */
int as_stem_pool_image_f32(const float* x, const int32_t* in_off, const int32_t* in_w, int Hin, const int32_t* out_off,
                           const int32_t* out_w, int Hout, int pool_h, const float* w, int Kp, const float* bias, int kh, int B, int C,
                           int max_out, uint16_t* yh, int n_out, as_stream_t stream);
/* One launch for up to AS_MAX_MULTI of the down-sampling steps above (independent problems: the style towers and dur_block march through
 * their ResBlks in step, models.py:385-411,530-535).  kind 0 = as_dwconv_down[_image]_f32 (w [C][kh*3], bias [C], lrelu = LeakyReLU on the
 * result), 1 = as_avgpool_down[_image]_f32 (res / ldr optional, lrelu = the image holds LeakyReLU(y)), 2 = as_stem_pool_image_f32 (w = the
 * stem's fp32 image [T][Kp][C], bias or NULL).  y (kinds 0, 1) and yh as there; the single entry points are this with n = 1. */
typedef struct AsDownArgs {
    int32_t kind;
    const float* x; int32_t ldx;
    const int32_t* in_off; const int32_t* in_w; int32_t Hin;
    float* y; int32_t ldy;
    const int32_t* out_off; const int32_t* out_w; int32_t Hout;
    const float* w; const float* bias;
    int32_t kh, pool_h, Kp;
    const float* res; int32_t ldr;
    int32_t B, C, max_out, lrelu;
    uint16_t* yh; int32_t n_out;
} AsDownArgs;
int as_down_multi_f32(const AsDownArgs* list_host, int n, as_stream_t stream);
int as_im2col_valid_image_f32(const float* x, int ldx, const int32_t* in_off, const int32_t* in_w, const int32_t* out_off,
                              const int32_t* out_w, int K, int stride, int lrelu, int B, int C, uint16_t* yh, as_stream_t stream);
int as_mean_pool_f32(const float* x, int ldx, const int32_t* col_off, int B, int C, int lrelu, float* y, int ldy,
                     as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * wav -> normalised log-mel front end, test.py:40-47: torchaudio MelSpectrogram(n_mels 80, n_fft 2048,
 * win 1200, hop 300) = framing (centre, reflect padding) -> windowed DFT (a conv GEMM with the basis as weights) -> power ->
 * mel filterbank (a conv GEMM) -> (log(1e-5 + mel) + 4) / 4.
 * as_frame_signal_f32: wave = utterances back to back (wav_off int32 [B+1]); X [n_fft][N] frames, utterance b's at columns
 *   frame_off[b] .. frame_off[b+1] (1 + L_b / hop of them).  as_spec_power_f32: Y [2F][N] (real rows, then imaginary) -> P [F][N].
 * ------------------------------------------------------------------------------------------- */
int as_frame_signal_f32(const float* wave, const int32_t* wav_off, const int32_t* frame_off, int B, int max_frames, int n_fft, int hop,
                        float* X, int ldx, as_stream_t stream);
int as_spec_power_f32(const float* Y, int ldy, int F, int N, float* P, int ldp, as_stream_t stream);
int as_log_norm_f32(const float* x, int ldx, int C, int N, float eps, float mean, float std, float* y, int ldy, as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * EMA_Predictor: what is not a GEMM in Utils/EMA/EMA_Predictor.py and its conformer blocks.
 * as_xl_attention_f32: RelativeMultiHeadAttention.forward (conformer/attention.py:77-109) on a fused [3C][N] q/k/v
 *   projection and pos [C][N] = pos_proj(PE[frame index]) per utterance; u_bias, v_bias [heads][64]; the reference's
 *   _relative_shift (:111-119) is reproduced entry for entry; inv_scale = 1/sqrt(d_model).  d_head must be 64.
 * as_glu_dwconv_bn_swish_f32: GLU -> depthwise conv1d (k odd <= 63, zero 'same' padding inside the utterance) -> BatchNorm
 *   (eval, scale/shift) -> Swish (conformer/convolution.py:140-144); a [2C][N] -> y [C][N]; w [C][k].
 * as_lstm_step0_f32: one LSTM step from the zero state for every column, both directions (EMA_Predictor.py:43,79: an
 *   nn.LSTM without batch_first fed [1, T, 256] treats the T frames as a batch of length-1 sequences); gx [8H][N] -> h [2H][N].
 * ------------------------------------------------------------------------------------------- */
int as_xl_attention_f32(const float* qkv, int ld, int C, int heads, const float* pos, int ldp, const float* u_bias,
                        const float* v_bias, float inv_scale, const int32_t* col_off, int B, int max_len, float* out, int ldo,
                        as_stream_t stream);
/* as_xl_attention_image_f32: the same attention on the matrix cores (csrc/xl_attention.hip), fed by the projection GEMMs' operand images.
 *   qkv fp32 [4C][N] (ld) and qkv_h its image (as_conv_gemm_f32 with Y and Yh): rows q + u_bias, q + v_bias, k, v -- the query weights
 *   stacked twice with the biases folded in; pos_h = the image of pos [C][N]; n_total = N (the images' column count); out fp32 [C][N]
 *   and / or out_h = the result as the operand image of the out-projection GEMM (as_split_f16x2_bytes(C, N) bytes).
 *   f16x3 products (fp32-accurate); agrees with as_xl_attention_f32 to fp32 rounding. */
int as_xl_attention_image_f32(const float* qkv, int ld, const uint16_t* qkv_h, const uint16_t* pos_h, int n_total, int C, int heads,
                              float inv_scale, const int32_t* col_off, int B, int max_len, float* out, int ldo, uint16_t* out_h,
                              as_stream_t stream);
int as_glu_dwconv_bn_swish_f32(const float* a, int lda, int C, const float* w, int k, const float* scale, const float* shift,
                               const int32_t* col_off, int B, float* y, int ldy, as_stream_t stream);
int as_lstm_step0_f32(const float* gx, int ldg, int H, int N, float* h, int ldh, as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Windowed relative-position attention (K5) on a fused [3C][N] q/k/v projection.
 * Replaces MultiHeadAttention.attention and its skew helpers, RelTransformerEnc.py:138-233.
 * emb_rel_k / emb_rel_v: [2*window+1][C/heads] (heads share them, :113-117).
 * ------------------------------------------------------------------------------------------- */
int as_relpos_attention_f32(const float* qkv, int ld, int C, int heads, int window, const float* emb_rel_k,
                            const float* emb_rel_v, const int32_t* col_off, int B, int max_len, float* out, int ldo,
                            as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BiLSTM recurrence (K9).  gx_tm [N][ldg >= 8H] time-major gate pre-activations W_ih x + b_ih + b_hh
 * (forward gates 0..4H-1, reverse 4H..8H-1), whh_t [2][H][4H] = W_hh transposed; out [2H][N].
 * Replaces nn.LSTM at models.py:526,555-561 and :589-591,606-618.
 * Up to AS_MAX_LSTM_JOBS independent LSTMs of the same H over the same layout (the F0 / energy / TV
 * branches, models.py:606-618) go in one launch so their sequential recurrences overlap.
 * ------------------------------------------------------------------------------------------- */
#define AS_MAX_LSTM_JOBS 4
typedef struct BiLstmJob {
    const float* gx_tm;    /* [N][ldg] */
    const float* whh_t;    /* [2][H][4H] */
    float* out;            /* [2H][ldo] */
    int32_t ldg, ldo;
} BiLstmJob;
int as_bilstm_f32(const BiLstmJob* jobs_host, int n_jobs, const int32_t* col_off, int B, int H, as_stream_t stream);
/* H = 256: the recurrence of an utterance (or a pair) split over a cluster of four workgroups that keep W_hh in registers and exchange h
 * every step through `xchg` -- a device buffer of as_bilstm_cluster_bytes(n_jobs, B) bytes that the caller zero-fills ONCE and then leaves
 * to the library (it carries launch epochs; launches that may overlap in time need separate buffers).  max_len = longest utterance.
 * Falls back to as_bilstm_f32 when the grid would not fit the chip at once, for other H, or with xchg == NULL. */
/* A member's poll is bounded (~1 s); one that gives up raises AS_STATUS_LSTM_TIMEOUT (as_device_status above): the output is then void. */
size_t as_bilstm_cluster_bytes(int n_jobs, int B);
int as_bilstm_cluster_f32(const BiLstmJob* jobs_host, int n_jobs, const int32_t* col_off, int B, int H, int max_len, void* xchg,
                          size_t xchg_bytes, as_stream_t stream);

/* test hooks: member `drop_member` (0..3; -1 = none) of every cluster exits at once and a poll gives up after spin_limit tries
 * (0 = default), so that the failure path can be exercised in milliseconds */
int as_bilstm_cluster_test_hooks(int drop_member, int spin_limit);

/* ---------------------------------------------------------------------------------------------
 * HiFi-GAN generator glue (Vocoder/vocoder.py:75-125).  Its convolutions run through
 * as_conv_gemm_f32 (dilated taps, LeakyReLU slopes 0.1 / 0.01, tanh epilogue).
 * ------------------------------------------------------------------------------------------- */
/* ConvTranspose1d(k = 2u, stride u) = one 3-tap conv with output rows (phase r, channel m), then
 * y[m][u*q + r] = z[r*C + m][q] + bias[m]                                     vocoder.py:84-89,102-103 */
int as_interleave_phases_f32(const float* z, int ldz, const float* bias, int C, int u, int Nin, float* y, int ldy,
                             as_stream_t stream);
/* y = (a + b + c) / 3 over [C][N]                                              vocoder.py:104-110 */
int as_mean3_f32(const float* a, const float* b, const float* c, int ld, int C, int N, float* y, int ldy, as_stream_t stream);
/* LeakyReLU((a + b + c) / 3, slope) as the split operand image (as_split_f16x2_bytes(C, N) bytes) of the conv that follows -- the next
 * stage's ConvTranspose1d (vocoder.py:101-110): the fp32 mean is read by nothing else. */
int as_mean3_image_f32(const float* a, const float* b, const float* c, int ld, int C, int N, float slope, uint16_t* xh, as_stream_t stream);
/* conv_post (vocoder.py:97, 111-113): y [N] = act(conv1d(LeakyReLU(x [C][N], in_slope), w fp32 [C][k]) + bias[0]) with ONE output channel,
 * zero padding per utterance (meta = the layout's column descriptors), act = tanh when tanh_out; k = 3, 5 or 7.  Plain fp32 FMAs: a read of x. */
int as_conv_post_f32(const float* x, int ldx, int C, int N, const float* w, const float* bias, int k, float in_slope, int tanh_out,
                     const uint64_t* meta, float* y, as_stream_t stream);
/* One residual step of ResBlock1 (vocoder.py:35-42) as ONE launch, for the stages with C = 32 or 64 channels:
 *     y = x + conv2(lrelu(conv1(lrelu(x))))        conv1: k taps with dilation dil, conv2: k taps with dilation 1, zero padding per utterance
 * x, y fp32 [C][N] (y != x: a workgroup reads its neighbours' columns of x); w1, w2 = the conv GEMM's weight images of the two
 * [C][C][k] weights (as_prep_weight_f16x2_host, G = 1), scale1 / scale2 = 1 / their scales, b1 / b2 [C] or NULL; slope = LeakyReLU's;
 * col_off int32 [B + 1] = the utterances' first columns (packed frames), max_w = the widest utterance; k odd <= 17, dil * (k / 2) <= 40.
 * add1 / add2 (both or neither; [C][N], ld_add): y = ((add1 + add2) + y) / out_div -- the mean of the three stacks of a stage
 * (vocoder.py:104-110) folded into the last step of the third.  yh: the result also (or, with y NULL, only) as the split operand image
 * of the next conv (as_split_f16x2_bytes(C, N) bytes; LeakyReLU(yh_slope) applied first).  Same f16x3 arithmetic as as_conv_gemm_f32; the tile's columns stay in
 * LDS between the two convs (csrc/respair.hip). */
typedef struct AsResPairArgs {
    const float* x; int32_t ldx;
    float* y; int32_t ldy;
    const uint16_t* w1; const uint16_t* w2;
    const float* b1; const float* b2;
    float scale1, scale2;
    int32_t C, N, k, dil;
    float slope;
    const int32_t* col_off; int32_t B, max_w;
    const float* add1; const float* add2; int32_t ld_add; float out_div;
    uint16_t* yh; float yh_slope;       /* optional: LeakyReLU(y, yh_slope) as the operand image of the conv that follows (y may then be NULL) */
    int32_t x_u; const float* x_bias;   /* x_u >= 2: x is the phase-major ConvTranspose1d output z [x_u * C][N / x_u] (ldx its row stride) and
                                         * x[ch][col] = z[(col % x_u) * C + ch][col / x_u] + x_bias[ch]: as_interleave_phases_f32 folded into the read */
} AsResPairArgs;
int as_respair_f32(const AsResPairArgs* a, as_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Module-level entry points: the acoustic model behind an opaque handle.  A C / C++ host runs
 * the whole path of models.py:356-371 -- ArtsSpeech.forward(step="test") -- or one sub-module at a time through these; the
 * Python mirror of the reference's classes (artspeech_amd/models.py) is a thin caller of the same functions.
 *
 * as_model  = the weights: immutable after as_model_create, shareable between host threads, one per GPU.
 *             Exception to the conventions above: as_model_create ALLOCATES device memory (the prepared weights,
 *             ~1.3 GB for the shipped configuration) on the current HIP device and synchronises; as_model_destroy frees it.
 * as_plan   = per-caller mutable state: the small device tables of the batch geometries it has seen (utterance offsets,
 *             column descriptors), side HIP streams and events for the independent branches.  Not thread-safe: one per
 *             host thread / stream.  The first call with a new geometry uploads its tables and synchronises the stream
 *             (do not capture that call into a hipGraph); later calls with the same geometry only enqueue kernels.
 * All activations are "packed frames" (above): every utterance of the batch back to back along the column axis, no padding.
 * Workspaces are caller-owned device memory (256-byte aligned); as_module_workspace_bytes says how much a call needs.
 * ------------------------------------------------------------------------------------------- */
typedef struct as_model as_model;
typedef struct as_plan as_plan;
#define AS_ENOSPC (-2)      /* an output buffer or workspace is too small */

/* models.py:680-683 build_model(args): the Munch fields the inference path reads (config.yml model_params), plus the
 * normalisation statistics of Data/stats.json that StyleEncoder.forward applies (models.py:447-449; utils.py:86-92). */
typedef struct as_model_cfg {
    int32_t hidden_dim;    /* 512: channels of the encoders / predictors / decoder */
    int32_t dim_in;        /* 64: first width of the style towers */
    int32_t style_dim;     /* 256 (the Style vector has 2 * style_dim entries; its slices are hard-coded, models.py:499,597-599) */
    int32_t n_mels;        /* 80 */
    int32_t n_token;       /* 178 */
    int32_t reserved;
    float stats[24];       /* energy_mean, energy_std, pitch_mean, pitch_std, EMA_mean[10], EMA_std[10] */
} as_model_cfg;

/* blob_host: the checkpoint's ArtsSpeech state_dict in the reference's own key layout (weight_norm g / v, spectral_norm
 * orig / u / v) serialised as
 *     "ASWBLOB1" | u32 n | n x { u16 name_len | name | u8 ndim | u32 dims[ndim] | u64 data_offset } | u64 data_bytes | fp32 data
 * (little endian; data_offset counts from the start of the data section; artspeech_amd/blob.py writes it from a
 * state_dict).  Folds weight_norm / spectral_norm (models.py:685-701 does it implicitly on every forward), lays every
 * weight out for the kernels and uploads it. */
int as_model_create(const void* blob_host, size_t blob_bytes, const as_model_cfg* cfg, as_model** out);
int as_model_destroy(as_model* m);
/* the configuration the model was created with */
int as_model_get_cfg(const as_model* m, as_model_cfg* out);
int as_plan_create(const as_model* m, as_plan** out);
int as_plan_destroy(as_plan* p);
/* on = 1: the independent branches of a forward run on the calling stream instead of on side streams (one chain per batch: the
 * arrangement as_lanes keeps several of in flight).  as_plan_set_merge (default on; it matters for serial plans only): the branches'
 * launches are recorded first and played out in an order that lets conv GEMMs of independent branches -- the towers, the triple encoder,
 * dur_block and the duration predictor's blocks: models.py:356-360, 417-424, 540-546 have no edge between them -- share launches
 * (as_conv_gemm_multi_f32): on one stream a step costs the sum of its kernels' durations, and a small conv beside a large one costs next to
 * nothing.  off: every branch whole, one after the other, one launch per conv.  Set both before asking for workspace sizes and do not
 * change them between as_forward_test_begin and _finish (the order of the workspace allocations follows the flags). */
int as_plan_set_serial(as_plan* p, int on);
int as_plan_set_merge(as_plan* p, int on);
/* on = 1: as_forward_test records phase marks (HIP events) on the calling stream; as_plan_phase_ms then returns the ms of the
 * last call's phases: [0] reference features + tower inputs, [1] the concurrent branches (encoders, towers, duration predictor),
 * [2] durations + AdaIN fc + articulatory predictors, [3] decoder.  Blocks until that call has finished. */
int as_plan_set_timing(as_plan* p, int on);
/* matrix-core products per fp32 product in this plan's forwards: 3 = f16x3 (fp32-accurate, default); 1 = plain fp16 operands
 * (the 16-bit-operand mode BASELINE.md names for config C2; its error is reported by bench.py and tests/test_net_gpu.py) */
int as_plan_set_operand_mode(as_plan* p, int n_prod);
int as_plan_phase_ms(as_plan* p, float* ms, int n);
/* The plan caches the device tables of every batch geometry it has seen (key: the whole length vector).  Above max_layouts entries
 * (default 4096, minimum 64) the next entry point first waits for the streams THIS plan has launched on (not for the device: other plans'
 * work goes on), drops the cache and reuses its table memory: memory stays bounded under ever new ragged batches.  The flush never runs
 * while the calling stream is being captured (it then waits for the next entry point).  A hipGraph captured from a plan holds table
 * addresses -- give captured geometries a plan of their own whose cap is never reached, and reset it (as_plan_reset_layouts) only
 * together with its graphs: as_lanes does.  as_plan_layout_flushes: how often the cache has been dropped; as_plan_layout_count: entries
 * held now.  as_plan_reset_layouts: drop the cache now -- the caller guarantees that no work launched from this plan is in flight and that
 * no graph captured from it will be launched again. */
int as_plan_set_layout_cap(as_plan* p, int max_layouts);
int as_plan_layout_flushes(const as_plan* p);
int as_plan_layout_count(const as_plan* p);
int as_plan_reset_layouts(as_plan* p);

/* geometry of one batch: HOST arrays */
typedef struct as_batch {
    int32_t B;
    const int32_t* tok_lens;   /* [B] tokens per utterance (text modules) */
    const int32_t* ref_lens;   /* [B] reference mel frames per utterance (style modules) */
    const int32_t* frames;     /* [B] half-rate frames per utterance = sum of its integer durations (predictors / decoder);
                                  as_forward_test: NULL = not known yet, read back from the device (one stream synchronisation) */
} as_batch;

enum { AS_MOD_FORWARD_A = 0, AS_MOD_FORWARD_B = 1, AS_MOD_ENCODER = 2, AS_MOD_STYLE = 3, AS_MOD_DURATION = 4, AS_MOD_ARTS = 5,
       AS_MOD_DECODER = 6, AS_MOD_FORWARD_B_CAP = 7 };
/* workspace bytes of one module call for this geometry (AS_MOD_FORWARD_A / _B: the two workspaces of as_forward_test; AS_MOD_FORWARD_B_CAP:
 * workspace B of a call with as_forward_io.frame_cap = the SUM of batch->frames, whose entries are then capacities, not counts) */
size_t as_module_workspace_bytes(const as_model* m, as_plan* p, int module, const as_batch* batch);

/* RelTransformerEncoder.forward (RelTransformerEnc.py:371-380).  which: 0 text_encoder, 1 arts_encoder, 2 the duration
 * predictor's text_encoder.  tokens int32 [sum tok_lens]; out fp32 [hidden_dim][ldo >= sum tok_lens] (the reference returns
 * its transpose, padded per utterance). */
int as_encoder_forward(const as_model* m, as_plan* p, int which, const as_batch* batch, const int32_t* tokens, float* out, int ldo,
                       void* ws, size_t ws_bytes, as_stream_t stream);
/* StyleEncoder.forward (models.py:426-472) behind the two frozen extractors: mel [n_mels][ldm], f0_raw [sum ref_lens],
 * ema_raw [10][lde] (models.py:432-433 outputs) -> feat12 [12][ldf] (row 0 energy n_ext, 1 f0_ext, 2..11 ema_ext, normalised),
 * style [B][2 * style_dim]. */
int as_style_forward(const as_model* m, as_plan* p, const as_batch* batch, const float* mel, int ldm, const float* f0_raw,
                     const float* ema_raw, int lde, float* feat12, int ldf, float* style, void* ws, size_t ws_bytes, as_stream_t stream);
/* DurationPredictor.forward (models.py:540-566): tokens, ema_ext [10][lde] (rows 2..11 of feat12) -> duration fp32 [sum tok_lens] */
int as_duration_forward(const as_model* m, as_plan* p, const as_batch* batch, const int32_t* tokens, const float* ema_ext, int lde,
                        float* duration, void* ws, size_t ws_bytes, as_stream_t stream);
/* ArtsPredictor.forward (models.py:596-621): a_ens [hidden_dim][lda >= sum frames], style [B][2 * style_dim] ->
 * F0, N [1][ldp], EMA [10][ldp], ldp >= 2 * sum frames */
int as_arts_forward(const as_model* m, as_plan* p, const as_batch* batch, const float* a_ens, int lda, const float* style, float* F0,
                    float* N, float* EMA, int ldp, void* ws, size_t ws_bytes, as_stream_t stream);
/* Decoder.forward (models.py:497-517): asr [hidden_dim][lda >= sum frames] (half rate; the x2 nearest up-sampling of
 * models.py:500 happens inside), style, F0 / N [1][ldp], EMA [10][ldp] -> mel [n_mels][ldo >= 2 * sum frames] */
int as_decoder_forward(const as_model* m, as_plan* p, const as_batch* batch, const float* asr, int lda, const float* style,
                       const float* F0, const float* N, const float* EMA, int ldp, float* mel, int ldo, void* ws, size_t ws_bytes,
                       as_stream_t stream);

/* ArtsSpeech.forward(step="test") (models.py:356-371), batched: every utterance gets exactly its batch-1 result.
 * Two halves sharing workspace A: _begin runs everything up to the integer durations (encoders, style towers, duration
 * predictor; results stay in ws_a), _finish the alignment expansion, the articulatory predictors and the decoder (needs
 * batch->frames).  as_forward_test = both; with batch->frames == NULL it synchronises the stream once in between to read the
 * frame counts (returns AS_ENOSPC if ld_out or ws_b turn out too small; frames_host_out, if given, then says what is needed). */
typedef struct as_forward_io {
    /* inputs (device) */
    const int32_t* tokens;               /* [sum tok_lens] */
    const float* mel; int32_t ld_mel;    /* [n_mels][ld_mel >= sum ref_lens] normalised log-mel of the reference utterances */
    const float* f0_raw;                 /* [sum ref_lens]      pitch extractor output (models.py:432) */
    const float* ema_raw; int32_t ld_ema;/* [10][ld_ema]        EMA extractor output (models.py:433) */
    const int32_t* forced_dur;           /* optional [sum tok_lens]: integer durations replacing the predictor's (benchmarks) */
    /* output (device) */
    float* mel_out; int32_t ld_out;      /* [n_mels][ld_out >= 2 * sum frames] */
    /* optional outputs (device; NULL = not wanted) */
    float* duration;                     /* [sum tok_lens] fp32 durations before rounding */
    int32_t* dur_i;                      /* [sum tok_lens] integer durations (round half even, clamp >= 1) */
    int32_t* frame_off;                  /* [B + 1] half-rate frame offsets */
    float* style;                        /* [B][2 * style_dim] */
    float* feat12; int32_t ld_feat;      /* [12][ld_feat >= sum ref_lens] */
    float* t_en; float* a_en; int32_t ld_en;          /* [hidden_dim][ld_en >= sum tok_lens] */
    float* F0; float* N; float* EMA; int32_t ld_pred; /* [1] / [1] / [10] x [ld_pred >= 2 * sum frames] */
    /* Predicted durations WITHOUT the host read-back (batch->frames == NULL and frame_cap > 0).  models.py:361-368 sizes the second half
     * from the durations the first half predicts; frame_cap = the half-rate frames the caller makes ROOM for, all utterances together
     * (ld_out >= 2 * frame_cap, ld_pred likewise; the workspace of AS_MOD_FORWARD_B_CAP).  Every launch of the second half is then sized by
     * the capacity, the utterances' real extents are derived on the device (frame_off, optional output here, tells the caller where each
     * utterance's frames lie: utterance b at columns [2 frame_off[b], 2 frame_off[b + 1]) of mel_out), columns behind them are filler that
     * costs next to nothing (ConvGemmArgs.n_valid) -- and the call neither synchronises nor depends on values the host does not have: it
     * can be captured into a hipGraph, and as_lanes replays and coalesces it.  More frames than room: AS_STATUS_CAPACITY (what was computed
     * was cut at the capacity; nothing is written out of bounds; run again with more room).  frames_host_out is not written.
     * Optional outputs in this mode: frame_off, dur_i, duration, style, feat12, t_en, a_en, F0 / N / EMA. */
    int32_t frame_cap;
    /* a merged call (as_lanes coalescing under frame_cap): the submissions it was made of.  Utterances [first[s], first[s + 1]) came with
     * submission s, whose own output buffer mel_out[s] (row stride ld_out[s] >= 2 cap[s]) gets its utterances from its first column on and
     * whose frame_off[s] (optional, [its utterances + 1]) counts from 0 -- every submission is served as if it had been alone.  NULL: the
     * call is one submission (mel_out / frame_off above). */
    const struct as_segments* segs;
} as_forward_io;
typedef struct as_segments {
    int32_t n;                      /* 1 .. AS_MAX_SEGMENTS */
    int32_t first[17];
    int32_t cap[16];                /* half-rate frames of room of submission s; their sum = frame_cap */
    float* mel_out[16]; int32_t ld_out[16];
    int32_t* frame_off[16];
} as_segments;
#define AS_MAX_SEGMENTS 16
int as_forward_test_begin(const as_model* m, as_plan* p, const as_batch* batch, const as_forward_io* io, void* ws_a, size_t ws_a_bytes,
                          as_stream_t stream);
int as_forward_test_finish(const as_model* m, as_plan* p, const as_batch* batch, const as_forward_io* io, void* ws_a, size_t ws_a_bytes,
                           void* ws_b, size_t ws_b_bytes, as_stream_t stream);
int as_forward_test(const as_model* m, as_plan* p, const as_batch* batch, const as_forward_io* io, void* ws_a, size_t ws_a_bytes,
                    void* ws_b, size_t ws_b_bytes, int32_t* frames_host_out, as_stream_t stream);

/* ---- batches in flight (csrc/lanes.hip; DESIGN.md section 5: the throughput arrangement) -----------------------------------------------
 * as_lanes = n lanes on ONE model: per lane a HIP stream of its own, its two workspaces (grown on demand), TWO serial plans
 * (as_plan_set_serial) -- one for eager calls, whose layout cache may be flushed at any entry point, and one that only ever sees the
 * geometries that are replayed from hipGraphs and is reset only together with them -- and the hipGraphs of the (geometry, io) pairs it
 * has replayed.  as_lanes_submit enqueues ONE batch (ArtsSpeech.forward(step="test"), as as_forward_test) on the next lane, round robin,
 * and returns at once; before it does it waits until that lane's PREVIOUS batch has finished -- so the device buffers named by a lane's
 * as_forward_io may be refilled once the submit that follows them on the same lane has been entered, or after as_lanes_wait.  Fill a
 * lane's input buffers on ITS stream (as_lanes_stream(q, as_lanes_next(q))) or synchronise before submitting.  With batch->frames given
 * (forced durations, or a second pass) a (geometry, io) pair runs eagerly twice on a lane (first on the eager plan, then on the graph
 * plan, which uploads its tables), the third submit is captured into a hipGraph and later ones are one hipGraphLaunch; a lane keeps at
 * most as_lanes_set_graph_cap graphs (default 256) and drops them all -- with the graph plan's tables -- when one more is wanted.  With
 * frames == NULL (predicted durations) and no frame capacity every submit runs as_forward_test eagerly, which synchronises that lane's stream
 * once to read the frame counts, and a workspace too small for them is re-sized and the call repeated (AS_ENOSPC only if io->ld_out itself
 * is too small: frames_host_out then says what is needed); with io->frame_cap > 0 (as_forward_io: the durations stay on the device) such a
 * submission is eager, captured, replayed -- and coalesced -- like one with known counts.  A workspace that is outgrown is kept until as_lanes_wait(q, -1) / as_lanes_destroy (freeing
 * it would synchronise the device under the other lanes).  Not thread-safe: one host thread per as_lanes.  as_lanes_wait(q, lane):
 * lane < 0 = all; AS_EDEVICE if a kernel raised a status bit (as_device_status). */
typedef struct as_lanes as_lanes;
int as_lanes_create(const as_model* m, int n_lanes, as_lanes** out);
int as_lanes_destroy(as_lanes* q);
int as_lanes_count(const as_lanes* q);
int as_lanes_next(const as_lanes* q);
as_stream_t as_lanes_stream(const as_lanes* q, int lane);
int as_lanes_submit(as_lanes* q, const as_batch* batch, const as_forward_io* io, int32_t* frames_host_out, int32_t* lane_out);
int as_lanes_wait(as_lanes* q, int lane);
/* Coalescing (k > 1; 1 = off, the default).  Every tensor of the path concatenates the utterances along its column axis, so submissions
 * whose buffers are ADJACENT views of one block -- the next one's tokens / mel / f0_raw / ema_raw / forced_dur / mel_out begin exactly where
 * the previous one's end, with the same leading dimensions -- are one batch as they lie: a lane holds such a submission back (frames given,
 * no optional outputs; or a frame capacity, io->frame_cap: then only the INPUTS have to be adjacent, every submission keeps its own mel_out
 * and may ask for its own frame_off, as_segments) until k neighbours have arrived and launches them as ONE as_forward_test call, without a copy (wider conv GEMM
 * launches; one fetch of the weights for k batches).  A submission that is not the neighbour of what waits on its lane sends that group out
 * first; as_lanes_wait and as_lanes_flush launch whatever waits.  With k > 1 a submit may therefore return before anything of it is
 * enqueued, and the status of a group's launch is returned by the call that triggers it; the host arrays of as_batch are copied at
 * submit.  DEVICE BUFFERS with k > 1: a submission's buffers are read when its GROUP is launched, not when it is submitted, and entering
 * the next submit of a lane no longer means the lane's previous work has finished (that submit may only be queued): refill or reuse a
 * lane's buffers after as_lanes_wait(q, lane) -- or keep two blocks per lane and alternate.  Re-submitting the SAME unchanged buffers (a
 * benchmark's replay) needs nothing: a group's launch first waits for the lane's previous group.  The turn passes to the next lane when
 * a group is launched; as_lanes_destroy launches what still waits before it tears the lanes down.
 * (models.py:361-362 processes one utterance at a time: any grouping is legal, and every utterance gets its batch-1 result.) */
int as_lanes_set_coalesce(as_lanes* q, int k);
int as_lanes_flush(as_lanes* q);
/* Debug mode (also switched on by AS_DEBUG=1 in the environment at as_lanes_create).  The buffer rule above is the caller's to keep, and a
 * caller that breaks it corrupts a batch without a sign.  With debug on, the device inputs of a submission that is held back for its group
 * (tokens, forced durations, f0, the EMA and mel rows) are checksummed when it is submitted -- the submit then WAITS for its lane's stream, so
 * the sum is of what was handed over -- and again when the group is launched: a difference raises AS_STATUS_BAD_LAYOUT and the call that
 * launches the group returns AS_EDEVICE.  One stream synchronisation per submission: not for production traffic. */
int as_lanes_set_debug(as_lanes* q, int on);
/* Host submissions: the boundary of the reference hands over HOST arrays (test.py:96-113 moves tokens / mel to the device inside
 * `synthesis` and the mel back), and under coalescing the device-buffer rule above is easy to get wrong -- so the lane can own the device
 * side.  as_lanes_submit_host copies the submission's inputs into the next free column range of the lane's own device block (one block per
 * lane: a group's submissions lie side by side in it, i.e. they are adjacent as as_lanes_set_coalesce wants them, with no gather), launches
 * the group when it is full (coalesce = 1: at once) and copies every submission's mel to ITS host array behind the launch.  A lane keeps
 * TWO such blocks and alternates between them from group to group, with a copy stream of its own for either direction: the copies of a
 * lane's next group run under the kernels of its current one, and a block is refilled only behind the kernels AND the result copies of
 * the group that used it last (events between the lane's three streams; nothing for the caller to keep).  batch->frames must be given (forced durations, or known
 * from an earlier pass).  Pointers are HOST pointers: pinned memory (hipHostMalloc) for copies that do not block the calling thread; the
 * input arrays must stay unchanged and the output array is valid after as_lanes_wait(q, lane) (lane = *lane_out) has returned.  A
 * submission that does not fit behind what waits on its lane's block (or follows a device submission there) sends that group out first. */
typedef struct as_host_io {
    const int32_t* tokens;                 /* [sum tok_lens] */
    const float* mel; int32_t ld_mel;      /* [n_mels][ld_mel >= sum ref_lens] */
    const float* f0_raw;                   /* [sum ref_lens] */
    const float* ema_raw; int32_t ld_ema;  /* [10][ld_ema >= sum ref_lens] */
    const int32_t* forced_dur;             /* optional [sum tok_lens] */
    float* mel_out; int32_t ld_out;        /* [n_mels][ld_out >= 2 * sum frames] */
    /* predicted durations (batch->frames == NULL; as_forward_io.frame_cap): the half-rate frames there is room for, and where the
     * utterances' frame offsets [B + 1] go (HOST; required: it is how the caller finds its utterances in mel_out, whose row stride is then
     * ld_out >= 2 * frame_cap and which is copied back whole) */
    int32_t frame_cap;
    int32_t* frame_off;
} as_host_io;
int as_lanes_submit_host(as_lanes* q, const as_batch* batch, const as_host_io* io, int32_t* lane_out);
int64_t as_lanes_merged_calls(const as_lanes* q, int lane);   /* as_forward_test calls of this lane that held more than one submission */
/* tuning / tests: graphs a lane keeps (>= 1), and the layout cap of every lane's eager plan (as_plan_set_layout_cap) */
int as_lanes_set_graph_cap(as_lanes* q, int max_graphs);
int as_lanes_set_layout_cap(as_lanes* q, int max_layouts);
/* every lane's workspaces A / B get at least this many bytes now: a server that knows its largest batch (as_module_workspace_bytes) never
 * re-sizes a workspace -- which would drop that lane's graphs -- in the middle of traffic */
int as_lanes_reserve(as_lanes* q, size_t bytes_a, size_t bytes_b);
/* counters of lane `lane`: [0] graphs held, [1] times all graphs were dropped, [2] layout flushes of the eager plan, [3] graph launches,
 * [4] eager calls, [5] captures */
int as_lanes_stats(const as_lanes* q, int lane, int64_t* out6);

#ifdef __cplusplus
}
#endif
#endif /* ARTSPEECH_HIP_H */
