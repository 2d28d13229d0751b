/*
 * efts_abi.h -- C ABI of libefts_hip.so: the MI355X (gfx950) implementation of the
 * EFTS-CNN acoustic-model hot path.
 *
 * The reference (liusongxiang/efficient_tts) is pure Python/PyTorch and has NO FFI or
 * plugin interface; its boundary for this path is the Python class
 * nntts.models.EfficientTTSCNN (nntts/models/efficient_tts.py:23) resolved by name at
 * nntts/bin/train.py:173-185.  This header therefore defines the C entry points a
 * maintainer would bind (ctypes) to replace the stock torch ops that class calls; each
 * entry cites the reference op(s) it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + explicit sizes; no torch types, no hidden allocation,
 *     no retained pointers.  The caller owns every buffer.
 *   - every launch goes to the hipStream_t passed as `void* stream`; no device sync.
 *   - return 0 on success, negative EFTS_E* on error; efts_last_error() gives the text
 *     (thread-local).  Nothing throws across the ABI.
 *
 * Row space.  Activations of B items x T steps live in a padded, item-major row space:
 *   row(b, t) = b * Tp + t,  Tp = T + EFTS_GAP, rows t >= T of every item are zero ("gap").
 * With a gap of (k-1)/2 = 2 zero rows between items a 1-D convolution over the whole
 * row space equals B independent zero-padded convolutions, so tiles may straddle items.
 * Buffers carry EFTS_GUARD_LO zero rows before row 0 and EFTS_GUARD_HI rows after the
 * last 128-row tile; pointers passed in point at row 0.
 *
 * MFMA operand planes.  Matrices consumed by the MFMA GEMM are bf16, row-major, K in
 * 128-byte chunks:
 *   split 1 ("bf16")   : row = [Kp] bf16,            Kp = roundup(K, 64); chunk = 64 k's
 *   split 2 ("bf16x3") : row = [Kp/32][hi32 | lo32], Kp = roundup(K, 32); chunk = 32 k's,
 *                        x = hi + lo (two bf16), product = hi*hi + hi*lo + lo*hi in fp32.
 */
#ifndef EFTS_ABI_H
#define EFTS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFTS_GAP 2
#define EFTS_GUARD_LO 8
#define EFTS_GUARD_HI 144
#define EFTS_TILE_M 128

#define EFTS_OK 0
#define EFTS_EINVAL (-1)
#define EFTS_ESHAPE (-2)
#define EFTS_EALIGN (-3)
#define EFTS_ELAUNCH (-4)
#define EFTS_EDEVICE (-5)

#define EFTS_ACT_NONE 0
#define EFTS_ACT_LEAKY 1 /* LeakyReLU, slope in args */
#define EFTS_ACT_RELU 2
#define EFTS_ACT_TANH 3 /* the vocoder's output layer */

/* ABI revision of THIS header.  Bumped whenever an argument block changes size / meaning or an export is removed, so that a consumer built
 * against another revision can tell: efts_version() returns the revision the library was built from, and a binding must refuse to run unless
 * it equals the EFTS_ABI_VERSION it was written against (efficient_tts_amd/lib.py does, tests/test_abi_cpu.py pins all three).
 *   100  rounds 1-4
 *   500  round 5: efts_resconv5_args grew by act_bwd_sign / act_bwd_bias_part / act_bwd_bias_rows / act_bwd_slope / kernel;
 *        efts_wgrad_tn, efts_wgrad_reduce_bias, efts_resconv5_kernel removed (efts_wgrad_tn_grouped / efts_wgrad_reduce_grouped instead)
 *   600  round 6: + efts_frame_pack_dit, efts_logmel_dit; efts_pack_item.plane may be NULL (dgrad plane only)
 *   601  round 6 (this header): efts_gemm_args grew by sqerr_target / ld_target / target_batch_stride / sqerr_part;
 *        + efts_losses_from_parts, efts_logmel_fft, efts_logmel_fft_pcm16 */
#define EFTS_ABI_VERSION 601
int efts_version(void);
const char* efts_last_error(void);
/* 0 when the current HIP device is gfx950. */
int efts_device_check(void);

/* ------------------------------------------------------------------------------------
 * The MFMA contraction: out = epilogue( alpha * sum_tap A[row + tap - pad, :] . Bw[tap][col, :] )
 * Replaces torch Conv1d (nntts/layers/efts_modules.py:32-35,48-51; duration_predictor.py:57),
 * Linear (efficient_tts.py:149-153,161,198) and bmm (efficient_tts.py:190,390).
 * epilogue: + bias[col]; activation; + resid[row, col]; * rowmask[row]; store fp32 and/or
 * bf16 operand plane(s) for the next contraction.
 * ---------------------------------------------------------------------------------- */
typedef struct efts_gemm_args {
    /* A operand plane (activations): row 0 pointer, row stride in bytes */
    const void* a;
    int64_t lda;
    int64_t a_batch_stride; /* bytes between batch items (batched mode), else 0 */
    /* B operand plane ("weights"): [taps][N rows][K], row stride / tap stride in bytes */
    const void* b;
    int64_t ldb;
    int64_t b_tap_stride;
    int64_t b_batch_stride;
    int32_t split;  /* 1 = bf16, 2 = bf16x3 (hi/lo interleaved) */
    int32_t taps;   /* 1, 3, 5, 7, 9 or 11 */
    int32_t m;      /* rows per batch item */
    int32_t n;      /* output columns */
    int32_t nchunk; /* 128-byte K chunks per row */
    int32_t batch;  /* >= 1 */
    float alpha;
    int32_t act;
    float slope;
    const float* bias;    /* [n] or NULL */
    const float* resid;   /* fp32 [m, ldr] or NULL */
    int64_t ldr;          /* elements */
    int64_t resid_batch_stride;
    const float* rowmask; /* fp32 [m] (per batch item: + z*m_mask_stride) or NULL */
    int64_t rowmask_batch_stride;
    float* out_f32; /* or NULL */
    int64_t ldo;    /* elements */
    int64_t out_batch_stride;
    void* out_bf16;    /* operand plane for the next contraction, or NULL */
    int64_t ldob;      /* bytes */
    int64_t outb_batch_stride;
    int32_t out_split; /* 1 or 2: format of out_bf16 */
    int32_t batch2;    /* optional outer batch (grid.z), 0/1 = none; e.g. the taps of a wgrad */
    int64_t a_batch2_stride, b_batch2_stride, out_batch2_stride; /* bytes, bytes, elements */
    /* dilated convolutions (HiFi-GAN residual blocks, nntts/vocoders/hifigan_model.py:30-58): rows between taps,
     * 0 / 1 = dense; (taps - 1) * dilation <= 64; taps may also be 7 or 11.  The A plane is read from
     * (taps - 1) / 2 * dilation rows before row 0 to 144 rows after row m - 1: the caller's buffer must hold zero rows
     * there (EFTS_GUARD_LO = 8 covers the dense taps of the acoustic model; the vocoder's row spaces carry 64). */
    int32_t dilation;
    /* 1: out_bf16 receives LeakyReLU(out, plane_slope) instead of out -- for consumers that apply the
     * activation to their INPUT (pre-activation residual blocks); out_f32 stays un-activated. */
    int32_t plane_act;
    float plane_slope;
    /* out_split 1 only, optional: a second plane of the layout of out_bf16 that receives the bf16 REMAINDER
     * out - bf16(out), so that a consumer can rebuild out to 16 mantissa bits (the residual input of efts_resconv5). */
    void* out_bf16_lo;
    /* which of the (bit-identical) kernels runs: EFTS_TILING_AUTO picks by shape; the explicit values are for A/B timing and
     * for the equality tests between the kernels (EFTS_TILING_SMALLM excepted, see below).  An explicit tiling the shape does not allow is an error. */
    int32_t tiling;
    /* training forward of a LeakyReLU / ReLU layer: the sign of every activated output BEFORE the residual add, as bit words
     * for efts_act_bwd mode 4 (instead of keeping y and x in fp32 for the backward: 1/8 B instead of 8 B per element read
     * there).  Row stride n / 8 bytes; within a row, 16 bytes per 128-column group: word u (0..3), bit q (0..31) = column
     * 128 * group + 4 * q + u is positive.  Needs n % 128 == 0, batch 1, 16-byte aligned fp32 / plane rows (the vector epilogue),
     * and runs on the generic or wide tiling only.  NULL: not written. */
    void* sign_mask;
    /* attention scores -> expected key index in the epilogue (scaled_dot_product_attention + the soft index of imv_generator,
     * nntts/models/efficient_tts.py:391-398, :312): soft_index[b][row] = sum_i i * softmax_i(out[b][row][i < key_len[b]]) for
     * row < query_len[b], 0 beyond -- what efts_attn_soft_index computes from the stored scores, without storing them
     * (out_f32 may be NULL then).  Needs n <= 128 (one column tile holds a whole row), generic tiling, no residual / plane
     * outputs.  soft_index: fp32 [batch][m]; key_len, query_len: int32 [batch].  NULL: not computed. */
    float* soft_index;
    const int32_t* key_len;
    const int32_t* query_len;
    /* train-mode Dropout on the activated value, BEFORE the residual add (ResConv1d / mel_prenet with dropout_rate > 0,
     * nntts/layers/efts_modules.py:38-47, efficient_tts.py:76-80): element (row, col) is kept and scaled by 1 / (1 - drop_p) iff
     * hash(drop_seed, row * n + col) >= drop_p * 2^32 -- a stateless counter-based mask efts_act_bwd_dropout regenerates (not
     * torch's Philox stream: same distribution, different draws).  batch 1, generic / wide tiling, vector epilogue.  0: none. */
    float drop_p;
    uint32_t drop_seed;
    /* the mel term of FastSpeechLoss in the epilogue of the mel head (nntts/losses/fastspeech_loss.py:54-67 behind
     * nntts/models/efficient_tts.py:198-200, :220): per workgroup and wave, the sum of (out - target)^2 over the rows of the tile whose
     * rowmask value is not zero -- sqerr_part[((batch item * row tiles) + row tile) * 4 + wave], row tiles = ceil(m / 128) --
     * in a fixed order (deterministic; efts_losses_from_parts adds the partial sums up).  target: fp32 [batch][m][ld_target] (what the
     * caller padded the frames with is never read past a zero rowmask value); needs rowmask, one tap, n <= 128, n % 4 == 0, no residual,
     * no dropout, 16-byte aligned target rows, and runs on the generic tiling only.  NULL: not computed. */
    const float* sqerr_target;
    int64_t ld_target;           /* elements */
    int64_t target_batch_stride; /* elements */
    float* sqerr_part;
} efts_gemm_args;

#define EFTS_TILING_AUTO 0
#define EFTS_TILING_GENERIC 1  /* 124-row x 128-column tiles, two 4-wave workgroups per CU: every shape */
#define EFTS_TILING_WIDE 2     /* 252-row x 128-column tiles: dense k5 launches */
#define EFTS_TILING_NARROW 3   /* 64- / 32-column tiles */
#define EFTS_TILING_RESIDENT 4 /* window + all taps resident in LDS: n <= 64, one K chunk, taps 3 / 7 / 11 */
#define EFTS_TILING_SMALLM 5   /* short row spaces (one utterance): 64 x 32 tiles, K split across the four waves, operand fragments
                                * straight from global memory; taps 1 / 3 / 5, batch 1, n % 32 == 0, the A plane readable up to the next
                                * multiple of 64 rows (+ pad).  Same operand rounding as the other tilings but a different summation
                                * order (equal to fp32 rounding, not bit for bit): never chosen by AUTO, the caller asks for it */

int efts_gemm(const efts_gemm_args* a, void* stream);

/* ------------------------------------------------------------------------------------
 * One residual convolution layer of the EFTS-CNN stacks (ResConv1d.forward,
 * nntts/layers/efts_modules.py:48-51 with the Conv1d k=5 of :32-35; the stack loop is :77-79):
 *   y[row, :] = ( x[row, :] + LeakyReLU( conv1d_k5(x)[row, :] + bias, slope ) ) * rowmask[row]
 * on a padded row space, for long row spaces (the mel-length stacks).  The contraction runs on the operand plane `x`
 * (format `split`); the RESIDUAL x is taken, in this order, from x_f32 (fp32 [m][ldr]) if given, else from the
 * planes as hi + lo: split 2 planes carry lo inside x; a split 1 plane may come with a separate lo plane x_lo of the
 * same layout (NULL: the residual is the bf16 value itself).  Outputs, any combination: y (operand plane of format
 * y_split; with y_split 1 an optional lo plane y_lo so that the next layer can rebuild its residual), y_f32.
 * Bit-identical to efts_gemm(taps 5, LeakyReLU, resid, rowmask) on the same operands.
 * n (= cout) must be a multiple of 256; x is read from 2 rows before row 0 to 144 rows after row m - 1 (the guard rows).
 * ---------------------------------------------------------------------------------- */
typedef struct efts_resconv5_args {
    const void* x;      /* operand plane of the layer input, row 0 */
    const void* x_lo;   /* split 1 only: bf16 remainder plane of x, or NULL */
    int64_t ldx;        /* bytes, both planes */
    const float* x_f32; /* fp32 residual stream or NULL */
    int64_t ldr;        /* elements */
    const void* w;      /* B operand plane [5][n][K] (efts_pack_weight) */
    int64_t ldw;
    int64_t w_tap_stride;
    int32_t split;      /* format of x and w: 1 = bf16, 2 = bf16x3 */
    int32_t m;          /* rows */
    int32_t n;          /* output channels, % 256 == 0 */
    int32_t nchunk;     /* 128-byte K chunks per row of x / w */
    const float* bias;  /* [n] or NULL */
    float slope;        /* LeakyReLU slope */
    const float* rowmask; /* [m] or NULL */
    float* y_f32;       /* fp32 [m][ldo] or NULL */
    int64_t ldo;        /* elements */
    void* y;            /* operand plane of the layer output or NULL */
    void* y_lo;         /* y_split 1 only: remainder plane of y, or NULL */
    int64_t ldy;        /* bytes, both planes */
    int32_t y_split;    /* 1 or 2 */
    const int32_t* plan; /* HOST pointer: explicit tile schedule (format of efts_resconv5_plan) or NULL = automatic */
    int32_t taps;        /* 0 / 5: the k5 layer; 3: a k3 convolution with a [3][n][K] weight plane (duration_predictor.py:57, or a
                          * ResConv1d stack built with k_size = 3) on the same kernel: window and tile geometry of the k5 layer */
    int32_t no_residual; /* 1: y = LeakyReLU(conv + bias, slope) * rowmask without the residual term (slope 0 = ReLU): the
                          * duration predictor's Conv1d + ReLU, riding in a decoder launch (efts_resconv5_multi) */
    void* sign_bits;     /* training forward, optional: the sign of every activated output BEFORE the residual add as plain bit rows for
                          * efts_act_bwd mode 5 -- row stride n / 8 bytes, bit j of byte c = column 8 c + j is positive (the same
                          * information as efts_gemm_args.sign_mask, in the order this kernel's epilogue holds it).  NULL: not written */
    /* training backward, optional: this launch is the DGRAD of stack layer l -- x = the operand plane of dZ_l, w = the transposed weights,
     * x_f32 = G, y_f32 = G' = d loss / d x_l, slope 1, no bias -- and its epilogue also runs the activation backward of layer l - 1
     * (efts_act_bwd mode 5 | EFTS_ACT_BWD_BIAS_PARTS) on the values it holds: y receives dZ_{l-1} = G' * (bit ? 1 : act_bwd_slope) as an
     * operand plane of format y_split = split, act_bwd_bias_part one row of column sums of dZ_{l-1} per tile and wave row
     * (efts_resconv5_bias_rows(m, n) rows of n floats; every row is written by every launch -- rows of tiles the schedule does not have
     * receive zeros -- so the table needs no clearing and may be re-used across shapes; NULL: no sums).  act_bwd_sign = the sign_bits layer l - 1's forward launch wrote.  NULL: a plain layer. */
    const void* act_bwd_sign;
    float* act_bwd_bias_part;
    int32_t act_bwd_bias_rows;
    float act_bwd_slope;
    int32_t kernel;      /* which kernel runs the launch (every layer of a grouped launch must name the same one; other values: EFTS_EINVAL).
                          * 0: the 8-wave ping-pong kernel -- the
                          * product path.  2: the one-wave-per-SIMD kernel with the generated main loop wherever it applies (bf16 planes, 5 taps,
                          * >= 2 K chunks, no fused activation backward; the 8-wave kernel elsewhere): same results bit for bit, same time on
                          * MI355X (DESIGN.md 4a'), kept for the bit-equality tests between the two and for A/B measurements */
} efts_resconv5_args;

int efts_resconv5(const efts_resconv5_args* a, void* stream);
/* rows of efts_resconv5_args.act_bwd_bias_part for an m x n layer on the current device (the automatic schedule) */
int efts_resconv5_bias_rows(int32_t m, int32_t n);

/* `count` (1 or 2) independent residual layers of the same geometry (split, n, nchunk, ldw) in ONE persistent launch: the rows of
 * the layers are laid end to end and scheduled over the compute units as one row space; a tile never crosses from one layer
 * into the next, and every tile brings its own operands, weights, bias, mask and outputs.  This is how a short stack rides
 * along with a long one -- text-encoder layer k + 2 (64 x 130 rows) in the launch of mel-encoder layer k (64 x 802 rows;
 * nntts/models/efficient_tts.py:148 and :162 are independent until :167) -- at the long launch's efficiency instead of a
 * launch of its own that cannot fill the chip.  Results per layer are those of efts_resconv5.  layers[0].plan (optional)
 * schedules the combined rows. */
int efts_resconv5_multi(const efts_resconv5_args* layers, int32_t count, void* stream);

/* A Linear with few input features applied to the caller's fp32 frames, written into the row space:
 *   y[b * Tp + t, :] = act(x[b][t][:cin] . W^T + bias), t < T (rows t >= T of the row space are not touched: they stay zero)
 * -- `mel_prenet` of the reference (nntts/models/efficient_tts.py:76-80, applied at :161; eval / Dropout-free) in one launch,
 * without an operand plane of the input.  Bit-identical to efts_pack_rows + efts_gemm.  cin % 8 == 0, cin <= 128, n % 128 == 0. */
typedef struct efts_frame_linear_args {
    const float* x;      /* [B][T][cin] fp32, contiguous, 16-byte aligned */
    const void* w;       /* packed B plane [n][ldw] (efts_pack_weight, one tap) of format `split` */
    int64_t ldw;
    int32_t split;       /* operand format the contraction runs in: 1 bf16, 2 bf16x3 (hi / lo) */
    const float* bias;   /* [n] or NULL */
    int32_t act;         /* EFTS_ACT_* */
    float slope;
    int32_t B, T, Tp, cin, n;
    float* y_f32;        /* [B * Tp][ldo] or NULL */
    int64_t ldo;         /* floats */
    void* y;             /* operand plane of the output (row 0) or NULL */
    void* y_lo;          /* y_split 1 only: bf16 remainder plane or NULL */
    int64_t ldy;         /* bytes, both planes */
    int32_t y_split;
    int32_t max_workgroups; /* 0: one workgroup per (item, 128-channel slice); > 0: at most this many (each then takes several units in
                             * turn), leaving compute units to a launch that runs beside this one -- the kernel is HBM-bound */
} efts_frame_linear_args;
int efts_frame_linear(const efts_frame_linear_args* a, void* stream);

/* The static tile schedule of efts_resconv5 for m rows x n columns on `cus` compute units (0: the current device).
 * One persistent workgroup per CU; the workgroups form groups (one workgroup per 256-column tile); group g belongs to
 * class g % classes and owns `rows` consecutive output rows, cut into `ntile` tiles of 32 * h - 4 rows, h = 2..8 half
 * units of 32 window rows (odd heights: the two wave rows of the workgroup take ceil(h/2) and floor(h/2) 32-row blocks).
 * plan (host memory, >= EFTS_RC_PLAN_INTS int32) = [groups, classes, then for each of 4 classes: rows, ntile, h[8]].
 * Returns the number of int32 written, or a negative EFTS_E* code.  No device work. */
#define EFTS_RC_PLAN_INTS 42
int efts_resconv5_plan(int32_t m, int32_t n, int32_t cus, int32_t* plan, int32_t cap);

/* ------------------------------------------------------------------------------------
 * Parameter preparation.
 * efts_pack_weight: w[cout][cin][taps] fp32 (torch Conv1d / Linear layout) -> B operand plane
 * [taps][cout][Kp].  With g != NULL the weight-norm fold w = g * v / ||v|| (norm over cin*taps)
 * is applied first (torch.nn.utils.weight_norm, nntts/layers/efts_modules.py:92-99;
 * remove_weight_norm efficient_tts.py:400-409); w_f32_out (optional) receives the folded fp32
 * weight in the input layout.  plane row stride = ldb bytes, tap stride = cout*ldb.
 * ---------------------------------------------------------------------------------- */
int efts_pack_weight(const float* w, const float* g, float* w_f32_out, void* plane, int64_t ldb,
                     int32_t cout, int32_t cin, int32_t taps, int32_t split, void* stream);

/* ------------------------------------------------------------------------------------
 * Row-space producers.  `rows` = B*Tp.  f32_out (optional) is [rows][c] fp32, plane (optional)
 * an A operand plane with row stride ld_plane bytes.  Gap rows are written as zero.
 * efts_row_masks : gapmask[row] = t < T ; lenmask[row] = t < len[b]
 *                  (make_non_pad_mask, nntts/utils/nets_utils.py:170-254, built on device).
 * efts_embed     : text_embedding_table(text) (efficient_tts.py:144); ids int64 [B,T].
 * efts_pack_rows : x fp32 [B,T,c] contiguous (e.g. speech [B,T2,80], efficient_tts.py:161 input).
 * ---------------------------------------------------------------------------------- */
int efts_row_masks(const int32_t* lengths, float* gapmask, float* lenmask, int32_t B, int32_t T,
                   int32_t Tp, void* stream);
/* both row spaces of a teacher-forced pass (text: T1, mel: T2) in ONE launch, straight from the caller's length tensors (int64 as
 * torch makes them -- is_int64 -- or int32): int32 copies of the lengths (what every later entry point takes) + the four masks. */
int efts_row_masks_pair(const void* len1, const void* len2, int32_t is_int64, int32_t* len1_i32, int32_t* len2_i32, float* gap1,
                        float* lenmask1, float* gap2, float* lenmask2, int32_t B, int32_t T1, int32_t Tp1, int32_t T2,
                        int32_t Tp2, void* stream);
int efts_embed(const int64_t* ids, const float* table, float* f32_out, void* plane, int64_t ld_plane,
               int32_t B, int32_t T, int32_t Tp, int32_t c, int32_t num_symbols, int32_t split,
               void* stream);
int efts_pack_rows(const float* x, float* f32_out, void* plane, int64_t ld_plane, int32_t B, int32_t T,
                   int32_t Tp, int32_t c, int32_t kp, int32_t split, void* stream);
/* Embedding AND the first residual convolution of the text encoder (efficient_tts.py:144 + the first ResConv1d of :148,
 * nntts/layers/efts_modules.py:48-51) as table look-ups: that layer's input takes only num_symbols distinct values per position and
 * the convolution is linear, so tap k's product with every symbol's embedding is tabulated once per weight set --
 * tap_table[k][v][:] = W_k . table[v] (fp32 [taps][num_symbols][c]; build it with `taps` one-tap efts_gemm launches over the
 * embedding rows, in the operand format the model runs in) -- and the layer becomes
 *   y[b * Tp + t] = table[id_t] + LeakyReLU( bias + sum_k tap_table[k][ id_{t + k - (taps-1)/2} ], slope )
 * with positions outside [0, lim) contributing nothing: lim = T (lengths NULL: padded ids are real symbols, the reference's
 * teacher-forced semantics) or lengths[b] (zero embedding and zero output beyond an item's length: batched free-running inference).
 * Same outputs as efts_embed (fp32 rows and / or operand plane, gap rows zero). */
int efts_embed_conv(const int64_t* ids, const int32_t* lengths, const float* table, const float* tap_table, const float* bias, float slope,
                    float* f32_out, void* plane, int64_t ld_plane, int32_t B, int32_t T, int32_t Tp, int32_t c, int32_t num_symbols,
                    int32_t taps, int32_t split, void* stream);

/* ------------------------------------------------------------------------------------
 * Alignment block (fp32 VALU, HBM-bound).
 * efts_attn_soft_index: scores[B][T2][ld] (= q.k/sqrt(D), from efts_gemm) -> softmax over the
 *   text_len[b] valid keys and soft_idx[b][j] = sum_i alpha_ij * i, 0 for j >= mel_len[b]
 *   (scaled_dot_product_attention :377-398 + mask :168 + the bmm of imv_generator :312).
 *   alpha_out (optional) [B][T1][T2] fp32 receives alpha itself.
 * efts_imv_scan: d_j = relu(s_j - s_{j-1}), d_0 = 0; pi = cumsum(d) * melmask;
 *   pi = pi / max(max_j pi, 1e-8) * (text_len - 1)            (imv_generator :314-323).
 * efts_aligned_positions: e[b][i] = sum_j softmax_j(-sigma_e (pi_j - p_i)^2) * q_j, masked
 *   (get_aligned_positions :326-345), and the duration target log(e_i - e_{i-1} + offset),
 *   0 at padded text (efficient_tts.py:203-216, delta_e_method_1).
 * efts_reconst_alpha: alpha'[b][i][j] = softmax_i(-sigma (q_j - e_i)^2), zero outside the
 *   text x mel mask (reconstruct_align_from_aligned_position :347-375 + :186).  Lengths NULL
 *   = inference (no masks, :270-274).  Writes the fp32 API tensor [B][T1][T2] (optional) and
 *   the split-2 A operand plane of alpha'^T: row (b*T2p + j), K = i  (for the expand bmm :190).
 * efts_pack_vt: V fp32 [B*T1p][c] -> per-item split-2 B operand plane V^T [B][c][K = i].
 * efts_cumsum_rows: e = cumsum(delta) over T1 (inference :260); x [B][T] contiguous.
 * ---------------------------------------------------------------------------------- */
int efts_attn_soft_index(const float* scores, int64_t ld, const int32_t* text_len, const int32_t* mel_len,
                         float* soft_idx, float* alpha_out, int32_t B, int32_t T1, int32_t T2, void* stream);
int efts_imv_scan(const float* soft_idx, const int32_t* text_len, const int32_t* mel_len, float* imv,
                  int32_t B, int32_t T2, void* stream);
int efts_aligned_positions(const float* imv, const int32_t* text_len, const int32_t* mel_len, float sigma_e,
                           float offset, float* e, float* log_delta_e, int32_t B, int32_t T1, int32_t T2,
                           void* stream);
/* the duration target alone: log(delta_e + offset) with delta_e_i = e_i - e_{i-1} (method1 != 0, efficient_tts.py:204) or
 * e_{i+1} - e_i with e_{len} = mel_len (method1 == 0, :205-213); 0 at padded text.  e, log_delta_e: [B][T1]. */
int efts_duration_target(const float* e, const int32_t* text_len, const int32_t* mel_len, float offset, int32_t method1,
                         float* log_delta_e, int32_t B, int32_t T1, void* stream);
int efts_reconst_alpha(const float* e, const int32_t* text_len, const int32_t* mel_len, float sigma,
                       float* alpha_out, void* plane, int64_t ld_plane, int32_t B, int32_t T1, int32_t T2,
                       int32_t T2p, void* stream);
int efts_pack_vt(const float* v, int64_t ldv, void* plane, int64_t ld_plane, int32_t B, int32_t T1,
                 int32_t T1p, int32_t c, void* stream);
int efts_cumsum_rows(const float* x, float* y, int32_t B, int32_t T, void* stream);

/* The middle of the alignment block in ONE launch (one workgroup per item, operands in LDS after the first load):
 * efts_imv_scan + efts_aligned_positions + efts_duration_target, bit-identical to that chain
 * (imv_generator efficient_tts.py:314-323, get_aligned_positions :326-345, duration target :203-216).
 * imv [B][T2], e [B][T1], log_delta_e [B][T1] or NULL.  2 * T2 floats must fit 160 KiB of LDS. */
int efts_imv_align(const float* soft_idx, const int32_t* text_len, const int32_t* mel_len, float sigma_e, float offset,
                   int32_t method1, float* imv, float* e, float* log_delta_e, int32_t B, int32_t T1, int32_t T2, void* stream);

/* Gaussian re-alignment + expansion in one launch:
 *   alpha'[b][i][j] = softmax_i( -sigma (q_j - e[b][i])^2 ), q_j = j for j < mel_len[b] else 0, keys i >= text_len[b] excluded,
 *                     zero outside the text x mel mask   (reconstruct_align_from_aligned_position efficient_tts.py:347-375, :186)
 *   H[b * T2p + j][:] = sum_i alpha'[b][i][j] * V[b * T1p + i][:]   (the expand bmm :190-194; frames j >= mel_len[b] are zero)
 * alpha' is produced in registers as the MFMA A operand (split-bf16: lo*hi + hi*lo + hi*hi, as efts_gemm on split-2 planes)
 * and never stored as an operand plane; V is read as fp32 from its row space (no V^T planes).  Lengths NULL = inference (no
 * masks, :270-274).  Outputs, any combination: alpha_out (the API tensor [B][T1][T2] fp32), y_f32 (row space [B*T2p][ldo]),
 * y (operand plane of format y_split; with y_split 1 an optional remainder plane y_lo).  Rows j >= T2 of the row space are
 * not touched.  T1 <= 256, n % 128 == 0. */
typedef struct efts_expand_args {
    const float* e;          /* [B][T1] aligned positions */
    const int32_t* text_len; /* [B] or NULL */
    const int32_t* mel_len;  /* [B] or NULL */
    float sigma;
    const float* v;          /* value projection, fp32 row space [B * T1p][ldv] */
    int64_t ldv;             /* elements */
    int32_t B, T1, T1p, T2, T2p;
    int32_t n;               /* channels */
    float* alpha_out;        /* [B][T1][T2] or NULL */
    float* y_f32;            /* [B * T2p][ldo] or NULL */
    int64_t ldo;             /* elements */
    void* y;                 /* operand plane (row 0) or NULL */
    void* y_lo;              /* y_split 1 only: bf16 remainder plane or NULL */
    int64_t ldy;             /* bytes, both planes */
    int32_t y_split;
} efts_expand_args;
int efts_expand(const efts_expand_args* a, void* stream);

/* Between the two phases of free-running synthesis (efficient_tts.py:258-270), one launch per batch: e[b][i] = cumulative sum of
 * the durations dur[b * ld + i] (i < T1; efts_cumsum_rows' scan), mel_len[b] = round-half-even(e[b][text_len[b] - 1]), and with
 * method1 == 0 the positions start at 0 (e[b][i] -= dur[b][i], :261-265).  force_delta >= 0 replaces every valid duration by that
 * value (benchmark hook: a synthetic batch then yields a known mel length); < 0: off. */
int efts_duration_positions(const float* dur, int64_t ld, const int32_t* text_len, float force_delta, int32_t method1, float* e,
                            int32_t* mel_len, int32_t B, int32_t T1, void* stream);

/* The fp32 -> bf16 rounding of every operand-plane producer in this library (round to nearest even; torch's
 * .to(torch.bfloat16)): mode 0 = as the kernels do it (gfx950's packed conversion instruction), mode 1 = the integer
 * reference form.  y[i] = bf16 bits of x[i].  Exists so that a test can sweep bit patterns through both. */
int efts_bf16_round(const float* x, uint16_t* y, int64_t n, int32_t mode, void* stream);

/* ------------------------------------------------------------------------------------
 * Duration predictor tail (nntts/layers/duration_predictor.py:57-88, layer_norm.py:6-30).
 * efts_layernorm_rows: y = LN_c(x) * gamma + beta (biased variance, eps), times rowmask[row];
 *   writes fp32 (optional) and/or an A operand plane.
 * efts_layernorm_dot: out[row] = dot(LN_c(x[row]), w) + b  (LayerNorm + Linear(c,1) + squeeze);
 *   mode 0: log domain, * rowmask (masked_fill(x_masks, 0), :85-86)
 *   mode 1: inference, max(exp(.) - offset, 0)  (:78-83, to_round=False), * rowmask if given.
 * drop_p > 0 applies the module's Dropout(0.1) after the LayerNorm (:61, train mode): a stateless
 * counter-based mask from (drop_seed, element index), regenerated identically by efts_layernorm_bwd.
 * drop_seed_add (optional, device): one word added to drop_seed when the kernel runs -- the step counter of a training step that
 * is replayed as a hipGraph, where by-value arguments are frozen at capture time.
 * ---------------------------------------------------------------------------------- */
int efts_layernorm_rows(const float* x, const float* gamma, const float* beta, float eps,
                        const float* rowmask, float* f32_out, void* plane, int64_t ld_plane,
                        int32_t rows, int32_t c, int32_t split, float drop_p, uint32_t drop_seed, const uint32_t* drop_seed_add, void* stream);
int efts_layernorm_dot(const float* x, const float* gamma, const float* beta, float eps, const float* w,
                       const float* b, const float* rowmask, int32_t mode, float offset, float* out,
                       int32_t rows, int32_t c, float drop_p, uint32_t drop_seed, const uint32_t* drop_seed_add, void* stream);

/* ------------------------------------------------------------------------------------
 * FastSpeechLoss with use_masking=True (nntts/losses/fastspeech_loss.py:54-67):
 * out[0] = loss, out[1] = mel MSE over valid frames, out[2] = duration L1 over valid tokens.
 * mel_pred: row space [B*T2p][ldm]; speech [B][T2][odim]; dur_pred/log_delta_e row space /
 * [B][T1] resp.  workspace: >= efts_losses_workspace_bytes() bytes.
 * ---------------------------------------------------------------------------------- */
size_t efts_losses_workspace_bytes(void);
int efts_masked_losses(const float* mel_pred, int64_t ldm, const float* speech, const int32_t* mel_len,
                       const float* dur_pred, const float* log_delta_e, const int32_t* text_len,
                       float* out3, void* workspace, int32_t B, int32_t T1, int32_t T1p, int32_t T2,
                       int32_t T2p, int32_t odim, void* stream);
/* The same losses when the mel head's launch already left the squared-error partial sums (efts_gemm_args.sqerr_part, n_part of
 * them): one single-workgroup launch adds them up in index order, takes the duration L1 itself and writes out[0..2] as above. */
int efts_losses_from_parts(const float* sqerr_part, int32_t n_part, const int32_t* mel_len, const float* dur_pred,
                           const float* log_delta_e, const int32_t* text_len, float* out3, int32_t B, int32_t T1,
                           int32_t T1p, int32_t T2, int32_t odim, void* stream);

/* ====================================================================================
 * Training step (backward + optimizer).  The reference backward is torch autograd of the
 * forward ops (loss.backward(), nntts/trainers/efficient_tts_trainer.py:153); these entry
 * points are its hand-written equivalents.  MFMA-shaped gradients (conv/linear dgrad and
 * wgrad, the four products of the alignment block) run through efts_gemm on operand planes
 * produced here.
 * ==================================================================================== */
/* folded weight w[cout][cin][taps] -> dgrad B plane [taps][cin rows][K = cout], taps flipped */
int efts_pack_weight_t(const float* w, void* plane, int64_t ldb, int32_t cout, int32_t cin, int32_t taps,
                       int32_t split, void* stream);
/* Weight preparation of a whole group of equally shaped Conv1d / Linear weights in one call (2 launches): per item
 * efts_pack_weight (plane and/or w_f32_out; either may be NULL) and, with with_t != 0, efts_pack_weight_t into plane_t
 * (NULL = skip the item) from the folded weight.  `items` is an array in DEVICE memory.  scale_ws: n_items * cout
 * floats of device scratch (weight-norm scales); with it, shapes with cout % 64 == 0, cin % 64 == 0, taps <= 5 take
 * the tiled path (one pass over the weights writes both planes); NULL or other shapes: row kernels, which need
 * w_f32_out for a weight-normed item that has a plane_t.  The reference does this work implicitly per layer and
 * step: the weight_norm hook (efts_modules.py:92-99) in every forward, the transposed operand inside cuDNN's dgrad. */
typedef struct efts_pack_item {
    const float* w;        /* [cout][cin][taps] fp32: weight_v, or the plain weight when g == NULL */
    const float* g;        /* weight-norm gain [cout] or NULL */
    float* w_f32_out;      /* folded fp32 copy [cout][cin][taps] or NULL */
    void* plane;           /* forward B plane [taps][cout][Kp(cin)], row stride ldb, or NULL */
    void* plane_t;         /* transposed + tap-flipped dgrad plane [taps][cin][Kp(cout)], row stride ldb_t, or NULL */
} efts_pack_item;
int efts_pack_weights_grouped(const efts_pack_item* items, int32_t n_items, float* scale_ws, int64_t ldb, int64_t ldb_t,
                              int32_t cout, int32_t cin, int32_t taps, int32_t split, int32_t with_t, void* stream);
/* d loss / d mel_pred (fp32 [B*T2p][odim] and/or operand plane) and d loss / d dur_pred [B*T1p]
 * of FastSpeechLoss(use_masking) (fastspeech_loss.py:54-67); gscale: device scalar or NULL (1). */
int efts_loss_bwd(const float* mel_pred, int64_t ldm, const float* speech, const int32_t* mel_len,
                  const float* dur_pred, const float* log_delta_e, const int32_t* text_len, const float* gscale,
                  float* dmel, void* dmel_plane, int64_t ld_plane, int32_t split, float* ddur, int32_t B, int32_t T1,
                  int32_t T1p, int32_t T2, int32_t T2p, int32_t odim, void* stream);
/* dZ = G * act'(.) * rowmask, bias grad += column sums.  mode 1: residual LeakyReLU layer
 * (sign from y - x); 2: ReLU (sign of y); 3: LeakyReLU without residual (sign of y); 0: identity;
 * 5: LeakyReLU with the plain sign bits efts_resconv5 wrote (`sign_bits`) passed as y (row stride c / 8 bytes);
 * 4: LeakyReLU with the sign words efts_gemm wrote (`sign_mask` of the forward launch) passed as y (row stride c / 8 bytes,
 * c % 128 == 0), x unused.
 * mode | EFTS_ACT_BWD_BIAS_PARTS: dbias is a [ceil(rows / 64)][c] workspace that receives one column sum per 64-row block
 * (plain stores, overwritten) instead of atomic adds into the gradient; efts_wgrad_reduce_grouped adds them up. */
#define EFTS_ACT_BWD_BIAS_PARTS 16
/* efts_act_bwd with the Dropout mask of the forward launch (efts_gemm_args.drop_p / drop_seed, c = that launch's n) applied too */
int efts_act_bwd_dropout(const float* g, const float* y, const float* x, const float* rowmask, float slope, int32_t mode,
                         float* dz, void* plane, int64_t ld_plane, int32_t split, float* dbias, int32_t rows, int32_t c,
                         float drop_p, uint32_t drop_seed, void* stream);
int efts_act_bwd(const float* g, const float* y, const float* x, const float* rowmask, float slope, int32_t mode,
                 float* dz, void* plane, int64_t ld_plane, int32_t split, float* dbias, int32_t rows, int32_t c,
                 void* stream);
/* The residual layer's non-linearity when it is not (Leaky)ReLU: the reference builds getattr(torch.nn, nonlinear_activation)(**params)
 * into every ResConv1d and into the mel prenet (nntts/layers/efts_modules.py:32-35, nntts/models/efficient_tts.py:76-80).  The contraction then
 * writes the pre-activation z = conv(x) + bias in fp32 (efts_gemm with EFTS_ACT_NONE) and these two elementwise launches do the rest.
 * p0 / p1: the module's scalar parameters, in the order of the comments below (unused ones are ignored). */
#define EFTS_ACTFN_IDENTITY 0
#define EFTS_ACTFN_RELU 1
#define EFTS_ACTFN_LEAKY_RELU 2  /* p0 = negative_slope */
#define EFTS_ACTFN_ELU 3         /* p0 = alpha */
#define EFTS_ACTFN_CELU 4        /* p0 = alpha */
#define EFTS_ACTFN_SELU 5
#define EFTS_ACTFN_GELU 6        /* approximate="none" (erf) */
#define EFTS_ACTFN_GELU_TANH 7   /* approximate="tanh" */
#define EFTS_ACTFN_SILU 8
#define EFTS_ACTFN_MISH 9
#define EFTS_ACTFN_TANH 10
#define EFTS_ACTFN_SIGMOID 11
#define EFTS_ACTFN_SOFTPLUS 12   /* p0 = beta, p1 = threshold */
#define EFTS_ACTFN_HARDTANH 13   /* p0 = min_val, p1 = max_val (ReLU6 = 0, 6) */
#define EFTS_ACTFN_HARDSWISH 14
#define EFTS_ACTFN_HARDSIGMOID 15
#define EFTS_ACTFN_SOFTSIGN 16
#define EFTS_ACTFN_TANHSHRINK 17
#define EFTS_ACTFN_LOGSIGMOID 18
#define EFTS_ACTFN_COUNT 19
/* y = (resid + Dropout(f(z))) * rowmask over [rows][c] fp32 (c % 4 == 0): y_f32 and / or the operand plane (either may be NULL, not both);
 * resid, rowmask may be NULL; Dropout as in the argument block of efts_gemm: drop_p / drop_seed, element index row * c + col */
int efts_act_apply(const float* z, const float* resid, const float* rowmask, int32_t act, float p0, float p1, float* y_f32, void* plane,
                   int64_t ld_plane, int32_t split, int32_t rows, int32_t c, float drop_p, uint32_t drop_seed, void* stream);
/* dZ = G * rowmask * Dropout'(.) * f'(z): dz (fp32) and / or the operand plane; dbias (may be NULL) += column sums of dZ (atomic adds) */
int efts_act_grad(const float* g, const float* z, const float* rowmask, int32_t act, float p0, float p1, float* dz, void* plane,
                  int64_t ld_plane, int32_t split, float* dbias, int32_t rows, int32_t c, float drop_p, uint32_t drop_seed, void* stream);
/* transposed operand planes out_s[ch][t] = x[t + shift0 + s][ch], s = 0..nshift-1 (plane s at
 * plane + s*plane_stride bytes), K = t padded with zeros to kpad: the wgrad operands of all taps
 * from one pass over x */
int efts_pack_t(const float* x, int64_t ldx, void* plane, int64_t ld_plane, int64_t plane_stride, int32_t split,
                int32_t rows, int32_t c, int32_t shift0, int32_t nshift, int32_t kpad, void* stream);
/* Reduction of the transposed-plane path's split-K partials (taps x split-K efts_gemm launches over efts_pack_t planes: the path of shapes
 * the direct kernel's tiles do not fit): dW[co][ci][k] = sum_s part[k][s][co][ci]; with g != NULL also the weight-norm backward
 * (dv -> dw_or_dv, dg) of w = g v / ||v|| */
int efts_wgrad_reduce(const float* part, int32_t nsplit, const float* v, const float* g, float* dw_or_dv, float* dg,
                      int32_t cout, int32_t cin, int32_t taps, void* stream);
/* Direct weight gradients of k5 / k3 convolutions and Linears (taps 5, 3, 1) from the ROW-MAJOR bf16 planes (no transposed copies):
 *   dW[co][ci][k] = sum_t dZ[t][co] * X[t + k - (taps - 1) / 2][ci]      (autograd of the Conv1d in nntts/layers/efts_modules.py:48-51)
 * dz_plane / x_plane: operand planes of the row space, both of format `split` (1 = bf16, 2 = bf16x3 hi/lo) (row 0 pointers; >= 2 zero rows
 * before row 0 and >= 144 after `rows`), cout % 128 == 0, cin % 64 == 0.
 * The weight gradients of up to EFTS_WGRAD_MAX_ITEMS layers (same rows, cout, cin, taps and plane
 * format; e.g. the Conv1d of every ResConv1d of a `ResConvBlock`, nntts/layers/efts_modules.py:77-79 under
 * nntts/trainers/efficient_tts_trainer.py:146) in ONE launch.  The (layer, tile, 64-row step) space is dealt out to `workgroups`
 * workgroups (0: two per CU) as equal contiguous ranges (stream-K); every workgroup leaves one fp32 slab per tile its range touches
 * in `part` (efts_wgrad_grouped_part_bytes() bytes; -1: bad arguments), and efts_wgrad_reduce_grouped -- called with the SAME count, rows, cout, cin,
 * taps, split and workgroups -- adds a tile's slabs in a fixed order, then per layer the weight-norm backward (g != NULL) and the bias
 * gradient dbias[co] += sum_i bias_part[i][co], i < nparts (the workspace efts_act_bwd fills in EFTS_ACT_BWD_BIAS_PARTS mode, or a dgrad
 * launch's act_bwd_bias_part), in a fixed order.  The launch leaves its geometry behind the slabs (the last 64 bytes of `part`); a reduction whose
 * arguments do not match the launch that filled `part` writes NaN into every output instead of a sum over slabs that were never written. */
#define EFTS_WGRAD_MAX_ITEMS 8
typedef struct efts_wgrad_item {
    const void* dz_plane;   /* operand plane of dZ, row 0 */
    int64_t ldz;
    const void* x_plane;    /* operand plane of the layer's input, row 0 */
    int64_t ldx;
    const float* v;         /* weight_v [cout][cin][taps] (with g) or NULL */
    const float* g;         /* weight_g [cout] or NULL: no weight-norm backward, dw_or_dv receives dW */
    float* dw_or_dv;        /* [cout][cin][taps] */
    float* dg;              /* [cout] or NULL */
    const float* bias_part; /* [nparts][cout] column sums efts_act_bwd left (EFTS_ACT_BWD_BIAS_PARTS), or NULL */
    float* dbias;           /* [cout], += */
    int32_t nparts;
    int32_t reserved;
} efts_wgrad_item;
int64_t efts_wgrad_grouped_part_bytes(int32_t count, int32_t rows, int32_t cout, int32_t cin, int32_t taps, int32_t split,
                                      int32_t workgroups);
int efts_wgrad_tn_grouped(const efts_wgrad_item* items, int32_t count, float* part, int32_t rows, int32_t cout, int32_t cin,
                          int32_t taps, int32_t split, int32_t workgroups, void* stream);
int efts_wgrad_reduce_grouped(const efts_wgrad_item* items, int32_t count, const float* part, int32_t rows, int32_t cout,
                              int32_t cin, int32_t taps, int32_t split, int32_t workgroups, void* stream);
/* backward of [ReLU ->] LayerNorm [-> Linear(c,1)] (duration_predictor.py:57-77); accumulates
 * dgamma, dbeta, conv-bias grad (dbias), and with ddur != NULL the Linear's dw, db. */
int efts_layernorm_bwd(const float* x, const float* gamma, const float* beta, float eps, const float* dy,
                       const float* ddur, const float* w, const float* rowmask, float* dz, void* plane,
                       int64_t ld_plane, int32_t split, float* dgamma, float* dbeta, float* dbias, float* dw, float* db,
                       int32_t rows, int32_t c, float drop_p, uint32_t drop_seed, const uint32_t* drop_seed_add, void* stream);
/* alignment block backward (efficient_tts.py:287-398 under autograd): */
int efts_alpha_bwd(const float* ralpha, const float* dalpha, const float* e, const int32_t* text_len,
                   const int32_t* mel_len, float sigma, float* r_ws /* [B*T2] */, float* de, int32_t B, int32_t T1,
                   int32_t T2, void* stream);
int efts_e_bwd(const float* imv, const float* e, const float* de, const int32_t* text_len, const int32_t* mel_len,
               float sigma_e, float* stats_ws /* [2*B*T1] */, float* dpi, int32_t B, int32_t T1, int32_t T2,
               void* stream);
int efts_imv_bwd(const float* soft_idx, const float* imv, const float* dpi, const int32_t* text_len,
                 const int32_t* mel_len, float* dsoft_idx, int32_t B, int32_t T2, void* stream);
int efts_attn_bwd(const float* scores, int64_t ld, const float* soft_idx, const float* dsoft_idx,
                  const int32_t* text_len, const int32_t* mel_len, float* dscores, int64_t ldd, void* plane,
                  int64_t ld_plane, int32_t B, int32_t T1, int32_t T2, int32_t T2p, void* stream);
int efts_embed_bwd(const int64_t* ids, const float* g, float* dtable, int32_t B, int32_t T, int32_t Tp, int32_t c,
                   int32_t num_symbols, void* stream);
/* clip_grad_norm_ + torch.optim.Adam(amsgrad=True, coupled weight decay) on flat fp32 buffers
 * (trainer.py:154-158; YAML :34-40).  efts_sumsq ACCUMULATES sum(g^2) into *out1 (zero it first) with a
 * deterministic two-stage reduction (workspace >= efts_sumsq_workspace_bytes());
 * efts_adam_amsgrad scales g by gscale * min(1, max_norm / (gscale*sqrt(sumsq) + 1e-6)). */
size_t efts_sumsq_workspace_bytes(void);
int efts_sumsq(const float* g, int64_t n, float* out1, void* workspace, void* stream);
/* x[0..n) *= *scale (device scalar), skipped on the device when *scale == 1: the `grad_output` factor of
 * loss.backward() on the flat gradient buffer (torch autograd semantics of nntts/trainers/efficient_tts_trainer.py:150). */
int efts_scale_unless_one(float* x, int64_t n, const float* scale, void* stream);
int efts_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, const float* sumsq,
                      float max_norm, float gscale, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int32_t step, void* stream);
/* The same update with its per-step scalars in DEVICE memory -- hyper = {lr, 1 - beta1^step, sqrt(1 - beta2^step)}, as
 * efts_adam_hyper (host) computes them, bit for bit what efts_adam_amsgrad derives from its by-value arguments -- so that a whole
 * training step (nntts/trainers/efficient_tts_trainer.py:139-160) can be captured once and replayed as a hipGraph: the host only
 * refreshes the words in front of every replay (efts_store_words: up to 8 32-bit words passed by value, stream-ordered; the same
 * call carries the dropout step word of efts_layernorm_rows / _dot / _bwd `drop_seed_add`). */
int efts_adam_hyper(float lr, float beta1, float beta2, int32_t step, float* out3 /* host */);
int efts_adam_amsgrad_dev(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, const float* sumsq,
                          float max_norm, float gscale, const float* hyper, float beta1, float beta2, float eps,
                          float weight_decay, void* stream);
int efts_store_words(uint32_t* dst /* device */, const uint32_t* words /* host */, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Log-mel front-end: the data format right before the path.
 * Replaces nntts/datasets/meldataset.py:49-82 (mel_spectrogram, computed per item on CPU dataloader
 * workers by TextMelLoader.get_mel, taco2_data.py:66-76) and the mel half of TextMelCollate
 * (taco2_data.py:122-139) with a batched device pipeline:
 *   efts_frame_pack : audio fp32 [B][ld_audio] in [-1,1], per-item sample counts `lengths` ->
 *                     A operand plane rows (b*Tp + t), K = n_fft: frame t of item b =
 *                     reflect_pad(audio_b, (n_fft-hop)/2)[t*hop + k] * window[k]  (:69-73);
 *                     frames past L_b / hop and gap rows are zero.
 *   efts_gemm       : taps 1, against a B plane of the real DFT (rows 0..n_bins-1 = cos, rows
 *                     n_bins..2*n_bins-1 = -sin), fp32 output [rows][ld_spec >= 2*n_bins].
 *   efts_logmel     : out[b][t][m] = log(max(sum_k basis[m][k] * sqrt(re^2+im^2+1e-9), 1e-5)) for
 *                     t < frames[b], else 0 (:75-78, :27-28; zero padding after the log as the
 *                     collate does).  basis fp32 [n_mels][n_bins]; ranges int32 [n_mels][2] = the
 *                     [lo, hi) span of non-zero basis entries of each filter.  out is the contiguous
 *                     [B][T][n_mels] tensor EfficientTTSCNN.forward takes as `speech`.
 * ---------------------------------------------------------------------------------- */
int efts_frame_pack(const float* audio, int64_t ld_audio, const int32_t* lengths, const float* window, void* plane,
                    int64_t ld_plane, int32_t B, int32_t T, int32_t Tp, int32_t n_fft, int32_t hop, int32_t split,
                    void* stream);
int efts_logmel(const float* spec, int64_t ld_spec, const float* basis, const int32_t* ranges, const int32_t* frames,
                float* out, int32_t B, int32_t T, int32_t Tp, int32_t n_bins, int32_t n_mels, void* stream);
/* The same pipeline with the DFT split by decimation in time (round 6): `radix` interleaved sub-sequences x_p[j] = x[radix j + p] of
 * M = n_fft / radix samples each.
 *   efts_frame_pack_dit : as efts_frame_pack, but column p * M + j of a row holds sample radix * j + p of the frame, so that ONE batched
 *                         efts_gemm (batch = radix, a_batch_stride = the bytes of M columns, nchunk = the chunks of M columns, n = M,
 *                         out_batch_stride = M) against the B plane of the real M-point DFT -- rows 0 .. M/2: cos(2 pi g j / M), rows
 *                         M/2 + g, g = 1 .. M/2 - 1: -sin(2 pi g j / M) -- leaves, per row, `radix` blocks [re Y_p[0 .. M/2] | im Y_p[1 .. M/2 - 1]]:
 *                         radix times fewer FLOPs than the dense n_fft-point product.
 *   efts_logmel_dit     : X[f] = sum_p W^(p f) Y_p[f mod M] (Y_p[M - g] = conj Y_p[g]) in front of the magnitude, then as efts_logmel.
 *                         twiddle: fp32 [radix][n_bins][2] = (cos, sin)(2 pi p f / n_fft).  spec rows hold n_fft floats. */
int efts_frame_pack_dit(const float* audio, int64_t ld_audio, const int32_t* lengths, const float* window, void* plane,
                        int64_t ld_plane, int32_t B, int32_t T, int32_t Tp, int32_t n_fft, int32_t hop, int32_t split,
                        int32_t radix, void* stream);
int efts_logmel_dit(const float* spec, int64_t ld_spec, const float* basis, const int32_t* ranges, const int32_t* frames,
                    const float* twiddle, float* out, int32_t B, int32_t T, int32_t Tp, int32_t n_bins, int32_t n_mels, int32_t radix,
                    void* stream);

/* The whole front-end as ONE launch (round 6; n_fft 1024, hop 256, n_mels <= 80: the reference's configuration): audio in, log-mels out, the
 * STFT as an fp32 FFT in registers and LDS -- one wave per pair of neighbouring frames (two real frames = one complex 1024-point FFT, taken apart
 * by conjugate symmetry), no operand plane and no spectrum in memory.  Arguments as efts_frame_pack (audio, lengths, window) and efts_logmel
 * (basis, ranges, out); the frame count of item b is lengths[b] / hop; out[b][t][:] = 0 for t past it.  Other configurations: the three-launch
 * pipeline above. */
int efts_logmel_fft(const float* audio, int64_t ld_audio, const int32_t* lengths, const float* window, const float* basis,
                    const int32_t* ranges, float* out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream);
/* ... straight from int16 PCM, as TextMelLoader.get_mel reads a waveform (taco2_data.py:66-76): every sample is multiplied by pcm_scale
 * (1 / max_wav_value = 1 / 32768: exact in fp32) at the load -- the same bits as converting the batch first, without the conversion pass. */
int efts_logmel_fft_pcm16(const int16_t* audio, int64_t ld_audio, float pcm_scale, const int32_t* lengths, const float* window, const float* basis,
                          const int32_t* ranges, float* out, int32_t B, int32_t T, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream);

/* ------------------------------------------------------------------------------------
 * HiFi-GAN generator (nntts/vocoders/hifigan_model.py:95-136): every Conv1d /
 * ConvTranspose1d of it is an efts_gemm call (dilated taps 3 / 7 / 11, `plane_act` for the pre-activation
 * residual blocks, a stride-u transposed convolution = a 2-tap convolution with u * cout output columns,
 * EFTS_ACT_TANH for the output layer).  The one remaining elementwise step:
 * efts_mean_act_rows: m = (a + b + c) * scale (b, c optional); out (optional) = m; plane (optional) =
 *   operand plane of LeakyReLU(m, slope): the multi-receptive-field mean `xs / num_kernels` (:123-131)
 *   and the activation its consumer applies to its input.  a, b, c: fp32 [rows][ld], c % 4 == 0.
 * ---------------------------------------------------------------------------------- */
int efts_mean_act_rows(const float* a, const float* b, const float* c3, int64_t ld, float scale, float slope, float* out,
                       int64_t ldo, void* plane, int64_t ld_plane, int32_t split, int32_t rows, int32_t c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EFTS_ABI_H */
