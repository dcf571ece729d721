/*
 * b2h.h -- C ABI of libb2h.so, the MI355X (gfx950) body->hand keypoint path.
 *
 * The reference (benoriol/hand_pose_sl) has no FFI: its boundary for this path
 * is the Python duck type of `ConvModel` (body2hand/src/models/HandPoseModels.py
 * :17-64).  Each entry point below names the reference interface it replaces;
 * the Python mirror that binds them with ctypes is hand_pose_sl_amd/conv_model.py
 * and INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*, NULL = the null stream);
 *   - every function returns B2H_OK (0) or a negative b2h_status; the message of
 *     the last failure on the calling thread is b2h_last_error();
 *   - "device pointer" = memory of the current HIP device (hipMalloc or a
 *     PyTorch-ROCm tensor's data_ptr()); the library never frees caller memory;
 *   - launches are asynchronous on `stream`; nothing here synchronises the
 *     device except b2h_load_weights (a one-off staging copy) and b2h_stream_sync.
 */
#ifndef B2H_H_
#define B2H_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define B2H_VERSION 100 /* 0.1.0 */

typedef enum b2h_status {
    B2H_OK = 0,
    B2H_ERR_INVALID = -1,     /* bad argument (the Python mirror raises ValueError)   */
    B2H_ERR_SHAPE = -2,       /* shape the model cannot take (RuntimeError)           */
    B2H_ERR_NO_WEIGHTS = -3,  /* forward before b2h_load_weights                      */
    B2H_ERR_HIP = -4,         /* a HIP runtime call failed                            */
    B2H_ERR_NO_DEVICE = -5,   /* no gfx950 device visible                             */
    B2H_ERR_UNSUPPORTED = -6  /* kernel variant cannot run this configuration         */
} b2h_status;

/* Which hand-written kernel computes the four-layer stack. */
typedef enum b2h_kernel {
    B2H_KERNEL_AUTO = 0,      /* the faster exact-fp32 kernel for the model's width: F32_MFMA, except
                                 F32_VALU at conv_channels <= 8 and 33..39 (measured crossovers) and above
                                 64 (the only kernel there) */
    B2H_KERNEL_F32_VALU = 1,  /* fp32 FMA on the vector ALU; any conv_channels <= 128 (cross-check kernel,
                                 and the whole path for 65..128 channels) */
    B2H_KERNEL_F32_MFMA = 2,  /* exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32); any conv_channels <= 64
                                 (33..64: a one-wave-per-SIMD wide variant) */
    B2H_KERNEL_BF16_MFMA = 3, /* bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16); any
                                 conv_channels <= 64 (33..64: the wide kernel, two k-steps per tap) */
    B2H_KERNEL_F16_MFMA = 4,  /* fp16 operands, fp32 accumulate (v_mfma_f32_16x16x32_f16); <= 64 likewise */
    B2H_KERNEL_F16X3_MFMA = 5 /* any conv_channels <= 64 (33..64: a one-wave-per-SIMD wide variant).
                                 fp32-grade: every operand split into f16 hi + lo, three f16 MFMAs per
                                 product (hi.hi + hi.lo + lo.hi), fp32 accumulate; needs |x| < 65504:
                                 a model with a weight outside that range is refused
                                 (B2H_ERR_UNSUPPORTED), an activation beyond it becomes inf / NaN */
} b2h_kernel;

/* Pre/post-processing fused around the stack (b2h_forward_fused). */
enum {
    B2H_PRE_CHEST_DIFF = 1,   /* body -= body[:, 1]  ChestDifference, steps/utils.py:203-210 */
    B2H_PRE_NORMALIZE = 2,    /* body /= factor      NormalizeFixedFactor, steps/utils.py:180-190 */
    B2H_POST_DENORMALIZE = 4, /* pred *= factor      steps/traintest.py:270-271,387-388 */
    B2H_POST_MASK_TAIL = 8    /* pred[i, n_frames[i]:] = 0   mask_output, steps/utils.py:309-312 */
};

typedef struct b2h_model b2h_model; /* opaque; owns the packed device weights */

/* Library / device ------------------------------------------------------- */

int b2h_version(void);
/* 0 for the shipped library.  Non-zero = a development build with parts of a kernel removed or instrumented
 * (csrc/dev/b2h_dev.h, B2H_ABLATE): its results may be WRONG by construction; the Python binding refuses to
 * load such a library unless B2H_ALLOW_ABLATE=1 is set (the measurement scripts under tools/ set it). */
int b2h_build_flags(void);
const char* b2h_last_error(void);
/* Number of visible HIP devices whose arch is gfx950 (0 when none). */
int b2h_device_count(void);

/* Model lifetime ---------------------------------------------------------
 * Replaces ConvModel.__init__(conv_channels, activation, pos_emb)
 * (HandPoseModels.py:18-37).  `activation` must be "ReLU" (B2H_ERR_INVALID
 * otherwise, mirroring the ValueError at :34-37).  1 <= conv_channels <= 128 (the reference's
 * --conv-channels is a free integer, default 30: run.py:37); the matrix-core kernels cover 1..64.
 * The model is bound to the HIP device current at creation; b2h_forward / b2h_forward_fused (and
 * b2h_tenc_forward for its model) return B2H_ERR_INVALID when called while another device is current. */
int b2h_create(int conv_channels, const char* activation, int pos_emb, b2h_model** out);
int b2h_destroy(b2h_model* m);

/* Replaces model.load_state_dict(...) (infer_utterance.py:109,
 * infer_utterance_h5.py:113, steps/traintest.py:62).  Tensors are fp32,
 * contiguous, in the reference's state_dict layout:
 *   w1 (C, 24|25, 5)  b1 (C)   w2, w3 (C, C, 5)  b2, b3 (C)   w4 (42, C, 5)  b4 (42)
 * `on_device` != 0: the eight pointers are device pointers, else host pointers.
 * Repacks into the kernels' fragment layouts (fp32 / bf16 / fp16) and uploads;
 * synchronous.  May be called again to replace the weights. */
int b2h_load_weights(b2h_model* m, const float* w1, const float* b1, const float* w2,
                     const float* b2, const float* w3, const float* b3, const float* w4,
                     const float* b4, int on_device);

/* Replaces ConvModel.forward(inp) (HandPoseModels.py:40-64), inference only.
 *   x : device, fp32, (B, T, 12, 2) contiguous  -- read only
 *   y : device, fp32, (B, T, 21, 2) contiguous  -- written (value-identical to
 *       the reference's non-contiguous view, :60-62)
 * T >= 1; pos_emb models require T == 100 (B2H_ERR_SHAPE, as torch.cat raises
 * in the reference, :78-84).  B == 0 is a no-op.  x and y must be 16-byte aligned and must
 * not overlap (B2H_ERR_INVALID otherwise). */
int b2h_forward(b2h_model* m, const float* x, float* y, int64_t B, int64_t T, int kernel,
                void* stream);

/* Forward with the reference's item transforms and de-normalisation fused in
 * (run.py:85-90,102 order):
 *   body : device fp32 (B, T, 12, 2) raw pixel keypoints
 *   y    : device fp32 (B, T, 21, 2)
 *   flags: OR of B2H_PRE_* / B2H_POST_*;  factor: 1280 in the reference
 *   n_frames: device int64 (B) valid lengths, or NULL (required by MASK_TAIL) */
int b2h_forward_fused(b2h_model* m, const float* body, float* y, int64_t B, int64_t T,
                      int flags, float factor, const int64_t* n_frames, int kernel,
                      void* stream);

/* Target transform of the training item (right hand relative to the wrist):
 *   hand_out = (hand - body[:, 4]) / factor     WristDifference + Normalize,
 * steps/utils.py:194-201,180-190.  flags: bit0 wrist diff, bit1 normalize.
 *   body (B,T,12,2), hand / hand_out (B,T,21,2), all device fp32. */
int b2h_target_transform(const float* body, const float* hand, float* hand_out, int64_t B,
                         int64_t T, int flags, float factor, void* stream);

/* Evaluation metric of the reference: maskedPoseL1 (steps/utils.py:413-428) --
 *   loss = mean_i( mean(|pred[i, :n_frames[i]] - target[i, :n_frames[i]]|) ),  i < B.
 * pred, target: device fp32 (B, T, 21, 2); n_frames: device int64 (B) or NULL (= T);
 * per_seq: device fp32 (B) scratch that receives the per-sequence means; loss: device fp32 (1).
 * A sequence with n_frames 0 contributes NaN, as torch's mean of an empty tensor does.
 * "Pixel distance" (L12Pixels, steps/utils.py:291-299) is loss / 21 * 1280 on the host. */
int b2h_masked_l1(const float* pred, const float* target, const int64_t* n_frames, int64_t B,
                  int64_t T, float* per_seq, float* loss, void* stream);

/* The evaluation loop's other loss, `--loss confL1` = poderatedPoseL1 (steps/utils.py:431-452;
 * chosen at traintest.py:41-42,207-208):
 *   loss = sum_i( mean(|pred[i, :n_i] * s[i, :n_i, :, None] - target[i, :n_i] * s[i, :n_i, :, None]|) )
 * -- a SUM over the batch (the class does not divide by B).  scores: device fp32 (B, T, 21), the
 * OpenPose confidence of each target joint; everything else as b2h_masked_l1. */
int b2h_weighted_l1(const float* pred, const float* target, const float* scores, const int64_t* n_frames,
                    int64_t B, int64_t T, float* per_seq, float* loss, void* stream);

/* TransformerEnc -------------------------------------------
 * The reference's second text-free body->hand model, `TransformerEnc(ninp, nhead, nhid, nout,
 * nlayers, dropout)` (HandPoseModels.py:118-178), as its CLIs build it: ninp = 24, nhead = 4,
 * nhid = 128, nout = 42 (infer_utterance.py:99-101).  Inference only (dropout = identity).
 * b2h_tenc_create accepts exactly that geometry, 1 <= nlayers <= 16, 1 <= max_len <= 128
 * (100 in the reference, :125) and returns B2H_ERR_UNSUPPORTED for anything else. */
typedef struct b2h_tenc b2h_tenc;
/* Arithmetic of the Linear layers (attention, softmax and LayerNorm are fp32 in both):
 *   B2H_TENC_F32   fp32 operands on v_mfma_f32_16x16x4_f32 (default);
 *   B2H_TENC_F16X3 every operand split into f16 hi + lo, three v_mfma_f32_16x16x32_f16 per product
 *                  (hi.hi + hi.lo + lo.hi, fp32 accumulate): fp32-grade error (22 significant
 *                  bits per operand) at 3/16 of the matrix cycles, valid while every activation
 *                  and weight is below 65504 in magnitude (f16 range): b2h_tenc_forward returns
 *                  B2H_ERR_UNSUPPORTED for a model with a parameter outside it. */
typedef enum b2h_tenc_kernel { B2H_TENC_F32 = 0, B2H_TENC_F16X3 = 1 } b2h_tenc_kernel;
int b2h_tenc_create(int ninp, int nhead, int nhid, int nout, int nlayers, int max_len, b2h_tenc** out);
int b2h_tenc_destroy(b2h_tenc* m);
/* Selects the kernel for later b2h_tenc_forward calls (no reload of the weights needed). */
int b2h_tenc_set_kernel(b2h_tenc* m, int kernel);
/* Replaces load_state_dict.  `tensors`: 5 + 12*nlayers fp32 contiguous arrays in this order
 * (state_dict names of the reference):
 *   pos_encoder.pe (max_len,1,24); pose2hidden_projection.weight (128,24), .bias (128);
 *   per layer i, transformer_encoder.layers.i.: self_attn.in_proj_weight (384,128),
 *     self_attn.in_proj_bias (384), self_attn.out_proj.weight (128,128), self_attn.out_proj.bias,
 *     linear1.weight (128,128), linear1.bias, linear2.weight (128,128), linear2.bias,
 *     norm1.weight, norm1.bias, norm2.weight, norm2.bias (128 each);
 *   hidden2pose_projection.weight (42,128), .bias (42). */
int b2h_tenc_load_weights(b2h_tenc* m, const float* const* tensors, int count, int on_device);
/* Bytes of device scratch b2h_tenc_forward needs for a (B, T) batch (2560 B per frame). */
size_t b2h_tenc_workspace_bytes(const b2h_tenc* m, int64_t B, int64_t T);
/* Replaces TransformerEnc.forward(src) (HandPoseModels.py:152-178): x (B,T,12,2) -> y (B,T,21,2),
 * device fp32.  T <= max_len (the reference's `src + pe[:T]` raises beyond it): B2H_ERR_SHAPE.
 * `workspace`: device memory of at least b2h_tenc_workspace_bytes(m, B, T), 16-byte aligned. */
int b2h_tenc_forward(b2h_tenc* m, const float* x, float* y, int64_t B, int64_t T, void* workspace,
                     size_t workspace_bytes, void* stream);

/* TransformerEnc forward with the reference's item transforms fused into its first and last kernel
 * (SURVEY.md 8f N1 for the second model; order of run.py:85-90,102 and traintest.py:270-271):
 *   body: device fp32 (B, T, 12, 2) raw pixel keypoints; flags / factor / n_frames as for
 *   b2h_forward_fused -- B2H_PRE_CHEST_DIFF and B2H_PRE_NORMALIZE are applied to the rows as they
 *   enter the model, BEFORE the positional encoding is added (HandPoseModels.py:167);
 *   B2H_POST_DENORMALIZE and B2H_POST_MASK_TAIL in the store of hidden2pose_projection's output. */
int b2h_tenc_forward_fused(b2h_tenc* m, const float* body, float* y, int64_t B, int64_t T, int flags,
                           float factor, const int64_t* n_frames, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Introspection / measurement -------------------------------------------- */

/* conv_channels, pos_emb and whether weights are loaded. */
int b2h_model_info(const b2h_model* m, int* conv_channels, int* pos_emb, int* has_weights);
/* 1 if `kernel` can run this model (width, pos_emb), else 0. */
int b2h_kernel_supported(const b2h_model* m, int kernel);
/* Name of the __global__ function `kernel` resolves to for this model (for
 * matching rocprofv3 kernel-trace rows); static storage.  The plain (unfused) instantiation; for the
 * persistent 16-bit kernel the streaming one, b2h_fwd_mfma16<PREC, false, true>, which every launch of
 * >= 1 MiB of traffic runs (smaller launches run its <PREC, false, false> twin: default cache policy). */
const char* b2h_kernel_name(const b2h_model* m, int kernel);
/* Time `iters` back-to-back launches of b2h_forward on `stream` with HIP events
 * recorded on that stream; returns the average milliseconds per launch. */
int b2h_time_forward(b2h_model* m, const float* x, float* y, int64_t B, int64_t T, int kernel,
                     int iters, void* stream, float* avg_ms);
int b2h_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* B2H_H_ */
