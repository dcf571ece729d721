/*
 * ORACLE -- test infrastructure, NOT product code.
 *
 * Plain-C CPU restatement of the one hot path of benoriol/hand_pose_sl:
 *   ConvModel.forward          body2hand/src/models/HandPoseModels.py:40-64
 *   LinearPositionalEmbedding  body2hand/src/models/HandPoseModels.py:66-84
 *   item transforms            body2hand/src/steps/utils.py:180-210,261-277
 *   de-normalise / tail mask   body2hand/src/steps/traintest.py:387-388,
 *                              body2hand/src/steps/utils.py:309-312
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (hand_pose_sl_amd/)
 * never links, imports or falls back to it.
 *
 * Parity pin: tests/test_oracle.py checks every function below against the
 * golden vectors in tests/golden/ (*.npz), which were produced by importing the
 * reference's own classes (tests/golden/make_golden.py).  The convolution
 * arithmetic itself lives in torch.nn.Conv1d (cross-correlation, symmetric
 * zero padding, HandPoseModels.py:24-32); this file restates exactly that.
 *
 * Build: gcc -O2 -fPIC -shared -o _build/liboracle.so convmodel_oracle.c
 *        (no -ffast-math: the fp32 summation order below is part of the oracle)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KSIZE 5 /* kernel_size=5, padding=2: HandPoseModels.py:24-32 */
#define PAD 2
#define N_BODY 12 /* 12 body joints x (x,y): HandPoseModels.py:28 */
#define N_HAND 21 /* 2*21 output channels:   HandPoseModels.py:32 */

/* round-to-nearest-even fp32 -> bf16 -> fp32 (finite inputs) */
static float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return f; /* inf / nan unchanged */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

/* round-to-nearest-even fp32 -> fp16 -> fp32 (gcc 11 has no _Float16 on x86) */
static float f16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return f; /* inf / nan unchanged */
    float a = fabsf(f);
    if (a < 6.103515625e-05f) /* below 2^-14: fp16 subnormal grid of 2^-24 */
        return rintf(f * 16777216.0f) / 16777216.0f;
    u += 0xfffu + ((u >> 13) & 1u); /* keep 10 mantissa bits */
    u &= 0xffffe000u;
    memcpy(&f, &u, 4);
    if (fabsf(f) > 65504.0f) return f > 0 ? INFINITY : -INFINITY;
    return f;
}

static float quant(float f, int mode) {
    if (mode == 1) return bf16_round(f);
    if (mode == 2) return f16_round(f);
    return f;
}

/*
 * One Conv1d(k=5, padding=2) over the time axis, channel-last storage.
 *   in  : (T, cin)   out : (T, cout)   w : (cout, cin, 5)   b : (cout)
 *   out[t][o] = b[o] + sum_i sum_k w[o][i][k] * in[t+k-2][i],  in == 0 outside [0,T)
 * HandPoseModels.py:55-58 (torch Conv1d == cross-correlation, no kernel flip).
 * acc64 != 0 accumulates in double (used to measure error, not for parity).
 */
static void conv1d_k5(const float* in, float* out, int T, int cin, int cout,
                      const float* w, const float* b, int relu, int acc64) {
    for (int t = 0; t < T; ++t) {
        for (int o = 0; o < cout; ++o) {
            if (acc64) {
                double acc = b[o];
                for (int i = 0; i < cin; ++i)
                    for (int k = 0; k < KSIZE; ++k) {
                        int tt = t + k - PAD;
                        if (tt < 0 || tt >= T) continue;
                        acc += (double)w[(o * cin + i) * KSIZE + k] * (double)in[tt * cin + i];
                    }
                float r = (float)acc;
                out[t * cout + o] = (relu && r < 0.f) ? 0.f : r;
            } else {
                float acc = b[o];
                for (int i = 0; i < cin; ++i)
                    for (int k = 0; k < KSIZE; ++k) {
                        int tt = t + k - PAD;
                        if (tt < 0 || tt >= T) continue;
                        acc += w[(o * cin + i) * KSIZE + k] * in[tt * cin + i];
                    }
                out[t * cout + o] = (relu && acc < 0.f) ? 0.f : acc;
            }
        }
    }
}

/*
 * ConvModel.forward (HandPoseModels.py:40-64).
 *   x : (B, T, 12, 2) fp32 contiguous  == (B, T, 24) channel c = joint*2 + xy
 *       (permute(0,2,3,1)+view at :43-46 is a stride change only)
 *   y : (B, T, 21, 2) fp32 contiguous  == (B, T, 42)   (:60-62)
 *   w1: (C, 24 or 25, 5) b1: (C)  w2,w3: (C, C, 5)  w4: (42, C, 5)  b4: (42)
 *   pos_emb != 0: channel 0 of the layer-1 input is t/100 and the keypoints
 *       move to channels 1..24 (LinearPositionalEmbedding, :66-84); requires
 *       T == 100 like the reference (torch.cat would raise otherwise).
 *   mode : 0 = fp32 operands (the parity oracle)
 *          1 = bf16-rounded operands (input, weights, inter-layer activations),
 *              fp32 accumulate -- models what a bf16-MFMA kernel computes
 *          2 = fp16-rounded operands, fp32 accumulate
 *   acc64: accumulate in double.
 * Returns 0, or -1 bad argument, -2 T != 100 with pos_emb, -3 out of memory.
 */
int b2h_oracle_forward(const float* x, float* y, int B, int T, int C, int pos_emb,
                       const float* w1, const float* b1, const float* w2, const float* b2,
                       const float* w3, const float* b3, const float* w4, const float* b4,
                       int mode, int acc64) {
    if (B < 0 || T < 1 || C < 1 || !y) return -1;
    if (pos_emb && T != 100) return -2;
    const int cin1 = 2 * N_BODY + (pos_emb ? 1 : 0);
    const int cout4 = 2 * N_HAND;
    const int cmax = C > cin1 ? (C > cout4 ? C : cout4) : (cin1 > cout4 ? cin1 : cout4);
    float* h0 = (float*)malloc(sizeof(float) * (size_t)T * cmax);
    float* h1 = (float*)malloc(sizeof(float) * (size_t)T * cmax);
    size_t nw[4] = {(size_t)C * cin1 * KSIZE, (size_t)C * C * KSIZE, (size_t)C * C * KSIZE,
                    (size_t)cout4 * C * KSIZE};
    const float* wsrc[4] = {w1, w2, w3, w4};
    float* wq[4] = {0, 0, 0, 0};
    int rc = 0;
    if (!h0 || !h1) rc = -3;
    for (int l = 0; l < 4 && rc == 0; ++l) {
        wq[l] = (float*)malloc(sizeof(float) * nw[l]);
        if (!wq[l]) { rc = -3; break; }
        for (size_t i = 0; i < nw[l]; ++i) wq[l][i] = quant(wsrc[l][i], mode);
    }
    for (int b = 0; b < B && rc == 0; ++b) {
        const float* xb = x + (size_t)b * T * 2 * N_BODY;
        for (int t = 0; t < T; ++t) {
            float* row = h0 + (size_t)t * cin1;
            int c0 = 0;
            if (pos_emb) row[c0++] = quant((float)t / 100.0f, mode); /* :71-75 */
            for (int c = 0; c < 2 * N_BODY; ++c) row[c0 + c] = quant(xb[t * 2 * N_BODY + c], mode);
        }
        conv1d_k5(h0, h1, T, cin1, C, wq[0], b1, 1, acc64); /* :55 */
        for (size_t i = 0; i < (size_t)T * C; ++i) h1[i] = quant(h1[i], mode);
        conv1d_k5(h1, h0, T, C, C, wq[1], b2, 1, acc64);    /* :56 */
        for (size_t i = 0; i < (size_t)T * C; ++i) h0[i] = quant(h0[i], mode);
        conv1d_k5(h0, h1, T, C, C, wq[2], b3, 1, acc64);    /* :57 */
        for (size_t i = 0; i < (size_t)T * C; ++i) h1[i] = quant(h1[i], mode);
        conv1d_k5(h1, y + (size_t)b * T * cout4, T, C, cout4, wq[3], b4, 0, acc64); /* :58 */
    }
    for (int l = 0; l < 4; ++l) free(wq[l]);
    free(h0);
    free(h1);
    return rc;
}

/*
 * Item transforms in the order run.py:85-90,102 composes them, batched.
 *   body  : (B, T, 12, 2) raw pixel keypoints   (in)
 *   hand  : (B, T, 21, 2) raw right-hand pixels (in, may be NULL)
 *   flags : bit0 WristDifference  hand -= body[:,4]   (utils.py:194-201; uses the
 *                                  body BEFORE the chest shift, it runs first)
 *           bit1 ChestDifference  body -= body[:,1]   (utils.py:203-210)
 *           bit2 NormalizeFixedFactor: / factor       (utils.py:180-190)
 *   input_kp  : (B, T, 12, 2) out == item["input_kp"]  (BuildRightHandItem :261-277)
 *   target_kp : (B, T, 21, 2) out == item["target_kp"] (NULL if hand is NULL)
 */
int b2h_oracle_preprocess(const float* body, const float* hand, float* input_kp, float* target_kp,
                          int B, int T, int flags, float factor) {
    if (B < 0 || T < 0 || !body || !input_kp) return -1;
    for (size_t f = 0; f < (size_t)B * T; ++f) {
        const float* bf = body + f * 2 * N_BODY;
        float wrist[2] = {bf[4 * 2 + 0], bf[4 * 2 + 1]};
        float chest[2] = {bf[1 * 2 + 0], bf[1 * 2 + 1]};
        if (hand && target_kp) {
            for (int j = 0; j < N_HAND; ++j)
                for (int d = 0; d < 2; ++d) {
                    float v = hand[f * 2 * N_HAND + j * 2 + d];
                    if (flags & 1) v = v - wrist[d];
                    if (flags & 4) v = v / factor;
                    target_kp[f * 2 * N_HAND + j * 2 + d] = v;
                }
        }
        for (int j = 0; j < N_BODY; ++j)
            for (int d = 0; d < 2; ++d) {
                float v = bf[j * 2 + d];
                if (flags & 2) v = v - chest[d];
                if (flags & 4) v = v / factor;
                input_kp[f * 2 * N_BODY + j * 2 + d] = v;
            }
    }
    return 0;
}

/*
 * Post-processing of a prediction (B, T, 21, 2), in place:
 *   pred *= factor                         traintest.py:270-271,387-388
 *   pred[i, n_frames[i]:, :] = 0           utils.py:309-312 (n_frames may be NULL)
 */
int b2h_oracle_postprocess(float* pred, int B, int T, float factor, const int64_t* n_frames) {
    if (B < 0 || T < 0 || !pred) return -1;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            float* row = pred + ((size_t)b * T + t) * 2 * N_HAND;
            int dead = n_frames && (int64_t)t >= n_frames[b];
            for (int c = 0; c < 2 * N_HAND; ++c) row[c] = dead ? 0.f : row[c] * factor;
        }
    return 0;
}

/*
 * maskedPoseL1.forward (steps/utils.py:413-428): per sequence the mean of |pred - target|
 * over its first n_frames[b] frames (x 21 joints x 2), then the mean over the batch.
 * Accumulates in double (the reference averages fp32 with torch's pairwise sums; the two
 * agree to fp32 rounding).  per_seq (B) may be NULL.  n_frames[b] == 0 gives NaN like torch.
 */
int b2h_oracle_masked_l1(const float* pred, const float* target, const int64_t* n_frames, int B,
                         int T, float* per_seq, float* loss) {
    if (B < 1 || T < 1 || !pred || !target || !loss) return -1;
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        int64_t n = n_frames ? n_frames[b] : T;
        if (n < 0) n = 0;
        if (n > T) n = T;
        double acc = 0.0;
        const size_t base = (size_t)b * T * 2 * N_HAND, cnt = (size_t)n * 2 * N_HAND;
        for (size_t i = 0; i < cnt; ++i) acc += fabs((double)pred[base + i] - (double)target[base + i]);
        const double mean = acc / (double)cnt; /* 0/0 -> NaN */
        if (per_seq) per_seq[b] = (float)mean;
        total += mean;
    }
    *loss = (float)(total / B);
    return 0;
}

/* poderatedPoseL1 (`--loss confL1`, steps/utils.py:431-452):
 *   loss = sum_i mean(|pred[i,:n_i] * s[i,:n_i,:,None] - target[i,:n_i] * s[i,:n_i,:,None]|)
 * -- a SUM over the batch, the class does not divide by B.  scores (B, T, 21) weight both coordinates
 * of a joint; the two products are rounded to fp32 before the subtraction, as torch does.
 * Accumulates in double.  per_seq (B) may be NULL.  n_frames[b] == 0 gives NaN like torch.
 */
int b2h_oracle_weighted_l1(const float* pred, const float* target, const float* scores, const int64_t* n_frames,
                           int B, int T, float* per_seq, float* loss) {
    if (B < 1 || T < 1 || !pred || !target || !scores || !loss) return -1;
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        int64_t n = n_frames ? n_frames[b] : T;
        if (n < 0) n = 0;
        if (n > T) n = T;
        double acc = 0.0;
        const size_t base = (size_t)b * T * 2 * N_HAND, cnt = (size_t)n * 2 * N_HAND;
        const size_t sbase = (size_t)b * T * N_HAND;
        for (size_t i = 0; i < cnt; ++i) {
            const float s = scores[sbase + i / 2];
            const float ps = pred[base + i] * s, ts = target[base + i] * s;
            acc += fabs((double)(float)(ps - ts));
        }
        const double mean = acc / (double)cnt; /* 0/0 -> NaN */
        if (per_seq) per_seq[b] = (float)mean;
        total += mean;
    }
    *loss = (float)total;
    return 0;
}
