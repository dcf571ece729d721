// Shared definitions of the gfx950 body->hand kernels (device + host).
//
// Path: ConvModel.forward, body2hand/src/models/HandPoseModels.py:40-64 of
// benoriol/hand_pose_sl: four Conv1d(k=5, padding=2) over the time axis,
// 24(25) -> C -> C -> C -> 42 channels, ReLU after the first three.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace b2h {

constexpr int kTaps = 5;       // kernel_size      (HandPoseModels.py:24-32)
constexpr int kPad = 2;        // padding
constexpr int kInCh = 24;      // 12 joints x (x,y) (HandPoseModels.py:28)
constexpr int kOutCh = 42;     // 21 joints x (x,y) (HandPoseModels.py:32)
constexpr int kHalo = 8;       // receptive field 17 frames = +-8
constexpr int kMaxWidth = 128;     // conv_channels supported at all (exact fp32 VALU kernel; its LDS tile fills at 128)
constexpr int kMfmaWideWidth = 64; // conv_channels supported by the matrix-core kernels (33..64: the wide variants)
constexpr int kMfmaWidth = 32;     // conv_channels of the narrow matrix-core geometry

// Pre/post-processing fused around the stack (values match include/b2h.h).
constexpr int kPreChest = 1, kPreNorm = 2, kPostDenorm = 4, kPostMask = 8;

struct FusedArgs {
    int flags;               // 0 = plain ConvModel.forward
    float factor;            // 1280 in the reference
    const int64_t* n_frames; // (B) or nullptr
};

// ---- fp32 VALU kernel ------------------------------------------------------
// Per layer: w[k][i][opad] (out-channel fastest, opad = cout rounded up to 8),
// b[opad].
struct ValuLayer {
    const float* w;
    const float* b;
    int cin, cout, opad;
};
struct ValuParams {
    ValuLayer L[4];
    int act_stride; // floats per LDS activation row (odd)
    int wbuf_floats;
    int pos_emb;
};

// ---- MFMA kernels ----------------------------------------------------------
// Channels are padded to 32 per layer input (one 16x16x32 k-step = one tap).
// A operand = weights in fragment order, B operand = activations (time on the
// MFMA column), so the result tile has out-channels in registers and time on
// lanes.  Out-channel slot -> channel map (`chan_of`):
//   hidden layers: M-tile mt, row 4q+r  <->  channel 8q + 4mt + r
//                  (a lane then owns 8 consecutive channels = one 16-B chunk)
//   last layer   : M-tile mt, row       <->  channel 16mt + row
struct MfmaParams {
    // bf16/f16: [mt][tap][lane] x 16 B (8 elements: in-channels 8(lane>>4)+j)
    // f32     : [mt][tap][g][lane] x 16 B (4 floats: in-channels 16g+4(lane>>4)+j)
    const void* w[4];
    const float* bias[4]; // [mt][q][4] fp32
    int pos_emb;          // layer-1 in-channel 24 carries t/100 (weights permuted)
};

__host__ __device__ inline int hidden_chan_of(int mt, int row) { return 8 * (row >> 2) + 4 * mt + (row & 3); }
__host__ __device__ inline int last_chan_of(int mt, int row) { return 16 * mt + row; }

} // namespace b2h
