// bf16 training path of the width-256 MLP on the register engine: shared constants.
//
// Forward (mlp_engine.hip, TRAIN): the inference kernel, which additionally writes every layer's
// output operand tiles and ReLU masks to the activation stash.  Backward (mlp_train16.hip): a data
// kernel on the same engine with the transposed weight stream, a modulation kernel, and one
// weight-gradient kernel that contracts stash tiles over the samples.
//
// Stash of a batch of M samples: blocks of 32 samples (two column blocks of 16), padded to the 8 blocks of a
// workgroup pass.  Per block:
//   tiles [kStashTiles][2][64 lanes] x 16 B   operand layout (lane = sample column + 16 x group, 8 bf16 =
//                                             positions 8g..8g+7 of the k-tile; mlp_plan.h ORDER_ACC)
//   masks [kStashMasks][2][64 lanes] x 8 B    bit 8 * k-tile + element = "output was > 0"
#pragma once
namespace zest {
// 8 trunk layers x 8 k-tiles, feature_linear 8, view layer 4; then the encoder's operands as the forward pass
// built them (the weight kernel contracts them like any other layer input): points (up to 3 k-tiles), features
// (up to 2), directions (1)
constexpr int kStashTiles = 82;
constexpr int kStashPts = 76, kStashFeat = 79, kStashViews = 81;
constexpr int kStashMasks = 9;        // 8 trunk layers, view layer
constexpr int kTrainBlock = 32;       // samples per block
inline long long train_blocks(long long M) { return (M + 255) / 256 * 8; }
}
