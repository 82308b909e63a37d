// Host-side plan of the width-256 NeRF MLP as the MFMA kernels execute it.
//
// The network (reference networks.py:150-221) is a sequence of D + 4 "ops" (12 at the default depth 8),
// each a Linear producing NJB row-blocks of 32 output features:
//   0..D-1  trunk layers  h = relu((W h + b) (*|+) m),  m = pts_bias(feats) recomputed per tile
//   D       head tile     rows: alpha, then w | sf(6) prob(2)      (on the trunk output)
//   D+1     feature_linear (no activation)
//   D+2     views_linears.0 on [feature | PE(dir)], relu
//   D+3     rgb_linear
// The ORDER_ACC engine and everything below "bf16 training path" are written for D = 8, W = 256,
// skips = [4] (their op ids 8..11 are literal); other depths / widths / skips exist in ORDER_NATURAL only.
// Work is done transposed, Y^T = W X^T: samples sit on the MFMA column/lane axis and the
// output features of a tile in its accumulator registers, so a tile's activations feed the
// next op's B operand without leaving the lane.
//
// fp32 kernel (ORDER_NATURAL).  Operands are sequences of SLOTS.  A slot holds one value per
// lane, i.e. two features: feature(slot, half) for the lane halves 0-31 / 32-63, which both own
// sample lane&31.  A TILE is 1 KiB of packed weights covering 4 slots (four 32x32x2 MFMAs):
// lane l, element e holds W[row0 + (l & 31)][col(feature(tile*4 + e, l >> 5))] (0 outside).
//
// bf16 engine (ORDER_ACC), built on v_mfma_f32_16x16x32_bf16 (measured ~5-15 % more FLOP/s than
// the 32x32x16 shape on this power-limited kernel at identical operand traffic): an operand is a
// sequence of k-tiles of 32 POSITIONS; lane l (sample column l & 15, group g = l >> 4) holds
// positions 8g .. 8g+7 of each k-tile as one bf16x8.  A stream unit is the 16 x 32 weight tile
// of one MFMA A operand: lane l, element e holds W[row0 + (l & 15)][feature(kt*32 + 8g + e)].
// A row block of 32 outputs = two row tiles; a finished pair of 16x16 accumulator tiles
// (lane: rows 4g..4g+3 of each) is, rounded to bf16, exactly one k-tile of the next layer.
// The stream is laid out in consumption order: for op, for row-block jb:
// [header] [modulation tiles over the feature operand] then one run of tiles per operand
// segment (ORDER_ACC: per k-tile the row tiles 0, 1).  A bias block is 128 bytes: ORDER_NATURAL
// the accumulator initialiser [half][16] = bias[32 jb + (i&3) + 8 (i>>2) + 4 half]; ORDER_ACC
// bias[32 jb .. 32 jb + 31] in order.
//   ORDER_NATURAL (fp32 kernel): no headers; blocks live in a separate bias area in front of
//     the tiles, the eight modulation blocks first, then one per (op, jb).
//   ORDER_ACC (bf16 engine): the header is one 1 KiB unit in the stream itself holding the
//     (op, jb) bias block at byte 0 and the modulation bias block of jb at byte 128, so the
//     stream is self-contained and can be fed through an LDS ring; it is padded with zero
//     units to a multiple of kStreamAlign units so a ring of that many units keeps a fixed
//     phase from one pass to the next.
// Operand types of the engine (include/zest_render.h): bf16, fp16 (same stream layout, 2-byte
// elements) and the split fp16 pair ZEST_PREC_F16X3: there every weight tile is TWO consecutive
// units, hi = fp16(w) and lo = fp16((w - hi) * 2^11) (MlpPlan::parts = 2; the tile counts nt_*
// stay logical, unit counts double), and the kernel forms a product as
// hi*hi + 2^-11 (hi*lo + lo*hi).
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/zest_render.h"

#ifdef __HIPCC__
#define ZEST_HD __host__ __device__
#else
#define ZEST_HD
#endif

namespace zest {

// ---- feature operand of the register engine (ORDER_ACC): which four input columns a QUAD holds ----------------
// The operand is a sequence of k-tiles of 8 quads (4 channels each); lane group g holds quads 8 kt + 2 g and
// 8 kt + 2 g + 1 of k-tile kt, so the four groups fill the operand in ROUNDS r = 2 kt + j of four quads (j = q & 1),
// one per group - and in the fused renderer a round is one gather step that all 64 lanes execute together.  The
// quads are therefore dealt so that the lanes of a round do the SAME kind of work: source view v of the V views
// (input columns 8 + 4 v ..) sits in round v / 4 at group v % 4 for the first 4 rv views; the two channel quads of the
// encoding volume (columns 0-3, 4-7) sit in round rv at groups 0 and 1, next to at most two more views at groups 2
// and 3 (rv = V / 4, or V / 4 + 1 when three views would be left over).  8 views: two rounds of four bilinear
// colour taps, one round with the trilinear lookup alone, nothing in the fourth (rounds 1-2 dealt the quads
// [vol | view] [view] [view] [view]: five serial steps, two of them for views that do not exist).
// -> first input column of quad q (channels col .. col + 3), or -1 for an all-zero quad.  One definition for the plan
// builder (weight packing), the operand loaders of the standalone / training kernels and the fused gather.
ZEST_HD inline int feat_volume_round(int V) { return V % 4 <= 2 ? V / 4 : V / 4 + 1; }
ZEST_HD inline int feat_quad_col(int q, int V) {
    const int r = 2 * (q >> 3) + (q & 1), g = (q & 7) >> 1, rv = feat_volume_round(V);
    if (r > rv) return -1;
    if (r == rv && g < 2) return 4 * g;
    const int view = r < rv ? 4 * r + g : 4 * rv + g - 2;
    return view < V ? 8 + 4 * view : -1;
}

constexpr int kW = 256;            // default trunk width (every shipped config: netwidth = 256), the engine's only one
constexpr int kNumOps = 12;        // op slots: depth <= 8
constexpr int kMaxSeg = 2;
constexpr int kStreamAlign = 128;  // units (KiB): ring size the bf16 stream is padded to

// slot-order conventions for operands that are produced from accumulator tiles
enum SlotOrder {
    ORDER_ACC = 0,      // position 32kt+8g+e <-> feature 32kt + (e<4 ? 4g+e : 16+4g+e-4)   (register engine)
    ORDER_NATURAL = 1,  // tile of 4 slots over features k0..k0+7: slot j <-> k0 + 4h + j
};

enum SegKind { SEG_PTS = 0, SEG_FEAT = 1, SEG_VIEWS = 2, SEG_H = 3 };

struct SegPlan {
    int kind;      // SegKind: which operand the tiles multiply
    int ntiles;    // tiles in this segment
};

struct OpPlan {
    int njb;                 // row blocks of 32 outputs
    int nseg;
    SegPlan seg[kMaxSeg];
    int mod;                 // 1: modulated trunk layer
    int relu;                // 1: relu in the epilogue
    int tile_base;           // first unit of the op in the stream
    int tiles_per_jb;        // including header and modulation tiles
    int bias_block;          // first bias block (128 B units)
};

// desc->depth / width / skip_mask resolved (all zero = 8 / 256 / skips [4]) and range-checked
struct MlpShape {
    int D, W, skip_mask;
    bool is_default;
};
bool mlp_shape(const zest_mlp_desc &d, MlpShape *s, const char **err);
// leading dimension (input width) of every ZEST_P_* weight
void mlp_param_ld(const zest_mlp_desc &d, const MlpShape &s, int *ld);

struct MlpPlan {
    zest_mlp_desc desc;
    MlpShape shape;
    int n_ops;                     // shape.D + 4
    int precision, order;
    int spt;                       // slots per tile
    int ns_pts, ns_feat, ns_views; // padded slot counts of the encoder operands
    int nt_pts, nt_feat, nt_views, nt_h, nt_h128;
    int tail_unit0;                // backward stream: first unit of the modulation^T tail (after the padded ring part)
    int headers;                   // 1: ORDER_ACC stream with inline header units
    int parts;                     // stream units per weight tile: 1, or 2 (hi, lo) for ZEST_PREC_F16X3
    int n_tiles, n_bias_blocks;    // n_tiles includes headers and tail padding
    size_t bias_bytes, bytes;      // bias area (padded to 1 KiB) and total
    OpPlan op[kNumOps];
    // feature index (column within the operand's own input range), -1 = pad:
    // ORDER_NATURAL per slot and half [ns][2], ORDER_ACC per position [ns]
    std::vector<int16_t> map_pts, map_feat, map_views;
    // gather tables for the packer: per packed weight element / bias float, source
    // (param_slot << 24 | element offset), 0xFFFFFFFF = zero
    std::vector<uint32_t> tile_src, bias_src;
    // headers mode: per unit, 64 fp32 sources of the header payload (only header units are set)
    std::vector<uint32_t> hdr_src;
    // per unit: 0 = plain / hi tile, 1 = lo tile of a split pair (header and padding units: 0)
    std::vector<uint8_t> unit_part;
};

inline bool prec_is_engine(int precision) {     // the register engine's operand types (ORDER_ACC)
    return precision == ZEST_PREC_BF16 || precision == ZEST_PREC_F16 || precision == ZEST_PREC_F16X3;
}

// ---- bf16 training path (mlp_train16.hip) ---------------------------------------------------
// Backward-data stream: the transposed weights in the order the backward walk consumes them
// (rgb, view layer, feature_linear + heads, trunk layers 7 .. 0), same unit format as the forward
// ORDER_ACC stream: per row block of 32 INPUT features a header (bias block zero, modulation bias
// block of the layer whose ReLU mask / modulation the epilogue applies), its modulation tiles, then
// per k-tile of OUTPUT-feature positions the row tiles 0, 1.  Behind the ring-aligned part follows a
// tail the data kernel does not stream, read directly by the finishing kernel: pts_bias^T (rows = feature
// positions, k = the 256 modulation features; 17 units per row block), then per row block of m its
// header + modulation tiles (1 + nt_feat units) as in the forward stream.  Built into an MlpPlan so the forward packer packs
// it (tile_src / hdr_src / n_tiles / tail_unit0 / bytes are the fields used).
bool build_bwd_plan(const zest_mlp_desc &d, MlpPlan *out, const char **err);
// row blocks / k-tiles of the backward ops, shared by the plan builder and the kernel's unrolled walk
constexpr int bwd_stream_units_raw(int nt_pts, int nt_feat) {
    const int kp = nt_pts / 2, mod = nt_feat;       // pts k-tiles; modulation units per row block (0 = off)
    return 4 * (1 + 2 * 1)                          // rgb^T: 4 row blocks (128 view-layer features) x 1 k-tile
           + 8 * (1 + 2 * 4)                        // view layer^T: 8 row blocks (feature_linear outputs) x 4 k-tiles
           + 8 * (1 + mod + 2 * 9)                  // feature_linear^T | heads^T: -> d h7, masked + modulated as layer 7
           + 6 * 8 * (1 + mod + 2 * 8)              // layers 7, 6, 4, 3, 2, 1 -> d h(l-1)
           + kp * (1 + 2 * 8) + 8 * (1 + mod + 2 * 8)   // layer 5: point rows, then d h4
           + kp * (1 + 2 * 8);                      // layer 0: point rows
}
constexpr int bwd_stream_units(int nt_pts, int nt_feat) {
    return (bwd_stream_units_raw(nt_pts, nt_feat) + kStreamAlign - 1) / kStreamAlign * kStreamAlign;
}

// One weight-gradient job of the training path: (op's output positions) x (a chunk of <= 256 input positions)
struct DwJob {
    int32_t out_tile0, n_out_tiles;      // gradient-stash tiles (k-tiles of 32 output positions)
    int32_t in_kind;                     // which operand: 0 a layer's output, 1 points, 2 features, 3 directions - all are
                                         // activation-stash tiles from in_tile0 on (the encoder's at kDwStash*)
    int32_t in_tile0, n_in_tiles;
    int32_t ld, col0, want_bias;
    int32_t wg0, n_wg;                   // workgroups of the weight kernel that share this job (set per device at upload)
    int16_t out_slot[256], out_row[256]; // output position -> (ZEST_P_* slot, row of its weight), -1: none
    int16_t in_col[256];                 // input position -> column (before col0), -1: none
};
// stash tile indices of the gradient stash (work buffer): d pre-activation of trunk layer l: 8 l + kt;
// feature_linear 64 + kt; view layer 72 + kt; head tile 76; rgb tile 77; modulation 78 + kt
constexpr int kGradTiles = 86;
// activation-stash tiles of the encoder's operands (mlp_train16.h kStashPts / kStashFeat / kStashViews)
constexpr int kDwStashPts = 76, kDwStashFeat = 79, kDwStashViews = 81;
int build_dw_jobs(const zest_mlp_desc &d, std::vector<DwJob> *jobs, const char **err);

// Builds the plan; returns false (with *err set) for shapes the kernels do not cover.
// with_tables = false skips the packer's gather tables (cheap: launch-time shape queries).
bool build_plan(const zest_mlp_desc &d, int precision, int order, MlpPlan *out, const char **err,
                bool with_tables = true);

}  // namespace zest
