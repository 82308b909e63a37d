// Host-only: builds the MLP execution plan and the packer's gather tables (see mlp_plan.h).
#include "mlp_plan.h"
#include <string.h>

namespace zest {

namespace {

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- ORDER_ACC operand maps: one feature index per POSITION = kt*32 + 8*g + e (k-tile kt of
// 32 operand values, lane group g = lane >> 4, element e of the lane's 8 bf16), -1 = zero pad.
//
// positional encoding of C coordinates with L (even) bands, reference column order
// [x(C), sin(2^0 x)(C), cos(2^0 x)(C), sin(2^1 x)(C), ...] (networks.py:60-65).
// m = 8*kt + e enumerates (band pair, coordinate): m = bp*C + c for m < (L/2)*C; the lane
// group picks the band of the pair (g >> 1) and sin / cos (g & 1), so a lane's eight values
// differ only in compile-time constants.  m = (L/2)*C holds the raw coordinates, one per group.
void pe_map_acc(int C, int L, int npos, std::vector<int16_t> &m) {
    m.assign((size_t)npos, -1);
    for (int pos = 0; pos < npos; pos++) {
        const int kt = pos / 32, g = (pos % 32) / 8, e = pos % 8, mm = 8 * kt + e;
        if (mm < (L / 2) * C) {
            const int band = 2 * (mm / C) + (g >> 1), coord = mm % C;
            m[pos] = (int16_t)(C + 2 * C * band + ((g & 1) ? C : 0) + coord);
        } else if (mm == (L / 2) * C) {
            m[pos] = (int16_t)(g < C ? g : -1);
        }
    }
}

// feature operand: 4-channel quads, quad index 8*kt + 2*g + (e >> 2); which columns a quad holds: mlp_plan.h
// feat_quad_col
void feat_map_acc(int V, int npos, std::vector<int16_t> &m) {
    m.assign((size_t)npos, -1);
    for (int pos = 0; pos < npos; pos++) {
        const int kt = pos / 32, g = (pos % 32) / 8, e = pos % 8, q = 8 * kt + 2 * g + (e >> 2), c = e & 3;
        const int col = feat_quad_col(q, V);
        if (col >= 0) m[pos] = (int16_t)(col + c);
    }
}

void natural_map(int width, int spt, int ntiles, std::vector<int16_t> &m) {
    m.assign((size_t)ntiles * spt * 2, -1);
    for (int t = 0; t < ntiles; t++)
        for (int j = 0; j < spt; j++)
            for (int h = 0; h < 2; h++) {
                const int f = t * 2 * spt + spt * h + j;
                m[2 * (t * spt + j) + h] = (int16_t)(f < width ? f : -1);
            }
}

// hidden operands: (slot, half) for ORDER_NATURAL; for ORDER_ACC `slot` is a position and the
// k-tile kt is the row block that produced it: e < 4 from its row tile 0, e >= 4 from row tile 1,
// accumulator rows 4g .. 4g+3 of the lane group
inline int h_feature(int order, int spt, int slot, int half) {
    if (order == ORDER_ACC) {
        const int kt = slot / 32, g = (slot % 32) / 8, e = slot % 8;
        return 32 * kt + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
    }
    const int t = slot / spt, j = slot % spt;
    return t * 2 * spt + spt * half + j;
}

struct RowSrc { int param, row; };       // param < 0: zero row

}  // namespace

bool build_plan(const zest_mlp_desc &d, int precision, int order, MlpPlan *P, const char **err,
                bool with_tables) {
    static const char *e_prec = "precision must be ZEST_PREC_F32, _BF16, _F16 or _F16X3";
    static const char *e_pts = "in_ch_pts must be 63 (xyz, L=10) or 84 (xyzt, L=10)";
    static const char *e_views = "in_ch_views must be 27 (L=4)";
    static const char *e_feat = "in_ch_feat must be 8 + 4V with 1 <= V <= 16 when use_feat is set";
    static const char *e_head = "bad head / net_type combination";
    if (precision != ZEST_PREC_F32 && !prec_is_engine(precision)) return *err = e_prec, false;
    if ((order == ORDER_ACC) != prec_is_engine(precision))
        return *err = "ORDER_ACC goes with the engine precisions, ORDER_NATURAL with fp32", false;
    if (d.in_ch_pts != 63 && d.in_ch_pts != 84) return *err = e_pts, false;
    if (d.in_ch_views != 27) return *err = e_views, false;
    const int F = d.use_feat ? d.in_ch_feat : 0;
    if (d.use_feat && (F < 12 || (F - 8) % 4 != 0 || F > 72)) return *err = e_feat, false;
    // net_type 3: the 'v2' trunk (additive modulation) with raw outputs - Renderer_linear.forward_alpha
    if (d.head < 0 || d.head > 2 || (d.net_type != 0 && d.net_type != 2 && d.net_type != 3) ||
        (d.net_type >= 2 && (d.head != 0 || !d.use_feat)))
        return *err = e_head, false;

    MlpShape sh;
    if (!mlp_shape(d, &sh, err)) return false;
    if (order == ORDER_ACC && !sh.is_default)
        return *err = "the MFMA engine covers depth 8 / width 256 / skips [4]; other shapes run in ZEST_PREC_F32", false;
    const int D = sh.D, Wd = sh.W;

    MlpPlan &p = *P;
    p.desc = d, p.precision = precision, p.order = order, p.shape = sh;
    p.n_ops = D + 4;
    p.spt = prec_is_engine(precision) ? 8 : 4;
    p.parts = precision == ZEST_PREC_F16X3 ? 2 : 1;
    const int spt = p.spt, C = d.in_ch_pts == 63 ? 3 : 4, V = d.use_feat ? (F - 8) / 4 : 0;
    if (order == ORDER_ACC) {
        // "slots" are positions here and a tile (16 rows x 32 positions) covers half a k-tile's
        // rows: ns / 16 units per row block of 32 outputs
        p.ns_pts = 32 * ((5 * C + 1 + 7) / 8);                 // 64 (xyz) / 96 (xyzt)
        p.ns_views = 32;
        p.ns_feat = d.use_feat ? round_up(4 * (2 + V), 32) : 0;  // 32 up to 6 views, 64 up to 14
        pe_map_acc(C, 10, p.ns_pts, p.map_pts);
        pe_map_acc(3, 4, p.ns_views, p.map_views);
        if (d.use_feat) feat_map_acc(V, p.ns_feat, p.map_feat);
    } else {
        p.ns_pts = round_up(d.in_ch_pts, 2 * spt) / 2;
        p.ns_views = round_up(27, 2 * spt) / 2;
        p.ns_feat = d.use_feat ? round_up(F, 2 * spt) / 2 : 0;
        natural_map(d.in_ch_pts, spt, p.ns_pts / spt, p.map_pts);
        natural_map(27, spt, p.ns_views / spt, p.map_views);
        if (d.use_feat) natural_map(F, spt, p.ns_feat / spt, p.map_feat);
    }
    const int spu = order == ORDER_ACC ? 16 : spt;          // operand slots / positions per stream unit
    p.nt_pts = p.ns_pts / spu, p.nt_views = p.ns_views / spu, p.nt_feat = p.ns_feat / spu;
    p.nt_h = order == ORDER_ACC ? 16 : Wd / 2 / spt, p.nt_h128 = order == ORDER_ACC ? 8 : Wd / 4 / spt;

    // ---- op table ----------------------------------------------------------------------
    p.headers = order == ORDER_ACC ? 1 : 0;
    // op ids: 0 .. D-1 trunk, D head tile, D+1 feature_linear, D+2 view layer, D+3 rgb_linear
    memset(p.op, 0, sizeof(p.op));
    int tile = 0, bblk = d.use_feat ? Wd / 32 : 0;
    for (int o = 0; o < p.n_ops; o++) {
        OpPlan &op = p.op[o];
        op.njb = Wd / 32, op.nseg = 1, op.seg[0] = {SEG_H, p.nt_h};
        if (o < D) op.mod = d.use_feat ? 1 : 0, op.relu = 1;
        if (o == 0) op.seg[0] = {SEG_PTS, p.nt_pts};
        if (o > 0 && o < D && (sh.skip_mask >> (o - 1) & 1))
            op.nseg = 2, op.seg[0] = {SEG_PTS, p.nt_pts}, op.seg[1] = {SEG_H, p.nt_h};
        if (o == D) op.njb = 1;
        if (o == D + 2) op.njb = Wd / 64, op.relu = 1, op.nseg = 2, op.seg[1] = {SEG_VIEWS, p.nt_views};
        if (o == D + 3) op.njb = 1, op.seg[0] = {SEG_H, p.nt_h128};
        op.tiles_per_jb = (op.mod ? p.nt_feat : 0) * p.parts + p.headers;
        for (int s = 0; s < op.nseg; s++) op.tiles_per_jb += op.seg[s].ntiles * p.parts;
        op.tile_base = tile, op.bias_block = bblk;
        tile += op.njb * op.tiles_per_jb, bblk += op.njb;
    }
    if (p.headers) tile = round_up(tile, kStreamAlign), bblk = 0;
    p.n_tiles = tile, p.n_bias_blocks = bblk;
    p.bias_bytes = (size_t)round_up(bblk * 128, 1024);
    p.bytes = p.bias_bytes + (size_t)tile * 1024;

    if (!with_tables) return true;
    // ---- gather tables -------------------------------------------------------------------
    int ld[ZEST_P_COUNT];
    mlp_param_ld(d, sh, ld);
    const uint32_t ZERO = 0xFFFFFFFFu;
    p.tile_src.assign((size_t)tile * 64 * spt, ZERO);
    p.bias_src.assign(p.bias_bytes / 4, ZERO);
    p.hdr_src.assign(p.headers ? (size_t)tile * 64 : 0, ZERO);   // 64 floats at the head of a header unit
    p.unit_part.assign((size_t)tile, 0);

    auto row_src = [&](int o, int row) -> RowSrc {
        if (o < D) return {ZEST_P_PTS0 + o, row};
        if (o == D + 1) return {ZEST_P_FEATURE, row};
        if (o == D + 2) return {row < Wd / 2 ? ZEST_P_VIEWS : -1, row};
        if (o == D + 3) return {row < 3 ? ZEST_P_RGB : -1, row};
        // head tile: row 0 alpha, then the extra heads in output order
        if (row == 0) return {ZEST_P_ALPHA, 0};
        if (d.head == ZEST_HEAD_BLEND && row == 1) return {ZEST_P_HEAD0, 0};
        if (d.head == ZEST_HEAD_DYNAMIC && row >= 1 && row <= 6) return {ZEST_P_HEAD0, row - 1};
        if (d.head == ZEST_HEAD_DYNAMIC && row >= 7 && row <= 8) return {ZEST_P_HEAD1, row - 7};
        return {-1, 0};
    };
    auto emit_tiles = [&](int &t, int param_of_rows_op, int jb, const SegPlan &sg, int col0,
                          bool is_mod) {
        for (int k = 0; k < sg.ntiles; k++)
          for (int part = 0; part < p.parts; part++, t++) {      // split pair: hi unit, then lo unit
            p.unit_part[t] = (uint8_t)part;
            for (int l = 0; l < 64; l++)
                for (int e = 0; e < spt; e++) {
                    int slot, half, row, feat;
                    if (order == ORDER_ACC) {
                        // unit k of the segment: k-tile k/2, row tile k%2 (16 rows x 32 positions);
                        // lane l, element e: row l&15, position 8*(l>>4) + e of the k-tile
                        slot = (k / 2) * 32 + 8 * (l >> 4) + e, half = 0, row = 32 * jb + 16 * (k % 2) + (l & 15);
                        switch (sg.kind) {
                            case SEG_PTS: feat = p.map_pts[slot]; break;
                            case SEG_FEAT: feat = p.map_feat[slot]; break;
                            case SEG_VIEWS: feat = p.map_views[slot]; break;
                            default: feat = h_feature(order, spt, slot, 0);
                        }
                    } else {
                        slot = k * spt + e, half = l >> 5, row = 32 * jb + (l & 31);
                        switch (sg.kind) {
                            case SEG_PTS: feat = p.map_pts[2 * slot + half]; break;
                            case SEG_FEAT: feat = p.map_feat[2 * slot + half]; break;
                            case SEG_VIEWS: feat = p.map_views[2 * slot + half]; break;
                            default: feat = h_feature(order, spt, slot, half);
                        }
                    }
                    RowSrc rs = is_mod ? RowSrc{ZEST_P_PTS_BIAS, row} : row_src(param_of_rows_op, row);
                    if (feat < 0 || rs.param < 0) continue;
                    p.tile_src[((size_t)t * 64 + l) * spt + e] =
                        ((uint32_t)rs.param << 24) | (uint32_t)(rs.row * ld[rs.param] + col0 + feat);
                }
          }
    };
    auto emit_bias_to = [&](std::vector<uint32_t> &dst, size_t at, int o, int jb, bool is_mod) {
        for (int h = 0; h < 2; h++)
            for (int i = 0; i < 16; i++) {
                // ORDER_ACC: the 32 biases of the row block in natural order (a lane reads rows
                // 16 rt + 4 g .. +3 of them); ORDER_NATURAL: 32x32 accumulator register order
                const int row = order == ORDER_ACC ? 32 * jb + 16 * h + i
                                                   : 32 * jb + (i & 3) + 8 * (i >> 2) + 4 * h;
                RowSrc rs = is_mod ? RowSrc{ZEST_P_PTS_BIAS, row} : row_src(o, row);
                if (rs.param < 0) continue;
                dst[at + h * 16 + i] = ((uint32_t)rs.param << 24) | rs.row;
            }
    };
    auto emit_bias = [&](int block, int o, int jb, bool is_mod) {
        emit_bias_to(p.bias_src, (size_t)block * 32, o, jb, is_mod);
    };
    if (d.use_feat && !p.headers)
        for (int jb = 0; jb < Wd / 32; jb++) emit_bias(jb, 0, jb, true);
    for (int o = 0; o < p.n_ops; o++) {
        const OpPlan &op = p.op[o];
        int t = op.tile_base;
        for (int jb = 0; jb < op.njb; jb++) {
            if (p.headers) {
                emit_bias_to(p.hdr_src, (size_t)t * 64, o, jb, false);
                if (op.mod) emit_bias_to(p.hdr_src, (size_t)t * 64 + 32, o, jb, true);
                t++;
            } else {
                emit_bias(op.bias_block + jb, o, jb, false);
            }
            if (op.mod) emit_tiles(t, o, jb, SegPlan{SEG_FEAT, p.nt_feat}, 0, true);
            int col0 = 0;
            for (int s = 0; s < op.nseg; s++) {
                emit_tiles(t, o, jb, op.seg[s], col0, false);
                // width of the operand just consumed, in the Linear's own column order
                col0 += op.seg[s].kind == SEG_PTS ? d.in_ch_pts : Wd;
            }
        }
    }
    return true;
}

bool mlp_shape(const zest_mlp_desc &d, MlpShape *s, const char **err) {
    const bool unset = d.depth == 0 && d.width == 0 && d.skip_mask == 0;
    s->D = unset ? 8 : d.depth, s->W = unset ? kW : d.width, s->skip_mask = unset ? 1 << 4 : d.skip_mask;
    if (s->D < 2 || s->D > 8) return *err = "depth must be 2 .. 8", false;
    if (s->W != 64 && s->W != 128 && s->W != 192 && s->W != 256) return *err = "width must be 64, 128, 192 or 256", false;
    // reference networks.py:93-100: `i in skips` widens layer i+1; an entry >= D-1 has no layer to widen
    if (s->skip_mask < 0 || (s->skip_mask >> (s->D - 1)) != 0) return *err = "skip_mask names a layer beyond depth", false;
    s->is_default = s->D == 8 && s->W == kW && s->skip_mask == 1 << 4;
    return true;
}

void mlp_param_ld(const zest_mlp_desc &d, const MlpShape &s, int *ld) {
    for (int l = 0; l < 8; l++)
        ld[ZEST_P_PTS0 + l] = l == 0 ? d.in_ch_pts : s.W + ((s.skip_mask >> (l - 1) & 1) ? d.in_ch_pts : 0);
    ld[ZEST_P_PTS_BIAS] = d.in_ch_feat, ld[ZEST_P_VIEWS] = s.W + d.in_ch_views, ld[ZEST_P_FEATURE] = s.W;
    ld[ZEST_P_ALPHA] = s.W, ld[ZEST_P_RGB] = s.W / 2, ld[ZEST_P_HEAD0] = s.W, ld[ZEST_P_HEAD1] = s.W;
}


// ---------------------------------------------------------------------------------------------
// bf16 training path: backward-data stream and weight-gradient jobs (see mlp_plan.h)
namespace {

struct BwdCtx {
    const zest_mlp_desc &d;
    MlpPlan &p;
    int ld[ZEST_P_COUNT];
    int t = 0;                       // next unit
    std::vector<uint32_t> tile_src, hdr_src;
    void grow(int units) {
        tile_src.resize((size_t)(t + units) * 512, 0xFFFFFFFFu);
        hdr_src.resize((size_t)(t + units) * 64, 0xFFFFFFFFu);
    }
};

// (param slot, row) of output feature `o` of a forward op; op ids as in build_plan (8 = head tile)
inline RowSrc out_row_of(const zest_mlp_desc &d, int op, int o) {
    if (op == 100) return {ZEST_P_PTS_BIAS, o};                  // the modulation Linear
    if (op < 8) return {ZEST_P_PTS0 + op, o};
    if (op == 9) return {ZEST_P_FEATURE, o};
    if (op == 10) return {o < kW / 2 ? ZEST_P_VIEWS : -1, o};
    if (op == 11) return {o < 3 ? ZEST_P_RGB : -1, o};
    if (o == 0) return {ZEST_P_ALPHA, 0};
    if (d.head == ZEST_HEAD_BLEND && o == 1) return {ZEST_P_HEAD0, 0};
    if (d.head == ZEST_HEAD_DYNAMIC && o >= 1 && o <= 6) return {ZEST_P_HEAD0, o - 1};
    if (d.head == ZEST_HEAD_DYNAMIC && o >= 7 && o <= 8) return {ZEST_P_HEAD1, o - 7};
    return {-1, 0};
}

// One row block of 32 input indices of a transposed op: header (+ modulation tiles of block `mod_jb`
// when mod_jb >= 0) + per k-tile of the op's output positions the row tiles 0, 1.
//   in_col(i): column of input index i (0..31 of this row block) in the op's weight, -1 = zero row
//   segs: {forward op id, k-tiles of its output positions}
template <class InCol>
void emit_bwd_row_block(BwdCtx &c, InCol in_col, int mod_jb, const int (*segs)[2], int nseg) {
    const MlpPlan &p = c.p;
    int units = 1 + (mod_jb >= 0 ? p.nt_feat : 0);
    for (int s = 0; s < nseg; s++) units += 2 * segs[s][1];
    c.grow(units);
    // header: accumulator initialiser 0; modulation bias block of block mod_jb
    if (mod_jb >= 0)
        for (int i = 0; i < 32; i++)
            c.hdr_src[(size_t)c.t * 64 + 32 + i] = ((uint32_t)ZEST_P_PTS_BIAS << 24) | (uint32_t)(32 * mod_jb + i);
    c.t++;
    if (mod_jb >= 0)
        for (int k = 0; k < p.nt_feat; k++, c.t++)
            for (int l = 0; l < 64; l++)
                for (int e = 0; e < 8; e++) {
                    const int pos = (k / 2) * 32 + 8 * (l >> 4) + e, row = 32 * mod_jb + 16 * (k % 2) + (l & 15);
                    const int feat = p.map_feat[pos];
                    if (feat < 0) continue;
                    c.tile_src[((size_t)c.t * 64 + l) * 8 + e] =
                        ((uint32_t)ZEST_P_PTS_BIAS << 24) | (uint32_t)(row * c.ld[ZEST_P_PTS_BIAS] + feat);
                }
    for (int s = 0; s < nseg; s++)
        for (int k = 0; k < 2 * segs[s][1]; k++, c.t++)
            for (int l = 0; l < 64; l++)
                for (int e = 0; e < 8; e++) {
                    const int pos = (k / 2) * 32 + 8 * (l >> 4) + e;            // position of the op's output operand
                    const int o = h_feature(ORDER_ACC, 8, pos, 0);               // ... is this output feature
                    const RowSrc rs = out_row_of(c.d, segs[s][0], o);
                    const int col = in_col(16 * (k % 2) + (l & 15));
                    if (rs.param < 0 || col < 0) continue;
                    c.tile_src[((size_t)c.t * 64 + l) * 8 + e] =
                        ((uint32_t)rs.param << 24) | (uint32_t)(rs.row * c.ld[rs.param] + col);
                }
}

}  // namespace

bool build_bwd_plan(const zest_mlp_desc &d, MlpPlan *P, const char **err) {
    if (!build_plan(d, ZEST_PREC_BF16, ORDER_ACC, P, err, false)) return false;
    if (d.net_type != 0) return *err = "the bf16 training path covers 'v0' nets", false;
    MlpPlan &p = *P;
    BwdCtx c{d, p, {d.in_ch_pts, kW, kW, kW, kW, kW + d.in_ch_pts, kW, kW, d.in_ch_feat, kW + d.in_ch_views, kW, kW,
                    kW / 2, kW, kW}};
    const bool mod = d.use_feat != 0;
    const int kp = p.nt_pts / 2;
    auto natural = [](int base) { return [base](int i) { return base + i; }; };
    // 1. rgb^T: rows = the 128 view-layer features
    for (int jb = 0; jb < 4; jb++) { const int sg[1][2] = {{11, 1}}; emit_bwd_row_block(c, natural(32 * jb), -1, sg, 1); }
    // 2. view layer^T: rows = feature_linear outputs (columns 0..255 of views_linears.0)
    for (int jb = 0; jb < 8; jb++) { const int sg[1][2] = {{10, 4}}; emit_bwd_row_block(c, natural(32 * jb), -1, sg, 1); }
    // 3. feature_linear^T | head^T: rows = h7; the epilogue applies layer 7's mask and modulation
    for (int jb = 0; jb < 8; jb++) { const int sg[2][2] = {{9, 8}, {8, 1}}; emit_bwd_row_block(c, natural(32 * jb), mod ? jb : -1, sg, 2); }
    // 4. trunk layers 7 .. 1: rows = inputs of layer l (layer 5: the point positions first), epilogue of layer l-1
    for (int l = 7; l >= 1; l--) {
        const int sg[1][2] = {{l, 8}};
        if (l == 5)
            for (int jb = 0; jb < kp; jb++)
                emit_bwd_row_block(c, [&](int i) { return (int)p.map_pts[32 * jb + i]; }, -1, sg, 1);
        const int h0 = l == 5 ? d.in_ch_pts : 0;         // columns of the hidden part
        for (int jb = 0; jb < 8; jb++) emit_bwd_row_block(c, natural(h0 + 32 * jb), mod ? jb : -1, sg, 1);
    }
    // 5. layer 0: rows = the point positions
    for (int jb = 0; jb < kp; jb++) {
        const int sg[1][2] = {{0, 8}};
        emit_bwd_row_block(c, [&](int i) { return (int)p.map_pts[32 * jb + i]; }, -1, sg, 1);
    }
    if (c.t != bwd_stream_units_raw(p.nt_pts, mod ? p.nt_feat : 0)) return *err = "backward stream: unit count mismatch", false;
    int total = round_up(c.t, kStreamAlign);
    c.grow(total - c.t);
    c.t = total;
    p.tail_unit0 = total;
    if (mod) {          // tail: pts_bias^T (rows = feature positions), then the forward's modulation units per row block
        for (int jb = 0; jb < p.nt_feat / 2; jb++) {
            const int sg[1][2] = {{100, 8}};
            emit_bwd_row_block(c, [&](int i) { return (int)p.map_feat[32 * jb + i]; }, -1, sg, 1);
        }
        for (int jb = 0; jb < 8; jb++) emit_bwd_row_block(c, [](int) { return -1; }, jb, nullptr, 0);
    }
    total = c.t;
    p.headers = 1, p.parts = 1, p.n_tiles = total, p.n_bias_blocks = 0, p.bias_bytes = 0;
    p.bytes = (size_t)total * 1024;
    p.tile_src.swap(c.tile_src), p.hdr_src.swap(c.hdr_src);
    p.bias_src.clear();
    p.unit_part.assign((size_t)total, 0);
    return true;
}

int build_dw_jobs(const zest_mlp_desc &d, std::vector<DwJob> *jobs, const char **err) {
    MlpPlan p;
    if (!build_plan(d, ZEST_PREC_BF16, ORDER_ACC, &p, err, false)) return -1;
    jobs->clear();
    const bool mod = d.use_feat != 0;
    const int kp = p.nt_pts / 2, kf = p.nt_feat / 2;
    auto add = [&](int op, int out_tile0, int n_out, int in_kind, int in_tile0, int n_in, int ld, int col0, int bias,
                   const std::vector<int16_t> *in_map) {
        DwJob j;
        memset(&j, 0, sizeof(j));
        j.out_tile0 = out_tile0, j.n_out_tiles = n_out, j.in_kind = in_kind, j.in_tile0 = in_tile0, j.n_in_tiles = n_in;
        j.ld = ld, j.col0 = col0, j.want_bias = bias;
        for (int pos = 0; pos < 256; pos++) {
            j.out_slot[pos] = -1, j.out_row[pos] = 0, j.in_col[pos] = -1;
            if (pos < 32 * n_out) {
                const int o = h_feature(ORDER_ACC, 8, pos, 0);
                const RowSrc rs = out_row_of(d, op, o);
                j.out_slot[pos] = (int16_t)rs.param, j.out_row[pos] = (int16_t)rs.row;
            }
            if (pos < 32 * n_in)
                j.in_col[pos] = in_map ? (*in_map)[pos] : (int16_t)h_feature(ORDER_ACC, 8, pos, 0);
        }
        jobs->push_back(j);
    };
    for (int l = 0; l < 8; l++) {
        const int ld = l == 0 ? d.in_ch_pts : (l == 5 ? kW + d.in_ch_pts : kW);
        if (l == 0 || l == 5) add(l, 8 * l, 8, 1, kDwStashPts, kp, ld, 0, l == 0, &p.map_pts);
        if (l > 0) add(l, 8 * l, 8, 0, 8 * (l - 1), 8, ld, l == 5 ? d.in_ch_pts : 0, 1, nullptr);
    }
    add(8, 76, 1, 0, 56, 8, kW, 0, 1, nullptr);                       // heads <- h7
    add(9, 64, 8, 0, 56, 8, kW, 0, 1, nullptr);                       // feature_linear <- h7
    add(10, 72, 4, 0, 64, 8, kW + d.in_ch_views, 0, 1, nullptr);      // view layer <- feature_linear output
    add(10, 72, 4, 3, kDwStashViews, 1, kW + d.in_ch_views, kW, 0, &p.map_views);   //   <- direction encoding
    add(11, 77, 1, 0, 72, 4, kW / 2, 0, 1, nullptr);                  // rgb <- view layer
    if (mod) add(100, 78, 8, 2, kDwStashFeat, kf, d.in_ch_feat, 0, 1, &p.map_feat);   // pts_bias <- features
    return (int)jobs->size();
}

}  // namespace zest
