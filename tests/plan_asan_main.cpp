#include "mlp_plan.h"
#include <cstdio>
int main() {
    int n = 0;
    for (int prec = 0; prec < 4; prec++)
        for (int order = 0; order < 2; order++)
            for (int pts : {63, 84})
                for (int use_feat = 0; use_feat < 2; use_feat++)
                    for (int V = 1; V <= 16; V += (use_feat ? 1 : 16))
                        for (int head = 0; head < 3; head++)
                            for (int nt : {0, 2}) {
                                zest_mlp_desc d{};
                                d.in_ch_pts = pts, d.in_ch_feat = 8 + 4 * V, d.in_ch_views = 27;
                                d.use_feat = use_feat, d.net_type = nt, d.head = head;
                                zest::MlpPlan p;
                                const char *err = nullptr;
                                const int precision = prec;      // ZEST_PREC_F32, _BF16, _F16, _F16X3
                                if (!zest::build_plan(d, precision, order, &p, &err, true)) continue;
                                n++;
                                // the feature operand holds every input column exactly once (mlp_plan.h
                                // feat_quad_col), the volume quads share a round with views only at lane groups
                                // 2 and 3, and no round after the volume's holds anything
                                if (use_feat && order == zest::ORDER_ACC) {
                                    std::vector<int> seen(d.in_ch_feat, 0);
                                    for (int16_t c : p.map_feat)
                                        if (c >= 0) {
                                            if (c >= d.in_ch_feat) return 9;
                                            seen[c]++;
                                        }
                                    for (int c : seen)
                                        if (c != 1) return 10;
                                    const int rv = zest::feat_volume_round(V);
                                    if (2 * (int)(p.map_feat.size() / 32) <= rv) return 11;
                                    if (zest::feat_quad_col(8 * (rv / 2) + (rv & 1), V) != 0 ||
                                        zest::feat_quad_col(8 * (rv / 2) + 2 + (rv & 1), V) != 4) return 12;
                                }
                                // a split plan doubles every tile unit and nothing else
                                if (precision == ZEST_PREC_F16X3) {
                                    zest::MlpPlan q;
                                    if (!zest::build_plan(d, ZEST_PREC_F16, order, &q, &err, false)) return 2;
                                    int hdr = 0, lo = 0;
                                    for (int o = 0; o < zest::kNumOps; o++) hdr += p.op[o].njb;
                                    for (size_t u = 0; u < p.unit_part.size(); u++) lo += p.unit_part[u];
                                    int raw_q = 0, raw_p = 0;
                                    for (int o = 0; o < zest::kNumOps; o++)
                                        raw_q += q.op[o].njb * q.op[o].tiles_per_jb, raw_p += p.op[o].njb * p.op[o].tiles_per_jb;
                                    if (raw_p - hdr != 2 * (raw_q - hdr) || lo != raw_q - hdr || p.parts != 2) return 3;
                                }
                            }
    // other depths / widths / skips: fp32 (ORDER_NATURAL) plans only; the engine order must refuse them
    int n_shapes = 0;
    for (int depth = 2; depth <= 8; depth++)
        for (int width : {64, 128, 192, 256})
            for (int mask = 0; mask < (1 << (depth - 1)); mask += (depth > 5 ? 5 : 1))
                for (int use_feat = 0; use_feat < 2; use_feat++) {
                    zest_mlp_desc d{};
                    d.in_ch_pts = 63, d.in_ch_feat = 20, d.in_ch_views = 27, d.use_feat = use_feat, d.head = use_feat ? 2 : 0;
                    d.depth = depth, d.width = width, d.skip_mask = mask;
                    zest::MlpPlan p;
                    const char *err = nullptr;
                    if (!zest::build_plan(d, ZEST_PREC_F32, zest::ORDER_NATURAL, &p, &err, true)) return 4;
                    if (p.n_ops != depth + 4) return 5;
                    const bool is_default = depth == 8 && width == 256 && mask == 16;
                    if (zest::build_plan(d, ZEST_PREC_BF16, zest::ORDER_ACC, &p, &err, false) != is_default) return 6;
                    n_shapes++;
                }
    {
        zest_mlp_desc d{};
        d.in_ch_pts = 63, d.in_ch_views = 27, d.depth = 4, d.width = 128, d.skip_mask = 1 << 3;   // no layer 4 to widen
        zest::MlpPlan p;
        const char *err = nullptr;
        if (zest::build_plan(d, ZEST_PREC_F32, zest::ORDER_NATURAL, &p, &err, false)) return 7;
        d.skip_mask = 0, d.width = 100;
        if (zest::build_plan(d, ZEST_PREC_F32, zest::ORDER_NATURAL, &p, &err, false)) return 8;
    }
    std::printf("plans built: %d\nshape plans built: %d\n", n, n_shapes);
    return n > 0 ? 0 : 1;
}
