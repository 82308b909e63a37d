#include "mlp_plan.h"
#include <cstdio>
int main() {
    int n = 0;
    for (int prec = 0; prec < 2; prec++)
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
                                const int precision = prec == 0 ? ZEST_PREC_F32 : ZEST_PREC_BF16;
                                if (zest::build_plan(d, precision, order, &p, &err, true)) n++;
                            }
    std::printf("plans built: %d\n", n);
    return n > 0 ? 0 : 1;
}
