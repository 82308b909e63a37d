// One instantiation of the fused renderer per object file so the variants compile in parallel:
// build_hip.py compiles this source once per (operand type, feature shape) with
//   -DZEST_V_PTAG=bf16|f16|x3  -DZEST_V_EP=ZEST_PREC_*  -DZEST_V_TAG=s4d2  -DZEST_V_NTS=4 -DZEST_V_DYN=true -DZEST_V_NTD=2 -DZEST_V_V2=false
// (NTS / NTD: stream tiles of the static / dynamic feature operand per row block, 0 = no features; V2: the static net is
// a 'v2' net - tags s2v, s4v).
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(ZEST_V_PTAG, ZEST_V_EP, ZEST_V_TAG, ZEST_V_NTS, ZEST_V_DYN, ZEST_V_NTD, ZEST_V_V2)
}
