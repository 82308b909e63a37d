// fused renderer variant: 0 static feature tiles, dynamic net false (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s0, 0, false, 0)
}
