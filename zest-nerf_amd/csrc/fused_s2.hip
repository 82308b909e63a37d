// fused renderer variant: 2 static feature tiles, dynamic net false (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s2, 2, false, 0)
}
