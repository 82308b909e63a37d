// fused renderer variant: 2 static feature tiles, dynamic net true (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s2d0, 2, true, 0)
}
