// fused renderer variant: 2 static feature tiles, dynamic net true (2 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s2d2, 2, true, 2)
}
