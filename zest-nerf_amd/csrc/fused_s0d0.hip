// fused renderer variant: 0 static feature tiles, dynamic net true (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s0d0, 0, true, 0)
}
