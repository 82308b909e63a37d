// fused renderer variant: 3 static feature tiles, dynamic net true (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s3d0, 3, true, 0)
}
