// fused renderer variant: 3 static feature tiles, dynamic net true (2 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s3d2, 3, true, 2)
}
