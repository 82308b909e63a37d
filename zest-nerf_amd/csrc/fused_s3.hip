// fused renderer variant: 3 static feature tiles, dynamic net false (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s3, 3, false, 0)
}
