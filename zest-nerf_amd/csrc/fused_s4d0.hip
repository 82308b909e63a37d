// fused renderer variant: 4 static feature units (two k-tiles), dynamic net true (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s4d0, 4, true, 0)
}
