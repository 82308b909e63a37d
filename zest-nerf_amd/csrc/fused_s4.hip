// fused renderer variant: 4 static feature units (two k-tiles), dynamic net false (0 feature tiles)
#include "fused.cuh"
namespace zest {
ZEST_FUSED_VARIANT(s4, 4, false, 0)
}
