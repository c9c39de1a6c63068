// The f16x2 policy of the split SDF kernels as its own translation unit (same source: sdf_mlp_split.hip), so that the two
// policies' long compiles run in parallel (build.sh).
#define SURF_SDF_TU_F16 1
#include "sdf_mlp_split.hip"
