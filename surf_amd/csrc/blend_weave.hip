// K10c: the woven bf16x3 blending kernel (blend_split.hip, section SURF_BLEND_WEAVE_TU) as its own translation unit: it is
// compiled with -mllvm -pre-RA-sched=source (hand-placed VALU work between the MFMAs, one scheduling barrier per gap), which
// the two-wavefronts-per-SIMD kernels of blend_split.hip are not.
#define SURF_BLEND_WEAVE_TU 1
#include "blend_split.hip"
