// Layout of the BlendingNetwork parameters as the host hands them over (shared by blend.hip and blend_split.hip).
#pragma once

namespace blend_raw {

constexpr int DF = 19;  // d_feature(16) + 3
// raw (state_dict order) buffer, floats: surf_amd.ops.blend_raw_weights concatenates in this order
// (blending_network.py:34-64 in declaration order, weight then bias, row-major)
constexpr int R_S = 0;
constexpr int R_RD0_W = R_S + 1, R_RD0_B = R_RD0_W + 16 * 4;
constexpr int R_RD2_W = R_RD0_B + 16, R_RD2_B = R_RD2_W + DF * 16;
constexpr int R_B0_W = R_RD2_B + DF, R_B0_B = R_B0_W + 64 * 57;
constexpr int R_B2_W = R_B0_B + 64, R_B2_B = R_B2_W + 32 * 64;
constexpr int R_V0_W = R_B2_B + 32, R_V0_B = R_V0_W + 32 * 32;
constexpr int R_V2_W = R_V0_B + 32, R_V2_B = R_V2_W + 33 * 32;
constexpr int R_W0_W = R_V2_B + 33, R_W0_B = R_W0_W + 32 * 32;
constexpr int R_W2_W = R_W0_B + 32, R_W2_B = R_W2_W + 32;
constexpr int R_R0_W = R_W2_B + 1, R_R0_B = R_R0_W + 16 * 37;
constexpr int R_R2_W = R_R0_B + 16, R_R2_B = R_R2_W + 8 * 16;
constexpr int R_R4_W = R_R2_B + 8, R_R4_B = R_R4_W + 8;
constexpr int RAW_FLOATS = R_R4_B + 1;


}  // namespace blend_raw
