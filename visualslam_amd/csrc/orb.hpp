// Internal launchers of the ORB description stage (orb_describe.hip), used by the pyramid driver in corner.hip.
#pragma once
#include "common.hpp"

namespace vslam {

vslam_status launch_gauss7(vslam_ctx *ctx, const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
// Keypoints kp[0 .. min(*count, max_count)) in level coordinates (count is a device word; negative = none).
vslam_status launch_orb_describe(vslam_ctx *ctx, const uint8_t *gray, int stride, const uint8_t *blur, int bstride, int w, int h,
                                 const vslam_feature *kp, const int32_t *count, int max_count, const int8_t *pattern,
                                 float *angle, uint8_t *desc);
void orb_default_pattern(int8_t *pattern);
bool orb_pattern_ok(const int8_t *pattern);

}  // namespace vslam
