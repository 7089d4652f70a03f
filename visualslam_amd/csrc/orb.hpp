// Internal launchers of the ORB description stage (orb_describe.hip), used by the pyramid driver in corner.hip.
#pragma once
#include "common.hpp"

namespace vslam {

// Frame f of a batch at src + f * sframe / dst + f * dframe.
vslam_status launch_gauss7(vslam_ctx *ctx, const uint8_t *src, int w, int h, int sstride, size_t sframe, uint8_t *dst, int dstride,
                           size_t dframe, int frames);
// Keypoints kp[f * max_count + (0 .. min(count[f], max_count))) of frame f in level coordinates (counts are device
// words; negative = none); angle and desc are laid out like kp (max_count entries per frame).
vslam_status launch_orb_describe(vslam_ctx *ctx, const uint8_t *gray, int stride, size_t gframe, const uint8_t *blur, int bstride,
                                 size_t bframe, int w, int h, int frames, const vslam_feature *kp, const int32_t *count, int max_count,
                                 const int8_t *pattern, float *angle, uint8_t *desc);
void orb_default_pattern(int8_t *pattern);
bool orb_pattern_ok(const int8_t *pattern);

}  // namespace vslam
