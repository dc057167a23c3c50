// GraphicConstants.h -- source/GraphicConstants.h:10-28 (the constants the path uses).
#pragma once

#include "nvrhi_lite.h"

namespace GraphicConstants
{
static constexpr uint32_t kMaxThreadGroupsPerDimension = 65535;
static constexpr bool kInversedDepthBuffer = true;
static constexpr bool kInfiniteDepthBuffer = true;
static constexpr float kNearDepth = kInversedDepthBuffer ? 1.0f : 0.0f;
static constexpr float kFarDepth = 1.0f - kNearDepth;
static constexpr nvrhi::Format kDepthStencilFormat = nvrhi::Format::D24S8;
static constexpr nvrhi::Format kHZBFormat = nvrhi::Format::R16_FLOAT;
} // namespace GraphicConstants
