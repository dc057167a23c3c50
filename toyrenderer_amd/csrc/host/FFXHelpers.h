// FFXHelpers.h -- source/FFXHelpers.h: the SPD (single-pass downsampler) wrapper used by GenerateHZB.
#pragma once

#include "RenderGraph.h"
#include "nvrhi_lite.h"

namespace FFXHelpers
{
class SPD
{
public:
    void CreateTransientResources(RenderGraph& renderGraph);
    void Execute(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, nvrhi::TextureHandle src,
                 nvrhi::TextureHandle dest, nvrhi::SamplerReductionType reductionType);
    RenderGraph::ResourceHandle m_AtomicRDGBufferHandle;
};
} // namespace FFXHelpers
