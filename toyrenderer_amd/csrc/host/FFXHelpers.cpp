// FFXHelpers.cpp -- SPD::Execute of the reference (source/FFXHelpers.cpp:25-115) against the HIP
// kernels registered as "ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=N".  The
// FidelityFX headers are an empty submodule in the reference; ffxSpdSetup's outputs (64x64 tiles per
// work group) are restated here.
#include "FFXHelpers.h"

#include <string>

#include "Graphic.h"
#include "../ShaderInterop.h"

namespace FFXHelpers
{

void SPD::CreateTransientResources(RenderGraph& renderGraph)
{
    nvrhi::BufferDesc desc;                                                   // FFXHelpers.cpp:27-33
    desc.byteSize = sizeof(uint32_t) * 6;
    desc.structStride = (uint32_t)desc.byteSize;
    desc.debugName = "SPD Global Atomic Buffer";
    desc.canHaveUAVs = true;
    renderGraph.CreateTransientResource(m_AtomicRDGBufferHandle, desc);
}

void SPD::Execute(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, nvrhi::TextureHandle srcTex,
                  nvrhi::TextureHandle destTex, nvrhi::SamplerReductionType reductionType)
{
    nvrhi::BufferHandle atomicBuffer = renderGraph.GetBuffer(m_AtomicRDGBufferHandle);
    commandList->clearBufferUInt(atomicBuffer, 0);                            // :48-49 (SPD's counter must start at 0)

    const nvrhi::TextureDesc& destDesc = destTex->getDesc();
    interop::SPDConstants passParameters{};
    // ffxSpdSetup (:58): one work group per 64x64 tile of the destination rectangle
    const uint32_t groupsX = (destDesc.width + 63) / 64, groupsY = (destDesc.height + 63) / 64;
    passParameters.mips = ComputeNbMips(destDesc.width, destDesc.height) - 1;
    passParameters.numWorkGroups = groupsX * groupsY;
    check(passParameters.mips == destDesc.mipLevels - 1);                     // :63-64

    nvrhi::BindingSetDesc bindingSetDesc;                                     // :66-72
    const uint32_t midMip = destDesc.mipLevels > 6 ? 6u : destDesc.mipLevels - 1;
    bindingSetDesc.bindings = {
        nvrhi::BindingSetItem::PushConstants(0, sizeof(interop::SPDConstants)),
        nvrhi::BindingSetItem::Texture_SRV(0, srcTex),
        nvrhi::BindingSetItem::StructuredBuffer_UAV(0, atomicBuffer),
        nvrhi::BindingSetItem::Texture_UAV(1, destTex, nvrhi::Format::UNKNOWN, nvrhi::TextureSubresourceSet{ midMip, 1, 0, 1 }),
        nvrhi::BindingSetItem::Texture_UAV(2, destTex, nvrhi::Format::UNKNOWN, nvrhi::TextureSubresourceSet{ 0, 1, 0, 1 }),
    };
    const uint32_t kStartUAVSlotForMips = 3;                                  // :74-81
    for (uint32_t i = 0; i + 1 < destDesc.mipLevels; ++i)
        bindingSetDesc.bindings.push_back(nvrhi::BindingSetItem::Texture_UAV(kStartUAVSlotForMips + i, destTex, nvrhi::Format::UNKNOWN,
                                                                             nvrhi::TextureSubresourceSet{ i + 1, 1, 0, 1 }));
    // (the reference pads unused UAV slots up to 12 with a dummy texture, :83-89: a D3D12 root-signature need)

    check(reductionType != nvrhi::SamplerReductionType::Comparison);          // :91-104
    const uint32_t filterIdx = reductionType == nvrhi::SamplerReductionType::Minimum ? 1u : reductionType == nvrhi::SamplerReductionType::Maximum ? 2u : 0u;

    Graphic::ComputePassParams computePassParams;                             // :106-114
    computePassParams.m_CommandList = commandList;
    computePassParams.m_ShaderName = "ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=" + std::to_string(filterIdx);
    computePassParams.m_BindingSetDesc = bindingSetDesc;
    computePassParams.m_DispatchGroupSize = Vector3U{ groupsX, groupsY, 1 };
    computePassParams.m_PushConstantsData = &passParameters;
    computePassParams.m_PushConstantsBytes = sizeof(passParameters);
    g_Graphic.AddComputePass(computePassParams);
}

} // namespace FFXHelpers
