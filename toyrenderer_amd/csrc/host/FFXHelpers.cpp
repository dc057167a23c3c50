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
    // FFXHelpers.cpp:27-33: SPD's global atomic counter, six words (one per slice of a cube / array source)
    nvrhi::BufferDesc counter;
    counter.debugName = "SPD Global Atomic Buffer";
    counter.structStride = counter.byteSize = 6 * sizeof(uint32_t);
    counter.canHaveUAVs = true;
    renderGraph.CreateTransientResource(m_AtomicRDGBufferHandle, counter);
}

void SPD::Execute(nvrhi::CommandListHandle commandList, const RenderGraph& renderGraph, nvrhi::TextureHandle srcTex,
                  nvrhi::TextureHandle destTex, nvrhi::SamplerReductionType reductionType)
{
    using Item = nvrhi::BindingSetItem;
    const nvrhi::TextureDesc& dest = destTex->getDesc();
    const uint32_t mipsBelowTop = dest.mipLevels - 1;
    check(ComputeNbMips(dest.width, dest.height) - 1 == mipsBelowTop);          // :63-64
    check(reductionType != nvrhi::SamplerReductionType::Comparison);             // :91

    nvrhi::BufferHandle counter = renderGraph.GetBuffer(m_AtomicRDGBufferHandle);
    commandList->clearBufferUInt(counter, 0);                                    // :47-49: the counter must start at 0

    // ffxSpdSetup (:58): one work group per 64 x 64 tile of the destination rectangle, every mip below mip 0 in one dispatch
    const Vector3U tiles = Vector3U{ (dest.width + 63) / 64, (dest.height + 63) / 64, 1 };
    interop::SPDConstants constants{};
    constants.mips = mipsBelowTop;
    constants.numWorkGroups = tiles.x * tiles.y;

    auto mip = [](uint32_t level) { return nvrhi::TextureSubresourceSet{ level, 1, 0, 1 }; };
    Graphic::ComputePassParams pass;                                             // :66-114
    pass.m_CommandList = commandList;
    // FFX_SPD_OPTION_DOWNSAMPLE_FILTER: 0 mean, 1 min, 2 max (:93-104)
    pass.m_ShaderName = std::string("ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=") +
                        (reductionType == nvrhi::SamplerReductionType::Minimum ? "1" : reductionType == nvrhi::SamplerReductionType::Maximum ? "2" : "0");
    pass.m_DispatchGroupSize = tiles;
    pass.m_PushConstantsData = &constants;
    pass.m_PushConstantsBytes = sizeof constants;
    std::vector<Item>& bindings = pass.m_BindingSetDesc.bindings;
    bindings = {
        Item::PushConstants(0, sizeof constants),
        Item::Texture_SRV(0, srcTex),
        Item::StructuredBuffer_UAV(0, counter),
        Item::Texture_UAV(1, destTex, nvrhi::Format::UNKNOWN, mip(mipsBelowTop < 6 ? mipsBelowTop : 6)),   // the mip SPD's last work group continues from
        Item::Texture_UAV(2, destTex, nvrhi::Format::UNKNOWN, mip(0)),
    };
    for (uint32_t level = 1; level <= mipsBelowTop; ++level)                     // u3.. = mips 1..N-1 (:74-81)
        bindings.push_back(Item::Texture_UAV(2 + level, destTex, nvrhi::Format::UNKNOWN, mip(level)));
    // (the reference pads the unused UAV slots up to 12 with a dummy texture, :83-89: a D3D12 root-signature need)
    g_Graphic.AddComputePass(pass);
}

} // namespace FFXHelpers
