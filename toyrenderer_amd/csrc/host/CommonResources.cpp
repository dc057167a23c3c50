#include "CommonResources.h"

#include "Graphic.h"

void CommonResources::Initialize()
{
    nvrhi::DeviceHandle device = g_Graphic.m_NVRHIDevice;
    nvrhi::SamplerDesc point;
    point.minFilter = point.magFilter = point.mipFilter = false;
    PointClampSampler = device->createSampler(point);
    nvrhi::SamplerDesc minReduction;                                          // CommonResources.cpp:276-287,298
    minReduction.reductionType = nvrhi::SamplerReductionType::Minimum;
    LinearClampMinReductionSampler = device->createSampler(minReduction);

    nvrhi::BufferDesc dummy;                                                  // CommonResources.cpp:270
    dummy.byteSize = 16;
    dummy.structStride = sizeof(uint32_t);
    dummy.canHaveUAVs = true;
    dummy.debugName = "DummyUIntStructuredBuffer";
    DummyUIntStructuredBuffer = device->createBuffer(dummy);

    nvrhi::TextureDesc black;                                                 // CommonResources.cpp:157
    black.width = black.height = 1;
    black.format = nvrhi::Format::R16_FLOAT;
    black.debugName = "BlackTexture";
    BlackTexture = device->createTexture(black);
    nvrhi::CommandListHandle cl = g_Graphic.AllocateCommandList();
    {
        SCOPED_COMMAND_LIST_AUTO_QUEUE(cl, "CommonResources::Initialize");
        cl->clearBufferUInt(DummyUIntStructuredBuffer, 0);
        cl->clearTextureFloat(BlackTexture, nvrhi::AllSubresources, nvrhi::Color{ 0.0f });
    }
}
