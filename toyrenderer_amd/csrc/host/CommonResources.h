// CommonResources.h -- the samplers and dummy resources the path binds
// (source/CommonResources.cpp:157,270,276-287,298).  Samplers carry their description only: the
// back end samples the HZB in software with exactly the LinearClampMinReduction semantics.
#pragma once

#include "nvrhi_lite.h"

class CommonResources
{
public:
    void Initialize();

    nvrhi::SamplerHandle PointClampSampler;
    nvrhi::SamplerHandle LinearClampMinReductionSampler;   // min/mag/mip linear, clamp, SamplerReductionType::Minimum
    nvrhi::BufferHandle DummyUIntStructuredBuffer;         // bound when occlusion culling is off (BasePassRenderers.cpp:318-320)
    nvrhi::TextureHandle BlackTexture;                     // bound as HZB when occlusion culling is off (:357)
};
#define g_CommonResources (*Graphic::GetInstance().m_CommonResources)
