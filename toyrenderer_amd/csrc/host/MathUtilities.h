// MathUtilities.h -- the few math helpers of the reference the path uses
// (source/MathUtilities.h:47-67, source/MathUtilities.cpp:3-38, SimpleMath wrappers).
#pragma once

#include <cmath>
#include <cstdint>

#include "../ShaderInterop.h"

using interop::Matrix;
using interop::Vector2U;
using interop::Vector3U;
using interop::Vector4;

// MathUtilities.h:47-61
constexpr uint32_t GetNextPow2(uint32_t x)
{
    if (x == 0) return 1;
    --x;
    x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
    return x + 1;
}

// MathUtilities.h:64-67
constexpr uint32_t DivideAndRoundUp(uint32_t dividend, uint32_t divisor) { return (dividend + divisor - 1) / divisor; }

inline Matrix Transpose(const Matrix& m)
{
    Matrix t;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t.m[i][j] = m.m[j][i];
    return t;
}

// Matrix product a * b in float32, every element summed left to right with separate multiplies and adds (the build
// is -ffp-contract=off): the ONE definition of m_WorldToClip = WorldToView * ViewToClip on every host side of this
// repo (toyrenderer_amd/interop.py world_to_clip), so that the oracle and the GPU receive the same 16 numbers.
inline Matrix MultiplyNoFMA(const Matrix& a, const Matrix& b)
{
    Matrix r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; ++k) { const float p = a.m[i][k] * b.m[k][j]; acc = acc + p; }
            r.m[i][j] = acc;
        }
    return r;
}

// Vector4::Normalize (SimpleMath.inl:1129-1135 -> XMVector4Normalize).  DirectXMath is absent; the
// build's convention is v / sqrt(dot4) with the dot product as an FMA chain (DESIGN.md "Arithmetic").
inline Vector4 Normalize(const Vector4& v)
{
    const float d = std::fmaf(v.w, v.w, std::fmaf(v.z, v.z, std::fmaf(v.y, v.y, v.x * v.x)));
    const float l = std::sqrt(d);
    return Vector4{ v.x / l, v.y / l, v.z / l, v.w / l };
}

// The four numbers FrustumCull reads (culling.hlsli:11-12): x and z of the normalised left/right plane, y and z of the
// normalised top/bottom plane, from the projection's columns 0, 1 and 3 (BasePassRenderers.cpp:557-563 and
// GIRenderer.cpp:691-695 spell this out with a transposed matrix; same arithmetic, same order).
inline Vector4 CullingFrustumOf(const Matrix& viewToClip)
{
    const float (*p)[4] = viewToClip.m;
    const Vector4 sideways = Normalize(Vector4{ p[0][3] + p[0][0], p[1][3] + p[1][0], p[2][3] + p[2][0], p[3][3] + p[3][0] });
    const Vector4 upright = Normalize(Vector4{ p[0][3] + p[0][1], p[1][3] + p[1][1], p[2][3] + p[2][1], p[3][3] + p[3][1] });
    return Vector4{ sideways.x, sideways.z, upright.y, upright.z };
}

// MathUtilities.cpp:3-38
void ModifyPerspectiveMatrix(Matrix& mat, float nearPlane, float farPlane, bool bReverseZ, bool bInfiniteZ);
// XMMatrixPerspectiveFovRH (SimpleMath.inl:2193-2199)
Matrix CreatePerspectiveFieldOfView(float fovY, float aspect, float nearPlane, float farPlane);
