#include "MathUtilities.h"

// Reverse-Z puts far at 0 and near at 1; the infinite variant drops the far plane
// (reference source/MathUtilities.cpp:3-38: only _33 and _43 change).
void ModifyPerspectiveMatrix(Matrix& mat, float nearPlane, float farPlane, bool bReverseZ, bool bInfiniteZ)
{
    float q1 = 0.0f, q2 = 0.0f;
    if (bReverseZ) {
        q1 = bInfiniteZ ? 0.0f : nearPlane / (farPlane - nearPlane);
        q2 = bInfiniteZ ? nearPlane : q1 * farPlane;
    } else {
        q1 = bInfiniteZ ? -1.0f : farPlane / (nearPlane - farPlane);
        q2 = bInfiniteZ ? -nearPlane : q1 * nearPlane;
    }
    mat.m[2][2] = q1;
    mat.m[3][2] = q2;
}

Matrix CreatePerspectiveFieldOfView(float fovY, float aspect, float nearPlane, float farPlane)
{
    const float h = 1.0f / std::tan(0.5f * fovY);
    const float w = h / aspect;
    const float range = farPlane / (nearPlane - farPlane);
    Matrix p{};
    p.m[0][0] = w;
    p.m[1][1] = h;
    p.m[2][2] = range;
    p.m[2][3] = -1.0f;
    p.m[3][2] = range * nearPlane;
    return p;
}
