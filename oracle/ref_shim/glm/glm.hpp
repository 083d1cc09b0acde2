// Minimal stand-in for the subset of g-truc/glm (pinned by the reference at 5c46b9c, install.sh:32-33; not vendored in
// /root/reference and not fetchable here) that ext/diff_gaussian_rasterization_hair/cuda_rasterizer uses:
// vec3 / vec4 / mat3 (column-major, m[col][row]), operator* in glm's term order (k = 0, 1, 2), transpose, dot, length,
// max(vec3, scalar).  TEST INFRASTRUCTURE ONLY: it exists so that the reference's own CUDA sources, hipified into
// oracle/_ref/ by oracle/Makefile.ref, compile; nothing in the product includes it.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define GLM_FN __host__ __device__ inline

namespace glm {

struct vec3 {
    float x, y, z;
    GLM_FN vec3() : x(0.f), y(0.f), z(0.f) {}
    GLM_FN vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    GLM_FN explicit vec3(float s) : x(s), y(s), z(s) {}
    GLM_FN float& operator[](int i) { return (&x)[i]; }
    GLM_FN const float& operator[](int i) const { return (&x)[i]; }
    GLM_FN vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    GLM_FN vec3& operator+=(float s) { x += s; y += s; z += s; return *this; }
    GLM_FN vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
GLM_FN vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
GLM_FN vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
GLM_FN vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
GLM_FN vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
GLM_FN vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
GLM_FN vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
GLM_FN vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }

struct vec4 {
    float x, y, z, w;
    GLM_FN vec4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    GLM_FN vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
};

// glm::dot: componentwise product first, then x + y + z
GLM_FN float dot(const vec3& a, const vec3& b)
{
    const vec3 t = a * b;
    return t.x + t.y + t.z;
}
GLM_FN float length(const vec3& v) { return sqrtf(dot(v, v)); }
GLM_FN vec3 max(const vec3& v, float s) { return vec3(fmaxf(v.x, s), fmaxf(v.y, s), fmaxf(v.z, s)); }

struct mat3 {
    vec3 c[3];  // columns
    GLM_FN mat3() {}
    GLM_FN explicit mat3(float d) { c[0] = vec3(d, 0.f, 0.f); c[1] = vec3(0.f, d, 0.f); c[2] = vec3(0.f, 0.f, d); }
    // nine scalars fill column 0, then column 1, then column 2
    GLM_FN mat3(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
    {
        c[0] = vec3(x0, y0, z0); c[1] = vec3(x1, y1, z1); c[2] = vec3(x2, y2, z2);
    }
    GLM_FN vec3& operator[](int i) { return c[i]; }
    GLM_FN const vec3& operator[](int i) const { return c[i]; }
};
// result[col][row] = a[0][row] * b[col][0] + a[1][row] * b[col][1] + a[2][row] * b[col][2]
GLM_FN mat3 operator*(const mat3& a, const mat3& b)
{
    mat3 r;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++)
            r[col][row] = a[0][row] * b[col][0] + a[1][row] * b[col][1] + a[2][row] * b[col][2];
    return r;
}
GLM_FN mat3 operator*(float s, const mat3& a)
{
    mat3 r;
    for (int col = 0; col < 3; col++) r[col] = a[col] * s;
    return r;
}
GLM_FN mat3 operator*(const mat3& a, float s) { return s * a; }
GLM_FN mat3 transpose(const mat3& a)
{
    mat3 r;
    for (int col = 0; col < 3; col++)
        for (int row = 0; row < 3; row++) r[col][row] = a[row][col];
    return r;
}

}  // namespace glm
