// compat/glm/glm.hpp — a small glm-compatible subset for the host drop-in.
//
// The reference's public API speaks glm types (glm::vec3, glm::mat4, ...; reference
// src/Pathtracing/scene.h:8-9, sampler.h:3, abstract_pathtracer.h:11-12), but glm is an
// un-vendored submodule there (.gitmodules:7-9) and is not installed in this image.  A
// deployment puts the real glm ahead of this directory on the include path; this header
// only exists so the host library and its callers build where glm is absent.  It is
// NOT used to compile any reference source and is not part of the oracle.
//
// Only what the rt:: API and its callers touch is provided.  Every formula is written
// in the operand order glm publishes (listed in oracle/pt_oracle.c's header), because
// node matrices built here feed hit decisions that must be bit-exact.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace glm {

struct vec2 {
    union { float x, r, s; };
    union { float y, g, t; };
    constexpr vec2() : x(0), y(0) {}
    constexpr explicit vec2(float v) : x(v), y(v) {}
    constexpr vec2(float a, float b) : x(a), y(b) {}
    float& operator[](int i) { return i == 0 ? x : y; }
    const float& operator[](int i) const { return i == 0 ? x : y; }
};

struct uvec2 {
    uint32_t x, y;
    constexpr uvec2() : x(0), y(0) {}
    template <typename A, typename B>
    constexpr uvec2(A a, B b) : x(static_cast<uint32_t>(a)), y(static_cast<uint32_t>(b)) {}
};

struct vec4;

struct vec3 {
    union { float x, r, s; };
    union { float y, g, t; };
    union { float z, b, p; };
    constexpr vec3() : x(0), y(0), z(0) {}
    constexpr explicit vec3(float v) : x(v), y(v), z(v) {}
    constexpr vec3(float a, float b_, float c) : x(a), y(b_), z(c) {}
    constexpr vec3(const vec2& v, float c) : x(v.x), y(v.y), z(c) {}
    inline vec3(const vec4& v); // truncating, like glm
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    const float& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3& operator-=(const vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3& operator*=(const vec3& o) { x *= o.x; y *= o.y; z *= o.z; return *this; }
    vec3& operator*=(float s_) { x *= s_; y *= s_; z *= s_; return *this; }
    vec3& operator/=(float s_) { x /= s_; y /= s_; z /= s_; return *this; }
};

struct vec4 {
    union { float x, r; };
    union { float y, g; };
    union { float z, b; };
    union { float w, a; };
    constexpr vec4() : x(0), y(0), z(0), w(0) {}
    constexpr explicit vec4(float v) : x(v), y(v), z(v), w(v) {}
    constexpr vec4(float a_, float b_, float c, float d) : x(a_), y(b_), z(c), w(d) {}
    constexpr vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
static_assert(sizeof(vec4) == 16 && sizeof(vec3) == 12 && sizeof(vec2) == 8, "packed vectors");

inline vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}

// ---- vec2
inline vec2 operator+(const vec2& a, const vec2& b) { return { a.x + b.x, a.y + b.y }; }
inline vec2 operator-(const vec2& a, const vec2& b) { return { a.x - b.x, a.y - b.y }; }
inline vec2 operator*(const vec2& a, float s) { return { a.x * s, a.y * s }; }
inline vec2 operator*(float s, const vec2& a) { return { s * a.x, s * a.y }; }

// ---- vec3
inline vec3 operator+(const vec3& a, const vec3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(const vec3& a, const vec3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator*(const vec3& a, const vec3& b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3 operator/(const vec3& a, const vec3& b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
inline vec3 operator*(const vec3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, const vec3& a) { return { s * a.x, s * a.y, s * a.z }; }
inline vec3 operator/(const vec3& a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline vec3 operator+(const vec3& a, float s) { return { a.x + s, a.y + s, a.z + s }; }
inline vec3 operator+(float s, const vec3& a) { return { s + a.x, s + a.y, s + a.z }; }
inline vec3 operator-(const vec3& a, float s) { return { a.x - s, a.y - s, a.z - s }; }
inline vec3 operator-(float s, const vec3& a) { return { s - a.x, s - a.y, s - a.z }; }
inline vec3 operator-(const vec3& a) { return { -a.x, -a.y, -a.z }; }
inline bool operator==(const vec3& a, const vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(const vec3& a, const vec3& b) { return !(a == b); }

// ---- vec4
inline vec4 operator+(const vec4& a, const vec4& b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline vec4 operator-(const vec4& a, const vec4& b) { return { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; }
inline vec4 operator*(const vec4& a, const vec4& b) { return { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; }
inline vec4 operator*(const vec4& a, float s) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }

// ---- geometric / common (glm/detail/func_geometric.inl, func_common.inl)
inline float dot(const vec3& a, const vec3& b)
{
    const vec3 t = a * b;
    return t.x + t.y + t.z;
}
inline float dot(const vec4& a, const vec4& b)
{
    const vec4 t = a * b;
    return (t.x + t.y) + (t.z + t.w);
}
inline vec3 cross(const vec3& a, const vec3& b)
{
    return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y };
}
inline float length(const vec3& v) { return std::sqrt(dot(v, v)); }
inline float length2(const vec3& v) { return dot(v, v); } // gtx/norm
inline float distance(const vec3& a, const vec3& b) { return length(b - a); }
inline vec3 normalize(const vec3& v) { return v * (1.0f / std::sqrt(dot(v, v))); } // v * inversesqrt(dot)
inline vec3 reflect(const vec3& I, const vec3& N) { return I - N * dot(N, I) * 2.0f; }

inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline vec3 mix(const vec3& x, const vec3& y, float a) { return x * (1.0f - a) + y * a; }
inline float min(float a, float b) { return (b < a) ? b : a; }
inline float max(float a, float b) { return (a < b) ? b : a; }
inline vec3 min(const vec3& a, const vec3& b) { return { min(a.x, b.x), min(a.y, b.y), min(a.z, b.z) }; }
inline vec3 max(const vec3& a, const vec3& b) { return { max(a.x, b.x), max(a.y, b.y), max(a.z, b.z) }; }
inline float abs(float v) { return std::fabs(v); }
inline float sqrt(float v) { return std::sqrt(v); }
inline float sin(float v) { return std::sin(v); }
inline float cos(float v) { return std::cos(v); }
inline float floor(float v) { return std::floor(v); }
inline float ceil(float v) { return std::ceil(v); }
inline float round(float v) { return std::round(v); }
inline float fract(float v) { return v - std::floor(v); }
inline vec2 fract(const vec2& v) { return { fract(v.x), fract(v.y) }; }
inline vec3 exp(const vec3& v) { return { std::exp(v.x), std::exp(v.y), std::exp(v.z) }; }
inline vec3 pow(const vec3& a, const vec3& b) { return { std::pow(a.x, b.x), std::pow(a.y, b.y), std::pow(a.z, b.z) }; }

template <typename T>
constexpr T pi() { return static_cast<T>(3.14159265358979323846264338327950288); }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// ---- mat4, column-major: m[c] is column c
struct mat4 {
    vec4 col[4];
    constexpr mat4() : col{ vec4(1, 0, 0, 0), vec4(0, 1, 0, 0), vec4(0, 0, 1, 0), vec4(0, 0, 0, 1) } {}
    constexpr explicit mat4(float d) : col{ vec4(d, 0, 0, 0), vec4(0, d, 0, 0), vec4(0, 0, d, 0), vec4(0, 0, 0, d) } {}
    constexpr mat4(const vec4& a, const vec4& b, const vec4& c, const vec4& d) : col{ a, b, c, d } {}
    vec4& operator[](int i) { return col[i]; }
    const vec4& operator[](int i) const { return col[i]; }
    inline mat4& operator*=(const mat4& o);
};
static_assert(sizeof(mat4) == 64, "mat4 is 16 packed floats");

template <typename M>
constexpr M identity() { return M(1.0f); }

inline vec4 operator*(const mat4& m, const vec4& v)
{
    // Mov0*m[0] + Mov1*m[1] added pairwise (type_mat4x4.inl operator*(mat, column))
    const vec4 a0 = m[0] * v.x + m[1] * v.y;
    const vec4 a1 = m[2] * v.z + m[3] * v.w;
    return a0 + a1;
}
inline mat4 operator*(const mat4& a, const mat4& b)
{
    mat4 r;
    for (int i = 0; i < 4; ++i) r[i] = ((a[0] * b[i][0] + a[1] * b[i][1]) + a[2] * b[i][2]) + a[3] * b[i][3];
    return r;
}
inline mat4& mat4::operator*=(const mat4& o) { return *this = *this * o; }
inline mat4 operator*(const mat4& m, float s) { return { m[0] * s, m[1] * s, m[2] * s, m[3] * s }; }

inline mat4 transpose(const mat4& m)
{
    mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) r[c][k] = m[k][c];
    return r;
}

// compute_inverse<4,4> (func_matrix.inl): cofactors, then one reciprocal of the determinant
inline mat4 inverse(const mat4& m)
{
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    const float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    const float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    const float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    const float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    const float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    const float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const vec4 fac0(c00, c00, c02, c03), fac1(c04, c04, c06, c07), fac2(c08, c08, c10, c11);
    const vec4 fac3(c12, c12, c14, c15), fac4(c16, c16, c18, c19), fac5(c20, c20, c22, c23);
    const vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]);
    const vec4 v1(m[1][1], m[0][1], m[0][1], m[0][1]);
    const vec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]);
    const vec4 v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    const vec4 inv0 = v1 * fac0 - v2 * fac1 + v3 * fac2;
    const vec4 inv1 = v0 * fac0 - v2 * fac3 + v3 * fac4;
    const vec4 inv2 = v0 * fac1 - v1 * fac3 + v3 * fac5;
    const vec4 inv3 = v0 * fac2 - v1 * fac4 + v2 * fac5;
    const vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
    const mat4 inv(inv0 * sa, inv1 * sb, inv2 * sa, inv3 * sb);
    const vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    const vec4 d0 = m[0] * row0;
    const float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    return inv * (1.0f / d1);
}

// gtx/transform one-argument forms = ext/matrix_transform applied to the identity
inline mat4 translate(const mat4& m, const vec3& v)
{
    mat4 r(m);
    r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    return r;
}
inline mat4 translate(const vec3& v) { return translate(mat4(1.0f), v); }

inline mat4 scale(const mat4& m, const vec3& v) { return { m[0] * v.x, m[1] * v.y, m[2] * v.z, m[3] }; }
inline mat4 scale(const vec3& v) { return scale(mat4(1.0f), v); }

inline mat4 rotate(const mat4& m, float angle, const vec3& v)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 axis = normalize(v);
    const vec3 temp = axis * (1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = temp.x * axis.y + s * axis.z;
    R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y;
    R[2][1] = temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    mat4 r;
    r[0] = m[0] * R[0][0] + m[1] * R[0][1] + m[2] * R[0][2];
    r[1] = m[0] * R[1][0] + m[1] * R[1][1] + m[2] * R[1][2];
    r[2] = m[0] * R[2][0] + m[1] * R[2][1] + m[2] * R[2][2];
    r[3] = m[3];
    return r;
}
inline mat4 rotate(float angle, const vec3& v) { return rotate(mat4(1.0f), angle, v); }

inline const float* value_ptr(const mat4& m) { return &m.col[0].x; }
inline float* value_ptr(mat4& m) { return &m.col[0].x; }
inline const float* value_ptr(const vec3& v) { return &v.x; }

} // namespace glm
