"""fp32 restatement of the glm calls the reference's scene graph makes on the host
(scene_node transforms, reference src/Pathtracing/scene.cpp:295-329, and the loader's
rotation order, src/PathtracingUtility/scene_loader.cpp:149-162).

Matrices are numpy float32 arrays m[c][r] (column-major like glm::mat4: m[c] is
column c); every operation rounds to fp32 in glm's published operand order.  These
feed the node table (inputs of the hot path); they are not on the hot path.
"""
import numpy as np

f32 = np.float32


def identity():
    return np.eye(4, dtype=np.float32)


def mat_mul(a, b):
    """glm operator*(mat4, mat4): column i = a[0]*b[i][0] + a[1]*b[i][1] + a[2]*b[i][2] + a[3]*b[i][3]."""
    out = np.empty((4, 4), dtype=np.float32)
    for i in range(4):
        out[i] = ((a[0] * b[i][0] + a[1] * b[i][1]) + a[2] * b[i][2]) + a[3] * b[i][3]
    return out


def translate(v):
    """glm::translate(vec3) (gtx/transform): identity with column 3 = m0*v0 + m1*v1 + m2*v2 + m3."""
    m = identity()
    v = np.asarray(v, dtype=np.float32)
    m[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
    return m


def scale(v):
    m = identity()
    v = np.asarray(v, dtype=np.float32)
    out = np.empty((4, 4), dtype=np.float32)
    out[0] = m[0] * v[0]
    out[1] = m[1] * v[1]
    out[2] = m[2] * v[2]
    out[3] = m[3]
    return out


def rotate(angle, axis):
    """glm::rotate(angle, axis) (gtx/transform) applied to the identity."""
    a = f32(angle)
    c = f32(np.cos(a))
    s = f32(np.sin(a))
    axis = np.asarray(axis, dtype=np.float32)
    axis = axis * (f32(1.0) / np.sqrt(f32(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2])))
    temp = (f32(1.0) - c) * axis
    r = np.zeros((4, 4), dtype=np.float32)
    r[0][0] = c + temp[0] * axis[0]
    r[0][1] = temp[0] * axis[1] + s * axis[2]
    r[0][2] = temp[0] * axis[2] - s * axis[1]
    r[1][0] = temp[1] * axis[0] - s * axis[2]
    r[1][1] = c + temp[1] * axis[1]
    r[1][2] = temp[1] * axis[2] + s * axis[0]
    r[2][0] = temp[2] * axis[0] + s * axis[1]
    r[2][1] = temp[2] * axis[1] - s * axis[0]
    r[2][2] = c + temp[2] * axis[2]
    m = identity()
    out = np.empty((4, 4), dtype=np.float32)
    out[0] = m[0] * r[0][0] + m[1] * r[0][1] + m[2] * r[0][2]
    out[1] = m[0] * r[1][0] + m[1] * r[1][1] + m[2] * r[1][2]
    out[2] = m[0] * r[2][0] + m[1] * r[2][1] + m[2] * r[2][2]
    out[3] = m[3]
    return out


def radians(deg):
    return f32(deg) * f32(0.01745329251994329576923690768489)


def inverse(m):
    """glm::inverse(mat4): cofactor expansion (glm/detail/func_matrix.inl compute_inverse<4,4>)."""
    m = np.asarray(m, dtype=np.float32)
    c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3]
    c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3]
    c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3]
    c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3]
    c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3]
    c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3]
    c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2]
    c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2]
    c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2]
    c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3]
    c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3]
    c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3]
    c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2]
    c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2]
    c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2]
    c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1]
    c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1]
    c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1]
    v = lambda *a: np.array(a, dtype=np.float32)
    fac0, fac1, fac2 = v(c00, c00, c02, c03), v(c04, c04, c06, c07), v(c08, c08, c10, c11)
    fac3, fac4, fac5 = v(c12, c12, c14, c15), v(c16, c16, c18, c19), v(c20, c20, c22, c23)
    vec0 = v(m[1][0], m[0][0], m[0][0], m[0][0])
    vec1 = v(m[1][1], m[0][1], m[0][1], m[0][1])
    vec2 = v(m[1][2], m[0][2], m[0][2], m[0][2])
    vec3 = v(m[1][3], m[0][3], m[0][3], m[0][3])
    inv0 = vec1 * fac0 - vec2 * fac1 + vec3 * fac2
    inv1 = vec0 * fac0 - vec2 * fac3 + vec3 * fac4
    inv2 = vec0 * fac1 - vec1 * fac3 + vec3 * fac5
    inv3 = vec0 * fac2 - vec1 * fac4 + vec2 * fac5
    sign_a = v(+1, -1, +1, -1)
    sign_b = v(-1, +1, -1, +1)
    inv = np.stack([inv0 * sign_a, inv1 * sign_b, inv2 * sign_a, inv3 * sign_b]).astype(np.float32)
    row0 = v(inv[0][0], inv[1][0], inv[2][0], inv[3][0])
    dot0 = m[0] * row0
    dot1 = (dot0[0] + dot0[1]) + (dot0[2] + dot0[3])
    return (inv * (f32(1.0) / dot1)).astype(np.float32)


def transpose(m):
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).T)


def normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return v * (f32(1.0) / np.sqrt(f32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
