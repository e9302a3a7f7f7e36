// ptk_device.h — device-side data layout and the intersect / shade functions
// shared by every kernel in ptk.hip (gfx950 only).
//
// Arithmetic contract: every expression that feeds a hit DECISION is written in
// the same operand order as the reference (file:line cited at each function,
// relative to /root/reference/src/Pathtracing) with glm's published evaluation
// order (see oracle/pt_oracle.c header), and the translation unit is built with
// -ffp-contract=off and correctly rounded fp32 divide / sqrt, so node and
// primitive ids replay the reference's x86-64 -O2 build bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptk.h"

namespace ptk {

// ---------------------------------------------------------------- HBM layout
// Node table: one 256-B record per scene node.  Indexed by a wave-uniform loop
// counter in intersect_scene, so hipcc fetches it through the scalar cache.
struct alignas(16) DNode {
    float inv[16]; // scene_node::m_inv_transform    (scene.h:289)
    float xf[16];  // scene_node::m_transform        (scene.h:288)
    float nrm[16]; // scene_node::m_normal_transform (scene.h:290)
    uint32_t kind, mesh;
    uint32_t albedo, emission, roughness, metallic;
    // Conservative world-space bounds of everything the node can be hit at (host: make_cull_bounds
    // in ptk.hip).  A ray that misses [wmin - m, wmax + m] with m = cull_k * |o|_1 + cull_c cannot be
    // reported as hitting this node by the reference's local-space tests; see intersect_scene_lanelist.
    float wmin[3], wmax[3];
    float cull_k, cull_c;
    // Work class for the per-lane candidate loops: lanes of a wave run in lockstep, so candidates
    // are processed class by class to keep the triangle-loop trip counts alike (kClass*).
    uint32_t cls;
    uint32_t uses_uv; // 1 if any of the node's four samplers reads an image (else raycast_result::uv is never looked at)
};
// Compact per-node record of the balanced traversal's cull loop (intersect_scene_balanced): the
// world box and margins of DNode, plus the common face normal of a planar mesh in world space
// (fn, unit length; zero if the node has none) and the cosine from which a ray direction is
// certain to face every triangle of the node from behind (ftau; 2 = never).
struct alignas(16) DCull {
    // { wmin - m, wmax + m } per axis, one packed operand each: the node's world box with the margin of a ray whose origin has
    // |o|_1 <= DScene::cull_o1max already applied, m = cull_k * cull_o1max + cull_c (rays further out are never culled)
    float bx[2], by[2], bz[2];
    float cull_k, cull_c;
    float fn[3], ftau;
    uint32_t cls;
    uint32_t tree_slot; // class kClassTree: position of the node in DScene::tree_nodes (split mode's pending bit)
    uint32_t pad[2];
};
static_assert(sizeof(DCull) == 64, "DCull must be 64 B");
enum { kClassSphere = 0, kClassSmallLeaf = 1, kClassLeaf = 2, kClassTree = 3, kNumClasses = 4 };
constexpr uint32_t kSmallLeafMax = 4u; // a one-leaf mesh with at most this many triangles (plane.obj: 2)
static_assert(sizeof(DNode) == 256, "DNode must be 256 B");

// Flattened KD node, 64 B = four 16-B words:
//   h = { a, b | axis<<30, right, split }   header
//        interior: a = skip index (first node after the subtree), b = 0, right = index of
//                  the right child (the left child is the next slot; right == own index + 1 means
//                  there is no left child, right == a no right child), split = the plane the
//                  node's box is cut at (scene.cpp:240-251), axis = depth % 3
//        leaf:     a = first leaf-ordered primitive, b = primitive count (> 0)
//   x = { cmin.x, cmin.y, cmin.z, cmax.x }   c* = the reference's cut box (scene.cpp:251):
//   y = { cmax.y, cmax.z, tmin.x, tmin.y }        decides whether the node is entered
//   z = { tmin.z, tmax.x, tmax.y, tmax.z }   t* = padded box of the triangles below the node
//                                                 (ptk_kd_node::tmin/tmax): pruning only
struct alignas(16) DKdNode {
    uint32_t a, b_axis, right;
    float split;
    float cmin[3];
    float cmax[3];
    float tmin[3];
    float tmax[3];
};
static_assert(sizeof(DKdNode) == 64, "DKdNode must be 64 B");
constexpr uint32_t kKdCountMask = 0x3FFFFFFFu;

// Per-mesh header.
struct alignas(16) DMesh {
    uint32_t kd_base;    // first DKdNode of this mesh
    uint32_t kd_count;   // number of DKdNodes
    uint32_t leaf_base;  // first leaf-ordered hit record / prim id
    uint32_t shade_base; // first shading record (original triangle order)
};

// Hit-test record, 64 B, stored once per (leaf, triangle) in leaf order so a
// leaf's triangles stream contiguously (the reference keeps by-value copies in
// every leaf too, scene.cpp:222-223):
//   q0 = v0.xyz, N.x   q1 = N.yz, e0.xy   q2 = e0.z, e1.xyz   q3 = d00 d01 d11 inv_den
// (triangle::update, scene.cpp:91-107).
// Shading record, 64 B, original triangle order, fetched for the winning hit only:
//   s0 = n0.xyz, n1.x  s1 = n1.yz, n2.xy  s2 = n2.z, uv0.xy, uv1.x  s3 = uv1.y, uv2.xy, 0

struct DSampler {
    uint32_t kind;
    float r, g, b;
    uint32_t texture, mode, pad0, pad1;
};

// Root leaf of a one-leaf mesh node: the reference's root box (scene.cpp:190), first leaf slot, triangle count.
struct alignas(16) DLeaf {
    float cmin[3];
    uint32_t first;
    float cmax[3];
    uint32_t count;
};

struct DTexture {
    const float* rgb;
    uint32_t w, h;
};

struct DScene {
    const DNode* nodes;
    const DCull* cull;
    const DMesh* meshes;
    const DKdNode* kd;
    const float4* tri_hit;     // 4 float4 per leaf-ordered primitive
    const float4* tri_pair;    // 8 float4 per pair of leaf slots (2k, 2k+1) of the one-leaf meshes, interleaved
    const uint32_t* leaf_prim; // original triangle index per leaf-ordered primitive
    // Copies that shorten dependent load chains (a wave waits for every link): what the balanced
    // traversal needs of a one-leaf mesh node (instead of node -> mesh -> KD root), the shading record
    // of a leaf slot (instead of node -> mesh, slot -> triangle), a node's four samplers (instead of
    // node -> index -> sampler).
    const struct DLeaf* leaf_info;    // per scene node
    const uint32_t* leaf_shade;       // per leaf slot: index into tri_shade (shade_base + triangle)
    const struct DSampler* node_samplers; // per scene node: albedo, emission, roughness, metallic
    const float4* tri_shade;   // 4 float4 per original triangle
    const DSampler* samplers;
    const DTexture* textures;
    const uint32_t* tree_nodes; // nodes of class kClassTree, in node order
    uint32_t num_trees;
    uint32_t num_nodes;
    uint32_t bg_kind;
    float bg_r, bg_g, bg_b;
    uint32_t bg_texture, bg_mode;
    float cam_x, cam_y, cam_z;
    uint32_t walk_stack; // pending far children a KD walk may hold before it sweeps a subtree without a stack (PTK_OPT_WALK_STACK)
    float cull_o1max;    // the |o|_1 the margins in DCull::bx.. were built for (every point of the scene's boxes, with slack)
#ifdef PTK_STATS
    unsigned long long* stats; // experiment builds only: work counters of the balanced traversal
#endif
#ifdef PTK_EXP
    // experiment builds only (scripts/stage_budget.sh): the launch stops after a stage of the bounce kernel, so that
    // per-dispatch instruction counters of successive builds give a per-stage budget.  0 = whole kernel, 1 = load +
    // cull loop only, 2 = + ray -> local / sphere / box stage, 3 = + triangle stage (whole cast), no shading.
    uint32_t exp_stage;
#endif
};
#ifdef PTK_STATS
#define PTK_STAT(S, k, v) do { if (lane_id() == 0u && (blockIdx.x & 15u) == 0u) atomicAdd((S).stats + (k), (unsigned long long)(v)); } while (0) // every 16th block: same-line atomics are slow enough to distort the clocks
#define PTK_CLOCK() ((unsigned long long)__builtin_readcyclecounter())
#ifdef PTK_STATS_COARSE // clocks only: the walker's per-pass counters are atomics too and stretch what they count
#define PTK_STAT_FINE(S, k, v) do { } while (0)
#else
#define PTK_STAT_FINE(S, k, v) PTK_STAT(S, k, v)
#endif
// inside divergent code: counts one event per wave (k) and the lanes taking part (k + 1)
#ifdef PTK_STATS_COARSE
#define PTK_STAT_ACTIVE(S, k) do { } while (0)
#else
#define PTK_STAT_ACTIVE(S, k) do { const uint64_t m_ = __builtin_amdgcn_ballot_w64(true); \
    if (lane_id() == (uint32_t)__builtin_ctzll(m_)) { atomicAdd((S).stats + (k), 1ull); atomicAdd((S).stats + (k) + 1, (unsigned long long)__popcll(m_)); } } while (0)
#endif
#define PTK_STAT_ACTIVE_NS(k) do { } while (0)
#else
#define PTK_STAT_ACTIVE(S, k) do { } while (0)
#define PTK_STAT_ACTIVE_NS(k) do { } while (0)
#define PTK_STAT(S, k, v) do { } while (0)
#define PTK_STAT_FINE(S, k, v) do { } while (0)
#define PTK_CLOCK() 0ull
#endif

// abstract_pathtracer.cpp:20-25
struct DView {
    float fx, fy, fz; // forward
    float rx, ry, rz; // right
    float ux, uy, uz; // up
    float h2, w2;
    uint32_t width, height;
};

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ uint32_t lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------- vec helpers
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{ x, y, z }; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 mulv(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b)
{
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
__device__ __forceinline__ V3 mix(V3 x, V3 y, float a) { return x * (1.0f - a) + y * a; }

// ---- shading arithmetic --------------------------------------------------------------------------------
// Everything above replays the reference's rounding because it feeds hit DECISIONS.  What follows a decided hit
// (normal, bounce direction, weights: pathtracer.cpp:34-60, rng.cpp:12-36) is held to a tolerance instead
// (tests/test_gpu_probes.py: 2e-5 on unit vectors and weights against the oracle), so it may use fused
// multiply-adds and the hardware's 1-ulp reciprocal square root / square root: about a quarter of the
// shading stage's instructions.
__device__ __forceinline__ float sh_dot(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 sh_normalize(V3 a) { return a * __builtin_amdgcn_rsqf(sh_dot(a, a)); }
__device__ __forceinline__ V3 sh_madd(V3 a, float s, V3 c) // a * s + c
{
    return mk(__builtin_fmaf(a.x, s, c.x), __builtin_fmaf(a.y, s, c.y), __builtin_fmaf(a.z, s, c.z));
}
__device__ __forceinline__ V3 sh_mix(V3 x, V3 y, float a) { return sh_madd(y, a, x * (1.0f - a)); }
__device__ __forceinline__ V3 sh_cross(V3 a, V3 b)
{
    return mk(__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)), __builtin_fmaf(a.x, b.y, -(b.x * a.y)));
}
// mat4 * vec4(p, 0) -> xyz
__device__ __forceinline__ V3 sh_mat_dir(const float* __restrict__ m, V3 p)
{
    return mk(__builtin_fmaf(m[0], p.x, __builtin_fmaf(m[4], p.y, m[8] * p.z)), __builtin_fmaf(m[1], p.x, __builtin_fmaf(m[5], p.y, m[9] * p.z)),
              __builtin_fmaf(m[2], p.x, __builtin_fmaf(m[6], p.y, m[10] * p.z)));
}

// sin and cos of 2 pi u for the hemisphere sample (rng.cpp:29-31), shading only: one argument
// reduction and two polynomials (error < 1e-7 on [0, 2 pi)) instead of the library's full-range
// sinf and cosf, which differ from the host libm's in the last place just the same.  (Since round 2 the
// shading normalisations use the 1-ulp hardware rsq / sqrt as well, see the sh_* helpers above; hit decisions
// never do.)
// atan2 and asin for the equirectangular background lookup (sampler.cpp:116-124): shading arithmetic, never a hit decision.
// atan(a) on [0, 1] as a * P(a^2) (degree 7 in a^2, near-minimax: 1.9e-7 rad in fp32), the octant folded back afterwards;
// asin(x) on [0, 1] as pi/2 - sqrt(1 - x) Q(x) (degree 7: 2.9e-7 rad).  The map coordinate u = atan2 / 2 pi therefore sits
// within 3e-8 of the exact one and v = asin / pi within 1e-7, less than an ulp of a coordinate near 1; the library routines
// (correctly rounded divides included) cost about four times the instructions.
__device__ __forceinline__ float sh_atan2(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mn * __builtin_amdgcn_rcpf(mx);
    a = mx > 0.0f ? a : 0.0f; // atan2(0, 0) = 0
    const float s = a * a;
    float p = -0.004668773151934147f;
    p = __builtin_fmaf(p, s, 0.02416618913412094f);
    p = __builtin_fmaf(p, s, -0.0593671016395092f);
    p = __builtin_fmaf(p, s, 0.09906096756458282f);
    p = __builtin_fmaf(p, s, -0.14016585052013397f);
    p = __builtin_fmaf(p, s, 0.19969235360622406f);
    p = __builtin_fmaf(p, s, -0.33331960439682007f);
    p = __builtin_fmaf(p, s, 0.9999998807907104f);
    float r = p * a;
    r = ay > ax ? 1.57079632679489661923f - r : r;
    r = x < 0.0f ? 3.14159265358979323846f - r : r;
    return __builtin_copysignf(r, y);
}
__device__ __forceinline__ float sh_asin(float x)
{
    const float ax = fabsf(x);
    float p = -0.0012370048789307475f;
    p = __builtin_fmaf(p, ax, 0.006580885034054518f);
    p = __builtin_fmaf(p, ax, -0.01696547120809555f);
    p = __builtin_fmaf(p, ax, 0.030808253213763237f);
    p = __builtin_fmaf(p, ax, -0.05014502629637718f);
    p = __builtin_fmaf(p, ax, 0.08897409588098526f);
    p = __builtin_fmaf(p, ax, -0.21459849178791046f);
    p = __builtin_fmaf(p, ax, 1.570796251296997f);
    const float r = __builtin_fmaf(-__builtin_amdgcn_sqrtf(fmaxf(1.0f - ax, 0.0f)), p, 1.57079632679489661923f);
    return __builtin_copysignf(r, x);
}

__device__ __forceinline__ void sincos_2pi_shade(float u, float& s, float& c)
{
    // u in [0, 1): quarter turns q = round(4u) in {0..4}, r = 2 pi u - q pi/2 in [-pi/4, pi/4]
    const float phi = 3.14159265358979323846f * 2.0f * u; // rng.cpp:29
    const float q = rintf(phi * 0.636619772367581343f);
    float r = __builtin_fmaf(q, -1.57079637050628662f, phi);
    r = __builtin_fmaf(q, 4.37113882867379e-8f, r);
    const float z = r * r;
    const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                                    z * z, __builtin_fmaf(-0.5f, z, 1.0f));
    const uint32_t qi = (uint32_t)(int)q;
    const float a = (qi & 1u) ? cp : sp, b = (qi & 1u) ? sp : cp;
    s = (qi & 2u) ? -a : a;
    c = ((qi + 1u) & 2u) ? -b : b;
}

// glm mat4 * vec4(p, w) -> xyz; column-major m[c*4+r].
__device__ __forceinline__ V3 mat_point(const float* __restrict__ m, V3 p, float w)
{
    V3 r;
    r.x = (m[0] * p.x + m[4] * p.y) + (m[8] * p.z + m[12] * w);
    r.y = (m[1] * p.x + m[5] * p.y) + (m[9] * p.z + m[13] * w);
    r.z = (m[2] * p.x + m[6] * p.y) + (m[10] * p.z + m[14] * w);
    return r;
}

// ---------------------------------------------------------------- intersectors

// scene.cpp:58-75  bounding_box::intersect (quirks kept: six divides, NaN-ignoring
// fmin/fmax, and the `t9 != -1` test that rejects an entry distance of exactly -1), in two
// steps: the six plane quotients, then the decision.  A KD child's box is its parent's box
// with one plane moved to the split (scene.cpp:31-56), so a walk that keeps the parent's
// six quotients only has to divide once per node to get both children's; the values are the
// ones the reference computes (same operands, same IEEE divide).
__device__ __forceinline__ bool slab_decide(float t1, float t2, float t3, float t4, float t5, float t6)
{
    const float t7 = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float t8 = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    const float t9 = (t8 < 0 || t7 > t8) ? -1.0f : t7;
    return t9 != -1.0f;
}

__device__ __forceinline__ bool box_intersect(float minx, float miny, float minz, float maxx, float maxy,
                                              float maxz, V3 o, V3 d)
{
    const float t1 = (minx - o.x) / d.x;
    const float t2 = (maxx - o.x) / d.x;
    const float t3 = (miny - o.y) / d.y;
    const float t4 = (maxy - o.y) / d.y;
    const float t5 = (minz - o.z) / d.z;
    const float t6 = (maxz - o.z) / d.z;
    return slab_decide(t1, t2, t3, t4, t5, t6);
}

// bounding_box::intersect decided from approximate quotients, when that is certain.  q = (b - o) * rcp(d) is within
// 2^-22 |q| of the IEEE quotient t the reference computes (b - o is the same subtraction; v_rcp_f32 is good to 1 ulp, the
// product and the reference's divide to half an ulp each), and min / max keep such bounds, so with
// delta = 5e-7 * max|q| + 1e-30 every entry distance e_a = min(q, q') and exit distance x_a = max(q, q') of an axis is
// within delta of the reference's.  The reference accepts iff  t8 >= 0, t7 <= t8 and t7 != -1  with t7 = max_a e_a,
// t8 = min_a x_a (scene.cpp:58-75).  t7 <= t8 means e_a <= x_b for all a, b; for a == b that holds by construction
// (also for a flat box, where both are the same quotient), so only the pairs a != b need a margin:
//   m_a = min(x_b, x_c) - e_a  must clear 2 delta,  t8' = min3(x) must clear delta,  t7' = max3(e) + 1 must clear delta,
// and every quotient must be finite and below 1e30 (a zero or denormal direction component, NaN, overflow: undecided).
// Returns the decision and sets `certain`; an undecided lane takes box_intersect.  A ray misses certainty only when it
// passes within a relative 1e-6 of a box edge, so a wave rarely holds one.
__device__ __forceinline__ bool box_intersect_fast(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                                                   V3 o, V3 d, bool& certain)
{
    const float rx = __builtin_amdgcn_rcpf(d.x), ry = __builtin_amdgcn_rcpf(d.y), rz = __builtin_amdgcn_rcpf(d.z);
    const float q1 = (minx - o.x) * rx, q2 = (maxx - o.x) * rx;
    const float q3 = (miny - o.y) * ry, q4 = (maxy - o.y) * ry;
    const float q5 = (minz - o.z) * rz, q6 = (maxz - o.z) * rz;
    const float ex = fminf(q1, q2), xx = fmaxf(q1, q2);
    const float ey = fminf(q3, q4), xy = fmaxf(q3, q4);
    const float ez = fminf(q5, q6), xz = fmaxf(q5, q6);
    const float big = fmaxf(fmaxf(fmaxf(fabsf(q1), fabsf(q2)), fabsf(q3)), fmaxf(fmaxf(fabsf(q4), fabsf(q5)), fabsf(q6)));
    const float delta = __builtin_fmaf(5e-7f, big, 1e-30f);
    const float mx = fminf(xy, xz) - ex, my = fminf(xx, xz) - ey, mz = fminf(xx, xy) - ez;
    const float t8 = fminf(fminf(xx, xy), xz), t7 = fmaxf(fmaxf(ex, ey), ez);
    const float m_lo = fminf(fminf(mx, my), mz);
    const float m_abs = fminf(fminf(fabsf(mx), fabsf(my)), fabsf(mz));
    // every comparison is written so that a NaN gives "not certain"
    certain = (big < 1e30f) && (m_abs > 2.0f * delta) && (fabsf(t8) > delta) && (fabsf(t7 + 1.0f) > delta);
    return m_lo >= 0.0f && t8 >= 0.0f;
}

// Closest-hit record carried through the node loop.  For a triangle a* holds the
// barycentrics (u,v,w of scene.cpp:78-89); for a sphere it holds the local hit
// position, from which normal and uv are rebuilt for the winner only.
struct Hit {
    float d0;   // length2(pos_world - origin), scene.cpp:383
    int node;   // index in scene.nodes, -1 = miss
    int prim;   // leaf-ordered primitive slot (mesh) or -1 (sphere)
    float a0, a1, a2;
    V3 pw;      // world position, scene.cpp:375
};

// scene.cpp:142-186 mesh::intersect_triangle up to the hit decision, and the squared
// distance intersect_internal ranks hits by (scene.cpp:196).  Returns true when the
// triangle is hit; bar = (u,v,w) of triangle::baricentric, p = the hit position.
__device__ __forceinline__ bool tri_eval(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o,
                                         V3 d, float& dd_out, V3& bar_out, V3& p_out)
{
    const V3 v0 = mk(q0.x, q0.y, q0.z);
    const V3 N = mk(q0.w, q1.x, q1.y);
    const V3 l = o - v0;
    const float distance = dot(l, N);
    if (distance < 0) return false;
    const float cosine = dot(d, N);
    if (cosine >= 0) return false;
    const V3 projection = o + d * (distance / -cosine);
    // triangle::baricentric, scene.cpp:78-89
    const V3 e0 = mk(q1.z, q1.w, q2.x);
    const V3 e1 = mk(q2.y, q2.z, q2.w);
    const V3 v2 = projection - v0;
    const float d20 = dot(v2, e0);
    const float d21 = dot(v2, e1);
    const float v = (q3.z * d20 - q3.y * d21) * q3.w;
    const float w = (q3.x * d21 - q3.y * d20) * q3.w;
    const float u = 1.0f - v - w;
    if (u >= 0 && v >= 0 && w >= 0) {
        const V3 dv = o - projection;
        dd_out = dot(dv, dv);
        bar_out = mk(u, v, w);
        p_out = projection;
        return true;
    }
    return false;
}

// The same without early returns, for callers that test a triangle per lane with the whole wave (the walker's pooled leaf
// stage): every lane runs every operation anyway, and as compiled the three nested exits of tri_eval cost an exec-mask region
// each plus a fresh set of zeroed result registers per level (about thirty moves per call).  Same operations in the same
// order, so a lane that hits gets the same values; the decision is the conjunction of the same comparisons (a NaN distance
// or cosine passes its test in both forms and fails the barycentric ones).
__device__ __forceinline__ bool tri_eval_flat(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o,
                                              V3 d, float& dd_out, V3& bar_out, V3& p_out)
{
    const V3 v0 = mk(q0.x, q0.y, q0.z);
    const V3 N = mk(q0.w, q1.x, q1.y);
    const V3 l = o - v0;
    const float distance = dot(l, N);
    const float cosine = dot(d, N);
    const V3 projection = o + d * (distance / -cosine);
    const V3 e0 = mk(q1.z, q1.w, q2.x);
    const V3 e1 = mk(q2.y, q2.z, q2.w);
    const V3 v2 = projection - v0;
    const float d20 = dot(v2, e0);
    const float d21 = dot(v2, e1);
    const float v = (q3.z * d20 - q3.y * d21) * q3.w;
    const float w = (q3.x * d21 - q3.y * d20) * q3.w;
    const float u = 1.0f - v - w;
    const V3 dv = o - projection;
    dd_out = dot(dv, dv);
    bar_out = mk(u, v, w);
    p_out = projection;
    return !(distance < 0) && !(cosine >= 0) && (u >= 0 && v >= 0 && w >= 0);
}

// tri_eval plus the bookkeeping of intersect_internal (scene.cpp:196-202).  Returns true
// and updates (best_d, best_slot, bar, p) when this triangle is hit and strictly closer,
// or exactly as close but earlier in the reference's visiting order (`slot` = position in
// the leaf-ordered primitive list = DFS order).  Walked in the reference's own order the
// slot only ever grows, so the second clause never fires and this is the strict `<` of
// scene.cpp:198; it is what makes any other visiting order return the same triangle.
// best_slot starts at -1 (no hit yet), so a distance equal to the initial FLT_MAX is
// rejected as the reference's strict compare rejects it.
__device__ __forceinline__ bool tri_test(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o,
                                         V3 d, int slot, float& best_d, int& best_slot, V3& bar_out, V3& p_out)
{
    float dd;
    V3 bar, p;
    if (!tri_eval(q0, q1, q2, q3, o, d, dd, bar, p)) return false;
    if (dd < best_d || (dd == best_d && slot < best_slot)) {
        best_d = dd;
        best_slot = slot;
        bar_out = bar;
        p_out = p;
        return true;
    }
    return false;
}

// One leaf's triangles, per-lane addresses (scene.cpp:192-203).
__device__ __forceinline__ void leaf_triangles(const DScene& S, uint32_t first, uint32_t count, V3 o, V3 d,
                                               float& best_d, int& best_slot, bool& found, V3& bar_out, V3& p_out)
{
    for (uint32_t k = 0; k < count; ++k) {
        PTK_STAT_ACTIVE(S, 18);
        const float4* t = S.tri_hit + (size_t)(first + k) * 4;
        const float4 q0 = t[0], q1 = t[1], q2 = t[2], q3 = t[3];
        if (tri_test(q0, q1, q2, q3, o, d, (int)(first + k), best_d, best_slot, bar_out, p_out)) found = true;
    }
}

// ---- KD walk, reference order -------------------------------------------------
// mesh::intersect + intersect_internal, scene.cpp:115-121,188-212, on the flattened
// preorder tree.  The reference's recursion "box hit -> own triangles, then left
// subtree, then right subtree" is the loop "box hit or leaf -> i+1, interior box miss ->
// skip"; visiting order and the strict `<` are unchanged.  No stack is needed because
// the reference never orders or prunes children.  `start` = 1 when the caller has
// already tested (and entered) the root.
__device__ __forceinline__ void kd_walk_reference(const DScene& S, const DMesh mesh, V3 o, V3 d, uint32_t start,
                                                  float& best_d, int& best_slot, bool& found, V3& bar_out, V3& p_out)
{
    const DKdNode* __restrict__ kd = S.kd + mesh.kd_base;
    uint32_t i = start;
    while (i < mesh.kd_count) {
        const float4* __restrict__ q = reinterpret_cast<const float4*>(kd + i);
        const float4 h = q[0], x = q[1], y = q[2];
        const uint32_t a = __float_as_uint(h.x);
        const uint32_t b = __float_as_uint(h.y) & kKdCountMask;
        const bool hit = box_intersect(x.x, x.y, x.z, x.w, y.x, y.y, o, d);
        if (hit && b != 0u) leaf_triangles(S, mesh.leaf_base + a, b, o, d, best_d, best_slot, found, bar_out, p_out);
        i = (hit || b != 0u) ? i + 1u : a;
    }
}

// ---- KD walk, pruned ----------------------------------------------------------
// Same result as kd_walk_reference, far fewer nodes.  What the reference returns is
// fully described by: among all (leaf slot, triangle) pairs whose leaf is reached through
// cut boxes that all pass bounding_box::intersect, the pair with the smallest squared
// distance, the lowest slot on exact ties.  So
//   * a node is entered only if the reference's own box test passes (same arithmetic);
//   * children are visited near-first: the two cut boxes of a node are separated by the
//     split plane, so the sign of the ray direction on the split axis gives front-to-back;
//   * a subtree is skipped when its padded triangle box (DKdNode::tmin/tmax) is missed, is
//     behind the origin, or starts farther than the best hit so far.  Every hit position
//     lies inside that box, so a skipped subtree holds no hit at all or only hits that are
//     strictly farther than the current best (margins below cover fp32 rounding of both the
//     slab arithmetic and the hit distance), hence neither a winner nor a tie;
//   * exact ties between visited candidates go to the lower slot (tri_test).
// The stack holds pending far children, one per level at most.  It is SHORT: the walker kernel keeps
// kWalkStack entries per lane in LDS, the per-lane loops of the other kernels a small local array.  A walk
// that needs more (chains of unsplittable triangles run the tree down to depth 100, scene.cpp:220) does not
// spill: at the node where the stack is full it stops ordering children and sweeps that node's subtree in
// preorder with the skip links ("box passes or leaf -> next slot, interior node fails -> its skip link"),
// every node still subject to the reference's box test and to the distance pruning.  The sweep visits the
// same leaves a stack would have reached, in another order, and the (distance, slot) rule makes the order
// irrelevant; when it ends the walk pops its short stack as usual.
constexpr int kWalkStack = 12;  // LDS entries per lane in k_walk (8 B each: 6 KB per wave)
constexpr int kKdStack = 16;    // local entries of the closed per-lane loops (kd_walk_pruned)

// Where a walk keeps its pending far children: (node index, lower bound of the squared hit distance below it).
struct LocalStack {
    uint32_t* idx;
    float* bound;
#ifdef PTK_STATS
    unsigned long long* stats = nullptr;
#endif
    __device__ __forceinline__ void put(int sp, uint32_t i, float b) const { idx[sp] = i; bound[sp] = b; }
    __device__ __forceinline__ uint32_t node(int sp) const { return idx[sp]; }
    __device__ __forceinline__ float dist(int sp) const { return bound[sp]; }
};
struct LdsStack { // entry e of lane l at [e * 64 + l]: lanes side by side, no bank conflicts
    uint32_t* idx;
    float* bound;
#ifdef PTK_STATS
    unsigned long long* stats = nullptr;
#endif
    __device__ __forceinline__ void put(int sp, uint32_t i, float b) const { idx[sp * 64] = i; bound[sp * 64] = b; }
    __device__ __forceinline__ uint32_t node(int sp) const { return idx[sp * 64]; }
    __device__ __forceinline__ float dist(int sp) const { return bound[sp * 64]; }
};

// Returns true when the subtree under a node can be skipped; `bound` receives a lower bound
// of the squared distance of any hit below it (0 when the origin may be inside the box).
// y = {.., .., tmin.x, tmin.y}, z = {tmin.z, tmax.x, tmax.y, tmax.z} of the node record.
__device__ __forceinline__ bool kd_prune(const float4 y, const float4 z, V3 o, V3 inv, float oabs, float best_d,
                                         float& bound)
{
    const float ax = (y.z - o.x) * inv.x, bx = (z.y - o.x) * inv.x;
    const float ay = (y.w - o.y) * inv.y, by = (z.z - o.y) * inv.y;
    const float az = (z.x - o.z) * inv.z, bz = (z.w - o.z) * inv.z;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    // every comparison is false on NaN, i.e. "do not prune"
    const bool missed = (t_in - t_out) > 1e-5f * (fabsf(t_in) + fabsf(t_out)) + 2.0f * oabs;
    const bool behind = t_out < 0.0f && t_out * 0.99999f < -oabs;
    const float near_lo = t_in * 0.99999f - oabs;
    bound = near_lo > 0.0f ? near_lo * near_lo : 0.0f;
    return missed || behind || bound > best_d;
}

// Loop shape ("while-while"): all lanes of a wave first descend through interior nodes,
// testing BOTH children of the current node per step (their loads are issued together, and
// a child that fails never touches the stack), then all lanes that reached a leaf test its
// triangles.  The six plane quotients of the current node's cut box are carried in
// registers: one IEEE divide per step yields both children's reference box tests
// (slab_decide).  A pending far child is pushed with its distance bound, so a pop can
// discard it against the best hit found meanwhile without touching memory; a node that
// is popped gets its six quotients recomputed from its own box.
//
// The walk is written as a resumable state (KdWalk) + one "round" (descend to leaves, test
// them), so that the same code serves the closed per-lane loop (kd_walk_pruned) and the
// walker kernel, which interleaves rounds with refilling idle lanes (k_walk in ptk.hip).
constexpr uint32_t kWalkNone = 0xFFFFFFFFu;
constexpr uint32_t kWalkPop = 0xFFFFFFFEu; // walker kernel only: the lane waits for its next stack entry (walk_pop)
struct KdWalk {
    const DKdNode* kd;     // the mesh's first node
    uint32_t leaf_base;    // DMesh::leaf_base
    V3 o, d, inv;          // local ray, approximate reciprocal direction
    float oabs;
    uint32_t neg;          // bits 0-2: sign bits of d; bit 3: a stackless sweep is under way (next node in cr, end in cs);
                           // bit 4: the ray's box tests always take the exact path (a zero, denormal or non-finite component)
    uint32_t cur, ca, cb, cr; // node being processed (kWalkNone = walk finished) and its header words
    float cs;
    // The node's cut box as per-axis entry / exit distances along the ray: e = min, x = max of the two plane quotients of
    // the axis, computed as (plane - o) * rcp(d) (see slab_certified).
    float ex, xx, ey, xy, ez, xz;
    int sp;
    uint32_t leaf_done;    // triangles of the current leaf already tested (walker kernel's pooled leaf stage)
    float best_d;
    int best_slot;
    bool found;
    V3 bar, p;
};
constexpr uint32_t kWalkExact = 16u; // KdWalk::neg bit: no approximate box decisions for this ray

// bounding_box::intersect (scene.cpp:58-75) decided from approximate quotients, when that is certain: the walker's form of
// box_intersect_fast above.  t7 = the largest entry distance, t8 = the smallest exit distance of the box, both built from
// quotients q = (plane - o) * rcp(d), each within eps |q| of the IEEE quotient q* the reference divides out, eps < 2^-21
// (same subtraction, then a 1-ulp reciprocal and a product instead of a correctly rounded divide).  The maximum of such
// values is within eps |t7| of the reference's t7*: t7* >= e*[a0] >= t7 - eps |t7| for the axis a0 that gives t7, and every
// e*[a] <= e[a] + eps |e[a]| <= t7 + eps |t7| (for e[a] >= 0 because e[a] <= t7; for e[a] < 0 because then e[a] (1 - eps) is
// below t7 (1 - eps) or below 0 <= t7); likewise t8.  The reference enters the box iff t8* >= 0, t7* <= t8* and t7* != -1.
// Entry and exit of ONE axis can never cross in either arithmetic (division and multiplication are monotone), so the
// decision is certain when t8 - t7, t8 and t7 + 1 clear delta = 5e-7 max(|t7|, |t8|) + 1e-30; anything else (a ray within a
// relative 1e-6 of a box edge or corner, an exit at t = 0, an entry at t = -1, a flat box, non-finite values) goes to the
// exact test.  Rays with a zero, denormal or non-finite component never come here (kWalkExact).
__device__ __forceinline__ bool slab_certified(float t7, float t8, bool& certain)
{
    const float delta = __builtin_fmaf(5e-7f, fmaxf(fabsf(t7), fabsf(t8)), 1e-30f);
    certain = (fabsf(t8 - t7) > 2.0f * delta) && (fabsf(t8) > delta) && (fabsf(t7 + 1.0f) > delta); // false on NaN / inf
    return t7 <= t8 && t8 >= 0.0f;
}

// v_max_f32 / v_min_f32 as they are (maxNum / minNum: a NaN operand yields the other one), without the compiler's quieting
// of operands it cannot prove canonical
__device__ __forceinline__ float raw_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float raw_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// (re)starts at node `idx`: header words + entry / exit distances from its own cut box
__device__ __forceinline__ void walk_enter(KdWalk& W, uint32_t idx)
{
    const float4* __restrict__ q = reinterpret_cast<const float4*>(W.kd + idx);
    const float4 h = q[0], x = q[1], y = q[2];
    W.cur = idx;
    W.ca = __float_as_uint(h.x);
    W.cb = __float_as_uint(h.y);
    W.cr = __float_as_uint(h.z);
    W.cs = h.w;
    const float q1 = (x.x - W.o.x) * W.inv.x, q2 = (x.w - W.o.x) * W.inv.x;
    const float q3 = (x.y - W.o.y) * W.inv.y, q4 = (y.x - W.o.y) * W.inv.y;
    const float q5 = (x.z - W.o.z) * W.inv.z, q6 = (y.y - W.o.z) * W.inv.z;
    W.ex = fminf(q1, q2); W.xx = fmaxf(q1, q2);
    W.ey = fminf(q3, q4); W.xy = fmaxf(q3, q4);
    W.ez = fminf(q5, q6); W.xz = fmaxf(q5, q6);
}

constexpr uint32_t kWalkSweep = 8u; // KdWalk::neg bit: stackless sweep

// Next node for a lane that has finished one: continues a stackless sweep if one is under way (stops at the
// next leaf whose boxes pass: the caller tests its triangles and comes back), else pops the short stack.
template <typename Stack>
__device__ __forceinline__ void walk_pop(KdWalk& W, const Stack& st)
{
    W.cur = kWalkNone;
    if (W.neg & kWalkSweep) {
        uint32_t i = W.cr;
        const uint32_t end = __float_as_uint(W.cs);
#if defined(PTK_STATS) && defined(PTK_STATS_COARSE)
        [[maybe_unused]] uint32_t sweep_iter = 0u;
#define PTK_SWEEP_COUNT() (++sweep_iter)
#define PTK_SWEEP_REPORT() do { atomicAdd(st.stats + 62, (unsigned long long)sweep_iter); atomicMax(st.stats + 63, (unsigned long long)sweep_iter); } while (0)
#else
#define PTK_SWEEP_COUNT() ((void)0)
#define PTK_SWEEP_REPORT() ((void)0)
#endif
        while (i < end) {
            PTK_SWEEP_COUNT();
            const float4* __restrict__ q = reinterpret_cast<const float4*>(W.kd + i);
            const float4 h = q[0], x = q[1], y = q[2], z = q[3];
            const uint32_t a = __float_as_uint(h.x), b = __float_as_uint(h.y);
            const float t1 = (x.x - W.o.x) / W.d.x, t2 = (x.w - W.o.x) / W.d.x;
            const float t3 = (x.y - W.o.y) / W.d.y, t4 = (y.x - W.o.y) / W.d.y;
            const float t5 = (x.z - W.o.z) / W.d.z, t6 = (y.y - W.o.z) / W.d.z;
            float bound;
            const bool pass = slab_decide(t1, t2, t3, t4, t5, t6) && !kd_prune(y, z, W.o, W.inv, W.oabs, W.best_d, bound);
            const bool leaf = (b & kKdCountMask) != 0u;
            if (pass && leaf) {
                W.cur = i;
                W.ca = a;
                W.cb = b;
                W.cr = i + 1u;
                PTK_SWEEP_REPORT();
                return;
            }
            i = (pass || leaf) ? i + 1u : a; // an interior node that fails takes its skip link
        }
        W.neg &= ~kWalkSweep;
        PTK_SWEEP_REPORT();
    }
    while (W.sp > 0) {
        --W.sp;
        if (st.dist(W.sp) > W.best_d) continue;
        walk_enter(W, st.node(W.sp));
        break;
    }
}

// Sets the walk up on `mesh` for local ray (o, d) and runs the root's two tests.  With work = false
// the walk is born finished.
__device__ __forceinline__ void walk_begin(KdWalk& W, const DScene& S, const DMesh mesh, V3 o, V3 d, bool work)
{
    W.kd = S.kd + mesh.kd_base;
    W.leaf_base = mesh.leaf_base;
    W.o = o;
    W.d = d;
    W.inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    W.oabs = 1e-5f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) + 1e-30f;
    W.neg = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    W.cur = kWalkNone;
    W.ca = W.cb = W.cr = 0u;
    W.cs = 0.0f;
    W.ex = W.xx = W.ey = W.xy = W.ez = W.xz = 0.0f;
    // rays the approximate quotients say nothing reliable about: a zero / denormal direction component (rcp = inf), any
    // non-finite or absurdly large coordinate
    const float mag = (fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) + (fabsf(W.inv.x) + fabsf(W.inv.y) + fabsf(W.inv.z));
    if (!(mag < 1e30f)) W.neg |= kWalkExact; // also on NaN
    W.sp = 0;
    W.leaf_done = 0u;
    W.best_d = 3.402823466e+38f;
    W.best_slot = -1;
    W.found = false;
    W.bar = W.p = mk(0.0f, 0.0f, 0.0f);
    if (work) {
        walk_enter(W, 0u);
        const float4* __restrict__ q = reinterpret_cast<const float4*>(W.kd);
        const float4 bx = q[1], by = q[2];
        bool certain;
        bool enter = slab_certified(fmaxf(fmaxf(W.ex, W.ey), W.ez), fminf(fminf(W.xx, W.xy), W.xz), certain);
        if (!certain || (W.neg & kWalkExact)) enter = box_intersect(bx.x, bx.y, bx.z, bx.w, by.x, by.y, o, d); // scene.cpp:190 as written
        float bound;
        if (!enter || kd_prune(q[2], q[3], o, W.inv, W.oabs, W.best_d, bound)) W.cur = kWalkNone;
    }
}

// One descend step for a lane that stands on an interior node (the caller checks that).  DEFER_POP: a
// lane whose two children both fail does not pop at once (a pop re-enters a node: six IEEE divides, for
// the one or two lanes that need it) but parks itself as kWalkPop; the walker kernel pops such lanes
// together.
template <bool DEFER_POP, typename Stack>
__device__ __forceinline__ void walk_descend_step(KdWalk& W, [[maybe_unused]] const DScene& S, const Stack& st, int cap)
{
    PTK_STAT_ACTIVE(S, 16);
    const DKdNode* __restrict__ kd = W.kd;
    const V3 o = W.o, d = W.d;
    const uint32_t cur = W.cur, ca = W.ca, cb = W.cb, cr = W.cr;
    const uint32_t li = cur + 1u;
    const bool has_r = cr < ca;           // ca = end of this subtree
    const bool has_l = li < ca && cr != li; // right == cur + 1: the next slot is a right-only child (kd_children in ptk.hip)
    const uint32_t ri = has_r ? cr : cur;
    const float4* __restrict__ ql = reinterpret_cast<const float4*>(kd + (has_l ? li : cur));
    const float4* __restrict__ qr = reinterpret_cast<const float4*>(kd + ri);
    const float4 lh = ql[0], ly = ql[2], lz = ql[3];
    const float4 rh = qr[0], ry = qr[2], rz = qr[3];
    // the one new plane: both children's boxes are this node's box cut at `cs` on axis k.  Seen along the ray, the left
    // child (its max plane moved to the cut) ends earlier on that axis when the ray runs upwards and starts later when it
    // runs downwards; the right child (min plane moved) the other way round.  The other two axes are shared.
    const uint32_t k = cb >> 30;
    const float ok = k == 0u ? o.x : (k == 1u ? o.y : o.z);
    // (a select between fields of W is turned into an indexed load, which pushes the whole walk state into scratch memory:
    // the empty asm hides where the three values come from)
    float ix = W.inv.x, iy = W.inv.y, iz = W.inv.z;
    asm volatile("" : "+v"(ix), "+v"(iy), "+v"(iz));
    const float ik = k == 0u ? ix : (k == 1u ? iy : iz);
    const float ts = (W.cs - ok) * ik;
    const bool down = ((W.neg >> k) & 1u) != 0u; // d[k] < 0
    // (every select below picks between computed values)
    const float ex = W.ex, ey = W.ey, ez = W.ez, xx = W.xx, xy = W.xy, xz = W.xz;
    // (raw_max / raw_min: the six distances come through selects and loop-carried registers, where the compiler no longer
    // knows them to be canonical and would put a quieting v_max x, x in front of every fminf / fmaxf)
    const float e_yz = raw_max(ey, ez), e_xz = raw_max(ex, ez), e_xy = raw_max(ex, ey);
    const float x_yz = raw_min(xy, xz), x_xz = raw_min(xx, xz), x_xy = raw_min(xx, xy);
    const float e12 = k == 0u ? e_yz : (k == 1u ? e_xz : e_xy); // entry / exit of the two other axes together
    const float x12 = k == 0u ? x_yz : (k == 1u ? x_xz : x_xy);
    const float e_all = raw_max(e_yz, ex), x_all = raw_min(x_yz, xx); // the node's own box
    // one child has its ENTRY on axis k moved to the cut, the other its EXIT
    const float a7 = raw_max(e12, ts), a8 = x_all; // entry moved
    const float b7 = e_all, b8 = raw_min(x12, ts); // exit moved
    // left / right picked as distances, before the decisions (a select between two decisions costs four instructions: the
    // compiler turns each into a 0 / 1 register and back)
    const float l7 = down ? a7 : b7, l8 = down ? a8 : b8;
    const float r7 = down ? b7 : a7, r8 = down ? b8 : a8;
    bool cl, cr_;
    bool bl_in = slab_certified(l7, l8, cl);
    bool br_in = slab_certified(r7, r8, cr_);
    const bool exact_only = (W.neg & kWalkExact) != 0u;
    const bool redo_l = has_l && (!cl || exact_only), redo_r = has_r && (!cr_ || exact_only);
    if (__builtin_amdgcn_ballot_w64(redo_l || redo_r) != 0ull) { // rare: the reference's own six divides on the child's own box
        if (redo_l) {
            const float4 lx = ql[1];
            bl_in = box_intersect(lx.x, lx.y, lx.z, lx.w, ly.x, ly.y, o, d);
        }
        if (redo_r) {
            const float4 rx = qr[1];
            br_in = box_intersect(rx.x, rx.y, rx.z, rx.w, ry.x, ry.y, o, d);
        }
    }
    float bl, br;
    // (both prune tests unconditionally: the wave runs them anyway, and without the short-circuit they are one straight
    // stretch of arithmetic instead of two exec-mask regions)
    const bool pl = kd_prune(ly, lz, o, W.inv, W.oabs, W.best_d, bl);
    const bool pr = kd_prune(ry, rz, o, W.inv, W.oabs, W.best_d, br);
    const bool el = has_l && bl_in && !pl;
    const bool er = has_r && br_in && !pr;
    const bool right_first = ((W.neg >> k) & 1u) != 0u;
    const bool take_r = er && (right_first || !el);
    if constexpr (DEFER_POP) {
        // The walker kernel's form: every field of the state is written once, at the end, as a select (as nested branches,
        // below, each path defines its own version of the state and the compiler joins them with more register copies).
        const bool both = el && er;
        const bool over = both && W.sp >= cap; // short stack full: sweep this node's subtree [cur + 1, ca) instead (see kWalkStack)
        const bool go = (el || er) && !over;
        if (both && !over) st.put(W.sp, take_r ? li : ri, take_r ? bl : br);
        W.sp += (both && !over) ? 1 : 0;
        const bool entry_moved = take_r != down; // the child taken: left has its entry moved when the ray runs down, right when up
        W.cur = go ? (take_r ? ri : li) : kWalkPop;
        W.ca = go ? __float_as_uint(take_r ? rh.x : lh.x) : W.ca;
        W.cb = go ? __float_as_uint(take_r ? rh.y : lh.y) : W.cb;
        W.cr = go ? __float_as_uint(take_r ? rh.z : lh.z) : (over ? li : W.cr);
        W.cs = go ? (take_r ? rh.w : lh.w) : (over ? __uint_as_float(ca) : W.cs);
        W.neg |= over ? kWalkSweep : 0u;
        W.ex = (go && k == 0u && entry_moved) ? ts : ex; W.xx = (go && k == 0u && !entry_moved) ? ts : xx;
        W.ey = (go && k == 1u && entry_moved) ? ts : ey; W.xy = (go && k == 1u && !entry_moved) ? ts : xy;
        W.ez = (go && k == 2u && entry_moved) ? ts : ez; W.xz = (go && k == 2u && !entry_moved) ? ts : xz;
        return;
    }
    if (el && er && W.sp >= cap) {
        // short stack full: no ordering below this node, sweep its subtree [cur + 1, ca) instead (see kWalkStack)
        W.neg |= kWalkSweep;
        W.cr = li;
        W.cs = __uint_as_float(ca);
        if (DEFER_POP) W.cur = kWalkPop;
        else walk_pop(W, st);
        return;
    }
    if (el && er) {
        st.put(W.sp, take_r ? li : ri, take_r ? bl : br);
        ++W.sp;
    }
    if (el || er) {
        W.cur = take_r ? ri : li;
        W.ca = __float_as_uint(take_r ? rh.x : lh.x);
        W.cb = __float_as_uint(take_r ? rh.y : lh.y);
        W.cr = __float_as_uint(take_r ? rh.z : lh.z);
        W.cs = take_r ? rh.w : lh.w;
        const bool entry_moved = take_r != down; // the child taken: left has its entry moved when the ray runs down, right when up
        W.ex = (k == 0u && entry_moved) ? ts : ex; W.xx = (k == 0u && !entry_moved) ? ts : xx;
        W.ey = (k == 1u && entry_moved) ? ts : ey; W.xy = (k == 1u && !entry_moved) ? ts : xy;
        W.ez = (k == 2u && entry_moved) ? ts : ez; W.xz = (k == 2u && !entry_moved) ? ts : xz;
    } else if (DEFER_POP) {
        W.cur = kWalkPop;
    } else {
        walk_pop(W, st);
    }
}

// One round of the closed per-lane loop: descend until the lane stands on a leaf (or has run out of
// nodes), then test that leaf and pop.
__device__ __forceinline__ void walk_round(KdWalk& W, const DScene& S, const LocalStack& st, int cap)
{
    while (W.cur != kWalkNone && (W.cb & kKdCountMask) == 0u) walk_descend_step<false>(W, S, st, cap);
    if (W.cur != kWalkNone) {
        leaf_triangles(S, W.leaf_base + W.ca, W.cb & kKdCountMask, W.o, W.d, W.best_d, W.best_slot, W.found, W.bar, W.p);
        walk_pop(W, st);
    }
}

__device__ __forceinline__ void kd_walk_pruned(const DScene& S, const DMesh mesh, V3 o, V3 d, bool work,
                                               float& best_d, int& best_slot, bool& found, V3& bar_out, V3& p_out)
{
    uint32_t stack_i[kKdStack];
    float stack_t[kKdStack];
    const LocalStack st{ stack_i, stack_t };
    const int cap = (int)S.walk_stack < kKdStack ? (int)S.walk_stack : kKdStack;
    KdWalk W;
    walk_begin(W, S, mesh, o, d, work);
    // results so far (a one-leaf root the caller already tested) carry over
    W.best_d = best_d;
    W.best_slot = best_slot;
    W.found = found;
    W.bar = bar_out;
    W.p = p_out;
    while (W.cur != kWalkNone) walk_round(W, S, st, cap);
    best_d = W.best_d;
    best_slot = W.best_slot;
    found = W.found;
    bar_out = W.bar;
    p_out = W.p;
}

// mesh::intersect (scene.cpp:115-121) for one lane.  `entered_root`: the caller has
// already run the reference's box test on the root and it passed.
template <bool PRUNE>
__device__ __forceinline__ bool mesh_traverse_lane(const DScene& S, const DMesh mesh, V3 o, V3 d, bool work,
                                                   bool entered_root, int& prim_out, V3& bar_out, V3& p_out)
{
    float best_d = 3.402823466e+38f;
    int best_slot = -1;
    bool found = false;
    const DKdNode* __restrict__ kd = S.kd + mesh.kd_base;
    if (PRUNE) {
        // a one-leaf mesh whose root the caller already entered needs no walk at all
        const bool flat = work && entered_root && mesh.kd_count == 1u;
        if (flat) leaf_triangles(S, mesh.leaf_base + kd->a, kd->b_axis & kKdCountMask, o, d, best_d, best_slot, found, bar_out, p_out);
        kd_walk_pruned(S, mesh, o, d, work && !flat, best_d, best_slot, found, bar_out, p_out);
    } else {
        uint32_t start = work ? 0u : mesh.kd_count;
        if (work && entered_root) {
            const uint32_t a = kd->a, b = kd->b_axis & kKdCountMask;
            if (b != 0u) leaf_triangles(S, mesh.leaf_base + a, b, o, d, best_d, best_slot, found, bar_out, p_out);
            start = 1u; // a leaf root has kd_count == 1
        }
        kd_walk_reference(S, mesh, o, d, start, best_d, best_slot, found, bar_out, p_out);
    }
    prim_out = found ? best_slot : prim_out;
    return found;
}

// mesh::intersect for a whole wave walking the same mesh node (traversal mode 0).
template <bool PRUNE>
__device__ __forceinline__ bool mesh_intersect(const DScene& S, const DMesh mesh, V3 o, V3 d, bool lane_on,
                                               int& prim_out, V3& bar_out, V3& p_out)
{
    if (mesh.kd_count == 1u) {
        // Single-leaf mesh (plane.obj, cube.obj: the 1.5x rule keeps them whole).
        // Every lane walks the same triangles, so the records come in through
        // wave-uniform (scalar) loads and only the box test is per lane.
        float best_d = 3.402823466e+38f;
        int best_slot = -1;
        bool found = false;
        const DKdNode* __restrict__ kn = S.kd + mesh.kd_base;
        const bool inside = lane_on && box_intersect(kn->cmin[0], kn->cmin[1], kn->cmin[2], kn->cmax[0], kn->cmax[1],
                                                     kn->cmax[2], o, d);
        if (__builtin_amdgcn_ballot_w64(inside) != 0ull) {
            const uint32_t first = mesh.leaf_base + kn->a;
            const uint32_t count = kn->b_axis & kKdCountMask;
            for (uint32_t k = 0; k < count; ++k) {
                const float4* t = S.tri_hit + (size_t)(first + k) * 4;
                const float4 q0 = t[0], q1 = t[1], q2 = t[2], q3 = t[3];
                if (inside && tri_test(q0, q1, q2, q3, o, d, (int)(first + k), best_d, best_slot, bar_out, p_out))
                    found = true;
            }
        }
        prim_out = found ? best_slot : prim_out;
        return found;
    }
    return mesh_traverse_lane<PRUNE>(S, mesh, o, d, lane_on, false, prim_out, bar_out, p_out);
}

// sphere::intersect up to the hit position, scene.cpp:401-432.
__device__ __forceinline__ bool sphere_intersect(V3 o, V3 d, V3& p_out)
{
    const float projection = dot(mk(0.0f, 0.0f, 0.0f) - o, d);
    const float sq_distance = dot(o, o) - projection * projection;
    if (sq_distance > 1.0f) return false;
    const float offset = sqrtf(1.0f - sq_distance);
    const float t1 = projection - offset;
    const float t2 = projection + offset;
    if (t1 < 0 && t2 < 0) return false;
    p_out = o + d * (t1 >= 0.0f ? t1 : t2);
    return true;
}

// The same without early returns (see tri_eval_flat): every lane of the wave runs every operation anyway; a line that misses
// the ball gives a NaN offset, which fails nothing that the first comparison has not already failed.
__device__ __forceinline__ bool sphere_intersect_flat(V3 o, V3 d, V3& p_out)
{
    const float projection = dot(mk(0.0f, 0.0f, 0.0f) - o, d);
    const float sq_distance = dot(o, o) - projection * projection;
    const float offset = sqrtf(1.0f - sq_distance);
    const float t1 = projection - offset;
    const float t2 = projection + offset;
    p_out = o + d * (t1 >= 0.0f ? t1 : t2);
    return !(sq_distance > 1.0f) && !(t1 < 0 && t2 < 0);
}

// scene::cast_ray, scene.cpp:355-399.  All lanes of a wave walk the node table
// together (there is no top-level structure in the reference: every ray tests
// every node), so the node record and the sphere/mesh branch are wave-uniform.
template <bool PRUNE>
__device__ __forceinline__ void intersect_scene(const DScene& S, V3 o, V3 d, bool lane_on, Hit& best)
{
    best.d0 = 3.402823466e+38f;
    best.node = -1;
    best.prim = -1;
    best.a0 = best.a1 = best.a2 = 0.0f;
    best.pw = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t n = 0; n < S.num_nodes; ++n) {
        const DNode* __restrict__ nd = S.nodes + n;
        // operator*(mat4, ray), scene.cpp:18-23
        const V3 lo = mat_point(nd->inv, o, 1.0f);
        const V3 ld = normalize(mat_point(nd->inv, d, 0.0f));
        bool hit;
        int prim = -1;
        V3 p = mk(0.0f, 0.0f, 0.0f), a = mk(0.0f, 0.0f, 0.0f);
        if (nd->kind == PTK_SHAPE_SPHERE) {
            hit = lane_on && sphere_intersect(lo, ld, p);
            a = p;
        } else {
            hit = mesh_intersect<PRUNE>(S, S.meshes[nd->mesh], lo, ld, lane_on, prim, a, p);
        }
        if (hit) {
            const V3 pw = mat_point(nd->xf, p, 1.0f); // scene.cpp:375
            const V3 dv = pw - o;
            const float d0 = dot(dv, dv);             // scene.cpp:383
            if (d0 < best.d0) {                       // scene.cpp:386
                best.d0 = d0;
                best.node = (int)n;
                best.prim = prim;
                best.a0 = a.x;
                best.a1 = a.y;
                best.a2 = a.z;
                best.pw = pw;
            }
        }
    }
}

// ---- traversal mode 1b: per-lane candidate lists --------------------------------
// The reference tests every ray against every node (scene.cpp:361): ray -> local (two
// mat-vecs + a normalise), then the shape.  In scenes like Room a ray can only hit one or
// two of the ten nodes, so mode 1 splits cast_ray in two phases:
//   1. wave-uniform loop over the node table: a cheap, CONSERVATIVE world-space slab test
//      of the ray against each node's padded bounds (DNode::wmin/wmax); a per-lane bit mask
//      records the nodes that survive;
//   2. per-lane loops over that mask (spheres, then meshes): every lane takes ITS next
//      candidate and runs the reference's arithmetic on it: ray -> local, sphere / box +
//      triangles / KD walk, hit -> world.
// Phase 1 never decides a hit, it only discards nodes the reference could not report:
// the reference reports a hit only if its fp32 local-space test passes, i.e. if the ray
// passes within rounding distance of the node's local box (meshes: mesh::m_bounds, which the
// root box test enforces, scene.cpp:190; spheres: the unit ball).  Mapped to world space that
// is the node's world box grown by at most (error of M^-1 and of the fp32 chain) ~
// cond(M) * 2^-23 * (|o| + |box|); the host stores 256 x that bound as cull_k / cull_c and
// disables the test (infinite box) for ill-conditioned or inconsistent matrices.
// The reference's "first node wins exact ties" (strict < in node order, scene.cpp:386) is
// kept by an explicit (d0, node index) comparison, so processing order does not matter.

__device__ __forceinline__ void best_update(Hit& best, float d0, int n, int prim, V3 a, V3 pw)
{
    if (d0 < best.d0 || (d0 == best.d0 && n < best.node)) {
        best.d0 = d0;
        best.node = n;
        best.prim = prim;
        best.a0 = a.x;
        best.a1 = a.y;
        best.a2 = a.z;
        best.pw = pw;
    }
}

// operator*(mat4, ray) (scene.cpp:18-23) with the matrix read per lane (node differs per lane)
__device__ __forceinline__ void lane_ray_to_local(const float4* __restrict__ nv, V3 o, V3 d, V3& lo, V3& ld)
{
    const float4 i0 = nv[0], i1 = nv[1], i2 = nv[2], i3 = nv[3];
    lo = mk((i0.x * o.x + i1.x * o.y) + (i2.x * o.z + i3.x * 1.0f),
            (i0.y * o.x + i1.y * o.y) + (i2.y * o.z + i3.y * 1.0f),
            (i0.z * o.x + i1.z * o.y) + (i2.z * o.z + i3.z * 1.0f));
    ld = normalize(mk((i0.x * d.x + i1.x * d.y) + (i2.x * d.z + i3.x * 0.0f),
                      (i0.y * d.x + i1.y * d.y) + (i2.y * d.z + i3.y * 0.0f),
                      (i0.z * d.x + i1.z * d.y) + (i2.z * d.z + i3.z * 0.0f)));
}

// hit -> world (scene.cpp:375) + distance (scene.cpp:383), matrix read per lane
__device__ __forceinline__ void lane_hit_to_world(const float4* __restrict__ nv, V3 o, V3 p, V3& pw, float& d0)
{
    const float4 x0 = nv[4], x1 = nv[5], x2 = nv[6], x3 = nv[7];
    pw = mk((x0.x * p.x + x1.x * p.y) + (x2.x * p.z + x3.x * 1.0f),
            (x0.y * p.x + x1.y * p.y) + (x2.y * p.z + x3.y * 1.0f),
            (x0.z * p.x + x1.z * p.y) + (x2.z * p.z + x3.z * 1.0f));
    const V3 dv = pw - o;
    d0 = dot(dv, dv);
}

// The bundle pre-cull's test of one node (see intersect_scene_balanced, "Bundle pre-cull"): can any ray of a wave whose
// rays lie within eo (origins) and ed (directions, both L1) of the central ray (oc, dc) touch the node's padded world box?
__device__ __forceinline__ bool bundle_keep(const DCull cu, V3 oc, V3 dc, float eo, float ed)
{
    const V3 invc = mk(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dc.x), -1e30f, 1e30f),
                       __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dc.y), -1e30f, 1e30f),
                       __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dc.z), -1e30f, 1e30f));
    const float o1c = fabsf(oc.x) + fabsf(oc.y) + fabsf(oc.z) + eo;
    const float m = __builtin_fmaf(cu.cull_k, o1c, cu.cull_c);
    const float cx = 0.5f * (cu.bx[0] + cu.bx[1]), cy = 0.5f * (cu.by[0] + cu.by[1]), cz = 0.5f * (cu.bz[0] + cu.bz[1]);
    const float R = 0.5f * ((cu.bx[1] - cu.bx[0]) + (cu.by[1] - cu.by[0]) + (cu.bz[1] - cu.bz[0])) + 3.0f * m;
    const float T = fabsf(oc.x - cx) + fabsf(oc.y - cy) + fabsf(oc.z - cz) + eo + R;
    const float g = 1.01f * (m + eo + T * ed) + 1e-30f; // total inflation, with slack for this arithmetic
    const float tx0 = (cu.bx[0] - g - oc.x) * invc.x, tx1 = (cu.bx[1] + g - oc.x) * invc.x;
    const float ty0 = (cu.by[0] - g - oc.y) * invc.y, ty1 = (cu.by[1] + g - oc.y) * invc.y;
    const float tz0 = (cu.bz[0] - g - oc.z) * invc.z, tz1 = (cu.bz[1] + g - oc.z) * invc.z;
    const float t_in = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1));
    const float t_out = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1));
    const bool missed = (t_in - t_out) > 1e-5f * (fabsf(t_in) + fabsf(t_out)); // false on NaN = keep
    const bool finite_box = g < 1e30f;                                         // false on NaN / overflow = keep
    return !(finite_box && (missed || t_out < 0.0f));
}

// BALLS: every node of the scene is a sphere that carries its bounding ball in place of its world box (DNode::wmin = centre,
// wmax[0] = radius = the largest singular value of the node's 3x3, rounded up; ptk_scene_upload sets uses_uv bit 1 and the
// host picks this instantiation only when all nodes have it: a per-node branch between the two tests costs more than either).
// The ball is the shape itself for a uniformly scaled sphere, so what survives the cull are (nearly) the true hits, and the
// test is cheaper than six slab planes: materials.json +7.7 %.
// BUNDLE (scenes of at most 64 nodes): the wave's rays are nearly one ray (primary rays: 64 samples of a pixel), so one test per
// node for the whole wave (lane = node, bundle_keep) first says which nodes any of them can touch; only those get the per-ray
// test.  The rays of a coherent wave then hold the same one or two candidates, and each lane working through its own list
// costs no more rounds than pooling them would.
template <bool PRUNE, bool BALLS = false, bool BUNDLE = false>
__device__ __forceinline__ void intersect_scene_lanelist(const DScene& S, V3 o, V3 d, bool lane_on, Hit& best)
{
    best.d0 = 3.402823466e+38f;
    best.node = -1;
    best.prim = -1;
    best.a0 = best.a1 = best.a2 = 0.0f;
    best.pw = mk(0.0f, 0.0f, 0.0f);
    const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const float o1 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
    // unit direction for the ball test (callers of the parity entry may pass any length; a zero or non-finite direction
    // gives NaN here, and a NaN never culls)
    const V3 dn = d * __builtin_amdgcn_rsqf(__builtin_fmaf(d.x, d.x, __builtin_fmaf(d.y, d.y, d.z * d.z)));
    uint64_t bundle_mask = ~0ull; // nodes 0..63 that some ray of the wave may touch
    if (BUNDLE && S.num_nodes <= 64u) {
        const uint32_t lane = lane_id();
        const uint64_t live = __builtin_amdgcn_ballot_w64(lane_on);
        const int first = live ? (int)__builtin_ctzll(live) : 0;
        const V3 oc = mk(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o.x), first)),
                         __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o.y), first)),
                         __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o.z), first)));
        const V3 dc = mk(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d.x), first)),
                         __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d.y), first)),
                         __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d.z), first)));
        float eo = lane_on ? fabsf(o.x - oc.x) + fabsf(o.y - oc.y) + fabsf(o.z - oc.z) : 0.0f;
        float ed = lane_on ? fabsf(d.x - dc.x) + fabsf(d.y - dc.y) + fabsf(d.z - dc.z) : 0.0f;
        const bool sane = o1 + (fabsf(d.x) + fabsf(d.y) + fabsf(d.z)) < 1e8f; // false on NaN
        const bool ok = !lane_on || (eo < 1e30f && ed < 1e30f && sane);
        for (int w = 32; w > 0; w >>= 1) { // wave maxima
            eo = fmaxf(eo, __shfl_xor(eo, w));
            ed = fmaxf(ed, __shfl_xor(ed, w));
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull && live != 0ull) {
            bool keep = lane < S.num_nodes;
            if (keep) keep = bundle_keep(S.cull[lane], oc, dc, eo, ed);
            bundle_mask = __builtin_amdgcn_ballot_w64(keep);
        }
    }
    for (uint32_t g = 0; g < S.num_nodes; g += 32u) {
        const uint32_t gend = (S.num_nodes - g < 32u) ? S.num_nodes : g + 32u;
        uint32_t pend_sphere = 0u, pend_small = 0u, pend_leaf = 0u, pend_tree = 0u;
        // phase 1: wave-uniform, conservative
        for (uint32_t n = g; n < gend; ++n) {
            if (BUNDLE && n < 64u && ((bundle_mask >> n) & 1ull) == 0ull) continue; // no ray of the wave can touch it
            const DNode* __restrict__ nd = S.nodes + n;
            const float m = nd->cull_k * o1 + nd->cull_c;
            bool missed, behind;
            if (BALLS) {
                // A hit lies within r + m of the centre c.  With v = c - o and b = v . dn (dn the unit direction), the line
                // passes the centre at squared distance q = |v|^2 - b^2 and comes nearest at parameter b: no point of the
                // ball is on the ray if q > (r + m)^2, none in front of the origin if b < -(r + m).  fp32 leaves q good to a
                // few 1e-7 |v|^2 (cancellation), dn to 1e-7: slacks of 1e-5 |v|^2 and 1e-4 (r + m) are far outside that.
                const float re = (nd->wmax[0] + m) * 1.0001f;
                const V3 v = mk(nd->wmin[0] - o.x, nd->wmin[1] - o.y, nd->wmin[2] - o.z);
                const float b = __builtin_fmaf(v.x, dn.x, __builtin_fmaf(v.y, dn.y, v.z * dn.z));
                const float vv = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z));
                const float q = __builtin_fmaf(-b, b, vv);
                missed = q > __builtin_fmaf(1e-5f, vv, re * re); // false on NaN = keep the node
                behind = b < -re;
            } else {
                const float ax = ((nd->wmin[0] - m) - o.x) * inv.x, bx = ((nd->wmax[0] + m) - o.x) * inv.x;
                const float ay = ((nd->wmin[1] - m) - o.y) * inv.y, by = ((nd->wmax[1] + m) - o.y) * inv.y;
                const float az = ((nd->wmin[2] - m) - o.z) * inv.z, bz = ((nd->wmax[2] + m) - o.z) * inv.z;
                const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
                const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
                // false on NaN = keep the node
                missed = (t_in - t_out) > 1e-5f * (fabsf(t_in) + fabsf(t_out));
                behind = t_out < 0.0f;
            }
            const uint32_t bit = (lane_on && !(missed || behind)) ? (1u << (n - g)) : 0u;
            if (BALLS) { // every node is a sphere: no routing by class (as compiled, eight selects per node)
                pend_sphere |= bit;
                continue;
            }
            const uint32_t cls = nd->cls; // wave-uniform
            if (cls == kClassSphere) pend_sphere |= bit;
            else if (cls == kClassSmallLeaf) pend_small |= bit;
            else if (cls == kClassLeaf) pend_leaf |= bit;
            else pend_tree |= bit;
        }
        // phase 2a: spheres
        while (__builtin_amdgcn_ballot_w64(pend_sphere != 0u) != 0ull) {
            const bool work = pend_sphere != 0u;
            const uint32_t n = g + (work ? (uint32_t)__builtin_ctz(pend_sphere) : 0u);
            pend_sphere &= pend_sphere - 1u;
            const float4* __restrict__ nv = reinterpret_cast<const float4*>(S.nodes + n);
            V3 lo, ld, p = mk(0.0f, 0.0f, 0.0f);
            lane_ray_to_local(nv, o, d, lo, ld);
            const bool sphere_hit = sphere_intersect_flat(lo, ld, p);
            if (work && sphere_hit) {
                V3 pw;
                float d0;
                lane_hit_to_world(nv, o, p, pw, d0);
                best_update(best, d0, (int)n, -1, p, pw);
            }
        }
        // phase 2b/2c: one-leaf meshes, small ones first (mesh::intersect on a root leaf:
        // box test, then the leaf's triangles, scene.cpp:190-203)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t& pend = pass == 0 ? pend_small : pend_leaf;
            while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
                const bool work = pend != 0u;
                const uint32_t n = g + (work ? (uint32_t)__builtin_ctz(pend) : 0u);
                pend &= pend - 1u;
                const float4* __restrict__ nv = reinterpret_cast<const float4*>(S.nodes + n);
                V3 lo, ld;
                lane_ray_to_local(nv, o, d, lo, ld);
                const DMesh mesh = S.meshes[work ? (S.nodes + n)->mesh : 0u];
                const DKdNode* __restrict__ kn = S.kd + mesh.kd_base;
                const bool inside = work && box_intersect(kn->cmin[0], kn->cmin[1], kn->cmin[2], kn->cmax[0],
                                                          kn->cmax[1], kn->cmax[2], lo, ld);
                float best_d = 3.402823466e+38f;
                int best_slot = -1;
                bool found = false;
                V3 p = mk(0.0f, 0.0f, 0.0f), a = mk(0.0f, 0.0f, 0.0f);
                const uint32_t count = inside ? (kn->b_axis & kKdCountMask) : 0u;
                leaf_triangles(S, mesh.leaf_base + kn->a, count, lo, ld, best_d, best_slot, found, a, p);
                if (found) {
                    V3 pw;
                    float d0;
                    lane_hit_to_world(nv, o, p, pw, d0);
                    best_update(best, d0, (int)n, best_slot, a, pw);
                }
            }
        }
        // phase 2d: meshes with a real KD-tree
        while (__builtin_amdgcn_ballot_w64(pend_tree != 0u) != 0ull) {
            const bool work = pend_tree != 0u;
            const uint32_t n = g + (work ? (uint32_t)__builtin_ctz(pend_tree) : 0u);
            pend_tree &= pend_tree - 1u;
            const float4* __restrict__ nv = reinterpret_cast<const float4*>(S.nodes + n);
            V3 lo, ld;
            lane_ray_to_local(nv, o, d, lo, ld);
            const uint32_t mesh_idx = work ? (S.nodes + n)->mesh : 0u;
            const DMesh mesh = S.meshes[mesh_idx];
            int prim = -1;
            V3 p = mk(0.0f, 0.0f, 0.0f), a = mk(0.0f, 0.0f, 0.0f);
            if (mesh_traverse_lane<PRUNE>(S, mesh, lo, ld, work, false, prim, a, p)) {
                V3 pw;
                float d0;
                lane_hit_to_world(nv, o, p, pw, d0);
                best_update(best, d0, (int)n, prim, a, pw);
            }
        }
    }
}

// ---- traversal mode 1a: per-lane node lists -------------------------------------
// For scenes with few nodes.  Phase 1 is the reference's node loop run by the whole wave
// (node matrices come in through scalar loads): ray -> local, spheres intersected in place,
// meshes only root-box tested (the reference's own test, scene.cpp:190); a per-lane bit mask
// records the meshes whose root box the lane's ray pierced.  Phase 2: every lane takes ITS
// next pierced mesh, rebuilds the same local ray (same operations, same bits) and walks it.
template <bool PRUNE>
__device__ __forceinline__ void intersect_scene_nodelist(const DScene& S, V3 o, V3 d, bool lane_on, Hit& best)
{
    best.d0 = 3.402823466e+38f;
    best.node = -1;
    best.prim = -1;
    best.a0 = best.a1 = best.a2 = 0.0f;
    best.pw = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t g = 0; g < S.num_nodes; g += 32u) {
        const uint32_t gend = (S.num_nodes - g < 32u) ? S.num_nodes : g + 32u;
        uint32_t pending = 0u;
        // phase 1: wave-uniform
        for (uint32_t n = g; n < gend; ++n) {
            const DNode* __restrict__ nd = S.nodes + n;
            const V3 lo = mat_point(nd->inv, o, 1.0f);
            const V3 ld = normalize(mat_point(nd->inv, d, 0.0f));
            if (nd->kind == PTK_SHAPE_SPHERE) {
                V3 p = mk(0.0f, 0.0f, 0.0f);
                if (lane_on && sphere_intersect(lo, ld, p)) {
                    const V3 pw = mat_point(nd->xf, p, 1.0f);
                    const V3 dv = pw - o;
                    best_update(best, dot(dv, dv), (int)n, -1, p, pw);
                }
            } else {
                const DMesh mesh = S.meshes[nd->mesh];
                const DKdNode* __restrict__ kn = S.kd + mesh.kd_base;
                const bool inside = lane_on && box_intersect(kn->cmin[0], kn->cmin[1], kn->cmin[2], kn->cmax[0],
                                                             kn->cmax[1], kn->cmax[2], lo, ld);
                pending |= inside ? (1u << (n - g)) : 0u;
            }
        }
        // phase 2: per-lane node list
        while (__builtin_amdgcn_ballot_w64(pending != 0u) != 0ull) {
            const bool work = pending != 0u;
            const uint32_t n = g + (work ? (uint32_t)__builtin_ctz(pending) : 0u);
            pending &= pending - 1u;
            const float4* __restrict__ nv = reinterpret_cast<const float4*>(S.nodes + n);
            const float4 i0 = nv[0], i1 = nv[1], i2 = nv[2], i3 = nv[3];
            const V3 lo = mk((i0.x * o.x + i1.x * o.y) + (i2.x * o.z + i3.x * 1.0f),
                             (i0.y * o.x + i1.y * o.y) + (i2.y * o.z + i3.y * 1.0f),
                             (i0.z * o.x + i1.z * o.y) + (i2.z * o.z + i3.z * 1.0f));
            const V3 ld = normalize(mk((i0.x * d.x + i1.x * d.y) + (i2.x * d.z + i3.x * 0.0f),
                                       (i0.y * d.x + i1.y * d.y) + (i2.y * d.z + i3.y * 0.0f),
                                       (i0.z * d.x + i1.z * d.y) + (i2.z * d.z + i3.z * 0.0f)));
            const uint32_t mesh_idx = work ? (S.nodes + n)->mesh : 0u;
            const DMesh mesh = S.meshes[mesh_idx];
            int prim = -1;
            V3 p = mk(0.0f, 0.0f, 0.0f), a = mk(0.0f, 0.0f, 0.0f);
            if (mesh_traverse_lane<PRUNE>(S, mesh, lo, ld, work, true, prim, a, p)) {
                const float4 x0 = nv[4], x1 = nv[5], x2 = nv[6], x3 = nv[7];
                const V3 pw = mk((x0.x * p.x + x1.x * p.y) + (x2.x * p.z + x3.x * 1.0f),
                                 (x0.y * p.x + x1.y * p.y) + (x2.y * p.z + x3.y * 1.0f),
                                 (x0.z * p.x + x1.z * p.y) + (x2.z * p.z + x3.z * 1.0f));
                const V3 dv = pw - o;
                best_update(best, dot(dv, dv), (int)n, prim, a, pw);
            }
        }
    }
}

// ---- traversal mode 2: wave-balanced work lists ----------------------------------
// Per-lane lists still leave most of a wave idle: the wave runs as many candidate steps as
// its busiest lane needs, and a lane that tests the twelve triangles of a cube keeps the
// other 63 waiting.  Here the work of a wave's 64 rays is pooled in LDS and dealt out evenly,
// as a three-stage pipeline inside the wave:
//   1. cull loop (wave-uniform over the compact DCull table): every (ray, node) pair that
//      survives the conservative tests is appended to a ring with ballot + mbcnt, in node order;
//   2. whenever 64 pairs are queued (or the nodes are exhausted), lane i takes pair i: it
//      fetches that ray from its owner lane (ds_bpermute) and runs the reference's arithmetic
//      for the node: ray -> local, then sphere / KD walk (committed at once) or the root box
//      test of a one-leaf mesh, whose survivors go to a second ring with their local ray;
//   3. whenever 64 such candidates are stored, their triangles are dealt to the lanes two at
//      a time (packed fp32: the two tests run in the two halves of v_pk_* operations, which
//      are element-wise IEEE, so each half replays the scalar expression tree), 64 items
//      at a time whatever the mix of leaf sizes, and only in whole chunks until the ray
//      batch runs dry; then lane i finishes candidate i (hit -> world).
// Results are combined with 64-bit LDS atomic minima on the keys the reference's strict `<`
// loops imply: (squared local distance, leaf slot) inside a mesh (scene.cpp:196-202), (squared
// world distance, node index) across nodes (scene.cpp:383-391).  Distances are non-negative
// floats, so their bit patterns order like the values; a key's low word makes it unique, so
// exactly one lane finds its own key after the minimum and writes the payload.  Any dealing
// order gives the same winner.  All cross-lane steps sit in wave-uniform control flow
// (conditions come from ballots).
constexpr unsigned long long kKeyNone = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kLeafItemMax = 4096u; // larger one-leaf meshes are walked per lane (class kClassTree)

struct alignas(16) WaveLds {
    unsigned long long ray_key[64];  // best (d0 bits, node) per ray
    unsigned long long cand_key[64]; // best (dd bits, leaf slot) per candidate of the window being finished
    float ray_pay[7][64];            // prim, a0..a2, pw of the ray's best hit
    float cand_pay[6][64];           // bar, p of the candidate's best triangle
    uint32_t queue[128];             // ring of (node << 6 | owner lane)
    float st_ray[6][128];            // ring of box-pierced one-leaf candidates: local ray,
    uint32_t st_first[128];          //   first leaf slot,
    uint32_t st_count[128];          //   triangle count,
    uint32_t st_item[128];           //   (node << 6 | owner lane)
    uint32_t owner_at[64];           // triangle stage: item i of the chunk being filled (pair record, second-triangle flag, window position)
};
static_assert(sizeof(WaveLds) <= 10224, "k_bounce: four workgroups of four waves per CU (160 KB of LDS)");

// value of `v` in lane `src` (every lane must be active)
__device__ __forceinline__ float lane_get(float v, uint32_t src)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src << 2), __float_as_int(v)));
}
__device__ __forceinline__ V3 lane_get(V3 v, uint32_t src)
{
    return mk(lane_get(v.x, src), lane_get(v.y, src), lane_get(v.z, src));
}
// orders this wave's LDS traffic around a cross-lane hand-over (LDS is in-order per wave:
// this only stops the compiler from moving accesses across it)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t lane_get_u32(uint32_t v, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}

// inclusive prefix sum over the 64 lanes (every lane active): four row_shr steps inside each row of 16, then the
// row totals through row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3); lanes without a source add 0
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (uint32_t)v;
}
// Two triangles per lane: tri_eval on both halves of packed fp32 operations.
typedef float f2 __attribute__((ext_vector_type(2)));
struct V3p {
    f2 x, y, z;
};
__device__ __forceinline__ V3p splat(V3 a) { return V3p{ f2{ a.x, a.x }, f2{ a.y, a.y }, f2{ a.z, a.z } }; }
__device__ __forceinline__ V3p operator+(V3p a, V3p b) { return V3p{ a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ V3p operator-(V3p a, V3p b) { return V3p{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ V3p operator*(V3p a, f2 s) { return V3p{ a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ f2 dot(V3p a, V3p b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// tri_eval (scene.cpp:142-186) for the triangles A and B of one pair record (DScene::tri_pair)
// against one ray.  r[m] = { A[2m], B[2m], A[2m+1], B[2m+1] } over the 16 words of a hit record
// (v0 N e0 e1 d00 d01 d11 inv_den).  Element .x of every packed value is triangle A's, .y
// triangle B's; the operations and their order are those of tri_eval.  Outputs are meaningful
// for the triangles whose hit flag is set.
__device__ __forceinline__ void tri_eval2(const float4 r0, const float4 r1, const float4 r2, const float4 r3,
                                          const float4 r4, const float4 r5, const float4 r6, const float4 r7, V3 o, V3 d,
                                          bool on_a, bool on_b, bool& hit_a, bool& hit_b, f2& dd_out, V3p& bar_out,
                                          V3p& p_out)
{
    // no early exits: after the cull almost every wave has a lane that needs the whole test
    const V3p v0 = { f2{ r0.x, r0.y }, f2{ r0.z, r0.w }, f2{ r1.x, r1.y } };
    const V3p N = { f2{ r1.z, r1.w }, f2{ r2.x, r2.y }, f2{ r2.z, r2.w } };
    const V3p O = splat(o), D = splat(d);
    const V3p l = O - v0;
    const f2 distance = dot(l, N);
    on_a = on_a && !(distance.x < 0);
    on_b = on_b && !(distance.y < 0);
    const f2 cosine = dot(D, N);
    on_a = on_a && !(cosine.x >= 0);
    on_b = on_b && !(cosine.y >= 0);
    const f2 tq = f2{ distance.x / -cosine.x, distance.y / -cosine.y };
    const V3p projection = O + D * tq;
    const V3p e0 = { f2{ r3.x, r3.y }, f2{ r3.z, r3.w }, f2{ r4.x, r4.y } };
    const V3p e1 = { f2{ r4.z, r4.w }, f2{ r5.x, r5.y }, f2{ r5.z, r5.w } };
    const V3p v2 = projection - v0;
    const f2 d20 = dot(v2, e0);
    const f2 d21 = dot(v2, e1);
    const f2 q_d00 = f2{ r6.x, r6.y }, q_d01 = f2{ r6.z, r6.w }, q_d11 = f2{ r7.x, r7.y }, q_inv = f2{ r7.z, r7.w };
    const f2 v = (q_d11 * d20 - q_d01 * d21) * q_inv;
    const f2 w = (q_d00 * d21 - q_d01 * d20) * q_inv;
    const f2 u = f2{ 1.0f, 1.0f } - v - w;
    hit_a = on_a && u.x >= 0 && v.x >= 0 && w.x >= 0;
    hit_b = on_b && u.y >= 0 && v.y >= 0 && w.y >= 0;
    const V3p dv = O - projection;
    dd_out = dot(dv, dv);
    bar_out = V3p{ u, v, w };
    p_out = projection;
}

// SPLIT: KD-tree nodes are not walked here; the ones that survive the cull are returned as a bit
// mask over DScene::tree_nodes (at most 64 of them) for the walker kernel (k_walk in ptk.hip).
// BUNDLE: the wave's rays are expected to be nearly the same ray (primary rays: a wave holds 64 samples
// of one pixel), so the cull loop only visits the nodes that the bundle as a whole can touch (below).
// NO_TREES: the scene has no node of class kClassTree (spheres and one-leaf meshes only), so the KD walk drops out of stage 2.
template <bool PRUNE, bool SPLIT, bool BUNDLE = false, bool NO_TREES = false>
__device__ __forceinline__ void intersect_scene_balanced(const DScene& S, V3 o, V3 d, bool lane_on, Hit& best,
                                                         WaveLds& L, unsigned long long& tree_mask)
{
    tree_mask = 0ull;
    const uint32_t lane = lane_id();
    best.d0 = 3.402823466e+38f;
    best.node = -1;
    best.prim = -1;
    best.a0 = best.a1 = best.a2 = 0.0f;
    best.pw = mk(0.0f, 0.0f, 0.0f);
    L.ray_key[lane] = kKeyNone;
    uint32_t qhead = 0u, qtail = 0u; // (ray, node) ring, wave-uniform
    uint32_t shead = 0u, stail = 0u; // candidate ring, wave-uniform
    [[maybe_unused]] const unsigned long long ic0 = PTK_CLOCK();

    // scene.cpp:383-391 across nodes: keep the smallest (d0, node)
    auto commit = [&](bool hit, uint32_t owner, float d0, uint32_t n, int prim, V3 a, V3 pw) {
        const bool ok = hit && d0 < 3.402823466e+38f; // scene.cpp:386 against the initial best
        const unsigned long long key = ((unsigned long long)__float_as_uint(d0) << 32) | n;
        if (ok) atomicMin(&L.ray_key[owner], key);
        wave_sync();
        if (ok && L.ray_key[owner] == key) {
            L.ray_pay[0][owner] = __int_as_float(prim);
            L.ray_pay[1][owner] = a.x;  L.ray_pay[2][owner] = a.y;  L.ray_pay[3][owner] = a.z;
            L.ray_pay[4][owner] = pw.x; L.ray_pay[5][owner] = pw.y; L.ray_pay[6][owner] = pw.z;
        }
    };

    // stage 3: triangles of the first `cnt` stored candidates, then their hit -> world.
    // The (candidate, pair) items of the window are dealt 64 at a time in level order: first pair 0 of every
    // candidate, then pair 1 of those that have one, and so on, whatever mix of leaf sizes the window holds.  Per
    // level the candidates that still have a pair push (level, window position) into the next free slots of a
    // 64-entry table (ballot + mbcnt); when the table is full, or nothing is left, lane i tests item i.  A chunk
    // therefore holds a candidate twice only when fewer than 64 items are left on a level, which keeps the LDS
    // minima below free of same-address collisions in the common case (and lets coherent waves read one triangle
    // pair per chunk).
    // Unless `all` is set only whole chunks are dealt: the window is cut back to the candidates that fill them (a
    // prefix, by the running sum of the pair counts), the others stay in the ring until more have arrived, so that
    // idle lanes appear only in a ray batch's last chunk.
    auto flush = [&](uint32_t cnt, bool all) {
        [[maybe_unused]] const unsigned long long tc0 = PTK_CLOCK();
        PTK_STAT(S, 3, 1);
        wave_sync();
        const uint32_t s = (shead + lane) & 127u;
        const uint32_t my_count = lane < cnt ? L.st_count[s] : 0u; // never 0 inside the window: empty leaves are not stored
        const uint32_t my_pairs = (my_count + 1u) >> 1;
        const uint32_t my_first = L.st_first[s] >> 1;              // first pair record (first slot is even, ptk_scene_upload)
        if (!all) {
            const uint32_t incl = wave_inclusive_sum(my_pairs);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total > 64u) {
                const uint32_t whole = total & ~63u;
                const uint32_t fit = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(lane < cnt && incl <= whole));
                cnt = fit != 0u ? fit : 1u; // a first candidate larger than the whole chunks goes alone
            }
        }
        const bool have = lane < cnt;
        const uint32_t pairs = have ? my_pairs : 0u;
        L.cand_key[lane] = kKeyNone;
        // the candidate's node -> world matrix and its ray's origin are requested now and used after the triangle
        // chunks (hit -> world of the best triangle): the memory round trip runs under them
        const uint32_t item = have ? L.st_item[s] : 0u;
        const uint32_t owner = item & 63u, n_item = item >> 6;
        const float4* __restrict__ nv_item = reinterpret_cast<const float4*>(S.nodes + n_item);
        const float4 wx0 = nv_item[4], wx1 = nv_item[5], wx2 = nv_item[6], wx3 = nv_item[7];
        uint32_t level = 0u, consumed = 0u, fill = 0u; // wave-uniform: current level, its candidates already dealt, slots taken
        uint64_t mask = __builtin_amdgcn_ballot_w64(pairs > 0u);
        while (mask != 0ull) {
            const uint32_t P = (uint32_t)__popcll(mask), rank = mbcnt64(mask);
            const uint32_t room = 64u - fill;
            const uint32_t take = P - consumed < room ? P - consumed : room;
            wave_sync();
            // item = (pair record << 7 | pair has a second triangle << 6 | window position): everything the testing lane
            // needs to request the record at once (pair records of one-leaf meshes < 2^25: ptk_scene_upload)
            if (pairs > level && rank - consumed < take)
                L.owner_at[fill + rank - consumed] = ((my_first + level) << 7) | (2u * level + 1u < my_count ? 64u : 0u) | lane;
            fill += take;
            consumed += take;
            if (consumed == P) {
                ++level;
                consumed = 0u;
                mask = __builtin_amdgcn_ballot_w64(pairs > level);
            }
            if (fill < 64u && mask != 0ull) continue;
            wave_sync();
            PTK_STAT(S, 4, 1);
            PTK_STAT(S, 7, fill);
            {
                const bool valid = lane < fill;
                const uint32_t it = valid ? L.owner_at[lane] : 0u;
                const uint32_t c = it & 63u, pair = it >> 7;
                const float4* tp = S.tri_pair + (size_t)pair * 8;
                const float4 r0 = tp[0], r1 = tp[1], r2 = tp[2], r3 = tp[3], r4 = tp[4], r5 = tp[5], r6 = tp[6], r7 = tp[7];
                const uint32_t cs = (shead + c) & 127u;
                const V3 co = mk(L.st_ray[0][cs], L.st_ray[1][cs], L.st_ray[2][cs]);
                const V3 cd = mk(L.st_ray[3][cs], L.st_ray[4][cs], L.st_ray[5][cs]);
                const bool on_b = valid && (it & 64u) != 0u;
                const uint32_t slot_a = 2u * pair, slot_b = slot_a + 1u;
                bool hit_a, hit_b;
                f2 dd2;
                V3p bar2, p2;
                tri_eval2(r0, r1, r2, r3, r4, r5, r6, r7, co, cd, valid, on_b, hit_a, hit_b, dd2, bar2, p2);
                hit_a = hit_a && dd2.x < 3.402823466e+38f; // scene.cpp:198 against the initial best
                hit_b = hit_b && dd2.y < 3.402823466e+38f;
                const bool take_b = hit_b && (!hit_a || dd2.y < dd2.x); // ties: the lower slot
                const bool hit = hit_a || hit_b;
                const float dd = take_b ? dd2.y : dd2.x;
                const uint32_t slot = take_b ? slot_b : slot_a;
                const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | slot;
                if (hit) atomicMin(&L.cand_key[c], key);
                wave_sync();
                if (hit && L.cand_key[c] == key) {
                    L.cand_pay[0][c] = take_b ? bar2.x.y : bar2.x.x;
                    L.cand_pay[1][c] = take_b ? bar2.y.y : bar2.y.x;
                    L.cand_pay[2][c] = take_b ? bar2.z.y : bar2.z.x;
                    L.cand_pay[3][c] = take_b ? p2.x.y : p2.x.x;
                    L.cand_pay[4][c] = take_b ? p2.y.y : p2.y.x;
                    L.cand_pay[5][c] = take_b ? p2.z.y : p2.z.x;
                }
            }
            fill = 0u;
        }
        wave_sync();
        const unsigned long long key = L.cand_key[lane];
        const bool found = have && key != kKeyNone;
        if (__builtin_amdgcn_ballot_w64(found) != 0ull) {
            const V3 ro = lane_get(o, owner);
            const V3 bar = mk(L.cand_pay[0][lane], L.cand_pay[1][lane], L.cand_pay[2][lane]);
            const V3 p = mk(L.cand_pay[3][lane], L.cand_pay[4][lane], L.cand_pay[5][lane]);
            // hit -> world (scene.cpp:375) + distance (scene.cpp:383), as lane_hit_to_world
            const V3 pw = mk((wx0.x * p.x + wx1.x * p.y) + (wx2.x * p.z + wx3.x * 1.0f),
                             (wx0.y * p.x + wx1.y * p.y) + (wx2.y * p.z + wx3.y * 1.0f),
                             (wx0.z * p.x + wx1.z * p.y) + (wx2.z * p.z + wx3.z * 1.0f));
            const V3 dv = pw - ro;
            const float d0 = dot(dv, dv);
            commit(found, owner, d0, n_item, (int)(uint32_t)key, bar, pw);
        }
        shead += cnt;
        PTK_STAT(S, 10, PTK_CLOCK() - tc0);
    };

    // stage 2: lane i < cnt takes the i-th queued (ray, node) pair
    auto process = [&](uint32_t cnt) {
        [[maybe_unused]] const unsigned long long pc0 = PTK_CLOCK();
        PTK_STAT(S, 2, 1);
        wave_sync();
        const bool have = lane < cnt;
        const uint32_t item = have ? L.queue[(qhead + lane) & 127u] : 0u;
        const uint32_t owner = item & 63u, n = item >> 6;
        const V3 ro = lane_get(o, owner), rd = lane_get(d, owner);
        const DNode* __restrict__ nd = S.nodes + n;
        const float4* __restrict__ nv = reinterpret_cast<const float4*>(nd);
        // the root box of a one-leaf mesh is requested together with the matrix (same dependence: n), not when the
        // leaf branch below needs it: one memory latency per batch instead of two
        const float4* __restrict__ lf = reinterpret_cast<const float4*>(S.leaf_info + n);
        const float4 l0 = lf[0], l1 = lf[1];
        V3 lo, ld;
        lane_ray_to_local(nv, ro, rd, lo, ld); // operator*(mat4, ray), scene.cpp:18-23
        const uint32_t cls = nd->cls;
        bool hit = false;
        int prim = -1;
        V3 p = mk(0.0f, 0.0f, 0.0f), a = mk(0.0f, 0.0f, 0.0f);
        const bool is_sphere = have && cls == kClassSphere;
        if (__builtin_amdgcn_ballot_w64(is_sphere) != 0ull) {
            V3 sp = mk(0.0f, 0.0f, 0.0f);
            const bool sphere_hit = sphere_intersect_flat(lo, ld, sp);
            if (is_sphere && sphere_hit) {
                hit = true;
                p = sp;
                a = sp;
            }
        }
        const bool is_tree = !SPLIT && !NO_TREES && have && cls == kClassTree;
        if (!SPLIT && !NO_TREES && __builtin_amdgcn_ballot_w64(is_tree) != 0ull) {
            const DMesh mesh = S.meshes[is_tree ? nd->mesh : 0u];
            int tprim = -1;
            V3 tb = mk(0.0f, 0.0f, 0.0f), tp = mk(0.0f, 0.0f, 0.0f);
            if (mesh_traverse_lane<PRUNE>(S, mesh, lo, ld, is_tree, false, tprim, tb, tp)) {
                hit = true;
                prim = tprim;
                a = tb;
                p = tp;
            }
        }
        if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
            V3 pw;
            float d0;
            lane_hit_to_world(nv, ro, p, pw, d0); // scene.cpp:375,383
            commit(hit, owner, d0, n, prim, a, pw);
        }
        const bool is_leaf = have && (cls == kClassSmallLeaf || cls == kClassLeaf);
        if (__builtin_amdgcn_ballot_w64(is_leaf) != 0ull) {
            // mesh::intersect on a root leaf: box test (scene.cpp:190); the triangles wait in the ring
            uint32_t count = 0u, first = 0u;
            bool pass = false, undecided = false;
            if (is_leaf) {
                bool certain;
                pass = box_intersect_fast(l0.x, l0.y, l0.z, l1.x, l1.y, l1.z, lo, ld, certain);
                undecided = !certain;
                count = __float_as_uint(l1.w);
                first = __float_as_uint(l0.w);
            }
            if (__builtin_amdgcn_ballot_w64(undecided) != 0ull) { // rare: a lane within rounding of a box edge
                if (undecided) pass = box_intersect(l0.x, l0.y, l0.z, l1.x, l1.y, l1.z, lo, ld);
            }
            if (!pass) count = 0u;
            const uint64_t pm = __builtin_amdgcn_ballot_w64(count != 0u);
            if (count != 0u) {
                const uint32_t sl = (stail + mbcnt64(pm)) & 127u;
                L.st_ray[0][sl] = lo.x; L.st_ray[1][sl] = lo.y; L.st_ray[2][sl] = lo.z;
                L.st_ray[3][sl] = ld.x; L.st_ray[4][sl] = ld.y; L.st_ray[5][sl] = ld.z;
                L.st_first[sl] = first;
                L.st_count[sl] = count;
                L.st_item[sl] = item;
            }
            stail += (uint32_t)__popcll(pm);
            PTK_STAT(S, 6, __popcll(pm));
        }
        qhead += cnt;
        PTK_STAT(S, 9, PTK_CLOCK() - pc0);
    };

    // stage 1: conservative cull.  A node is dropped only if the reference cannot report a hit
    // on it: the ray misses the node's padded world box (see intersect_scene_lanelist), or the
    // node is a planar one-sided mesh that the ray direction faces from behind (every triangle
    // then fails `cosine >= 0`, scene.cpp:158; DCull::fn / ftau, make_cull_record in ptk.hip).
    // The slab quotients are (b -+ m - o) / d = (b -+ m) * inv - o * inv: one packed add and one packed
    // fma per axis.  (Explicit fma: this test is ours, not the reference's; its rounding is far inside
    // the margins.)  The reciprocals are clamped to +-1e30 so that a ray parallel to an axis gives huge
    // finite quotients of the right sign instead of inf - inf; rays with non-finite or absurdly large
    // coordinates are not culled at all (`sane`).
    const V3 inv = mk(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e30f, 1e30f),
                      __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e30f, 1e30f),
                      __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e30f, 1e30f));
    const V3 noi = mk(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    const float o1 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
    const bool sane = o1 + (fabsf(d.x) + fabsf(d.y) + fabsf(d.z)) < 1e8f; // false on NaN
    // (a ray that starts further out than the margins in the table allow for, e.g. at a camera far from the scene, is not
    // culled per ray either; the bundle test below, which builds its own margin, still applies to it)
    const float lane_bias = !lane_on ? -__builtin_inff() : (sane && o1 <= S.cull_o1max) ? 0.0f : __builtin_inff();
    // Bundle pre-cull (BUNDLE): one test per node for the whole wave, 64 nodes at a time, lane = node.
    // Take the first live lane's ray (oc, dc) as the central ray and let eo = max |o_i - oc|_1,
    // ed = max |d_i - dc|_1 over the wave.  A ray i that reaches a point p of a node's padded world box
    // does so at t_i <= |oc - c|_1 + eo + R (c the box centre, R its L1 half-extent), and the central
    // ray's point at the same parameter is within eo + t_i ed of p.  So if ray i can touch the box, the
    // central ray touches the box inflated by eo + T ed, T = |oc - c|_1 + eo + R: nodes failing that
    // test cannot be hit by any ray of the wave and are skipped; the others still get the per-ray test.
    // Incoherent waves just inflate every box until all nodes pass.
    uint64_t bundle_mask = ~0ull; // candidate nodes among [bundle_base, bundle_base + 64)
    uint32_t bundle_base = 0u;
    V3 oc = o, dc = d;
    float eo = 0.0f, ed = 0.0f;
    bool bundle_sane = false;
    if (BUNDLE) {
        const uint64_t live = __builtin_amdgcn_ballot_w64(lane_on);
        const int first = live ? (int)__builtin_ctzll(live) : 0;
        oc = mk(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o.x), first)),
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o.y), first)),
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o.z), first)));
        dc = mk(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d.x), first)),
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d.y), first)),
                __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d.z), first)));
        const float my_eo = lane_on ? fabsf(o.x - oc.x) + fabsf(o.y - oc.y) + fabsf(o.z - oc.z) : 0.0f;
        const float my_ed = lane_on ? fabsf(d.x - dc.x) + fabsf(d.y - dc.y) + fabsf(d.z - dc.z) : 0.0f;
        // wave maxima through LDS (non-negative floats order like their bit patterns); NaN -> no pre-cull
        const bool finite = !lane_on || (my_eo < 1e30f && my_ed < 1e30f);
        uint32_t* const wave_max = L.owner_at; // free until the first triangle stage
        wave_max[0] = 0u;
        wave_max[1] = 0u;
        wave_sync();
        atomicMax(&wave_max[0], __float_as_uint(my_eo));
        atomicMax(&wave_max[1], __float_as_uint(my_ed));
        wave_sync();
        eo = __uint_as_float(wave_max[0]);
        ed = __uint_as_float(wave_max[1]);
        bundle_sane = __builtin_amdgcn_ballot_w64(!(finite && sane)) == 0ull && live != 0ull;
        wave_sync();
    }
    auto bundle_chunk = [&](uint32_t base) {
        // lane = node base + lane: central ray against the inflated box, same slab formulation as below
        const uint32_t nn = base + lane;
        bool keep = nn < S.num_nodes;
        if (bundle_sane && keep) keep = bundle_keep(S.cull[nn], oc, dc, eo, ed);
        bundle_base = base;
        bundle_mask = __builtin_amdgcn_ballot_w64(keep);
    };
    uint32_t n = 0u;
    if (BUNDLE) {
        bundle_chunk(0u);
        n = bundle_mask ? (uint32_t)__builtin_ctzll(bundle_mask) : 64u;
    }
    // BUNDLE: the next candidate node after n (walks the 64-node chunks)
    auto advance = [&]() {
        if (!BUNDLE) {
            ++n;
            return;
        }
        for (;;) {
            const uint32_t bit = n - bundle_base + 1u; // bits above the current node
            const uint64_t rest = bit < 64u ? (bundle_mask >> bit) << bit : 0ull;
            if (rest != 0ull) {
                n = bundle_base + (uint32_t)__builtin_ctzll(rest);
                return;
            }
            if (bundle_base + 64u >= S.num_nodes) {
                n = S.num_nodes;
                return;
            }
            bundle_chunk(bundle_base + 64u);
            if (bundle_mask & 1ull) {
                n = bundle_base;
                return;
            }
            n = bundle_base; // bit 0 is clear: look above it
        }
    };
    if (BUNDLE && n >= 64u) { // no candidate in the first chunk
        n = 0u;
        bundle_mask &= ~1ull; // (it is 0 already) make advance() start above bit 0
        advance();
    }
    for (;;) {
        while (n < S.num_nodes && qtail - qhead < 64u) {
            // constant address space: the record comes in through scalar loads even though the
            // kernel writes other global memory (ray state) in between
            const __attribute__((address_space(4))) DCull* cu =
                (const __attribute__((address_space(4))) DCull*)(S.cull + n);
            const f2 tx = __builtin_elementwise_fma(f2{ cu->bx[0], cu->bx[1] }, f2{ inv.x, inv.x }, f2{ noi.x, noi.x });
            const f2 ty = __builtin_elementwise_fma(f2{ cu->by[0], cu->by[1] }, f2{ inv.y, inv.y }, f2{ noi.y, noi.y });
            const f2 tz = __builtin_elementwise_fma(f2{ cu->bz[0], cu->bz[1] }, f2{ inv.z, inv.z }, f2{ noi.z, noi.z });
            const float t_in = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
            const float t_out = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
            // the three reasons to drop a node as slacks that are negative when they apply: missed
            // (t_in - t_out > 1e-5 (|t_in| + |t_out|)), behind (t_out < 0), facing away (d . fn > ftau); their minimum,
            // plus the lane's bias (0; +inf for a ray that is never culled; -inf for a lane without a ray), decides in
            // one compare, which the ballot below reads directly.  NaN compares as "keep".
            const float ok_hit = __builtin_fmaf(1e-5f, fabsf(t_in) + fabsf(t_out), t_out - t_in);
            const float ok_face = cu->ftau - __builtin_fmaf(d.x, cu->fn[0], __builtin_fmaf(d.y, cu->fn[1], d.z * cu->fn[2]));
            const bool keep = !(fminf(fminf(ok_hit, t_out), ok_face) + lane_bias < 0.0f);
            if (SPLIT && cu->cls == kClassTree) { // wave-uniform branch
                tree_mask |= keep ? (1ull << (cu->tree_slot & 63u)) : 0ull;
                advance();
                continue;
            }
            const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
            if (keep) L.queue[(qtail + mbcnt64(km)) & 127u] = (n << 6) | lane;
            qtail += (uint32_t)__popcll(km);
            advance();
        }
        const uint32_t qn = qtail - qhead;
#ifdef PTK_EXP
        if (S.exp_stage == 1u) qhead = qtail;
        else
#endif
        if (qn != 0u) process(qn < 64u ? qn : 64u);
        const bool drained = n >= S.num_nodes && qtail == qhead;
        const uint32_t sn = stail - shead;
#ifdef PTK_EXP
        if (S.exp_stage == 1u || S.exp_stage == 2u) shead = stail;
        else
#endif
        {
            for (uint32_t w = sn; w >= 64u; w = stail - shead) flush(64u, false); // leaves fewer than 64 behind
            if (drained && stail != shead) flush(stail - shead, true);
        }
        if (drained && stail == shead) break;
    }
    PTK_STAT(S, 0, 1);
    PTK_STAT(S, 1, qtail);
    PTK_STAT(S, 11, PTK_CLOCK() - ic0);

    wave_sync();
    const unsigned long long key = L.ray_key[lane];
    if (key != kKeyNone) {
        best.d0 = __uint_as_float((uint32_t)(key >> 32));
        best.node = (int)(uint32_t)key;
        best.prim = __float_as_int(L.ray_pay[0][lane]);
        best.a0 = L.ray_pay[1][lane];
        best.a1 = L.ray_pay[2][lane];
        best.a2 = L.ray_pay[3][lane];
        best.pw = mk(L.ray_pay[4][lane], L.ray_pay[5][lane], L.ray_pay[6][lane]);
    }
}

// ---- pooled leaf stage of the walker kernel --------------------------------------------------
// Leaves hold anything from one to dozens of triangles, so a per-lane triangle loop runs at the
// pace of the largest leaf in the wave.  The walker (k_walk in ptk.hip) lets lanes that reached a
// leaf wait until few lanes are still descending, then tests all waiting leaves together: the
// (lane, triangle) pairs are numbered through an exclusive prefix sum of the leaf sizes and dealt
// 64 at a time, item t going to lane t % 64.  The owner of an item is found without a search: every
// waiting lane sets bit (start % 64) in the 64-bit word (start / 64) of `start_mask`; the owner of
// item t is then the (number of starts at or before t)-th waiting lane.  Results are folded with
// the same (squared distance, slot) LDS minimum as in the balanced traversal, then compared with
// the lane's best so far exactly as tri_test does.  A lane contributes at most 64 triangles per
// stage, so 64 words always suffice; a larger leaf simply takes several stages (KdWalk::leaf_done).
struct alignas(16) WalkLds {
    unsigned long long key[64];        // best (dd bits, slot) per lane this stage
    unsigned long long start_mask[64]; // per 64-item chunk: bit i set = an owner's items start at item 64 * chunk + i
    float pay[6][64];                  // bar, p of the lane's best triangle this stage
    uint8_t owner[64];                 // k-th waiting lane
    uint32_t stack_i[kWalkStack][64];  // short stack of the lanes' pending far children (LdsStack) ...
    float stack_t[kWalkStack][64];     // ... and their distance bounds
};

// at_leaf: the lane stands on a leaf with triangles left to test.  Every lane of the wave calls this.
__device__ __forceinline__ void walk_leaf_stage(KdWalk& W, const DScene& S, bool at_leaf, WalkLds& L)
{
    const uint32_t lane = lane_id();
    const uint32_t left = at_leaf ? (W.cb & kKdCountMask) - W.leaf_done : 0u;
    const uint32_t cnt = left < 64u ? left : 64u;
    const uint32_t first = W.leaf_base + W.ca + W.leaf_done; // slot of the lane's first triangle this stage
    // exclusive prefix sum of cnt over the lanes
    const uint32_t incl = wave_inclusive_sum(cnt);
    const uint32_t start = incl - cnt;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint64_t wm = __builtin_amdgcn_ballot_w64(cnt != 0u);
    if (total == 0u) return;
    PTK_STAT_FINE(S, 28, total);
    L.key[lane] = kKeyNone;
    L.start_mask[lane] = 0ull;
    if (cnt != 0u) L.owner[mbcnt64(wm)] = (uint8_t)lane;
    wave_sync();
    if (cnt != 0u) atomicOr(&L.start_mask[start >> 6], 1ull << (start & 63u));
    wave_sync();
    uint32_t before = 0u; // owners whose items start before the current chunk (wave-uniform)
    for (uint32_t t0 = 0u; t0 < total; t0 += 64u) {
        const unsigned long long m = L.start_mask[t0 >> 6];
        const uint32_t t = t0 + lane;
        const bool valid = t < total;
        // starts at or before item t; at least one, because item 0 starts the first owner
        const uint32_t k = before + (uint32_t)__popcll(m & ((2ull << lane) - 1ull));
        const uint32_t c = L.owner[valid ? k - 1u : 0u];
        const uint32_t c_start = lane_get_u32(start, c), c_first = lane_get_u32(first, c);
        const V3 co = lane_get(W.o, c), cd = lane_get(W.d, c);
        const uint32_t slot = c_first + (valid ? t - c_start : 0u);
        const float4* tp = S.tri_hit + (size_t)slot * 4;
        const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
        float dd;
        V3 bar, p;
        PTK_STAT_FINE(S, 18, 1);
#ifdef PTK_STATS
        { // how many items the two side tests of scene.cpp:149,158 alone would reject
            const V3 v0s = mk(q0.x, q0.y, q0.z), Ns = mk(q0.w, q1.x, q1.y);
            const bool side_out = valid && (dot(co - v0s, Ns) < 0 || dot(cd, Ns) >= 0);
            const uint64_t so = __builtin_amdgcn_ballot_w64(side_out);
            PTK_STAT_FINE(S, 31, __popcll(so));
        }
#endif
        const bool tri_hit = tri_eval_flat(q0, q1, q2, q3, co, cd, dd, bar, p);
        const bool hit = valid && tri_hit && dd < 3.402823466e+38f;
        const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | slot;
        if (hit) atomicMin(&L.key[c], key);
        wave_sync();
        if (hit && L.key[c] == key) {
            L.pay[0][c] = bar.x; L.pay[1][c] = bar.y; L.pay[2][c] = bar.z;
            L.pay[3][c] = p.x;   L.pay[4][c] = p.y;   L.pay[5][c] = p.z;
        }
        before += (uint32_t)__popcll(m);
    }
    wave_sync();
    if (cnt != 0u) {
        const unsigned long long key = L.key[lane];
        if (key != kKeyNone) {
            // the bookkeeping of tri_test (scene.cpp:196-202) against the lane's best so far
            const float dd = __uint_as_float((uint32_t)(key >> 32));
            const int slot = (int)(uint32_t)key;
            if (dd < W.best_d || (dd == W.best_d && slot < W.best_slot)) {
                W.best_d = dd;
                W.best_slot = slot;
                W.found = true;
                W.bar = mk(L.pay[0][lane], L.pay[1][lane], L.pay[2][lane]);
                W.p = mk(L.pay[3][lane], L.pay[4][lane], L.pay[5][lane]);
            }
        }
        W.leaf_done += cnt;
    }
}

// MODE bit 4 (with bits 0 and 2): every node is a sphere with a bounding ball, culled against that (intersect_scene_lanelist).
// MODE bit 0: per-lane lists instead of the all-lanes-every-node loop (PTK_OPT_TRAVERSAL), bit 1: pruned KD
// walk (PTK_OPT_KD_PRUNE), bit 2 (with bit 0): candidate lists after the conservative cull (PTK_OPT_NODE_CULL),
// bit 3: wave-balanced work lists (PTK_OPT_BALANCE; needs the wave's WaveLds); bit 5 (with bit 3): the scene has no KD-tree
// node (k_bounce_fused for Room-like scenes).
template <int MODE, bool BUNDLE = false>
__device__ __forceinline__ void intersect_scene_mode(const DScene& S, V3 o, V3 d, bool lane_on, Hit& best, WaveLds* L)
{
    if (MODE & 8) {
        unsigned long long unused;
        intersect_scene_balanced<(MODE & 2) != 0, false, BUNDLE, (MODE & 32) != 0>(S, o, d, lane_on, best, *L, unused);
    }
    else if ((MODE & 5) == 5) intersect_scene_lanelist<(MODE & 2) != 0, (MODE & 16) != 0, BUNDLE && (MODE & 16) != 0>(S, o, d, lane_on, best);
    else if (MODE & 1) intersect_scene_nodelist<(MODE & 2) != 0>(S, o, d, lane_on, best);
    else intersect_scene<(MODE & 2) != 0>(S, o, d, lane_on, best);
}

// Rebuilds raycast_result::normal / uv for the winning hit only (the reference
// computes them for every candidate, scene.cpp:173-182,433-437; the values for
// the winner are identical).  n_world: scene.cpp:378.
// ALWAYS_UV = false: the sphere's lat-long uv (an atan2) is only computed for nodes whose material
// reads an image; a colour sampler ignores uv (sampler.h:58), so the shaded result is the same.
template <bool ALWAYS_UV>
__device__ __forceinline__ void hit_attributes(const DScene& S, const Hit& h, V3& n_world, float& uvx, float& uvy,
                                               int& prim_id, float& u_out, float& v_out)
{
    const DNode* __restrict__ nd = S.nodes + h.node;
    // the vector to normalise is picked per shape first, so a wave with both shapes normalises once
    V3 nl = mk(h.a0, h.a1, h.a2); // sphere: the local hit position, scene.cpp:433
    uvx = uvy = 0.0f;
    prim_id = -1;
    if (h.prim >= 0) {
        const uint32_t tri = ALWAYS_UV ? S.leaf_prim[h.prim] : 0u; // the triangle id is only reported by the parity entry
        const float4* s = S.tri_shade + (size_t)S.leaf_shade[h.prim] * 4;
        const float4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
        const V3 n0 = mk(s0.x, s0.y, s0.z), n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x);
        nl = n0 * h.a0 + n1 * h.a1 + n2 * h.a2;                           // scene.cpp:175-178
        uvx = h.a0 * s2.y + h.a1 * s2.w + h.a2 * s3.y;                    // scene.cpp:179-182
        uvy = h.a0 * s2.z + h.a1 * s3.x + h.a2 * s3.z;
        prim_id = (int)tri;
        u_out = h.a1;
        v_out = h.a2;
    }
    nl = ALWAYS_UV ? normalize(nl) : sh_normalize(nl); // the parity entry reports the reference's rounding
    if (h.prim < 0) {
        // sphere: scene.cpp:434-437
        if (ALWAYS_UV || (nd->uses_uv & 1u) != 0u) uvx = atan2f(nl.x, nl.z) / 3.14159265358979323846f + 0.5f;
        uvy = nl.y * 0.5f + 0.5f;
        u_out = uvx;
        v_out = uvy;
    }
    n_world = ALWAYS_UV ? normalize(mat_point(nd->nrm, nl, 0.0f)) : sh_normalize(sh_mat_dir(nd->nrm, nl));
}

// ---------------------------------------------------------------- samplers

__device__ __forceinline__ V3 tex_pixel(const DTexture t, uint32_t x, uint32_t y)
{
    const float* p = t.rgb + ((size_t)y * t.w + x) * 3; // sampler.cpp:67-70
    return mk(p[0], p[1], p[2]);
}

// i % w for an index that is almost always below 2w (texel coordinates come from fract(u) * w, so they
// are at most w): one compare-and-subtract instead of the ~25-instruction unsigned remainder; the general
// remainder stays for whatever a non-finite uv turns into.
__device__ __forceinline__ uint32_t wrap_index(uint32_t i, uint32_t w)
{
    if (i >= w) {
        i -= w;
        if (i >= w) i %= w;
    }
    return i;
}

// image::sample, sampler.cpp:36-61
__device__ __forceinline__ V3 image_sample(const DTexture t, uint32_t mode, float u, float v)
{
    const float u0 = u - floorf(u);
    const float v0 = v - floorf(v);
    const float x = u0 * (float)t.w;
    const float y = v0 * (float)t.h;
    if (mode == PTK_MODE_LINEAR) {
        const uint32_t x0 = wrap_index((uint32_t)floorf(x), t.w);
        const uint32_t x1 = wrap_index((uint32_t)ceilf(x), t.w);
        const uint32_t y0 = wrap_index((uint32_t)floorf(y), t.h);
        const uint32_t y1 = wrap_index((uint32_t)ceilf(y), t.h);
        const V3 a = mix(tex_pixel(t, x0, y0), tex_pixel(t, x1, y0), x - (float)x0);
        const V3 b = mix(tex_pixel(t, x0, y1), tex_pixel(t, x1, y1), x - (float)x0);
        return mix(a, b, y - (float)y0);
    }
    const uint32_t ix = wrap_index((uint32_t)roundf(x), t.w);
    const uint32_t iy = wrap_index((uint32_t)roundf(y), t.h);
    return tex_pixel(t, ix, iy);
}

// sampler_2d::sample: color_sampler (sampler.h:58) or image
__device__ __forceinline__ V3 sampler_sample(const DScene& S, const DSampler sd, float u, float v)
{
    if (sd.kind == PTK_SAMPLER_COLOR) return mk(sd.r, sd.g, sd.b);
    return image_sample(S.textures[sd.texture], sd.mode, u, v);
}

// scene.background->sample(dir): color_sampler or equirectangular_map (sampler.cpp:116-124)
__device__ __forceinline__ V3 background_sample(const DScene& S, V3 dir)
{
    if (S.bg_kind == PTK_BACKGROUND_COLOR) return mk(S.bg_r, S.bg_g, S.bg_b);
#ifdef PTK_BG_LIBM // the library's atan2f / asinf and correctly rounded divides, for A/B runs
    const V3 n = normalize(dir);
    const float u = atan2f(n.x, n.z) / (2.0f * 3.14159265358979323846f) + 0.5f;
    const float v = asinf(n.y) / 3.14159265358979323846f - 0.5f;
#else
    const V3 n = sh_normalize(dir);
    const float u = __builtin_fmaf(sh_atan2(n.x, n.z), 0.15915494309189533577f, 0.5f);
    const float v = __builtin_fmaf(sh_asin(n.y), 0.31830988618379067154f, -0.5f);
#endif
    return image_sample(S.textures[S.bg_texture], S.bg_mode, u, v);
}

// ---------------------------------------------------------------- random numbers
// Counter-based Philox4x32-10: counter = (pixel, sample_lo, sample_hi, block),
// key = seed.  One block yields four of the path's 12 dimensions in the
// reference's draw order (fx, fy, then z, phi per bounce: abstract_pathtracer.cpp:76-77,
// rng.cpp:27-29).  Replaces the thread_local mt19937 of rng.h:15.
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

// Dimension pair (2*j, 2*j+1) of path (pixel, sample): j = 0 primary jitter,
// j = 1 + bounce for the hemisphere draw of that bounce.
__device__ __forceinline__ void rng_pair(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t j, float& u1,
                                         float& u2)
{
    const uint4 r = philox4x32_10(pixel, (uint32_t)sample, (uint32_t)(sample >> 32), j >> 1, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    u1 = u01((j & 1u) ? r.z : r.x);
    u2 = u01((j & 1u) ? r.w : r.y);
}

// rng::hemisphere, rng.cpp:12-36 (shading arithmetic)
__device__ __forceinline__ V3 hemisphere_dir(V3 n, float u1, float u2)
{
    // rng.cpp:16-21: one of two tangents; the operands are selected first so that a wave whose
    // lanes disagree runs the normalisation once (same operations on the same values either way)
    const bool xc = fabsf(n.x) > fabsf(n.y);
    const V3 t = sh_normalize(mk(xc ? n.z : 0.0f, xc ? 0.0f : -n.z, xc ? -n.x : n.y));
    const V3 b = sh_cross(n, t);
    const float z = u1;
    const float r = __builtin_amdgcn_sqrtf(__builtin_fmaf(-z, z, 1.0f));
    float sn, cs;
    sincos_2pi_shade(u2, sn, cs);
    const float tx = r * cs, ty = r * sn;
    return sh_madd(t, tx, sh_madd(b, ty, n * z));
}

// abstract_pathtracer.cpp:74-83
__device__ __forceinline__ V3 primary_dir(const DView& V, uint32_t x, uint32_t y, float u1, float u2)
{
    const float fx = u1 - 0.5f + (float)x;
    const float fy = u2 - 0.5f + (float)y;
    const float x_factor = fx / (float)V.width * 2.0f - 1.0f;
    const float y_factor = 1.0f - fy / (float)V.height * 2.0f;
    const V3 fwd = mk(V.fx, V.fy, V.fz), right = mk(V.rx, V.ry, V.rz), up = mk(V.ux, V.uy, V.uz);
    return normalize(fwd + right * x_factor * V.w2 + up * y_factor * V.h2);
}

} // namespace ptk
