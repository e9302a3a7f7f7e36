/*
 * pt_oracle.c — CPU restatement of the reference's per-pixel / per-sample trace
 * loop (erbuka/pathtracing).  TEST INFRASTRUCTURE ONLY: nothing under
 * pathtracing_amd/ may import, link or call this; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PARITY UNPINNED by a reference build: the reference cannot be compiled in this
 * image (its glm and spdlog submodules are empty, .gitmodules:7-12, and the
 * round rules forbid stand-in headers), and it ships no tests or golden vectors.
 * What pins this file instead: the furnace known answer (furnace.json, E[L]=0.18),
 * hand-derived KD topologies of plane.obj / cube.obj, analytic sphere / slab
 * cases, and the two tonemapped PNGs the reference holds (the PNGs under results/) — see
 * tests/test_oracle_*.py and DESIGN.md.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).  glm is absent; its arithmetic is restated from glm's
 * published formulae (g-truc/glm, version unpinned by the reference):
 *   dot(vec3)  = a.x*b.x + a.y*b.y + a.z*b.z             (left-assoc)
 *   dot(vec4)  = (x + y) + (z + w)
 *   mat4*vec4  = (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
 *   normalize  = v * (1 / sqrt(dot(v, v)))
 *   cross(a,b) = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
 *   reflect    = I - N * dot(N, I) * 2
 *   mix(x,y,a) = x*(1-a) + y*a ;  fract(x) = x - floor(x) ;  length2 = dot(v,v)
 * Build: gcc -O2 -ffp-contract=off (x86-64 baseline, no FMA), matching the
 * reference's premake Release (-O2, no -march, no fast-math; premake5.lua:12-15).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ptk.h"

/* Dynamic operation counts for the algorithmic-flop model (bench.py). */
enum {
    ST_KD_BOX = 0,     /* bounding_box::intersect calls                      */
    ST_TRI_A = 1,      /* triangles reaching the plane-side test             */
    ST_TRI_B = 2,      /* ... passing distance >= 0 (cosine computed)        */
    ST_TRI_FULL = 3,   /* ... passing cosine < 0 (projection + barycentrics) */
    ST_TRI_HIT = 4,    /* ... inside the triangle                            */
    ST_SPHERE = 5,     /* sphere::intersect calls                            */
    ST_SPHERE_HIT = 6,
    ST_NODE = 7,       /* ray -> node-local transforms                       */
    ST_NODE_HIT = 8,   /* hits transformed back to world space               */
    ST_RAY = 9,        /* scene::cast_ray calls                              */
    ST_SHADE = 10,     /* rays that hit (material + bounce direction)        */
    ST_MISS = 11,      /* rays that sampled the background                   */
    ST_PATH = 12,      /* primary rays generated                             */
    ST_MESH_WIN = 13,  /* rays whose closest hit is a triangle               */
    ST_TEXEL = 14,     /* texels read by image::sample (4 per linear lookup) */
    ST_COUNT = 16
};
static __thread uint64_t* tl_stats = 0;
#define STAT(i) do { if (tl_stats) tl_stats[i]++; } while (0)

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;
typedef struct { v3 o, d; } ray_t;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b)
{
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline v3 mix3(v3 x, v3 y, float a) { return add3(scale3(x, 1.0f - a), scale3(y, a)); }

/* glm mat4 * vec4 -> xyz (column-major m[c*4+r]). */
static inline v3 mat_mul_point(const float* m, v3 p, float w)
{
    v3 r;
    r.x = (m[0] * p.x + m[4] * p.y) + (m[8] * p.z + m[12] * w);
    r.y = (m[1] * p.x + m[5] * p.y) + (m[9] * p.z + m[13] * w);
    r.z = (m[2] * p.x + m[6] * p.y) + (m[10] * p.z + m[14] * w);
    return r;
}

/* scene.cpp:18-23  ray operator*(mat4, ray) */
static inline ray_t ray_transform(const float* m, ray_t r)
{
    ray_t o;
    o.o = mat_mul_point(m, r.o, 1.0f);
    o.d = normalize3(mat_mul_point(m, r.d, 0.0f));
    return o;
}

/* ---- geometry ----------------------------------------------------------- */

typedef struct {
    ptk_vertex v[3];
    v3 e0, e1, e2, n;       /* scene.h:123-124 m_edges, m_face_normal */
    float d00, d01, d11, inv_den;
    uint32_t id;            /* index in mesh::get_triangles() */
} tri_t;

typedef struct kd_node {
    float bmin[3], bmax[3];
    uint32_t depth;
    tri_t* tris;            /* by-value copies, scene.cpp:222-223,274 */
    uint64_t ntris;
    struct kd_node *left, *right;
    /* Only for the model of the library's pruned walk (below): padded box of the triangles
     * under this node and the DFS position of its first leaf triangle.  Not in the reference. */
    float tmin[3], tmax[3];
    uint64_t first_slot;
} kd_node;

typedef struct {
    tri_t* tris;
    uint64_t ntris;
    float bmin[3], bmax[3];
    kd_node* root;
} mesh_t;

typedef struct {
    int hit;
    v3 position, normal;
    v2 uv;
    /* instrumentation (not in raycast_result, scene.h:47-53) */
    int32_t prim;
    float a, b;             /* (u,v) per SURVEY §8c */
} hit_t;

static inline v3 P3(const float* p) { return V3(p[0], p[1], p[2]); }

/* scene.cpp:91-107  triangle::update */
static void tri_update(tri_t* t)
{
    v3 p0 = P3(t->v[0].position), p1 = P3(t->v[1].position), p2 = P3(t->v[2].position);
    t->n = normalize3(cross3(sub3(p1, p0), sub3(p2, p1)));
    t->e0 = sub3(p1, p0);
    t->e1 = sub3(p2, p0);
    t->e2 = sub3(p2, p1);
    t->d00 = dot3(t->e0, t->e0);
    t->d01 = dot3(t->e0, t->e1);
    t->d11 = dot3(t->e1, t->e1);
    t->inv_den = 1.0f / (t->d00 * t->d11 - t->d01 * t->d01);
}

/* scene.cpp:78-89  triangle::baricentric */
static inline v3 tri_baricentric(const tri_t* t, v3 point)
{
    const v3 v2_ = sub3(point, P3(t->v[0].position));
    const float d20 = dot3(v2_, t->e0);
    const float d21 = dot3(v2_, t->e1);
    const float v = (t->d11 * d20 - t->d01 * d21) * t->inv_den;
    const float w = (t->d00 * d21 - t->d01 * d20) * t->inv_den;
    const float u = 1.0f - v - w;
    return V3(u, v, w);
}

/* scene.cpp:58-75  bounding_box::intersect */
static inline int box_intersect(const float* bmin, const float* bmax, const ray_t* r)
{
    const float t1 = (bmin[0] - r->o.x) / r->d.x;
    const float t2 = (bmax[0] - r->o.x) / r->d.x;
    const float t3 = (bmin[1] - r->o.y) / r->d.y;
    const float t4 = (bmax[1] - r->o.y) / r->d.y;
    const float t5 = (bmin[2] - r->o.z) / r->d.z;
    const float t6 = (bmax[2] - r->o.z) / r->d.z;
    const float t7 = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float t8 = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    const float t9 = (t8 < 0 || t7 > t8) ? -1 : t7;
    return t9 != -1;
}

/* scene.cpp:142-186  mesh::intersect_triangle */
static inline hit_t tri_intersect(const ray_t* r, const tri_t* t)
{
    hit_t res;
    res.hit = 0;
    STAT(ST_TRI_A);
    const v3 l = sub3(r->o, P3(t->v[0].position));
    const float distance = dot3(l, t->n);
    if (distance < 0) return res;
    STAT(ST_TRI_B);
    const float cosine = dot3(r->d, t->n);
    if (cosine >= 0) return res;
    STAT(ST_TRI_FULL);
    const v3 projection = add3(r->o, scale3(r->d, distance / -cosine));
    const v3 bar = tri_baricentric(t, projection);
    if (bar.x >= 0 && bar.y >= 0 && bar.z >= 0) {
        STAT(ST_TRI_HIT);
        res.hit = 1;
        res.position = projection;
        res.normal = normalize3(add3(add3(scale3(P3(t->v[0].normal), bar.x),
                                          scale3(P3(t->v[1].normal), bar.y)),
                                     scale3(P3(t->v[2].normal), bar.z)));
        res.uv.x = bar.x * t->v[0].uv[0] + bar.y * t->v[1].uv[0] + bar.z * t->v[2].uv[0];
        res.uv.y = bar.x * t->v[0].uv[1] + bar.y * t->v[1].uv[1] + bar.z * t->v[2].uv[1];
        res.prim = (int32_t)t->id;
        res.a = bar.y;
        res.b = bar.z;
    }
    return res;
}

/* scene.cpp:188-212  mesh::intersect_internal */
static void mesh_intersect_internal(const ray_t* r, const kd_node* node, hit_t* result, float* distance,
                                    uint64_t* stats)
{
    if (stats) stats[0]++;
    STAT(ST_KD_BOX);
    if (box_intersect(node->bmin, node->bmax, r)) {
        for (uint64_t i = 0; i < node->ntris; ++i) {
            if (stats) stats[1]++;
            hit_t h = tri_intersect(r, &node->tris[i]);
            if (h.hit) {
                const v3 dv = sub3(r->o, h.position);
                const float d = dot3(dv, dv);
                if (d < *distance) {
                    *result = h;
                    *distance = d;
                }
            }
        }
        if (node->left) mesh_intersect_internal(r, node->left, result, distance, stats);
        if (node->right) mesh_intersect_internal(r, node->right, result, distance, stats);
    }
}

/* ---- model of the LIBRARY's pruned KD walk (not reference code) ----------------
 * Restates, on the pointer tree, what kd_walk_pruned in pathtracing_amd/csrc/ptk_device.h
 * does: enter a node only if the reference's box test passes, near child first by the sign
 * of the direction on the split axis, skip subtrees whose padded triangle box is missed,
 * behind, or farther than the best hit, lowest DFS slot wins exact ties.  It exists so that
 * (a) tests can check on the CPU, over millions of rays, that this walk returns exactly what
 * mesh_intersect_internal returns, and (b) bench.py can count the nodes and triangles the
 * walk touches (the algorithmic bytes of the HBM roofline).  orc_set_walk_model(1) makes
 * every entry point of this file use it. */
static int g_walk_model = 0;
void orc_set_walk_model(int model) { g_walk_model = model; }

typedef struct {
    v3 inv;
    float oabs;
    uint32_t neg;
    float best_d;
    uint64_t best_slot;
    hit_t best;
} walk_t;

static int model_prune(const kd_node* n, const ray_t* r, const walk_t* w, float* bound)
{
    const float ax = (n->tmin[0] - r->o.x) * w->inv.x, bx = (n->tmax[0] - r->o.x) * w->inv.x;
    const float ay = (n->tmin[1] - r->o.y) * w->inv.y, by = (n->tmax[1] - r->o.y) * w->inv.y;
    const float az = (n->tmin[2] - r->o.z) * w->inv.z, bz = (n->tmax[2] - r->o.z) * w->inv.z;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const int missed = (t_in - t_out) > 1e-5f * (fabsf(t_in) + fabsf(t_out)) + 2.0f * w->oabs;
    const int behind = t_out < 0.0f && t_out * 0.99999f < -w->oabs;
    const float near_lo = t_in * 0.99999f - w->oabs;
    *bound = near_lo > 0.0f ? near_lo * near_lo : 0.0f;
    return missed || behind || *bound > w->best_d;
}

static void model_walk(const ray_t* r, const kd_node* node, walk_t* w, uint64_t* stats)
{
    if (node->ntris || (!node->left && !node->right)) {
        for (uint64_t i = 0; i < node->ntris; ++i) {
            if (stats) stats[1]++;
            hit_t h = tri_intersect(r, &node->tris[i]);
            if (!h.hit) continue;
            const v3 dv = sub3(r->o, h.position);
            const float d = dot3(dv, dv);
            const uint64_t slot = node->first_slot + i;
            if (d < w->best_d || (d == w->best_d && slot < w->best_slot)) {
                w->best = h;
                w->best_d = d;
                w->best_slot = slot;
            }
        }
        return;
    }
    const kd_node* kid[2] = { node->left, node->right };
    int enter[2] = { 0, 0 };
    float bound[2] = { 0.0f, 0.0f };
    for (int c = 0; c < 2; ++c) {
        if (!kid[c]) continue;
        if (stats) stats[0]++;
        STAT(ST_KD_BOX);
        enter[c] = box_intersect(kid[c]->bmin, kid[c]->bmax, r) && !model_prune(kid[c], r, w, &bound[c]);
    }
    const int first = ((w->neg >> (node->depth % 3)) & 1u) ? 1 : 0; /* right first when the direction is negative */
    const int second = 1 - first;
    if (enter[first]) model_walk(r, kid[first], w, stats);
    if (enter[second] && !(bound[second] > w->best_d)) model_walk(r, kid[second], w, stats);
}

/* scene.cpp:115-121  mesh::intersect */
static hit_t mesh_intersect(const mesh_t* m, const ray_t* r, uint64_t* stats)
{
    hit_t res;
    res.hit = 0;
    float distance = 3.402823466e+38f;
    if (g_walk_model == 0) {
        mesh_intersect_internal(r, m->root, &res, &distance, stats);
        return res;
    }
    walk_t w;
    w.inv = V3(1.0f / r->d.x, 1.0f / r->d.y, 1.0f / r->d.z);
    w.oabs = 1e-5f * (fabsf(r->o.x) + fabsf(r->o.y) + fabsf(r->o.z)) + 1e-30f;
    w.neg = (r->d.x < 0.0f ? 1u : 0u) | (r->d.y < 0.0f ? 2u : 0u) | (r->d.z < 0.0f ? 4u : 0u);
    w.best_d = distance;
    w.best_slot = ~0ull;
    w.best.hit = 0;
    float bound;
    if (stats) stats[0]++;
    STAT(ST_KD_BOX);
    if (box_intersect(m->root->bmin, m->root->bmax, r) && !model_prune(m->root, r, &w, &bound))
        model_walk(r, m->root, &w, stats);
    return w.best;
}

/* scene.cpp:401-441  sphere::intersect */
static hit_t sphere_intersect(const ray_t* r)
{
    hit_t res;
    res.hit = 0;
    STAT(ST_SPHERE);
    const float projection = dot3(sub3(V3(0.0f, 0.0f, 0.0f), r->o), r->d);
    const float sq_distance = dot3(r->o, r->o) - projection * projection;
    if (sq_distance > 1.0f) return res;
    const float offset = sqrtf(1.0f - sq_distance);
    const float t1 = projection - offset;
    const float t2 = projection + offset;
    if (t1 < 0 && t2 < 0) return res;
    STAT(ST_SPHERE_HIT);
    res.hit = 1;
    res.position = add3(r->o, scale3(r->d, (t1 >= 0.0f ? t1 : t2)));
    res.normal = normalize3(res.position);
    res.uv.x = atan2f(res.normal.x, res.normal.z) / 3.14159265358979323846f + 0.5f;
    res.uv.y = res.normal.y * 0.5f + 0.5f;
    res.prim = -1;
    res.a = res.uv.x;
    res.b = res.uv.y;
    return res;
}

/* ---- KD build: scene.cpp:214-286 (kd_tree_node ctor), 31-56 (split) -------- */

static kd_node* kd_build(const tri_t* tris, uint64_t n, const float* bmin, const float* bmax, uint32_t depth)
{
    kd_node* node = (kd_node*)calloc(1, sizeof(kd_node));
    memcpy(node->bmin, bmin, 12);
    memcpy(node->bmax, bmax, 12);
    node->depth = depth;
    if (n <= 1 || depth == PTK_KD_MAX_DEPTH) {
        node->tris = (tri_t*)malloc((n ? n : 1) * sizeof(tri_t));
        memcpy(node->tris, tris, n * sizeof(tri_t));
        node->ntris = n;
        return node;
    }
    const uint32_t ax = depth % 3;
    float median = 0;
    for (uint64_t i = 0; i < n; ++i) {
        median += tris[i].v[0].position[ax];
        median += tris[i].v[1].position[ax];
        median += tris[i].v[2].position[ax];
    }
    median /= (float)(3 * n); /* size_t -> float conversion, scene.cpp:249 */

    float lmin[3], lmax[3], rmin[3], rmax[3];
    memcpy(lmin, bmin, 12); memcpy(lmax, bmax, 12);
    memcpy(rmin, bmin, 12); memcpy(rmax, bmax, 12);
    lmax[ax] = median;
    rmin[ax] = median;

    tri_t* lt = (tri_t*)malloc(n * sizeof(tri_t));
    tri_t* rt = (tri_t*)malloc(n * sizeof(tri_t));
    uint64_t nl = 0, nr = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const float a = tris[i].v[0].position[ax], b = tris[i].v[1].position[ax], c = tris[i].v[2].position[ax];
        if (a <= median || b <= median || c <= median) lt[nl++] = tris[i];
        if (a >= median || b >= median || c >= median) rt[nr++] = tris[i];
    }
    if ((double)(nl + nr) > 1.5 * (double)n) {
        node->tris = (tri_t*)malloc(n * sizeof(tri_t));
        memcpy(node->tris, tris, n * sizeof(tri_t));
        node->ntris = n;
    } else {
        if (nl > 0) node->left = kd_build(lt, nl, lmin, lmax, depth + 1);
        if (nr > 0) node->right = kd_build(rt, nr, rmin, rmax, depth + 1);
    }
    free(lt);
    free(rt);
    return node;
}

static void kd_free(kd_node* n)
{
    if (!n) return;
    kd_free(n->left);
    kd_free(n->right);
    free(n->tris);
    free(n);
}

/* Walk-model annotations (see model_walk): DFS slot numbers and padded triangle boxes,
 * same padding as fill_triangle_boxes in pathtracing_amd/csrc/kd_build.cpp. */
static void kd_annotate(kd_node* n, uint64_t* slot)
{
    const float big = 3.402823466e+38f;
    for (int a = 0; a < 3; ++a) { n->tmin[a] = big; n->tmax[a] = -big; }
    n->first_slot = *slot;
    *slot += n->ntris;
    for (uint64_t i = 0; i < n->ntris; ++i)
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) {
                const float p = n->tris[i].v[v].position[a];
                if (p < n->tmin[a]) n->tmin[a] = p;
                if (p > n->tmax[a]) n->tmax[a] = p;
            }
    kd_node* kid[2] = { n->left, n->right };
    for (int c = 0; c < 2; ++c) {
        if (!kid[c]) continue;
        kd_annotate(kid[c], slot);
    }
    /* children are padded already; take the union before padding this node */
    for (int c = 0; c < 2; ++c) {
        if (!kid[c]) continue;
        for (int a = 0; a < 3; ++a) {
            if (kid[c]->tmin[a] < n->tmin[a]) n->tmin[a] = kid[c]->tmin[a];
            if (kid[c]->tmax[a] > n->tmax[a]) n->tmax[a] = kid[c]->tmax[a];
        }
    }
    if (n->ntris || (!n->left && !n->right))
        for (int a = 0; a < 3; ++a) {
            if (!(n->tmin[a] <= n->tmax[a])) continue;
            const float ext = n->tmax[a] - n->tmin[a];
            const float mag = fmaxf(fabsf(n->tmin[a]), fabsf(n->tmax[a]));
            const float pad = 1e-4f * ext + 1e-5f * mag + 1e-30f;
            n->tmin[a] -= pad;
            n->tmax[a] += pad;
        }
}

/* scene.cpp:123-140  mesh::compile (bounds grow from the default (0,0,0) box, scene.h:67) */
static void mesh_compile(mesh_t* m)
{
    for (int k = 0; k < 3; ++k) m->bmin[k] = m->bmax[k] = 0.0f;
    for (uint64_t i = 0; i < m->ntris; ++i) {
        tri_update(&m->tris[i]);
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const float p = m->tris[i].v[j].position[k];
                m->bmin[k] = (p < m->bmin[k]) ? p : m->bmin[k]; /* glm::min(x,y) = (y<x)?y:x */
                m->bmax[k] = (m->bmax[k] < p) ? p : m->bmax[k]; /* glm::max(x,y) = (x<y)?y:x */
            }
    }
    m->root = kd_build(m->tris, m->ntris, m->bmin, m->bmax, 0);
    uint64_t slot = 0;
    kd_annotate(m->root, &slot);
}

/* ---- samplers -------------------------------------------------------------- */

typedef struct {
    uint32_t w, h;
    float* rgb;
} tex_t;

static inline v3 tex_pixel(const tex_t* t, uint32_t x, uint32_t y)
{
    const float* p = t->rgb + ((size_t)y * t->w + x) * 3; /* sampler.cpp:67-70 */
    return V3(p[0], p[1], p[2]);
}

/* sampler.cpp:36-61  image::sample */
static v3 image_sample(const tex_t* t, uint32_t mode, v2 uv)
{
    const float u0 = uv.x - floorf(uv.x);
    const float v0 = uv.y - floorf(uv.y);
    const float x = u0 * (float)t->w;
    const float y = v0 * (float)t->h;
    if (tl_stats) tl_stats[ST_TEXEL] += (mode == PTK_MODE_LINEAR) ? 4 : 1;
    if (mode == PTK_MODE_LINEAR) {
        const uint32_t x0 = (uint32_t)floorf(x) % t->w;
        const uint32_t x1 = (uint32_t)ceilf(x) % t->w;
        const uint32_t y0 = (uint32_t)floorf(y) % t->h;
        const uint32_t y1 = (uint32_t)ceilf(y) % t->h;
        const v3 a = mix3(tex_pixel(t, x0, y0), tex_pixel(t, x1, y0), x - (float)x0);
        const v3 b = mix3(tex_pixel(t, x0, y1), tex_pixel(t, x1, y1), x - (float)x0);
        return mix3(a, b, y - (float)y0);
    } else {
        const uint32_t ix = (uint32_t)roundf(x) % t->w;
        const uint32_t iy = (uint32_t)roundf(y) % t->h;
        return tex_pixel(t, ix, iy);
    }
}

/* sampler.cpp:116-124  equirectangular_map::sample */
static v3 equirect_sample(const tex_t* t, uint32_t mode, v3 dir)
{
    const v3 n = normalize3(dir);
    v2 uv;
    uv.x = atan2f(n.x, n.z) / (2.0f * 3.14159265358979323846f) + 0.5f;
    uv.y = asinf(n.y) / 3.14159265358979323846f - 0.5f;
    return image_sample(t, mode, uv);
}

/* ---- scene ------------------------------------------------------------------ */

typedef struct orc_scene {
    ptk_node_desc* nodes;
    uint32_t num_nodes;
    mesh_t* meshes;
    uint32_t num_meshes;
    ptk_sampler_desc* samplers;
    uint32_t num_samplers;
    tex_t* textures;
    uint32_t num_textures;
    ptk_background_desc background;
    ptk_camera_desc camera;
} orc_scene;

void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    for (uint32_t i = 0; i < s->num_meshes; ++i) {
        kd_free(s->meshes[i].root);
        free(s->meshes[i].tris);
    }
    for (uint32_t i = 0; i < s->num_textures; ++i) free(s->textures[i].rgb);
    free(s->nodes); free(s->meshes); free(s->samplers); free(s->textures);
    free(s);
}

orc_scene* orc_scene_create(const ptk_scene_desc* d)
{
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->num_nodes = d->num_nodes;
    s->nodes = (ptk_node_desc*)malloc((d->num_nodes ? d->num_nodes : 1) * sizeof(ptk_node_desc));
    memcpy(s->nodes, d->nodes, d->num_nodes * sizeof(ptk_node_desc));
    s->num_samplers = d->num_samplers;
    s->samplers = (ptk_sampler_desc*)malloc((d->num_samplers ? d->num_samplers : 1) * sizeof(ptk_sampler_desc));
    memcpy(s->samplers, d->samplers, d->num_samplers * sizeof(ptk_sampler_desc));
    s->num_textures = d->num_textures;
    s->textures = (tex_t*)calloc(d->num_textures ? d->num_textures : 1, sizeof(tex_t));
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const size_t n = (size_t)d->textures[i].width * d->textures[i].height * 3;
        s->textures[i].w = d->textures[i].width;
        s->textures[i].h = d->textures[i].height;
        s->textures[i].rgb = (float*)malloc(n * sizeof(float));
        memcpy(s->textures[i].rgb, d->textures[i].rgb, n * sizeof(float));
    }
    s->num_meshes = d->num_meshes;
    s->meshes = (mesh_t*)calloc(d->num_meshes ? d->num_meshes : 1, sizeof(mesh_t));
    for (uint32_t i = 0; i < d->num_meshes; ++i) {
        mesh_t* m = &s->meshes[i];
        m->ntris = d->meshes[i].num_triangles;
        m->tris = (tri_t*)calloc(m->ntris ? m->ntris : 1, sizeof(tri_t));
        for (uint64_t k = 0; k < m->ntris; ++k) {
            memcpy(m->tris[k].v, d->meshes[i].triangles[k].v, sizeof(ptk_triangle));
            m->tris[k].id = (uint32_t)k;
        }
        mesh_compile(m);
    }
    s->background = d->background;
    s->camera = d->camera;
    return s;
}

/* scene.cpp:355-399  scene::cast_ray (return_on_first_hit / avoid_nodes unused by any caller) */
static hit_t scene_cast_ray(const orc_scene* s, const ray_t* r, int32_t* hit_node, float* dist2, uint64_t* stats)
{
    float distance = 3.402823466e+38f;
    hit_t best;
    best.hit = 0;
    *hit_node = -1;
    STAT(ST_RAY);
    for (uint32_t i = 0; i < s->num_nodes; ++i) {
        const ptk_node_desc* nd = &s->nodes[i];
        STAT(ST_NODE);
        const ray_t lr = ray_transform(nd->inv_transform, *r);
        hit_t r0 = (nd->shape_kind == PTK_SHAPE_SPHERE) ? sphere_intersect(&lr)
                                                        : mesh_intersect(&s->meshes[nd->mesh], &lr, stats);
        if (r0.hit) {
            STAT(ST_NODE_HIT);
            r0.position = mat_mul_point(nd->transform, r0.position, 1.0f);
            r0.normal = normalize3(mat_mul_point(nd->normal_transform, r0.normal, 0.0f));
            const v3 dv = sub3(r0.position, r->o);
            const float d0 = dot3(dv, dv);
            if (d0 < distance) {
                distance = d0;
                best = r0;
                *hit_node = (int32_t)i;
            }
        }
    }
    if (dist2) *dist2 = distance;
    if (best.hit) { STAT(ST_SHADE); if (best.prim >= 0) STAT(ST_MESH_WIN); } else STAT(ST_MISS);
    return best;
}

/* sampler_2d::sample dispatch: color_sampler (sampler.h:58) or image (sampler.cpp:36) */
static inline v3 sampler_sample(const orc_scene* s, uint32_t idx, v2 uv)
{
    const ptk_sampler_desc* sd = &s->samplers[idx];
    if (sd->kind == PTK_SAMPLER_COLOR) return V3(sd->color[0], sd->color[1], sd->color[2]);
    return image_sample(&s->textures[sd->texture], sd->mode, uv);
}

static inline v3 background_sample(const orc_scene* s, v3 dir)
{
    if (s->background.kind == PTK_BACKGROUND_COLOR)
        return V3(s->background.color[0], s->background.color[1], s->background.color[2]);
    return equirect_sample(&s->textures[s->background.texture], s->background.mode, dir);
}

/* ---- random numbers ----------------------------------------------------------
 * mode 0: counter-based Philox4x32-10 keyed by (seed; pixel, sample, block) — the
 *         build's own generator (the reference's thread_local mt19937, rng.h:15,
 *         cannot be reproduced across GPU threads; SURVEY §7 "Hard parts").
 * mode 1: std::mt19937 + libstdc++ uniform_real_distribution<float> in the
 *         reference's draw order (rng.h:25, rng.cpp:27-29), for --threads 1 replays. */
typedef struct {
    int mode;
    /* philox */
    uint32_t key[2], ctr[3];
    uint32_t dim;
    uint32_t block[4];
    /* mt19937 */
    uint32_t mt[624];
    int mti;
} rng_t;

static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void mt_seed(rng_t* g, uint32_t s)
{
    g->mt[0] = s;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->mti = 624;
}

static uint32_t mt_next_u32(rng_t* g)
{
    if (g->mti >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->mti = 0;
    }
    uint32_t y = g->mt[g->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static inline void rng_begin_path(rng_t* g, uint64_t seed, uint32_t pixel, uint64_t sample)
{
    g->key[0] = (uint32_t)seed;
    g->key[1] = (uint32_t)(seed >> 32);
    g->ctr[0] = pixel;
    g->ctr[1] = (uint32_t)sample;
    g->ctr[2] = (uint32_t)(sample >> 32);
    g->dim = 0;
}

/* rng::next(), rng.h:25 */
static inline float rng_next(rng_t* g)
{
    if (g->mode == 1) {
        /* libstdc++ generate_canonical<float,24>: float(x) / 2^32, clamped below 1 */
        float r = (float)mt_next_u32(g) / 4294967296.0f;
        if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
        return r;
    }
    if ((g->dim & 3u) == 0) {
        uint32_t c[4] = { g->ctr[0], g->ctr[1], g->ctr[2], g->dim >> 2 };
        philox4x32_10(c, g->key, g->block);
    }
    const uint32_t x = g->block[g->dim & 3u];
    g->dim++;
    return (float)(x >> 8) * 5.9604644775390625e-8f; /* 2^-24: uniform on [0,1) */
}

/* rng.cpp:12-36  rng::hemisphere with the two draws supplied */
static inline v3 hemisphere_dir(v3 n, float u1, float u2)
{
    v3 t;
    if (fabsf(n.x) > fabsf(n.y)) t = normalize3(V3(n.z, 0.0f, -n.x));
    else t = normalize3(V3(0.0f, -n.z, n.y));
    const v3 b = cross3(n, t);
    const float z = u1;
    const float r = sqrtf(1.0f - z * z);
    const float phi = 3.14159265358979323846f * 2.0f * u2;
    const float tx = r * cosf(phi), ty = r * sinf(phi), tz = z;
    return add3(add3(scale3(t, tx), scale3(b, ty)), scale3(n, tz));
}

/* pathtracer.cpp:19-69  pathtracer::trace_recursive */
static v3 trace_recursive(const orc_scene* s, ray_t r, uint32_t recursion, rng_t* g, uint64_t* nrays)
{
    if (recursion == 0) return V3(0.0f, 0.0f, 0.0f);
    int32_t node;
    if (nrays) (*nrays)++;
    const hit_t res = scene_cast_ray(s, &r, &node, NULL, NULL);
    if (res.hit) {
        const ptk_node_desc* nd = &s->nodes[node];
        const v3 albedo = sampler_sample(s, nd->albedo, res.uv);
        const v3 emission = sampler_sample(s, nd->emission, res.uv);
        const float roughness = sampler_sample(s, nd->roughness, res.uv).x;
        const float metallic = sampler_sample(s, nd->metallic, res.uv).x;
        const float u1 = rng_next(g);
        const float u2 = rng_next(g);
        const v3 hemi_dir = hemisphere_dir(res.normal, u1, u2);
        const v3 reflect_dir = sub3(r.d, scale3(scale3(res.normal, dot3(res.normal, r.d)), 2.0f));
        const v3 dir = normalize3(mix3(reflect_dir, hemi_dir, roughness));
        ray_t rr;
        rr.o = add3(res.position, scale3(dir, PTK_EPSILON));
        rr.d = dir;
        const float dn = dot3(rr.d, res.normal);
        const float cos_theta = (0.0f < dn) ? dn : 0.0f; /* glm::max(0, x) */
        const v3 radiance = trace_recursive(s, rr, recursion - 1, g, nrays);
        const v3 m = mix3(albedo, V3(1.0f, 1.0f, 1.0f), metallic);
        return add3(emission, scale3(scale3(mul3(m, radiance), cos_theta), 2.0f));
    }
    return background_sample(s, r.d);
}

/* abstract_pathtracer.cpp:20-25 */
typedef struct { v3 fwd, right, up; float h2, w2; } view_t;

static view_t make_view(const orc_scene* s, uint32_t w, uint32_t h, float fov_y)
{
    view_t v;
    v.fwd = normalize3(P3(s->camera.direction));
    v.right = normalize3(cross3(v.fwd, V3(0.0f, 1.0f, 0.0f)));
    v.up = cross3(v.right, v.fwd);
    v.h2 = atanf(fov_y / 2.0f);
    v.w2 = v.h2 * (float)w / (float)h;
    return v;
}

/* abstract_pathtracer.cpp:74-83 */
static inline ray_t primary_ray(const orc_scene* s, const view_t* v, uint32_t w, uint32_t h, uint32_t x,
                                uint32_t y, float u1, float u2)
{
    const float fx = u1 - 0.5f + (float)x;
    const float fy = u2 - 0.5f + (float)y;
    const float x_factor = fx / (float)w * 2.0f - 1.0f;
    const float y_factor = 1.0f - fy / (float)h * 2.0f;
    ray_t r;
    r.o = P3(s->camera.position);
    r.d = normalize3(add3(add3(v->fwd, scale3(scale3(v->right, x_factor), v->w2)),
                          scale3(scale3(v->up, y_factor), v->h2)));
    return r;
}

/* ---- exported entry points ----------------------------------------------------- */

/* SoA in/out exactly as ptk_intersect_batch (include/ptk.h). stats (optional):
 * [0] += KD nodes whose box was tested, [1] += triangles tested. */
void orc_intersect_batch(const orc_scene* s, uint64_t n, const float* o, const float* d, int32_t* node,
                         int32_t* prim, float* t, float* u, float* v, float* position, float* normal,
                         float* uv, uint64_t* stats)
{
    for (uint64_t i = 0; i < n; ++i) {
        ray_t r;
        r.o = V3(o[i], o[n + i], o[2 * n + i]);
        r.d = V3(d[i], d[n + i], d[2 * n + i]);
        int32_t hn;
        float d2;
        const hit_t h = scene_cast_ray(s, &r, &hn, &d2, stats);
        if (node) node[i] = hn;
        if (prim) prim[i] = h.hit ? h.prim : -1;
        if (t) t[i] = h.hit ? sqrtf(d2) : 0.0f;
        if (u) u[i] = h.hit ? h.a : 0.0f;
        if (v) v[i] = h.hit ? h.b : 0.0f;
        if (position) {
            position[i] = h.hit ? h.position.x : 0.0f;
            position[n + i] = h.hit ? h.position.y : 0.0f;
            position[2 * n + i] = h.hit ? h.position.z : 0.0f;
        }
        if (normal) {
            normal[i] = h.hit ? h.normal.x : 0.0f;
            normal[n + i] = h.hit ? h.normal.y : 0.0f;
            normal[2 * n + i] = h.hit ? h.normal.z : 0.0f;
        }
        if (uv) {
            uv[i] = h.hit ? h.uv.x : 0.0f;
            uv[n + i] = h.hit ? h.uv.y : 0.0f;
        }
    }
}

typedef struct {
    const orc_scene* s;
    view_t view;
    uint32_t w, h;
    uint64_t seed;
    uint64_t pixel_begin, pixel_end;
    uint64_t sample_begin;
    uint32_t sample_count;
    float* sum;
    int accumulate;
    uint64_t rays;
    int want_stats;
    uint64_t stats[ST_COUNT];
} sum_job;

static void* sum_worker(void* arg)
{
    sum_job* j = (sum_job*)arg;
    rng_t g;
    g.mode = 0;
    tl_stats = j->want_stats ? j->stats : 0;
    for (uint64_t p = j->pixel_begin; p < j->pixel_end; ++p) {
        const uint32_t x = (uint32_t)(p % j->w), y = (uint32_t)(p / j->w);
        v3 color = V3(0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < j->sample_count; ++k) {
            rng_begin_path(&g, j->seed, (uint32_t)p, j->sample_begin + k);
            const float u1 = rng_next(&g);
            const float u2 = rng_next(&g);
            STAT(ST_PATH);
            const ray_t r = primary_ray(j->s, &j->view, j->w, j->h, x, y, u1, u2);
            color = add3(color, trace_recursive(j->s, r, PTK_MAX_DEPTH, &g, &j->rays));
        }
        float* o = j->sum + p * 3;
        if (j->accumulate) { o[0] += color.x; o[1] += color.y; o[2] += color.z; }
        else { o[0] = color.x; o[1] = color.y; o[2] = color.z; }
    }
    tl_stats = 0;
    return NULL;
}

/* Mirror of ptk_render_samples_dev: per-pixel SUM over a sample range, Philox RNG,
 * samples added in increasing sample index (abstract_pathtracer.cpp:70-86).
 * threads > 1 splits the pixel range (results do not depend on it). Returns rays cast. */
static uint64_t render_samples_impl(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t seed,
                                    uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                                    uint32_t sample_count, float* sum_rgb, int accumulate, int threads,
                                    uint64_t* stats_out);

uint64_t orc_render_samples(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t seed,
                            uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                            uint32_t sample_count, float* sum_rgb, int accumulate, int threads)
{
    return render_samples_impl(s, w, h, fov_y, seed, pixel_begin, pixel_count, sample_begin, sample_count, sum_rgb,
                               accumulate, threads, NULL);
}

/* Same, also returning the ST_* operation counts (stats_out[ST_COUNT]). */
uint64_t orc_render_samples_stats(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t seed,
                                  uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                                  uint32_t sample_count, float* sum_rgb, int accumulate, int threads,
                                  uint64_t* stats_out)
{
    return render_samples_impl(s, w, h, fov_y, seed, pixel_begin, pixel_count, sample_begin, sample_count, sum_rgb,
                               accumulate, threads, stats_out);
}

static uint64_t render_samples_impl(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t seed,
                                    uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                                    uint32_t sample_count, float* sum_rgb, int accumulate, int threads,
                                    uint64_t* stats_out)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > pixel_count) threads = pixel_count ? (int)pixel_count : 1;
    sum_job* jobs = (sum_job*)calloc((size_t)threads, sizeof(sum_job));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    const view_t view = make_view(s, w, h, fov_y);
    /* interleave small pixel blocks so threads get similar work */
    for (int i = 0; i < threads; ++i) {
        jobs[i].s = s; jobs[i].view = view; jobs[i].w = w; jobs[i].h = h; jobs[i].seed = seed;
        jobs[i].pixel_begin = pixel_begin + pixel_count * (uint64_t)i / (uint64_t)threads;
        jobs[i].pixel_end = pixel_begin + pixel_count * (uint64_t)(i + 1) / (uint64_t)threads;
        jobs[i].sample_begin = sample_begin; jobs[i].sample_count = sample_count;
        jobs[i].sum = sum_rgb; jobs[i].accumulate = accumulate;
        jobs[i].want_stats = stats_out != NULL;
    }
    if (threads == 1) sum_worker(&jobs[0]);
    else {
        for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, sum_worker, &jobs[i]);
        for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
    }
    uint64_t rays = 0;
    for (int i = 0; i < threads; ++i) rays += jobs[i].rays;
    if (stats_out) {
        memset(stats_out, 0, ST_COUNT * sizeof(uint64_t));
        for (int i = 0; i < threads; ++i)
            for (int k = 0; k < ST_COUNT; ++k) stats_out[k] += jobs[i].stats[k];
    }
    free(jobs);
    free(th);
    return rays;
}

/* abstract_pathtracer.cpp:66,88-90: running mean over iterations. */
void orc_resolve_iteration(const float* sum_rgb, float* image_rgb, uint64_t num_pixels,
                           uint64_t samples_per_iteration, uint64_t iteration)
{
    const float t = (float)iteration / (float)(iteration + 1);
    const float spi = (float)samples_per_iteration;
    for (uint64_t i = 0; i < num_pixels * 3; ++i) {
        const float c = sum_rgb[i] / spi;
        image_rgb[i] = c * (1.0f - t) + image_rgb[i] * t;
    }
}

/* Whole render loop (abstract_pathtracer.cpp:14-117) with Philox sampling. */
uint64_t orc_render_image(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t seed,
                          uint64_t iterations, uint32_t spi, float* image_rgb, int threads)
{
    const uint64_t np = (uint64_t)w * h;
    float* sum = (float*)malloc(np * 3 * sizeof(float));
    memset(image_rgb, 0, np * 3 * sizeof(float));
    uint64_t rays = 0;
    for (uint64_t it = 0; it < iterations; ++it) {
        rays += orc_render_samples(s, w, h, fov_y, seed, 0, np, it * spi, spi, sum, 0, threads);
        orc_resolve_iteration(sum, image_rgb, np, spi, it);
    }
    free(sum);
    return rays;
}

/* The reference's own sampling order with --threads 1: the driver thread's
 * default-seeded mt19937 (seed 5489) draws one worker seed per iteration
 * (abstract_pathtracer.cpp:100, rng.h:27-30), the single worker reseeds
 * (abstract_pathtracer.cpp:61) and walks scan lines top to bottom. */
void orc_render_image_mt19937(const orc_scene* s, uint32_t w, uint32_t h, float fov_y,
                              uint64_t iterations, uint32_t spi, float* image_rgb)
{
    rng_t driver, g;
    driver.mode = 1;
    g.mode = 1;
    mt_seed(&driver, 5489u);
    const view_t view = make_view(s, w, h, fov_y);
    memset(image_rgb, 0, (size_t)w * h * 3 * sizeof(float));
    for (uint64_t it = 0; it < iterations; ++it) {
        const uint32_t seed = (uint32_t)(0u + (float)4294967295u * rng_next(&driver));
        mt_seed(&g, seed);
        const float t = (float)it / (float)(it + 1);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                v3 color = V3(0.0f, 0.0f, 0.0f);
                for (uint32_t k = 0; k < spi; ++k) {
                    const float u1 = rng_next(&g);
                    const float u2 = rng_next(&g);
                    const ray_t r = primary_ray(s, &view, w, h, x, y, u1, u2);
                    color = add3(color, trace_recursive(s, r, PTK_MAX_DEPTH, &g, NULL));
                }
                float* px = image_rgb + ((size_t)y * w + x) * 3;
                const v3 prev = V3(px[0], px[1], px[2]);
                const float fs = (float)spi;
                const v3 mean = V3(color.x / fs, color.y / fs, color.z / fs);
                const v3 nc = mix3(mean, prev, t);
                px[0] = nc.x; px[1] = nc.y; px[2] = nc.z;
            }
    }
}

/* Flattened DFS-preorder dump of the pointer tree, in ptk_kd_node form, so the
 * library's flattener can be compared node for node. */
typedef struct { ptk_kd_node* nodes; uint64_t ncap, n; uint32_t* prims; uint64_t pcap, np; uint32_t leaves, maxd; } dump_t;

static void kd_dump(const kd_node* k, dump_t* d)
{
    const uint64_t me = d->n++;
    if (k->depth > d->maxd) d->maxd = k->depth;
    const uint64_t first = d->np;
    if (k->ntris || (!k->left && !k->right)) d->leaves++;
    for (uint64_t i = 0; i < k->ntris; ++i) {
        if (d->prims && d->np < d->pcap) d->prims[d->np] = k->tris[i].id;
        d->np++;
    }
    if (k->left) kd_dump(k->left, d);
    if (k->right) kd_dump(k->right, d);
    if (d->nodes && me < d->ncap) {
        ptk_kd_node* o = &d->nodes[me];
        memcpy(o->bmin, k->bmin, 12);
        memcpy(o->bmax, k->bmax, 12);
        o->skip = (uint32_t)d->n;
        o->first_prim = (uint32_t)first;
        o->num_prims = (uint32_t)k->ntris;
        o->depth = k->depth;
    }
}

int orc_kd_export(const orc_scene* s, uint32_t mesh, ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap,
                  uint32_t* prims, uint64_t prims_cap)
{
    if (mesh >= s->num_meshes) return 1;
    const mesh_t* m = &s->meshes[mesh];
    dump_t d;
    memset(&d, 0, sizeof(d));
    d.nodes = nodes; d.ncap = nodes_cap; d.prims = prims; d.pcap = prims_cap;
    kd_dump(m->root, &d);
    if (info) {
        info->num_nodes = (uint32_t)d.n;
        info->num_leaves = d.leaves;
        info->num_leaf_prims = d.np;
        info->max_depth = d.maxd;
        memcpy(info->bmin, m->bmin, 12);
        memcpy(info->bmax, m->bmax, 12);
    }
    return 0;
}

/* Probes for unit parity tests. */
void orc_sample_image(const float* rgb, uint32_t w, uint32_t h, uint32_t mode, uint64_t n, const float* uv,
                      float* out)
{
    tex_t t = { w, h, (float*)rgb };
    for (uint64_t i = 0; i < n; ++i) {
        v2 q = { uv[2 * i], uv[2 * i + 1] };
        const v3 c = image_sample(&t, mode, q);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

void orc_sample_equirect(const float* rgb, uint32_t w, uint32_t h, uint32_t mode, uint64_t n, const float* dir,
                         float* out)
{
    tex_t t = { w, h, (float*)rgb };
    for (uint64_t i = 0; i < n; ++i) {
        const v3 c = equirect_sample(&t, mode, V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

void orc_hemisphere(uint64_t n, const float* normal, const float* u12, float* out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const v3 h = hemisphere_dir(V3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]), u12[2 * i], u12[2 * i + 1]);
        out[3 * i] = h.x; out[3 * i + 1] = h.y; out[3 * i + 2] = h.z;
    }
}

/* One level of pathtracer::trace_recursive (pathtracer.cpp:29-68) with the two hemisphere draws supplied: cast, then
 * for a hit the material lookups (:34-37), hemisphere / reflect / mix direction (:40-45), the next ray (:47-51) and
 * the factor mix(albedo, 1, metallic) * cos_theta * 2 of :60; for a miss the background (:67).  SoA arrays of n. */
void orc_shade_probe(const orc_scene* s, uint64_t n, const float* o, const float* d, const float* u12, int32_t* node_out,
                     float* emission, float* weight, float* next_o, float* next_d)
{
    for (uint64_t i = 0; i < n; ++i) {
        ray_t r;
        r.o = V3(o[i], o[n + i], o[2 * n + i]);
        r.d = V3(d[i], d[n + i], d[2 * n + i]);
        int32_t node;
        const hit_t res = scene_cast_ray(s, &r, &node, NULL, NULL);
        v3 e, w = V3(0.0f, 0.0f, 0.0f), no = r.o, nd = r.d;
        if (res.hit) {
            const ptk_node_desc* ndp = &s->nodes[node];
            const v3 albedo = sampler_sample(s, ndp->albedo, res.uv);
            e = sampler_sample(s, ndp->emission, res.uv);
            const float roughness = sampler_sample(s, ndp->roughness, res.uv).x;
            const float metallic = sampler_sample(s, ndp->metallic, res.uv).x;
            const v3 hemi_dir = hemisphere_dir(res.normal, u12[i], u12[n + i]);
            const v3 reflect_dir = sub3(r.d, scale3(scale3(res.normal, dot3(res.normal, r.d)), 2.0f));
            const v3 dir = normalize3(mix3(reflect_dir, hemi_dir, roughness));
            no = add3(res.position, scale3(dir, PTK_EPSILON));
            nd = dir;
            const float dn = dot3(dir, res.normal);
            const float cos_theta = (0.0f < dn) ? dn : 0.0f;
            w = scale3(scale3(mix3(albedo, V3(1.0f, 1.0f, 1.0f), metallic), cos_theta), 2.0f);
        } else {
            e = background_sample(s, r.d);
            node = -1;
        }
        node_out[i] = node;
        emission[i] = e.x; emission[n + i] = e.y; emission[2 * n + i] = e.z;
        weight[i] = w.x; weight[n + i] = w.y; weight[2 * n + i] = w.z;
        next_o[i] = no.x; next_o[n + i] = no.y; next_o[2 * n + i] = no.z;
        next_d[i] = nd.x; next_d[n + i] = nd.y; next_d[2 * n + i] = nd.z;
    }
}

/* CLI/main.cpp:91-103: tone map 1 - exp(-c), gamma pow(c, 1/2.2), truncating uint32(c * 255) per channel, alpha 255.
 * Also returns the pre-truncation products c * 255 (3 floats per pixel) so that a test can tell a last-place
 * difference of a device expf / powf at an integer boundary from a real one. */
void orc_tonemap(const float* rgb, uint64_t n, uint32_t* rgba, float* scaled)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t px = 255u << 24;
        for (int c = 0; c < 3; ++c) {
            float v = 1.0f - expf(-rgb[3 * i + c]);
            v = powf(v, 1.0f / 2.2f);
            if (scaled) scaled[3 * i + c] = v * 255;
            px |= ((uint32_t)(v * 255)) << (8 * c);
        }
        rgba[i] = px;
    }
}

/* rt::utility::debug_pathtracer::trace (PathtracingUtility/debug_pathtracer.cpp:8-32) summed over the jittered primary
 * rays of a sample range, like orc_render_samples (single thread).  aov: 0 albedo, 1 emission, 2 roughness,
 * 3 metallic, 4 normal (debug_pathtracer.h:16-23). */
void orc_render_aov(const orc_scene* s, uint32_t aov, uint32_t w, uint32_t h, float fov_y, uint64_t seed,
                    uint64_t sample_begin, uint32_t sample_count, float* sum_rgb)
{
    const view_t view = make_view(s, w, h, fov_y);
    rng_t g;
    g.mode = 0;
    for (uint64_t p = 0; p < (uint64_t)w * h; ++p) {
        v3 color = V3(0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < sample_count; ++k) {
            rng_begin_path(&g, seed, (uint32_t)p, sample_begin + k);
            const float u1 = rng_next(&g);
            const float u2 = rng_next(&g);
            const ray_t r = primary_ray(s, &view, w, h, (uint32_t)(p % w), (uint32_t)(p / w), u1, u2);
            int32_t node;
            const hit_t res = scene_cast_ray(s, &r, &node, NULL, NULL);
            v3 c;
            if (res.hit) {
                const ptk_node_desc* nd = &s->nodes[node];
                if (aov == 4) c = add3(scale3(res.normal, 0.5f), V3(0.5f, 0.5f, 0.5f));
                else c = sampler_sample(s, aov == 0 ? nd->albedo : aov == 1 ? nd->emission : aov == 2 ? nd->roughness : nd->metallic, res.uv);
            } else {
                c = background_sample(s, r.d);
            }
            color = add3(color, c);
        }
        sum_rgb[p * 3] = color.x; sum_rgb[p * 3 + 1] = color.y; sum_rgb[p * 3 + 2] = color.z;
    }
}

void orc_primary_rays(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t n, const uint32_t* xy,
                      const float* u12, float* o, float* d)
{
    const view_t view = make_view(s, w, h, fov_y);
    for (uint64_t i = 0; i < n; ++i) {
        const ray_t r = primary_ray(s, &view, w, h, xy[2 * i], xy[2 * i + 1], u12[2 * i], u12[2 * i + 1]);
        o[i] = r.o.x; o[n + i] = r.o.y; o[2 * n + i] = r.o.z;
        d[i] = r.d.x; d[n + i] = r.d.y; d[2 * n + i] = r.d.z;
    }
}

void orc_philox(uint64_t seed, uint32_t pixel, uint64_t sample, uint32_t block, uint32_t out[4])
{
    const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    const uint32_t ctr[4] = { pixel, (uint32_t)sample, (uint32_t)(sample >> 32), block };
    philox4x32_10(ctr, key, out);
}

/* First-bounce secondary rays of the Philox render (SURVEY §8d parity batch ii):
 * for path (pixel, sample) returns the reflected ray of pathtracer.cpp:47-51, or
 * valid=0 when the primary ray misses. */
void orc_secondary_rays(const orc_scene* s, uint32_t w, uint32_t h, float fov_y, uint64_t seed, uint64_t n,
                        const uint32_t* pixel, const uint64_t* sample, float* o, float* d, int32_t* valid)
{
    const view_t view = make_view(s, w, h, fov_y);
    rng_t g;
    g.mode = 0;
    for (uint64_t i = 0; i < n; ++i) {
        rng_begin_path(&g, seed, pixel[i], sample[i]);
        const float u1 = rng_next(&g), u2 = rng_next(&g);
        const ray_t r = primary_ray(s, &view, w, h, pixel[i] % w, pixel[i] / w, u1, u2);
        int32_t node;
        const hit_t res = scene_cast_ray(s, &r, &node, NULL, NULL);
        valid[i] = res.hit;
        ray_t rr = r;
        if (res.hit) {
            const ptk_node_desc* nd = &s->nodes[node];
            const float roughness = sampler_sample(s, nd->roughness, res.uv).x;
            const float a1 = rng_next(&g), a2 = rng_next(&g);
            const v3 hemi_dir = hemisphere_dir(res.normal, a1, a2);
            const v3 reflect_dir = sub3(r.d, scale3(scale3(res.normal, dot3(res.normal, r.d)), 2.0f));
            const v3 dir = normalize3(mix3(reflect_dir, hemi_dir, roughness));
            rr.o = add3(res.position, scale3(dir, PTK_EPSILON));
            rr.d = dir;
        }
        o[i] = rr.o.x; o[n + i] = rr.o.y; o[2 * n + i] = rr.o.z;
        d[i] = rr.d.x; d[n + i] = rr.d.y; d[2 * n + i] = rr.d.z;
    }
}

/* n-th (1-based) output of std::mt19937 seeded with `seed` — known-answer hook
 * (the C++ standard fixes the 10000th output of the default-seeded engine). */
uint32_t orc_mt19937_nth(uint32_t seed, uint32_t n)
{
    rng_t g;
    g.mode = 1;
    mt_seed(&g, seed);
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; ++i) v = mt_next_u32(&g);
    return v;
}

/* rng::next() of the mt19937 mode: libstdc++ uniform_real_distribution<float>. */
void orc_mt19937_floats(uint32_t seed, uint32_t n, float* out)
{
    rng_t g;
    g.mode = 1;
    mt_seed(&g, seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_next(&g);
}
