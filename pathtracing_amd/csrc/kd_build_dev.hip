// kd_build_dev.hip — the reference's KD-tree build (kd_tree_node ctor, reference
// src/Pathtracing/scene.cpp:214-286; bounds loop of mesh::compile, scene.cpp:123-137) on the GPU.
//
// Output is byte-identical to the host builder (kd_build.cpp): same topology, same boxes, same
// DFS-preorder numbering, same leaf-ordered primitive list.  The reference builds depth-first with one
// recursion per node; here the tree grows one LEVEL per pass, every node of the level in parallel:
//
//   k_split_value  one wave per node: the split is the fp32 running sum of all 3n vertex coordinates
//                  of the node's triangles IN ORDER, divided by float(3n) (scene.cpp:240-249).  fp32
//                  addition does not reassociate, so the chain stays sequential: the wave loads 256
//                  values at a time (coalesced, next block prefetched) and adds them in order straight
//                  from the registers, through DPP row shifts (chain_64): about 1.3 instructions per value.
//                  Only the top few levels have chains long enough to matter (15 M adds at the root of
//                  a 5 M-triangle mesh); below that thousands of chains run side by side.
//   k_classify     one thread per (node, triangle) entry: left if any vertex <= split, right if any
//                  vertex >= split (scene.cpp:257-265), as a packed pair of counters for one 64-bit scan.
//   k_decide       one thread per node: the "too many in common" rule in double (scene.cpp:271), the
//                  stop rule (scene.cpp:220); sizes of what the node hands to the next level.
//   k_scatter      stable partition: an entry's rank among its node's left (right) entries is the
//                  difference of two scan values, so the children keep the parent's triangle order
//                  exactly as the reference's push_back loops do.  Leaves park their list in a store.
//   k_children     child boxes = parent box cut at the split (scene.cpp:31-56), next level's node table.
//
// After the last level: subtree sizes bottom-up, preorder positions and first-primitive offsets
// top-down (left subtree first, as mesh::intersect_internal recurses, scene.cpp:188-212), triangle
// boxes for the pruned traversal (kd_build.cpp: fill_triangle_boxes), then one scatter into the
// ptk_kd_node array.  Scans are rocPRIM device scans; one small read-back per level sizes the next
// level's buffers.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "kd_build_dev.h"

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr float kBig = 3.402823466e+38f;

#define KD_HIP(call)                                                                                  \
    do {                                                                                              \
        const hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                       \
            if (err) *err = std::string("ptk_kd_build_dev: ") + #call + ": " + hipGetErrorString(e_); \
            return e_ == hipErrorOutOfMemory ? PTK_ERR_NOMEM : PTK_ERR_HIP;                           \
        }                                                                                             \
    } while (0)

// growable device array
template <typename T> struct DVec {
    T* p = nullptr;
    size_t cap = 0;
    ~DVec() { if (p) (void)hipFree(p); }
    T* release()
    {
        T* r = p;
        p = nullptr;
        cap = 0;
        return r;
    }
    // room for `count` elements; `keep` of the old ones survive
    hipError_t reserve(size_t count, size_t keep, hipStream_t s)
    {
        if (count <= cap) return hipSuccess;
        size_t ncap = cap ? cap : 1024;
        while (ncap < count) ncap += ncap / 2 + 1024;
        T* np = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&np), ncap * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep) {
            e = hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(np);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = ncap;
        return hipSuccess;
    }
};

struct Box {
    float lo[3], hi[3];
};

// what one node passes on: entries for the next level, primitives parked in the leaf store,
// child nodes, leaves
struct Quad {
    uint32_t ent, prims, nodes, leaves;
};
struct QuadAdd {
    __host__ __device__ Quad operator()(const Quad& a, const Quad& b) const
    {
        return Quad{ a.ent + b.ent, a.prims + b.prims, a.nodes + b.nodes, a.leaves + b.leaves };
    }
};

// ---- root: positions, bounds, level 0 ---------------------------------------------------------------

__global__ __launch_bounds__(256) void k_positions(const ptk_triangle* __restrict__ tris, uint32_t n, float* __restrict__ pos)
{
    // 9 floats per triangle: v0.xyz v1.xyz v2.xyz
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint64_t)n * 9) return;
    const uint32_t t = (uint32_t)(i / 9), r = (uint32_t)(i % 9);
    pos[i] = tris[t].v[r / 3].position[r % 3];
}

// mesh::compile bounds (scene.cpp:125-136): min / max over all vertices, NaN never wins a comparison
__global__ __launch_bounds__(256) void k_bounds_partial(const float* __restrict__ pos, uint32_t n, float* __restrict__ part)
{
    __shared__ float red[4][6];
    float lo[3] = { kBig, kBig, kBig }, hi[3] = { -kBig, -kBig, -kBig };
    // from +-FLT_MAX; k_bounds_final folds in the default box (0,0,0)-(0,0,0)
    for (uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x; v < (uint64_t)n * 3; v += (uint64_t)gridDim.x * 256)
        for (int a = 0; a < 3; ++a) {
            const float p = pos[v * 3 + a];
            lo[a] = fminf(lo[a], p);
            hi[a] = fmaxf(hi[a], p);
        }
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) {
            red[threadIdx.x >> 6][a] = lo[a];
            red[threadIdx.x >> 6][3 + a] = hi[a];
        }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        part[blockIdx.x * 6 + a] = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
        part[blockIdx.x * 6 + 3 + a] = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
    }
}

__global__ void k_bounds_final(const float* __restrict__ part, uint32_t blocks, Box* __restrict__ out)
{
    const int a = threadIdx.x;
    if (a >= 3) return;
    float lo = kBig, hi = -kBig;
    for (uint32_t b = 0; b < blocks; ++b) {
        lo = fminf(lo, part[b * 6 + a]);
        hi = fmaxf(hi, part[b * 6 + 3 + a]);
    }
    // the running box starts at (0,0,0)-(0,0,0): `p < bmin ? p : bmin`, `bmax < p ? p : bmax`
    // (compared as integers: the compiler turns the float form into a min, which keeps a -0.0; lo and hi
    // are never NaN here)
    const uint32_t lb = __float_as_uint(lo), hb = __float_as_uint(hi);
    out->lo[a] = __uint_as_float(lb > 0x80000000u ? lb : 0u);
    out->hi[a] = __uint_as_float((int32_t)hb > 0 ? hb : 0u);
}

__global__ __launch_bounds__(256) void k_root_entries(uint32_t n, const float* __restrict__ pos, uint32_t axis,
                                                      uint32_t* __restrict__ idx, uint32_t* __restrict__ owner, float* __restrict__ val)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    idx[i] = i;
    owner[i] = 0u;
    for (int v = 0; v < 3; ++v) val[(uint64_t)i * 3 + v] = pos[(uint64_t)i * 9 + v * 3 + axis];
}

__global__ void k_root_node(uint32_t n, const Box* __restrict__ box, uint32_t* __restrict__ start, uint32_t* __restrict__ count, Box* __restrict__ nbox)
{
    start[0] = 0u;
    count[0] = n;
    nbox[0] = *box;
}

// ---- one level ------------------------------------------------------------------------------------------

// One step of the sequential sum over the 64 values a wave holds one per lane (lane l = value l).
// Lane 15 of each 16-lane row adds its row's values in order: DPP row_shr:n lets lane 15 read lane
// 15 - n, and the shifted copies do not depend on the running sum (the compiler makes the fifteen of
// them once per register and reuses them for all four rows), so the chain itself is one v_add_f32 per
// value.  row_bcast:15 hands the sum to the next row, readlane 63 to the next register.  Other lanes
// compute values nobody reads.
template <int N> __device__ __forceinline__ float dpp_row_shr(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xF, 0xF, true));
}
__device__ __forceinline__ float chain_row(float acc, float x)
{
    acc += dpp_row_shr<15>(x); acc += dpp_row_shr<14>(x); acc += dpp_row_shr<13>(x); acc += dpp_row_shr<12>(x);
    acc += dpp_row_shr<11>(x); acc += dpp_row_shr<10>(x); acc += dpp_row_shr<9>(x);  acc += dpp_row_shr<8>(x);
    acc += dpp_row_shr<7>(x);  acc += dpp_row_shr<6>(x);  acc += dpp_row_shr<5>(x);  acc += dpp_row_shr<4>(x);
    acc += dpp_row_shr<3>(x);  acc += dpp_row_shr<2>(x);  acc += dpp_row_shr<1>(x);  acc += x;
    return acc;
}
template <int ROW_MASK> __device__ __forceinline__ float chain_next_row(float acc)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, acc), __builtin_bit_cast(int, acc),
                                                                 0x142 /* row_bcast:15 */, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float chain_64(float acc, float x)
{
    acc = chain_row(acc, x); // lane 15
    acc = chain_next_row<0x2>(acc);
    acc = chain_row(acc, x); // lane 31
    acc = chain_next_row<0x4>(acc);
    acc = chain_row(acc, x); // lane 47
    acc = chain_next_row<0x8>(acc);
    acc = chain_row(acc, x); // lane 63
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
}

__global__ __launch_bounds__(256) void k_split_value(uint32_t K, const uint32_t* __restrict__ start, const uint32_t* __restrict__ count,
                                                     const float* __restrict__ val, uint32_t depth, float* __restrict__ med,
                                                     uint32_t* __restrict__ stop)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + wave;
    if (k >= K) return;
    const uint32_t n = count[k];
    if (n <= 1u || depth == PTK_KD_MAX_DEPTH) { // scene.cpp:220
        if (lane == 0u) {
            stop[k] = 1u;
            med[k] = 0.0f;
        }
        return;
    }
    const float* __restrict__ v = val + 3ull * start[k];
    const uint64_t total = 3ull * n;
    float acc = 0.0f;
    // Values past the end read as +0: acc is never -0 (it starts at +0 and x + (-x) = +0), so acc + 0 = acc.
    // The loads are unconditional (clamped index, then a select): with branches around them the compiler
    // cannot count them and waits for the prefetch as well.
    const uint64_t last = total - 1u;
    auto fetch = [&](uint64_t i) {
        const float t = v[i < total ? i : last];
        return i < total ? t : 0.0f;
    };
    float x0 = fetch(lane), x1 = fetch(64u + lane), x2 = fetch(128u + lane), x3 = fetch(192u + lane);
    for (uint64_t base = 0; base < total; base += 256u) {
        const float c0 = x0, c1 = x1, c2 = x2, c3 = x3;
        const uint64_t nb = base + 256u; // the next block is on its way while this one is added
        x0 = fetch(nb + lane);
        x1 = fetch(nb + 64u + lane);
        x2 = fetch(nb + 128u + lane);
        x3 = fetch(nb + 192u + lane);
        const uint64_t left = total - base;
        acc = chain_64(acc, c0);
        if (left > 64u) acc = chain_64(acc, c1);
        if (left > 128u) acc = chain_64(acc, c2);
        if (left > 192u) acc = chain_64(acc, c3);
    }
    if (lane == 0u) {
        stop[k] = 0u;
        med[k] = acc / static_cast<float>(total); // scene.cpp:249
    }
}

__global__ __launch_bounds__(256) void k_classify(uint32_t E, const uint32_t* __restrict__ owner, const float* __restrict__ val,
                                                  const float* __restrict__ med, const uint32_t* __restrict__ stop,
                                                  unsigned long long* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > E) return;
    unsigned long long f = 0ull;
    if (i < E) {
        const uint32_t k = owner[i];
        if (!stop[k]) {
            const float m = med[k];
            const float p0 = val[(uint64_t)i * 3], p1 = val[(uint64_t)i * 3 + 1], p2 = val[(uint64_t)i * 3 + 2];
            if (p0 <= m || p1 <= m || p2 <= m) f |= 1ull;       // scene.cpp:257
            if (p0 >= m || p1 >= m || p2 >= m) f |= 1ull << 32; // scene.cpp:262
        }
    }
    flags[i] = f; // flags[E] = 0: the exclusive scan's last element is the total
}

__global__ __launch_bounds__(256) void k_decide(uint32_t K, const uint32_t* __restrict__ start, const uint32_t* __restrict__ count,
                                                const unsigned long long* __restrict__ sflags, const uint32_t* __restrict__ stop,
                                                uint32_t* __restrict__ is_leaf, uint32_t* __restrict__ nl, uint32_t* __restrict__ nr,
                                                Quad* __restrict__ give)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k > K) return;
    Quad g = { 0u, 0u, 0u, 0u };
    if (k < K) {
        const uint32_t s = start[k], n = count[k];
        bool leaf = stop[k] != 0u;
        uint32_t l = 0u, r = 0u;
        if (!leaf) {
            const unsigned long long d = sflags[s + n] - sflags[s]; // both halves stay below 2^31
            l = (uint32_t)d;
            r = (uint32_t)(d >> 32);
            leaf = static_cast<double>((unsigned long long)l + r) > 1.5 * static_cast<double>(n); // scene.cpp:271
        }
        is_leaf[k] = leaf ? 1u : 0u;
        nl[k] = leaf ? 0u : l;
        nr[k] = leaf ? 0u : r;
        if (leaf)
            g = Quad{ 0u, n, 0u, 1u };
        else
            g = Quad{ l + r, 0u, (l ? 1u : 0u) + (r ? 1u : 0u), 0u };
    }
    give[k] = g; // give[K] = 0
}

__global__ __launch_bounds__(256) void k_scatter(uint32_t E, uint32_t next_axis, uint32_t bfs_base, const float* __restrict__ pos,
                                                 const uint32_t* __restrict__ idx, const uint32_t* __restrict__ owner,
                                                 const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ sflags,
                                                 const uint32_t* __restrict__ start, const uint32_t* __restrict__ is_leaf,
                                                 const uint32_t* __restrict__ nl, const Quad* __restrict__ off, uint32_t leaf_base,
                                                 uint32_t* __restrict__ idx_next, uint32_t* __restrict__ owner_next, float* __restrict__ val_next,
                                                 uint32_t* __restrict__ leaf_store, uint32_t* __restrict__ leaf_owner)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    const uint32_t k = owner[i], s = start[k], tri = idx[i];
    const Quad o = off[k];
    if (is_leaf[k]) {
        const uint32_t j = leaf_base + o.prims + (i - s);
        leaf_store[j] = tri;
        leaf_owner[j] = bfs_base + k;
        return;
    }
    const unsigned long long f = flags[i];
    if (f == 0ull) return; // NaN coordinates: in neither child
    const unsigned long long d = sflags[i] - sflags[s];
    const float p0 = pos[(uint64_t)tri * 9 + next_axis], p1 = pos[(uint64_t)tri * 9 + 3 + next_axis], p2 = pos[(uint64_t)tri * 9 + 6 + next_axis];
    const uint32_t l = nl[k];
    if (f & 1ull) {
        const uint32_t j = o.ent + (uint32_t)d;
        idx_next[j] = tri;
        owner_next[j] = o.nodes;
        val_next[(uint64_t)j * 3] = p0;
        val_next[(uint64_t)j * 3 + 1] = p1;
        val_next[(uint64_t)j * 3 + 2] = p2;
    }
    if (f >> 32) {
        const uint32_t j = o.ent + l + (uint32_t)(d >> 32);
        idx_next[j] = tri;
        owner_next[j] = o.nodes + (l ? 1u : 0u);
        val_next[(uint64_t)j * 3] = p0;
        val_next[(uint64_t)j * 3 + 1] = p1;
        val_next[(uint64_t)j * 3 + 2] = p2;
    }
}

// what is kept of every node until the tree is laid out
struct Kept {
    Box cut;            // the reference's cut box
    uint32_t left, right; // breadth-first ids, kNone if absent
    uint32_t leaf_off;  // leaves: offset in the leaf store
    uint32_t count;     // leaves: primitives
    uint32_t is_leaf;
};

__global__ __launch_bounds__(256) void k_children(uint32_t K, uint32_t axis, uint32_t bfs_base, uint32_t next_bfs_base, uint32_t leaf_base,
                                                  const uint32_t* __restrict__ count, const Box* __restrict__ box,
                                                  const float* __restrict__ med, const uint32_t* __restrict__ is_leaf,
                                                  const uint32_t* __restrict__ nl, const uint32_t* __restrict__ nr, const Quad* __restrict__ off,
                                                  Kept* __restrict__ kept, uint32_t* __restrict__ start_next, uint32_t* __restrict__ count_next,
                                                  Box* __restrict__ box_next)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const Quad o = off[k];
    const Box b = box[k];
    Kept kp;
    kp.cut = b;
    kp.left = kp.right = kNone;
    kp.leaf_off = leaf_base + o.prims;
    kp.is_leaf = is_leaf[k];
    kp.count = kp.is_leaf ? count[k] : 0u;
    if (!kp.is_leaf) {
        const uint32_t l = nl[k], r = nr[k];
        const float m = med[k];
        uint32_t c = o.nodes;
        if (l) { // scene.cpp:280 with bounding_box::split's lower half (scene.cpp:35-52)
            Box lb = b;
            lb.hi[axis] = m;
            start_next[c] = o.ent;
            count_next[c] = l;
            box_next[c] = lb;
            kp.left = next_bfs_base + c;
            ++c;
        }
        if (r) {
            Box rb = b;
            rb.lo[axis] = m;
            start_next[c] = o.ent + l;
            count_next[c] = r;
            box_next[c] = rb;
            kp.right = next_bfs_base + c;
        }
    }
    kept[bfs_base + k] = kp;
}

// ---- layout -------------------------------------------------------------------------------------------

// fill_triangle_boxes (kd_build.cpp), leaves: one wave per node
__global__ __launch_bounds__(256) void k_leaf_boxes(uint32_t B, const Kept* __restrict__ kept, const uint32_t* __restrict__ leaf_store,
                                                    const float* __restrict__ pos, Box* __restrict__ tbox)
{
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (b >= B) return;
    const Kept kp = kept[b];
    if (!kp.is_leaf) return;
    float lo[3] = { kBig, kBig, kBig }, hi[3] = { -kBig, -kBig, -kBig };
    for (uint32_t t = lane; t < kp.count; t += 64u) {
        const float* p = pos + (uint64_t)leaf_store[kp.leaf_off + t] * 9;
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], p[v * 3 + a]);
                hi[a] = fmaxf(hi[a], p[v * 3 + a]);
            }
    }
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    if (lane == 0u) {
        Box r;
        for (int a = 0; a < 3; ++a) {
            r.lo[a] = lo[a];
            r.hi[a] = hi[a];
        }
        tbox[b] = r;
    }
}

// bottom-up over one level: subtree node count, subtree primitive count, interior triangle boxes
__global__ __launch_bounds__(256) void k_sizes(uint32_t base, uint32_t K, const Kept* __restrict__ kept, uint32_t* __restrict__ size,
                                               uint32_t* __restrict__ sub_prims, Box* __restrict__ tbox)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const uint32_t b = base + k;
    const Kept kp = kept[b];
    if (kp.is_leaf) {
        size[b] = 1u;
        sub_prims[b] = kp.count;
        return;
    }
    uint32_t sz = 1u, pr = 0u;
    Box t;
    for (int a = 0; a < 3; ++a) {
        t.lo[a] = kBig;
        t.hi[a] = -kBig;
    }
    const uint32_t ch[2] = { kp.left, kp.right };
    for (int c = 0; c < 2; ++c)
        if (ch[c] != kNone) {
            sz += size[ch[c]];
            pr += sub_prims[ch[c]];
            const Box cb = tbox[ch[c]];
            for (int a = 0; a < 3; ++a) {
                t.lo[a] = fminf(t.lo[a], cb.lo[a]);
                t.hi[a] = fmaxf(t.hi[a], cb.hi[a]);
            }
        }
    size[b] = sz;
    sub_prims[b] = pr;
    tbox[b] = t;
}

// top-down over one level: preorder position and first primitive of the children
__global__ __launch_bounds__(256) void k_place(uint32_t base, uint32_t K, const Kept* __restrict__ kept, const uint32_t* __restrict__ size,
                                               const uint32_t* __restrict__ sub_prims, uint32_t* __restrict__ pre, uint32_t* __restrict__ first)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const uint32_t b = base + k;
    const Kept kp = kept[b];
    uint32_t p = pre[b] + 1u, f = first[b];
    if (kp.left != kNone) {
        pre[kp.left] = p;
        first[kp.left] = f;
        p += size[kp.left];
        f += sub_prims[kp.left];
    }
    if (kp.right != kNone) {
        pre[kp.right] = p;
        first[kp.right] = f;
    }
}

__global__ void k_place_root(uint32_t* __restrict__ pre, uint32_t* __restrict__ first)
{
    pre[0] = 0u;
    first[0] = 0u;
}

__global__ __launch_bounds__(256) void k_emit_nodes(uint32_t base, uint32_t K, uint32_t depth, const Kept* __restrict__ kept,
                                                    const uint32_t* __restrict__ size, const uint32_t* __restrict__ pre,
                                                    const uint32_t* __restrict__ first, const Box* __restrict__ tbox, ptk_kd_node* __restrict__ out)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const uint32_t b = base + k;
    const Kept kp = kept[b];
    ptk_kd_node n;
    for (int a = 0; a < 3; ++a) {
        n.bmin[a] = kp.cut.lo[a];
        n.bmax[a] = kp.cut.hi[a];
    }
    n.skip = pre[b] + size[b];
    n.first_prim = first[b];
    n.num_prims = kp.count;
    n.depth = depth;
    const Box t = tbox[b];
    for (int a = 0; a < 3; ++a) {
        float lo = t.lo[a], hi = t.hi[a];
        if (lo <= hi) { // kd_build.cpp: the padding of fill_triangle_boxes
            const float ext = hi - lo;
            const float mag = fmaxf(fabsf(lo), fabsf(hi));
            const float pad = 1e-4f * ext + 1e-5f * mag + 1e-30f;
            lo -= pad;
            hi += pad;
        }
        n.tmin[a] = lo;
        n.tmax[a] = hi;
    }
    out[pre[b]] = n;
}

__global__ __launch_bounds__(256) void k_emit_prims(uint32_t P, const uint32_t* __restrict__ leaf_store, const uint32_t* __restrict__ leaf_owner,
                                                    const Kept* __restrict__ kept, const uint32_t* __restrict__ first, uint32_t* __restrict__ out)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    const uint32_t b = leaf_owner[j];
    out[first[b] + (j - kept[b].leaf_off)] = leaf_store[j];
}

inline uint32_t blocks_for(uint64_t n, uint32_t per) { return (uint32_t)((n + per - 1) / per); }

} // namespace

int ptk_kd_build_device_keep(hipStream_t stream, const ptk_triangle* triangles, uint64_t num_triangles, const float* bounds,
                             uint32_t root_depth, PtkDeviceTree* out, std::string* err)
{
    if (num_triangles == 0 || num_triangles > 0xFFFFFFFFull / 4 || root_depth > PTK_KD_MAX_DEPTH) {
        if (err) *err = "ptk_kd_build_dev: triangle count or root depth out of range";
        return PTK_ERR_INVALID;
    }
    const uint32_t T = (uint32_t)num_triangles;
    const bool trace = std::getenv("PTK_KD_TRACE") != nullptr; // phase times on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_begin = now();
    DVec<ptk_triangle> d_tris;
    DVec<float> d_pos, d_part, d_val[2], d_med;
    DVec<Box> d_root_box, d_box[2], d_tbox;
    DVec<uint32_t> d_idx[2], d_owner[2], d_start[2], d_count[2], d_stop, d_is_leaf, d_nl, d_nr, d_leaf_store, d_leaf_owner;
    DVec<uint32_t> d_size, d_sub, d_pre, d_first, d_prims_out;
    DVec<unsigned long long> d_flags, d_sflags;
    DVec<Quad> d_give, d_off;
    DVec<Kept> d_kept;
    DVec<unsigned char> d_tmp;
    DVec<ptk_kd_node> d_nodes_out;

    KD_HIP(d_tris.reserve(T, 0, stream));
    KD_HIP(hipMemcpyAsync(d_tris.p, triangles, (size_t)T * sizeof(ptk_triangle), hipMemcpyHostToDevice, stream));
    KD_HIP(d_pos.reserve((size_t)T * 9, 0, stream));
    k_positions<<<blocks_for((uint64_t)T * 9, 256), 256, 0, stream>>>(d_tris.p, T, d_pos.p);
    KD_HIP(d_root_box.reserve(1, 0, stream));
    if (bounds) {
        KD_HIP(hipMemcpyAsync(d_root_box.p, bounds, sizeof(Box), hipMemcpyHostToDevice, stream));
    } else {
        const uint32_t pb = 1024;
        KD_HIP(d_part.reserve((size_t)pb * 6, 0, stream));
        k_bounds_partial<<<pb, 256, 0, stream>>>(d_pos.p, T, d_part.p);
        k_bounds_final<<<1, 64, 0, stream>>>(d_part.p, pb, d_root_box.p);
    }

    if (trace) (void)hipStreamSynchronize(stream);
    const auto t_uploaded = now();
    int cur = 0;
    uint32_t E = T, K = 1, B = 0, leaf_total = 0, leaves = 0;
    std::vector<uint32_t> level_base, level_count;
    KD_HIP(d_idx[0].reserve(E, 0, stream));
    KD_HIP(d_owner[0].reserve(E, 0, stream));
    KD_HIP(d_val[0].reserve((size_t)E * 3, 0, stream));
    KD_HIP(d_start[0].reserve(1, 0, stream));
    KD_HIP(d_count[0].reserve(1, 0, stream));
    KD_HIP(d_box[0].reserve(1, 0, stream));
    k_root_entries<<<blocks_for(E, 256), 256, 0, stream>>>(T, d_pos.p, root_depth % 3u, d_idx[0].p, d_owner[0].p, d_val[0].p);
    k_root_node<<<1, 1, 0, stream>>>(T, d_root_box.p, d_start[0].p, d_count[0].p, d_box[0].p);

    for (uint32_t level = 0; K != 0; ++level) {
        const uint32_t depth = root_depth + level;
        const int nxt = cur ^ 1;
        KD_HIP(d_med.reserve(K, 0, stream));
        KD_HIP(d_stop.reserve(K, 0, stream));
        KD_HIP(d_is_leaf.reserve(K, 0, stream));
        KD_HIP(d_nl.reserve(K, 0, stream));
        KD_HIP(d_nr.reserve(K, 0, stream));
        KD_HIP(d_give.reserve((size_t)K + 1, 0, stream));
        KD_HIP(d_off.reserve((size_t)K + 1, 0, stream));
        KD_HIP(d_flags.reserve((size_t)E + 1, 0, stream));
        KD_HIP(d_sflags.reserve((size_t)E + 1, 0, stream));

        k_split_value<<<blocks_for(K, 4), 256, 0, stream>>>(K, d_start[cur].p, d_count[cur].p, d_val[cur].p, depth, d_med.p, d_stop.p);
        k_classify<<<blocks_for((uint64_t)E + 1, 256), 256, 0, stream>>>(E, d_owner[cur].p, d_val[cur].p, d_med.p, d_stop.p, d_flags.p);
        size_t tmp_bytes = 0;
        KD_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_flags.p, d_sflags.p, 0ull, (size_t)E + 1, rocprim::plus<unsigned long long>(), stream));
        KD_HIP(d_tmp.reserve(tmp_bytes, 0, stream));
        KD_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, d_flags.p, d_sflags.p, 0ull, (size_t)E + 1, rocprim::plus<unsigned long long>(), stream));
        k_decide<<<blocks_for((uint64_t)K + 1, 256), 256, 0, stream>>>(K, d_start[cur].p, d_count[cur].p, d_sflags.p, d_stop.p, d_is_leaf.p,
                                                                        d_nl.p, d_nr.p, d_give.p);
        KD_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_give.p, d_off.p, Quad{ 0u, 0u, 0u, 0u }, (size_t)K + 1, QuadAdd(), stream));
        KD_HIP(d_tmp.reserve(tmp_bytes, 0, stream));
        KD_HIP(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, d_give.p, d_off.p, Quad{ 0u, 0u, 0u, 0u }, (size_t)K + 1, QuadAdd(), stream));
        Quad tot;
        KD_HIP(hipMemcpyAsync(&tot, d_off.p + K, sizeof(Quad), hipMemcpyDeviceToHost, stream));
        KD_HIP(hipStreamSynchronize(stream));
        if ((uint64_t)leaf_total + tot.prims > 0x7FFFFFF0ull || (uint64_t)B + K + tot.nodes > 0xFFFFFFF0ull) {
            if (err) *err = "ptk_kd_build_dev: tree exceeds 32-bit table indices";
            return PTK_ERR_INVALID;
        }

        KD_HIP(d_idx[nxt].reserve(tot.ent, 0, stream));
        KD_HIP(d_owner[nxt].reserve(tot.ent, 0, stream));
        KD_HIP(d_val[nxt].reserve((size_t)tot.ent * 3, 0, stream));
        KD_HIP(d_start[nxt].reserve(tot.nodes, 0, stream));
        KD_HIP(d_count[nxt].reserve(tot.nodes, 0, stream));
        KD_HIP(d_box[nxt].reserve(tot.nodes, 0, stream));
        KD_HIP(d_leaf_store.reserve((size_t)leaf_total + tot.prims, leaf_total, stream));
        KD_HIP(d_leaf_owner.reserve((size_t)leaf_total + tot.prims, leaf_total, stream));
        KD_HIP(d_kept.reserve((size_t)B + K, B, stream));

        k_scatter<<<blocks_for(E, 256), 256, 0, stream>>>(E, (depth + 1u) % 3u, B, d_pos.p, d_idx[cur].p, d_owner[cur].p, d_flags.p, d_sflags.p,
                                                          d_start[cur].p, d_is_leaf.p, d_nl.p, d_off.p, leaf_total, d_idx[nxt].p,
                                                          d_owner[nxt].p, d_val[nxt].p, d_leaf_store.p, d_leaf_owner.p);
        k_children<<<blocks_for(K, 256), 256, 0, stream>>>(K, depth % 3u, B, B + K, leaf_total, d_count[cur].p, d_box[cur].p, d_med.p,
                                                           d_is_leaf.p, d_nl.p, d_nr.p, d_off.p, d_kept.p, d_start[nxt].p, d_count[nxt].p,
                                                           d_box[nxt].p);
        level_base.push_back(B);
        level_count.push_back(K);
        B += K;
        leaf_total += tot.prims;
        leaves += tot.leaves;
        E = tot.ent;
        K = tot.nodes;
        cur = nxt;
    }

    // layout
    const auto t_levels = now();
    const uint32_t levels = (uint32_t)level_base.size();
    KD_HIP(d_tbox.reserve(B, 0, stream));
    KD_HIP(d_size.reserve(B, 0, stream));
    KD_HIP(d_sub.reserve(B, 0, stream));
    KD_HIP(d_pre.reserve(B, 0, stream));
    KD_HIP(d_first.reserve(B, 0, stream));
    KD_HIP(d_nodes_out.reserve(B, 0, stream));
    KD_HIP(d_prims_out.reserve(leaf_total ? leaf_total : 1, 0, stream));
    k_leaf_boxes<<<blocks_for(B, 4), 256, 0, stream>>>(B, d_kept.p, d_leaf_store.p, d_pos.p, d_tbox.p);
    for (uint32_t l = levels; l-- > 0;)
        k_sizes<<<blocks_for(level_count[l], 256), 256, 0, stream>>>(level_base[l], level_count[l], d_kept.p, d_size.p, d_sub.p, d_tbox.p);
    k_place_root<<<1, 1, 0, stream>>>(d_pre.p, d_first.p);
    for (uint32_t l = 0; l < levels; ++l)
        k_place<<<blocks_for(level_count[l], 256), 256, 0, stream>>>(level_base[l], level_count[l], d_kept.p, d_size.p, d_sub.p, d_pre.p, d_first.p);
    for (uint32_t l = 0; l < levels; ++l)
        k_emit_nodes<<<blocks_for(level_count[l], 256), 256, 0, stream>>>(level_base[l], level_count[l], root_depth + l, d_kept.p, d_size.p,
                                                                          d_pre.p, d_first.p, d_tbox.p, d_nodes_out.p);
    if (leaf_total)
        k_emit_prims<<<blocks_for(leaf_total, 256), 256, 0, stream>>>(leaf_total, d_leaf_store.p, d_leaf_owner.p, d_kept.p, d_first.p, d_prims_out.p);
    KD_HIP(hipGetLastError());
    if (trace) (void)hipStreamSynchronize(stream);
    const auto t_layout = now();

    Box root_box;
    KD_HIP(hipMemcpyAsync(&root_box, d_root_box.p, sizeof(Box), hipMemcpyDeviceToHost, stream));
    KD_HIP(hipStreamSynchronize(stream));
    if (trace)
        std::fprintf(stderr, "ptk_kd_build_dev: %u triangles: upload+bounds %.1f ms, %u levels %.1f ms, layout %.1f ms\n", T,
                     ms(t_begin, t_uploaded), levels, ms(t_uploaded, t_levels), ms(t_levels, t_layout));
    out->info.num_nodes = B;
    out->info.num_leaves = leaves;
    out->info.num_leaf_prims = leaf_total;
    out->info.max_depth = root_depth + levels - 1u;
    std::memcpy(out->info.bmin, root_box.lo, 12);
    std::memcpy(out->info.bmax, root_box.hi, 12);
    out->num_triangles = T;
    out->tris = d_tris.release();
    out->nodes = d_nodes_out.release();
    out->prims = d_prims_out.release();
    return PTK_OK;
}

int ptk_kd_tree_download(hipStream_t stream, const PtkDeviceTree& tree, std::vector<ptk_kd_node>& nodes_out,
                         std::vector<uint32_t>& prims_out, std::string* err)
{
    nodes_out.resize(tree.info.num_nodes);
    prims_out.resize(tree.info.num_leaf_prims);
    KD_HIP(hipMemcpyAsync(nodes_out.data(), tree.nodes, nodes_out.size() * sizeof(ptk_kd_node), hipMemcpyDeviceToHost, stream));
    if (!prims_out.empty())
        KD_HIP(hipMemcpyAsync(prims_out.data(), tree.prims, prims_out.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    KD_HIP(hipStreamSynchronize(stream));
    return PTK_OK;
}

int ptk_kd_build_device(hipStream_t stream, const ptk_triangle* triangles, uint64_t num_triangles, const float* bounds,
                        uint32_t root_depth, ptk_kd_info* info, std::vector<ptk_kd_node>& nodes_out,
                        std::vector<uint32_t>& prims_out, std::string* err)
{
    PtkDeviceTree tree;
    int r = ptk_kd_build_device_keep(stream, triangles, num_triangles, bounds, root_depth, &tree, err);
    if (r == PTK_OK) r = ptk_kd_tree_download(stream, tree, nodes_out, prims_out, err);
    if (r == PTK_OK && info) *info = tree.info;
    tree.release();
    return r;
}
