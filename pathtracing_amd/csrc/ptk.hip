// ptk.hip — gfx950 kernels and the C ABI of include/ptk.h.
//
// Pipeline (wavefront path tracing; reference loop: abstract_pathtracer.cpp:63-90
// + pathtracer.cpp:14-69, paths relative to /root/reference/src/Pathtracing):
//
//   k_bounce<true>   bounce 0: generate the jittered primary ray of every
//                    (pixel, sample) of the chunk, cast it, shade, emit the next ray
//   k_bounce<false>  bounces 1..4: read the surviving rays of a 256-slot queue
//                    segment, cast, shade, re-compact survivors inside the segment
//   k_accumulate     per-pixel sum of the chunk's per-sample radiance
//   k_tally          device path / ray counters
//
// Ray queue: SoA float arrays (origin, direction, throughput, radiance, path id),
// 256-slot segments, one workgroup per segment.  Survivors of a bounce are packed to
// the front of their own segment with wave64 ballot + mbcnt prefix and a 4-entry LDS
// scan across the workgroup's waves, in place: no global atomics, no ping-pong
// buffer, deterministic slot order.  A path that ends (miss, depth 5, zero
// throughput) writes its radiance once to sample_L[path id].
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kd_build_dev.h"
#include "ptk_device.h"

void ptk_kd_build_vectors(const ptk_triangle* triangles, uint64_t num_triangles, ptk_kd_info* info,
                          std::vector<ptk_kd_node>& nodes, std::vector<uint32_t>& prims);

using namespace ptk;

// ---- static scratch of the entry point that allocates nothing (ptk_render_samples_noalloc_dev) ----
// Ray queues, per-sample radiance, segment counters and the parked-ray records of one launch chain of at
// most PTK_NOALLOC_MAX_PATHS paths, laid out by NoallocLayout below.  Every GPU the library runs on holds
// its own copy until the process ends.  Host code never uses the array's name as an address: the per-device
// address comes from hipGetSymbolAddress (noalloc_scratch_for) and is what kernels and memsets receive.
constexpr uint64_t kNoallocPaths = PTK_NOALLOC_MAX_PATHS;
constexpr uint64_t kNoallocSegments = kNoallocPaths / 256;
constexpr uint64_t kNoallocWordsPerPath = 13 + 13 + 3 + 1 + 2 + 9; // two queues, sample_L, pend_slot, pend_mask, pend_best
#ifndef PTK_WALK_HEADS
#define PTK_WALK_HEADS 32
#endif
#ifndef PTK_OVERLAP_MIN_PATHS
#define PTK_OVERLAP_MIN_PATHS (1ull << 20)
#endif
#ifndef PTK_WALK_BLOCK
#define PTK_WALK_BLOCK 64
#endif
#ifndef PTK_OVERLAP_FIRST_PCT
#define PTK_OVERLAP_FIRST_PCT 50
#endif
#ifndef PTK_OVERLAP_WALK_BLOCKS
#define PTK_OVERLAP_WALK_BLOCKS 0
#endif
#ifndef PTK_WALK_UNIT
#define PTK_WALK_UNIT 64u
#endif
#ifndef PTK_WALK_RAYS
#define PTK_WALK_RAYS 16ull
#endif
#ifndef PTK_WALK_GRABS
#define PTK_WALK_GRABS 96u
#endif
// the walker's head counters (SplitBufs::walk_head): kWalkHeads per bounce, each on a 128-byte line, then the planned chunk per bounce
constexpr uint32_t kWalkHeads = PTK_WALK_HEADS, kWalkHeadStride = 32u;
constexpr uint32_t kWalkUnit = PTK_WALK_UNIT, kWalkUnitsPerSegment = 256u / kWalkUnit; // the walker's work unit: this many slots of a segment's parked list
constexpr uint32_t kWalkPlanBase = PTK_MAX_DEPTH * kWalkHeads * kWalkHeadStride;
constexpr uint32_t kSplitCounters = kWalkPlanBase + PTK_MAX_DEPTH;
constexpr uint64_t kNoallocBytes =
    kNoallocPaths * kNoallocWordsPerPath * 4 + (kNoallocSegments * (PTK_MAX_DEPTH + 1) + kSplitCounters) * 4 + 8 * 256;
static __device__ __align__(256) unsigned char scratch[kNoallocBytes];

namespace {

constexpr int kBlock = 256;          // one queue segment = one workgroup = 4 waves
constexpr int kWalkBlock = PTK_WALK_BLOCK; // workgroup of the persistent walker: its waves share nothing, and a one-wave workgroup
                                           // hands its registers and LDS back the moment its wave ends (with PTK_OPT_OVERLAP the
                                           // other chain's workgroups move in: Mario +2 % over 256)
// The split-mode kernels (k_bounce_split, k_walk_plan, k_walk, k_shade_parked, k_batch_parked) hand out a segment's
// parked rays in quarters of 64 (unit = segment * 4 + quarter) and scan four wave counts per workgroup; the bounce
// kernels' LDS budget (WaveLds) is sized for four waves per workgroup.  Another segment size needs those revisited.
static_assert(kBlock == 256, "queue segment = 256 slots = four waves: quarter-segment walker units, 4-entry wave scans");
constexpr int kStateFields = 13;     // o(3) d(3) T(3) L(3) path id
constexpr uint64_t kDefaultMaxPaths = 1ull << 24;

// ------------------------------------------------------------------ kernels

// Outputs of the parity entry (ptk_intersect_batch*): SoA arrays of n, any may be NULL.
struct BatchIO {
    uint64_t n;
    const float* o;
    const float* d;
    int32_t* node;
    int32_t* prim;
    float* t;
    float* u;
    float* v;
    float* position;
    float* normal;
    float* uv;
};

__device__ __forceinline__ void write_batch_hit(const DScene& S, const BatchIO& B, uint64_t i, const Hit& h)
{
    const uint64_t n = B.n;
    V3 nw = mk(0.0f, 0.0f, 0.0f);
    float uvx = 0.0f, uvy = 0.0f, uu = 0.0f, vv = 0.0f;
    int prim_id = -1;
    const bool hit = h.node >= 0;
    if (hit) hit_attributes<true>(S, h, nw, uvx, uvy, prim_id, uu, vv);
    if (B.node) B.node[i] = h.node;
    if (B.prim) B.prim[i] = prim_id;
    if (B.t) B.t[i] = hit ? sqrtf(h.d0) : 0.0f;
    if (B.u) B.u[i] = uu;
    if (B.v) B.v[i] = vv;
    if (B.position) {
        B.position[i] = hit ? h.pw.x : 0.0f;
        B.position[n + i] = hit ? h.pw.y : 0.0f;
        B.position[2 * n + i] = hit ? h.pw.z : 0.0f;
    }
    if (B.normal) {
        B.normal[i] = nw.x;
        B.normal[n + i] = nw.y;
        B.normal[2 * n + i] = nw.z;
    }
    if (B.uv) {
        B.uv[i] = uvx;
        B.uv[n + i] = uvy;
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_intersect_batch(const DScene S, const BatchIO B)
{
    __shared__ WaveLds wave_lds[(MODE & 8) ? kBlock / 64 : 1];
    const uint64_t n = B.n;
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool on = i < n;
    const uint64_t j = on ? i : 0;
    const V3 ro = mk(B.o[j], B.o[n + j], B.o[2 * n + j]);
    const V3 rd = mk(B.d[j], B.d[n + j], B.d[2 * n + j]);
    Hit h;
    intersect_scene_mode<MODE>(S, ro, rd, on, h, &wave_lds[(MODE & 8) ? threadIdx.x >> 6 : 0]);
    if (on) write_batch_hit(S, B, i, h);
}

struct BounceArgs {
    float* state;        // kStateFields arrays of `cap` floats
    float* sample_L;     // 3 arrays of `cap` floats, indexed by path id
    uint32_t* seg_count; // [PTK_MAX_DEPTH][num_segments]; row b = live rays entering bounce b
    uint64_t cap;
    uint32_t num_segments;
    uint32_t n_paths;       // paths in this chunk (<= cap)
    uint32_t samples;       // samples per pixel in this chunk
    uint64_t pixel_begin;   // first pixel of the chunk
    uint64_t sample_begin;  // first global sample index of the chunk
    uint64_t seed;
    uint32_t bounce;
    // PTK_OPT_COMPACT across segments (k_bounce only): survivors are appended to the other queue, packed per bucket
    // (bucket = segment % 64, whose k-th packed segment is segment k * 64 + bucket) through one atomic per workgroup
    // on the bucket's counter; a bounce reads its segment's fill from that counter instead of seg_count.
    float* state_other;   // the second queue; bounce b reads queue b % 2 and writes queue (b + 1) % 2
    uint32_t* bucket;     // [PTK_MAX_DEPTH + 1][kBuckets] rays entering bounce b per bucket; zeroed per chunk
    uint32_t dense;
};
constexpr uint32_t kBuckets = 64u;

// pathtracer::trace_recursive for one hit (pathtracer.cpp:34-60) with the two hemisphere draws supplied: samples the
// node's material at the hit, returns its emission, the factor mix(albedo, 1, metallic) * cos_theta * 2 the
// radiance of the next ray is weighted with, and that next ray (position + dir * epsilon, dir).
__device__ __forceinline__ void shade_hit(const DScene& S, const Hit& h, V3 d, float u1, float u2, V3& emission, V3& weight,
                                          V3& o_next, V3& d_next)
{
    V3 n;
    float uvx, uvy, uu, vv;
    int prim_id;
    hit_attributes<false>(S, h, n, uvx, uvy, prim_id, uu, vv);
    // pathtracer.cpp:34-37
    const DSampler* __restrict__ ms = S.node_samplers + (size_t)h.node * 4;
    const DSampler s_albedo = ms[0], s_emission = ms[1], s_roughness = ms[2], s_metallic = ms[3];
    const V3 albedo = sampler_sample(S, s_albedo, uvx, uvy);
    emission = sampler_sample(S, s_emission, uvx, uvy);
    const float roughness = sampler_sample(S, s_roughness, uvx, uvy).x;
    const float metallic = sampler_sample(S, s_metallic, uvx, uvy).x;
    // pathtracer.cpp:40-54
    const V3 hemi = hemisphere_dir(n, u1, u2);
    const V3 refl = sh_madd(n, -2.0f * sh_dot(n, d), d);
    const V3 dir = sh_normalize(sh_mix(refl, hemi, roughness));
    const float dn = sh_dot(dir, n);
    const float cos_theta = (0.0f < dn) ? dn : 0.0f;
    // pathtracer.cpp:60
    const V3 m = sh_mix(albedo, mk(1.0f, 1.0f, 1.0f), metallic);
    weight = m * (cos_theta * 2.0f);
    o_next = sh_madd(dir, PTK_EPSILON, h.pw);
    d_next = dir;
}

// pathtracer::trace_recursive for one cast (pathtracer.cpp:29-68) in throughput form: updates L, and
// for a hit before the last bounce turns (o, d, T) into the next ray.  Returns whether the path goes
// on; a path that ends (miss, depth 5, zero throughput) writes its radiance to sample_L[path].
// RNG: a Philox block holds the draws of two consecutive dimension pairs (rng_pair).  A kernel that keeps a path over
// several bounces (k_bounce_fused) hands in the second half of the block it drew last (`carry`) and gets the second half
// of a newly drawn block back: three blocks per path instead of five.  The values are the same either way.
struct RngCarry {
    uint32_t z, w;
};
template <bool CARRY = false>
__device__ __forceinline__ bool shade_ray(const DScene& S, const BounceArgs& A, const uint32_t bounce, const Hit& h, V3& o, V3& d,
                                          V3& T, V3& L, uint32_t path, RngCarry* carry = nullptr)
{
    bool alive = false;
    if (h.node < 0) {
        // pathtracer.cpp:67: miss -> background
        L = L + mulv(T, background_sample(S, d));
    } else {
        // the two hemisphere draws need no memory: Philox is placed ahead of the attribute / sampler loads so that it can
        // run under their round trips
        float u1 = 0.0f, u2 = 0.0f;
        if (bounce + 1 < PTK_MAX_DEPTH) {
            const uint32_t pl = path / A.samples;
            const uint32_t sm = path - pl * A.samples;
            if (CARRY) {
                const uint32_t j = 1u + bounce; // wave-uniform in the kernels that carry
                if (j & 1u) {
                    u1 = u01(carry->z);
                    u2 = u01(carry->w);
                } else {
                    const uint64_t sample = A.sample_begin + sm;
                    const uint4 r = philox4x32_10((uint32_t)(A.pixel_begin + pl), (uint32_t)sample, (uint32_t)(sample >> 32), j >> 1,
                                                  (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
                    u1 = u01(r.x);
                    u2 = u01(r.y);
                    carry->z = r.z;
                    carry->w = r.w;
                }
            } else {
                rng_pair(A.seed, (uint32_t)(A.pixel_begin + pl), A.sample_begin + sm, 1u + bounce, u1, u2);
            }
        }
        V3 n;
        float uvx, uvy, uu, vv;
        int prim_id;
        hit_attributes<false>(S, h, n, uvx, uvy, prim_id, uu, vv);
        // pathtracer.cpp:34-37
        const DSampler* __restrict__ ms = S.node_samplers + (size_t)h.node * 4;
        const DSampler s_albedo = ms[0], s_emission = ms[1], s_roughness = ms[2], s_metallic = ms[3];
        const V3 albedo = sampler_sample(S, s_albedo, uvx, uvy);
        const V3 emission = sampler_sample(S, s_emission, uvx, uvy);
        const float roughness = sampler_sample(S, s_roughness, uvx, uvy).x;
        const float metallic = sampler_sample(S, s_metallic, uvx, uvy).x;
        L = mk(__builtin_fmaf(T.x, emission.x, L.x), __builtin_fmaf(T.y, emission.y, L.y), __builtin_fmaf(T.z, emission.z, L.z));
        if (bounce + 1 < PTK_MAX_DEPTH) {
            // pathtracer.cpp:40-54
            const V3 hemi = hemisphere_dir(n, u1, u2);
            const V3 refl = sh_madd(n, -2.0f * sh_dot(n, d), d);
            const V3 dir = sh_normalize(sh_mix(refl, hemi, roughness));
            const float dn = sh_dot(dir, n);
            const float cos_theta = (0.0f < dn) ? dn : 0.0f;
            // pathtracer.cpp:60 in throughput form
            const V3 m = sh_mix(albedo, mk(1.0f, 1.0f, 1.0f), metallic);
            T = mulv(T, m) * (cos_theta * 2.0f);
            o = sh_madd(dir, PTK_EPSILON, h.pw);
            d = dir;
            alive = (T.x != 0.0f) || (T.y != 0.0f) || (T.z != 0.0f);
        }
    }
    if (!alive) {
        float* __restrict__ sl = A.sample_L;
        sl[0 * A.cap + path] = L.x;
        sl[1 * A.cap + path] = L.y;
        sl[2 * A.cap + path] = L.z;
    }
    return alive;
}

// abstract_pathtracer.cpp:74-83: the jittered camera ray of path `path` of the chunk
__device__ __forceinline__ void primary_ray(const DScene& S, const DView& V, const BounceArgs& A, uint32_t path, V3& o, V3& d,
                                            RngCarry* carry = nullptr)
{
    const uint32_t pl = path / A.samples;
    const uint32_t sm = path - pl * A.samples;
    const uint32_t pixel = (uint32_t)(A.pixel_begin + pl);
    float u1, u2;
    if (carry) { // block 0: the jitter, and bounce 0's hemisphere draw for the caller to keep
        const uint64_t sample = A.sample_begin + sm;
        const uint4 r = philox4x32_10(pixel, (uint32_t)sample, (uint32_t)(sample >> 32), 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
        u1 = u01(r.x);
        u2 = u01(r.y);
        carry->z = r.z;
        carry->w = r.w;
    } else {
        rng_pair(A.seed, pixel, A.sample_begin + sm, 0u, u1, u2);
    }
    o = mk(S.cam_x, S.cam_y, S.cam_z);
    d = primary_dir(V, pixel % V.width, pixel / V.width, u1, u2);
}

template <bool FIRST, int MODE, bool DENSE = false>
// at most 128 VGPRs (4 waves per SIMD, what LDS allows anyway), except for the variants that keep a KD stack in registers
__global__ __launch_bounds__(kBlock, (MODE & 2) ? 1 : 4) void k_bounce(const DScene S, const DView V, const BounceArgs A)
{
    __shared__ uint32_t wave_count[kBlock / 64];
    __shared__ WaveLds wave_lds[(MODE & 8) ? kBlock / 64 : 1];
    [[maybe_unused]] const unsigned long long kc0 = PTK_CLOCK();
    const uint32_t seg = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = seg * kBlock + tid;
    uint32_t cnt;
    if (FIRST) {
        const uint32_t base = seg * kBlock;
        cnt = (A.n_paths > base) ? min((uint32_t)kBlock, A.n_paths - base) : 0u;
    } else if (DENSE) {
        const uint32_t in_bucket = A.bucket[A.bounce * kBuckets + (seg % kBuckets)];
        const uint32_t before = (seg / kBuckets) * kBlock; // slots of the bucket's earlier segments
        cnt = in_bucket > before ? min((uint32_t)kBlock, in_bucket - before) : 0u;
    } else {
        cnt = A.seg_count[(size_t)A.bounce * A.num_segments + seg];
    }
    if (cnt == 0u) {
        if (tid == 0 && A.bounce + 1 < PTK_MAX_DEPTH) A.seg_count[(size_t)(A.bounce + 1) * A.num_segments + seg] = 0u;
        return;
    }
    // A wave without a live ray skips the cast and the shading (a wave-uniform branch) but stays for the
    // workgroup barrier of the compaction below: every wave of the group reaches it.  Occupancy is bounded by
    // the workgroup's LDS, which is held until its last wave ends either way, so the idle wave costs nothing.
    // After bounce 0 of an open scene most segments are a quarter full.
    const uint32_t live_waves = (cnt + 63u) / 64u;
    const bool wave_live = (tid >> 6) < live_waves;
    const bool on = tid < cnt;
    const float* __restrict__ st = (DENSE && (A.bounce & 1u)) ? A.state_other : A.state;
    float* __restrict__ st_out = (DENSE && !(A.bounce & 1u)) ? A.state_other : A.state;
    const uint64_t cap = A.cap;

    V3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f), T = o, L = o;
    uint32_t path = 0u;
    bool alive = false;
    [[maybe_unused]] RngCarry carry{ 0u, 0u };
    if (wave_live) {
        if (FIRST) {
            path = slot;
            primary_ray(S, V, A, path, o, d, &carry); // block 0 also holds bounce 0's hemisphere draw
            T = mk(1.0f, 1.0f, 1.0f);
            L = mk(0.0f, 0.0f, 0.0f);
        } else {
            const uint32_t rs = on ? slot : seg * kBlock;
            o = mk(st[0 * cap + rs], st[1 * cap + rs], st[2 * cap + rs]);
            d = mk(st[3 * cap + rs], st[4 * cap + rs], st[5 * cap + rs]);
            T = mk(st[6 * cap + rs], st[7 * cap + rs], st[8 * cap + rs]);
            L = mk(st[9 * cap + rs], st[10 * cap + rs], st[11 * cap + rs]);
            path = __float_as_uint(st[12 * cap + rs]);
        }

        Hit h;
        intersect_scene_mode<MODE, FIRST>(S, o, d, on, h, &wave_lds[(MODE & 8) ? threadIdx.x >> 6 : 0]);

#ifdef PTK_EXP
        if (S.exp_stage != 0u) { // no shading: the hit goes out through the radiance slots so that the cast is not dead code
            if (on) {
                A.sample_L[0 * A.cap + path] = h.d0;
                A.sample_L[1 * A.cap + path] = __int_as_float(h.node);
                A.sample_L[2 * A.cap + path] = h.a0 + h.a1 + h.a2 + h.pw.x + h.pw.y + h.pw.z + __int_as_float(h.prim);
            }
        } else
#endif
        if (on) alive = FIRST ? shade_ray<true>(S, A, 0u, h, o, d, T, L, path, &carry) : shade_ray(S, A, A.bounce, h, o, d, T, L, path);
    }

    PTK_STAT(S, 12, PTK_CLOCK() - kc0);
    if (A.bounce + 1 >= PTK_MAX_DEPTH) return;

    // In-segment compaction: wave64 ballot + mbcnt prefix, then a scan over the
    // workgroup's four wave counts in LDS.
    const uint64_t mask = __builtin_amdgcn_ballot_w64(alive);
    const uint32_t lane_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t wave = tid >> 6;
    if ((tid & 63u) == 0u) wave_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads(); // also orders every wave's state reads before the in-place writes
    uint32_t offset = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) {
        const uint32_t c = wave_count[w]; // idle waves wrote 0
        offset += (w < wave) ? c : 0u;
        total += c;
    }
    if (tid == 0) A.seg_count[(size_t)(A.bounce + 1) * A.num_segments + seg] = total; // dense: for k_tally only
    uint32_t ws = seg * kBlock + offset + lane_rank;
    if (DENSE) {
        __shared__ uint32_t bucket_base;
        if (tid == 0) bucket_base = total ? atomicAdd(&A.bucket[(A.bounce + 1) * kBuckets + (seg % kBuckets)], total) : 0u;
        __syncthreads();
        const uint32_t p = bucket_base + offset + lane_rank; // position in the bucket's packed list
        ws = ((p / kBlock) * kBuckets + (seg % kBuckets)) * kBlock + (p % kBlock);
    }
    if (alive) {
        st_out[0 * cap + ws] = o.x;  st_out[1 * cap + ws] = o.y;  st_out[2 * cap + ws] = o.z;
        st_out[3 * cap + ws] = d.x;  st_out[4 * cap + ws] = d.y;  st_out[5 * cap + ws] = d.z;
        st_out[6 * cap + ws] = T.x;  st_out[7 * cap + ws] = T.y;  st_out[8 * cap + ws] = T.z;
        st_out[9 * cap + ws] = L.x;  st_out[10 * cap + ws] = L.y; st_out[11 * cap + ws] = L.z;
        st_out[12 * cap + ws] = __uint_as_float(path);
    }
}

// ---- closed scenes: all five bounces in one kernel ---------------------------------------------------------------
// In a closed scene (Room: the camera looks into a box) every path lives its five bounces, so the queue between bounces
// moves nothing around: a segment's 256 rays come back as the same 256 rays.  k_bounce_fused keeps them in registers
// instead: it generates the primary rays, then casts and shades bounce after bounce; the ray never goes to HBM (52 B
// written and 52 B read per ray and bounce otherwise), there is no compaction, no workgroup barrier, one launch instead of
// five.  A path that ends early (a miss, zero throughput) writes its radiance as everywhere else and its lane idles for the
// rest of the segment, which is why the host only picks this kernel once the counters of earlier launch chains have shown
// nearly five rays per path (PTK_OPT_FUSE).  Rays for k_tally: wave w of segment s leaves the number of rays it cast after
// bounce 0 in seg_count[1 + w][s] (the rows the per-bounce kernels use for their live counts).
template <int MODE>
__global__ __launch_bounds__(kBlock, 4) void k_bounce_fused(const DScene S, const DView V, const BounceArgs A)
{
    __shared__ WaveLds wave_lds[(MODE & 8) ? kBlock / 64 : 1];
    const uint32_t seg = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = seg * kBlock;
    const uint32_t cnt = (A.n_paths > base) ? min((uint32_t)kBlock, A.n_paths - base) : 0u;
    WaveLds* const lds = &wave_lds[(MODE & 8) ? tid >> 6 : 0];
    uint32_t rays = 0u; // casts of this wave after bounce 0 (wave-uniform)
    if ((tid & ~63u) < cnt) { // the wave has a path
        bool on = tid < cnt;
        const uint32_t path = base + tid;
        V3 o, d, T = mk(1.0f, 1.0f, 1.0f), L = mk(0.0f, 0.0f, 0.0f);
        RngCarry carry;
        primary_ray(S, V, A, on ? path : base, o, d, &carry);
        {
            Hit h;
            intersect_scene_mode<MODE, true>(S, o, d, on, h, lds);
            if (on) on = shade_ray<true>(S, A, 0u, h, o, d, T, L, path, &carry);
        }
#pragma unroll 1
        for (uint32_t b = 1u; b < PTK_MAX_DEPTH; ++b) {
            const uint64_t live = __builtin_amdgcn_ballot_w64(on);
            if (live == 0ull) break;
            rays += (uint32_t)__popcll(live);
            Hit h;
            intersect_scene_mode<MODE, false>(S, o, d, on, h, lds);
            if (on) on = shade_ray<true>(S, A, b, h, o, d, T, L, path, &carry);
        }
    }
    if ((tid & 63u) == 0u) A.seg_count[(size_t)(1u + (tid >> 6)) * A.num_segments + seg] = rays;
}

// ---- split mode: scenes with KD-trees -------------------------------------------------------
// A ray's KD walk takes anything from zero to a few hundred node steps; inside k_bounce the wave
// waits for its slowest lane.  Split mode (PTK_OPT_SPLIT) cuts a bounce in three kernels:
//   k_bounce_split  casts every ray against the spheres and one-leaf meshes (balanced traversal) and
//                   finds the KD-tree nodes it may hit (conservative cull).  Rays without such a node are
//                   shaded at once; the others are parked: (queue slot, tree bit mask, best hit so far),
//                   packed per 256-slot segment like the survivors (no global atomics).
//   k_walk          persistent waves take parked rays (a quarter segment at a time, one atomic per grab)
//                   and walk their trees.  A lane whose walk ends hands its hit back to the parked record;
//                   when 24 lanes are free the wave fetches new rays, so the walk loop keeps 28 or more
//                   lanes busy until the lists run dry ("replace terminated rays").
//   k_shade_parked  shades the parked rays with every lane busy and packs the survivors behind the ones
//                   k_bounce_split left in the segment of the next queue.
// Two queues alternate between bounces.  Everything is packed in a fixed order, so split mode is as
// deterministic as k_bounce.
struct SplitBufs {
    float* q[2];                   // ray queues: kStateFields arrays of `cap` floats each, 256-slot segments
    uint32_t* seg_count;           // [PTK_MAX_DEPTH][num_segments] live rays of a segment entering bounce b
    uint32_t* seg_pend;            // [num_segments] parked rays of a segment (current bounce)
    uint32_t* walk_head;           // [kSplitCounters]: per bounce kWalkHeads counters of chunks handed out (a cache line each), then
                                   // at [kWalkPlanBase + b] the units per chunk (k_walk_plan)
    uint32_t* pend_slot;           // per parked ray (segment * 256 + i): its queue slot (render) / ray index (batch)
    unsigned long long* pend_mask; //   its tree nodes: bits over DScene::tree_nodes
    float* pend_best;              //   9 arrays of `cap`: d0, node, prim, a0..a2, pw of its best hit so far
    uint64_t cap;
    uint32_t num_segments;
    uint32_t walk_chunk;           // quarter-segment units the walker takes per atomic (host: ~96 grabs per wave)
};

__device__ __forceinline__ void queue_store(float* __restrict__ q, uint64_t cap, uint32_t ws, V3 o, V3 d, V3 T, V3 L,
                                            uint32_t path)
{
    q[0 * cap + ws] = o.x;  q[1 * cap + ws] = o.y;  q[2 * cap + ws] = o.z;
    q[3 * cap + ws] = d.x;  q[4 * cap + ws] = d.y;  q[5 * cap + ws] = d.z;
    q[6 * cap + ws] = T.x;  q[7 * cap + ws] = T.y;  q[8 * cap + ws] = T.z;
    q[9 * cap + ws] = L.x;  q[10 * cap + ws] = L.y; q[11 * cap + ws] = L.z;
    q[12 * cap + ws] = __uint_as_float(path);
}

__device__ __forceinline__ void park_ray(const SplitBufs& Q, uint32_t p, uint32_t slot, unsigned long long mask,
                                         const Hit& h)
{
    const uint64_t cap = Q.cap;
    Q.pend_slot[p] = slot;
    Q.pend_mask[p] = mask;
    float* __restrict__ b = Q.pend_best;
    b[0 * cap + p] = h.d0;
    b[1 * cap + p] = __int_as_float(h.node);
    b[2 * cap + p] = __int_as_float(h.prim);
    b[3 * cap + p] = h.a0;   b[4 * cap + p] = h.a1;   b[5 * cap + p] = h.a2;
    b[6 * cap + p] = h.pw.x; b[7 * cap + p] = h.pw.y; b[8 * cap + p] = h.pw.z;
}

// Positions of the flagged lanes packed over the whole workgroup (wave64 ballot + mbcnt prefix, then a
// scan of the four wave counts through `counts`); needs one __syncthreads between fill and read.
__device__ __forceinline__ uint32_t block_rank_fill(bool flag, uint32_t* counts, uint32_t& lane_rank)
{
    const uint64_t mask = __builtin_amdgcn_ballot_w64(flag);
    lane_rank = mbcnt64(mask);
    if ((threadIdx.x & 63u) == 0u) counts[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    return (uint32_t)__popcll(mask);
}
__device__ __forceinline__ void block_rank_read(const uint32_t* counts, uint32_t& offset, uint32_t& total)
{
    const uint32_t wave = threadIdx.x >> 6;
    offset = total = 0u;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; ++w) {
        const uint32_t c = counts[w];
        offset += (w < wave) ? c : 0u;
        total += c;
    }
}

template <bool FIRST>
__global__ __launch_bounds__(kBlock) void k_bounce_split(const DScene S, const DView V, const BounceArgs A, const SplitBufs Q)
{
    __shared__ WaveLds wave_lds[kBlock / 64];
    __shared__ uint32_t alive_count[kBlock / 64], park_count[kBlock / 64];
    const uint32_t seg = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = seg * kBlock + tid;
    const bool last = A.bounce + 1 >= PTK_MAX_DEPTH;
    uint32_t cnt;
    if (FIRST) {
        const uint32_t base = seg * kBlock;
        cnt = (A.n_paths > base) ? min((uint32_t)kBlock, A.n_paths - base) : 0u;
    } else {
        cnt = Q.seg_count[(size_t)A.bounce * Q.num_segments + seg];
    }
    if (cnt == 0u) {
        if (tid == 0) {
            Q.seg_pend[seg] = 0u;
            if (!last) Q.seg_count[(size_t)(A.bounce + 1) * Q.num_segments + seg] = 0u;
        }
        return;
    }
    const bool on = tid < cnt;
    const uint64_t cap = A.cap;
    const float* __restrict__ qin = Q.q[A.bounce & 1u];
    float* __restrict__ qout = Q.q[(A.bounce + 1u) & 1u];

    V3 o, d, T, L;
    uint32_t path;
    [[maybe_unused]] RngCarry carry{ 0u, 0u };
    if (FIRST) {
        path = slot;
        primary_ray(S, V, A, path, o, d, &carry);
        T = mk(1.0f, 1.0f, 1.0f);
        L = mk(0.0f, 0.0f, 0.0f);
    } else {
        const uint32_t rs = on ? slot : seg * kBlock;
        o = mk(qin[0 * cap + rs], qin[1 * cap + rs], qin[2 * cap + rs]);
        d = mk(qin[3 * cap + rs], qin[4 * cap + rs], qin[5 * cap + rs]);
        T = mk(qin[6 * cap + rs], qin[7 * cap + rs], qin[8 * cap + rs]);
        L = mk(qin[9 * cap + rs], qin[10 * cap + rs], qin[11 * cap + rs]);
        path = __float_as_uint(qin[12 * cap + rs]);
    }

    Hit h;
    unsigned long long trees;
    intersect_scene_balanced<true, true, FIRST>(S, o, d, on, h, wave_lds[tid >> 6], trees);

    const bool parked = on && trees != 0ull;
    // bounce 0 has no input queue: the walker reads the generated ray from queue 0
    if (FIRST && parked) queue_store(Q.q[0], cap, slot, o, d, T, L, path);
    bool alive = false;
    if (on && !parked) alive = FIRST ? shade_ray<true>(S, A, 0u, h, o, d, T, L, path, &carry) : shade_ray(S, A, A.bounce, h, o, d, T, L, path);

    // parked rays and survivors are packed inside the segment (no global atomics, as in k_bounce)
    uint32_t park_rank, alive_rank, park_off, park_total, alive_off, alive_total;
    block_rank_fill(parked, park_count, park_rank);
    block_rank_fill(alive, alive_count, alive_rank);
    __syncthreads();
    block_rank_read(park_count, park_off, park_total);
    block_rank_read(alive_count, alive_off, alive_total);
    if (tid == 0) {
        Q.seg_pend[seg] = park_total;
        if (!last) Q.seg_count[(size_t)(A.bounce + 1) * Q.num_segments + seg] = alive_total; // the walker adds its own
    }
    if (parked) park_ray(Q, seg * kBlock + park_off + park_rank, slot, trees, h);
    if (alive) queue_store(qout, cap, seg * kBlock + alive_off + alive_rank, o, d, T, L, path);
}

// Parity-entry twin of k_bounce_split: rays from the caller's arrays, hits written out instead of shaded.
__global__ __launch_bounds__(kBlock) void k_intersect_batch_split(const DScene S, const BatchIO B, const SplitBufs Q)
{
    __shared__ WaveLds wave_lds[kBlock / 64];
    __shared__ uint32_t park_count[kBlock / 64];
    const uint64_t n = B.n;
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool on = i < n;
    const uint64_t j = on ? i : 0;
    const V3 ro = mk(B.o[j], B.o[n + j], B.o[2 * n + j]);
    const V3 rd = mk(B.d[j], B.d[n + j], B.d[2 * n + j]);
    Hit h;
    unsigned long long trees;
    intersect_scene_balanced<true, true>(S, ro, rd, on, h, wave_lds[threadIdx.x >> 6], trees);
    const bool parked = on && trees != 0ull;
    uint32_t park_rank, park_off, park_total;
    block_rank_fill(parked, park_count, park_rank);
    __syncthreads();
    block_rank_read(park_count, park_off, park_total);
    if (threadIdx.x == 0) Q.seg_pend[blockIdx.x] = park_total;
    if (parked) park_ray(Q, blockIdx.x * kBlock + park_off + park_rank, (uint32_t)i, trees, h);
    if (on && !parked) write_batch_hit(S, B, i, h);
}

// The walker.  Lane states: IDLE (no ray), START (has a ray, next tree node not begun), WALK, FIN (walk of
// one node ended, result not yet folded into the ray's best hit), DONE (all nodes walked, waiting to
// be shaded).  Every pass of the outer loop makes progress: with fewer than 32 lanes walking there
// are at least 33 lanes IDLE or DONE after the START/FIN step, so either 16 are IDLE (refill) or 17 are
// DONE (shade); once the list is dry the walk loop runs the remaining lanes to the end.
// How many units the walker takes per atomic, from what k_bounce_split parked.  Every grab is one atomic on
// one of kWalkHeads addresses, about 8 ns each back to back at one address: 65536 grabs were half a millisecond
// even when nearly nothing is parked (an open scene after bounce 0).  With P parked rays the walker aims at
// ~16 rays per grab (48 while there was one head: small grabs even the waves out, and spread over 32 heads they
// no longer queue), but no fewer than two grabs per wave and no more than the host's figure for full segments
// (base_chunk).
__global__ __launch_bounds__(1024) void k_walk_plan(const uint32_t* __restrict__ seg_pend, uint32_t num_segments, uint32_t base_chunk,
                                                   uint32_t waves, uint32_t* __restrict__ chunk_out)
{
    __shared__ unsigned long long part[16];
    unsigned long long sum = 0ull;
    for (uint32_t i = threadIdx.x; i < num_segments; i += 1024u) sum += seg_pend[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0u) {
        unsigned long long parked = 0ull;
        for (int w = 0; w < 16; ++w) parked += part[w];
        const unsigned long long units = (unsigned long long)num_segments * kWalkUnitsPerSegment;
        const unsigned long long most = units / (base_chunk ? base_chunk : 1u); // grabs at the host's chunk
        unsigned long long grabs = parked / PTK_WALK_RAYS;
        if (grabs < 2ull * waves) grabs = 2ull * waves;
        if (grabs > most) grabs = most;
        if (grabs < 1ull) grabs = 1ull;
        unsigned long long chunk = units / grabs;
        if (chunk < base_chunk) chunk = base_chunk;
        if (chunk >= kWalkUnitsPerSegment) chunk &= ~(unsigned long long)(kWalkUnitsPerSegment - 1u); // whole segments once a grab spans one
        if (chunk < 1ull) chunk = 1ull;
        *chunk_out = (uint32_t)(chunk > 0x40000000ull ? 0x40000000ull : chunk);
    }
}

#ifndef PTK_REFILL_IDLE
#define PTK_REFILL_IDLE 24
#endif
#ifndef PTK_WALK_MIN
#define PTK_WALK_MIN 28
#endif
constexpr uint32_t kRefillIdle = PTK_REFILL_IDLE; // idle lanes that trigger a refill
constexpr uint32_t kWalkMin = PTK_WALK_MIN;       // walking lanes below which the walk loop yields to the refill

template <bool BATCH>
__global__ __launch_bounds__(kWalkBlock, 4) void k_walk(const DScene S, const uint32_t bounce_in, const SplitBufs Q, const BatchIO B)
{
    enum : uint32_t { IDLE = 0u, START = 1u, WALK = 2u, FIN = 3u, DONE = 4u };
    __shared__ WalkLds wave_lds[kWalkBlock / 64];
    const uint32_t lane = lane_id();
    const uint32_t bounce = BATCH ? 0u : bounce_in;
    const uint32_t nunits = Q.num_segments * kWalkUnitsPerSegment; // work units: kWalkUnit slots of a segment's parked list
    const uint32_t planned = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q.walk_head[kWalkPlanBase + bounce]);
    const uint32_t walk_chunk = planned ? planned : Q.walk_chunk; // k_walk_plan's figure for this bounce
    const uint64_t cap = Q.cap;
    const float* __restrict__ qin = Q.q[bounce & 1u];
    // Work is handed out in chunks of walk_chunk units by atomics.  One head counter would see every grab of all 4096 waves
    // (5-7 us per grab in the queue at that address, 6 % of the walker's time); chunk j belongs to head j % kWalkHeads
    // instead, each head on a cache line of its own, and a wave starts at the head of its workgroup's number and moves on
    // to the next head when one runs dry: after kWalkHeads dry heads the lists are empty.
    const uint32_t nchunks = (nunits + walk_chunk - 1u) / walk_chunk;
    uint32_t head = (blockIdx.x * (kWalkBlock / 64) + (threadIdx.x >> 6)) % kWalkHeads, heads_dry = 0u;
    uint32_t* __restrict__ const heads = Q.walk_head + bounce * kWalkHeads * kWalkHeadStride;

    WalkLds& my_lds = wave_lds[threadIdx.x >> 6];
#ifdef PTK_STATS
    const LdsStack stack{ &my_lds.stack_i[0][lane], &my_lds.stack_t[0][lane], S.stats };
#else
    const LdsStack stack{ &my_lds.stack_i[0][lane], &my_lds.stack_t[0][lane] };
#endif
    const int stack_cap = (int)S.walk_stack < kWalkStack ? (int)S.walk_stack : kWalkStack;
    KdWalk W;
    walk_begin(W, S, DMesh{ 0u, 0u, 0u, 0u }, mk(0.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 1.0f), false);
    uint32_t state = IDLE;
    // wave-uniform cursor over the parked lists: units come in chunks of walk_chunk per atomic
    bool dry = false;
    uint32_t next_unit = 0u, chunk_end = 0u, seg_cur = 0u, pos = 0u, cnt = 0u;
    uint32_t mine = 0u, node = 0u; // parked record of the lane's ray, tree node being walked
    unsigned long long trees = 0ull;
    V3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 1.0f);
    // The ray's best hit so far stays in its parked record (pend_best): a walk works in the node's local space
    // and never looks at it; it is read and, if beaten, rewritten when a tree node's walk ends (FIN).

    [[maybe_unused]] const unsigned long long clk_begin = PTK_CLOCK();
    [[maybe_unused]] unsigned long long acc_d = 0ull, acc_p = 0ull, acc_l = 0ull, acc_r = 0ull, acc_f = 0ull, acc_a = 0ull, acc_n = 0ull, clk_dry = 0ull; // -DPTK_STATS: clocks per kind of pass
    for (;;) {
        // refill
        [[maybe_unused]] const unsigned long long clk_r0 = PTK_CLOCK();
        const uint64_t idle = __builtin_amdgcn_ballot_w64(state == IDLE);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (!dry && nidle >= kRefillIdle) {
            uint64_t idle_left = idle;
            while (idle_left != 0ull) {
                while (!dry && pos >= cnt) { // on to the next unit (a quarter of a segment's parked list) with rays
                    if (next_unit >= chunk_end) {
                        uint32_t k = 0u;
                        [[maybe_unused]] const unsigned long long clk_a0 = PTK_CLOCK();
                        if (lane == 0u) k = atomicAdd(heads + head * kWalkHeadStride, 1u);
                        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                        acc_a += PTK_CLOCK() - clk_a0;
                        acc_n += 1ull;
                        const uint32_t chunk = k * kWalkHeads + head;
                        if (chunk >= nchunks) { // this head has nothing left
                            head = head + 1u == kWalkHeads ? 0u : head + 1u;
                            if (++heads_dry == kWalkHeads) {
                                dry = true;
                                clk_dry = PTK_CLOCK();
                                break;
                            }
                            continue;
                        }
                        const uint32_t base = chunk * walk_chunk;
                        next_unit = base;
                        chunk_end = base + walk_chunk < nunits ? base + walk_chunk : nunits;
                    }
                    const uint32_t unit = next_unit++;
                    seg_cur = unit / kWalkUnitsPerSegment;
                    const uint32_t in_seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)Q.seg_pend[seg_cur]);
                    pos = (unit % kWalkUnitsPerSegment) * kWalkUnit;
                    cnt = in_seg < pos + kWalkUnit ? in_seg : pos + kWalkUnit;
                    if (pos >= in_seg) next_unit = (unit | (kWalkUnitsPerSegment - 1u)) + 1u; // the segment's later units are empty too
                }
                if (dry) break;
                const uint32_t avail = cnt - pos, left = (uint32_t)__popcll(idle_left);
                const uint32_t take = left < avail ? left : avail;
                const bool mine_now = ((idle_left >> lane) & 1ull) != 0ull && mbcnt64(idle_left) < take;
                if (mine_now) {
                    mine = seg_cur * kBlock + pos + mbcnt64(idle_left);
                    const uint32_t slot = Q.pend_slot[mine];
                    trees = Q.pend_mask[mine];
                    if (BATCH) {
                        o = mk(B.o[slot], B.o[B.n + slot], B.o[2 * B.n + slot]);
                        d = mk(B.d[slot], B.d[B.n + slot], B.d[2 * B.n + slot]);
                    } else {
                        o = mk(qin[0 * cap + slot], qin[1 * cap + slot], qin[2 * cap + slot]);
                        d = mk(qin[3 * cap + slot], qin[4 * cap + slot], qin[5 * cap + slot]);
                    }
                    state = START;
                }
                PTK_STAT_FINE(S, 24, take);
                PTK_STAT_FINE(S, 25, 1);
                idle_left &= ~__builtin_amdgcn_ballot_w64(mine_now);
                pos += take;
            }
        }

        [[maybe_unused]] const unsigned long long clk_r1 = PTK_CLOCK();
        acc_r += clk_r1 - clk_r0;
        // FIN -> fold the node's result into the ray's best hit; START -> next tree node or DONE
        while (__builtin_amdgcn_ballot_w64(state == FIN || state == START) != 0ull) {
            const float4* __restrict__ nv = reinterpret_cast<const float4*>(S.nodes + node);
            if (state == FIN) {
                if (W.found) {
                    V3 pw;
                    float d0;
                    lane_hit_to_world(nv, o, W.p, pw, d0); // scene.cpp:375,383
                    // scene.cpp:386 with the node-order tie rule of best_update, against the parked record
                    float* __restrict__ pb = Q.pend_best;
                    const float bd0 = pb[0 * cap + mine];
                    const int bnode = __float_as_int(pb[1 * cap + mine]);
                    if (d0 < bd0 || (d0 == bd0 && (int)node < bnode)) {
                        pb[0 * cap + mine] = d0;
                        pb[1 * cap + mine] = __int_as_float((int)node);
                        pb[2 * cap + mine] = __int_as_float(W.best_slot);
                        pb[3 * cap + mine] = W.bar.x;
                        pb[4 * cap + mine] = W.bar.y;
                        pb[5 * cap + mine] = W.bar.z;
                        pb[6 * cap + mine] = pw.x;
                        pb[7 * cap + mine] = pw.y;
                        pb[8 * cap + mine] = pw.z;
                    }
                }
                state = START;
            }
            if (state == START) {
                if (trees == 0ull) {
                    state = DONE;
                } else {
                    const uint32_t k = (uint32_t)__builtin_ctzll(trees);
                    trees &= trees - 1ull;
                    node = S.tree_nodes[k];
                    const DNode* __restrict__ nd = S.nodes + node;
                    V3 lo, ld;
                    lane_ray_to_local(reinterpret_cast<const float4*>(nd), o, d, lo, ld); // scene.cpp:18-23
                    walk_begin(W, S, S.meshes[nd->mesh], lo, ld, true);
                    state = (W.cur != kWalkNone) ? WALK : START; // root missed: straight to the next node
                }
            }
        }

        // walk.  A walking lane is in one of three places: on an interior node (next: a descend step), on a
        // leaf (next: its triangles), or between nodes (next: a pop, which re-enters a node with six IEEE
        // divides).  Each pass serves the most populated place for all its lanes at once, so the pooled
        // leaf stage and the pops run for many lanes instead of being dragged along by every descend step.
        acc_f += PTK_CLOCK() - clk_r1;
        const uint32_t walk_min = dry ? 1u : kWalkMin;
        for (;;) {
            [[maybe_unused]] const unsigned long long clk_p0 = PTK_CLOCK();
            if (state == WALK && W.cur == kWalkNone) state = FIN;
            const bool walking = state == WALK;
            const bool at_pop = walking && W.cur == kWalkPop;
            const bool at_leaf = walking && !at_pop && (W.cb & kKdCountMask) != 0u;
            const bool at_node = walking && !at_pop && !at_leaf;
            // (the three head counts from the ballots of the plain compares, combined as 64-bit masks: a ballot of a derived
            // condition is compiled as condition -> 0 / 1 register -> compare)
            const uint64_t m_walk = __builtin_amdgcn_ballot_w64(walking), m_cpop = __builtin_amdgcn_ballot_w64(W.cur == kWalkPop);
            const uint64_t m_cleaf = __builtin_amdgcn_ballot_w64((W.cb & kKdCountMask) != 0u);
            const uint32_t n_node = (uint32_t)__popcll(m_walk & ~m_cpop & ~m_cleaf);
            const uint32_t n_leaf = (uint32_t)__popcll(m_walk & ~m_cpop & m_cleaf);
            const uint32_t n_pop = (uint32_t)__popcll(m_walk & m_cpop);
            const uint32_t n_walk = n_node + n_leaf + n_pop;
            if (n_walk < walk_min || n_walk == 0u) break;
            // lanes served per instruction spent: a descend step costs about 135 VALU instructions, a pop 90,
            // a leaf stage 250 or more
            if (2u * n_node >= n_leaf && 2u * n_node >= 3u * n_pop) {
                // a RUN of descend passes, in a loop of its own: as long as the choice stays the same the state stays in
                // the registers it is in (as one pass per trip of the walk loop every pass copies the state into the
                // registers the three kinds of pass share, and back)
                bool on_node = at_node;
                uint32_t nn = n_node;
                for (;;) {
                    PTK_STAT_FINE(S, 22, 1);
                    PTK_STAT_FINE(S, 23, nn);
                    if (on_node) walk_descend_step<true>(W, S, stack, stack_cap);
                    const uint64_t r_pop = __builtin_amdgcn_ballot_w64(W.cur == kWalkPop);
                    const uint64_t r_leaf = __builtin_amdgcn_ballot_w64((W.cb & kKdCountMask) != 0u);
                    nn = (uint32_t)__popcll(m_walk & ~r_pop & ~r_leaf);
                    const uint32_t nl = (uint32_t)__popcll(m_walk & ~r_pop & r_leaf), np = (uint32_t)__popcll(m_walk & r_pop);
                    if (nn == 0u || !(2u * nn >= nl && 2u * nn >= 3u * np)) break; // (a descend pass ends no walk: n_walk stands)
                    on_node = walking && W.cur != kWalkPop && (W.cb & kKdCountMask) == 0u;
                }
                acc_d += PTK_CLOCK() - clk_p0;
            } else if (3u * n_pop >= n_leaf) {
                PTK_STAT_FINE(S, 29, 1);
                PTK_STAT_FINE(S, 30, n_pop);
                if (at_pop) walk_pop(W, stack);
                acc_p += PTK_CLOCK() - clk_p0;
            } else {
                PTK_STAT_FINE(S, 26, 1);
                PTK_STAT_FINE(S, 27, n_leaf);
                walk_leaf_stage(W, S, at_leaf, my_lds);
                if (at_leaf && W.leaf_done >= (W.cb & kKdCountMask)) {
                    W.leaf_done = 0u;
#if defined(PTK_WALK_INLINE_POP) && PTK_WALK_INLINE_POP >= 2
                    walk_pop(W, stack);
#else
                    W.cur = kWalkPop;
#endif
                }
                acc_l += PTK_CLOCK() - clk_p0;
            }
        }

        // a ray that is through frees its lane (its parked record already holds its best hit; k_shade_parked /
        // k_batch_parked pick it up with every lane busy)
        if (state == DONE) state = IDLE;
        if (dry && __builtin_amdgcn_ballot_w64(state != IDLE) == 0ull) break;
    }
    PTK_STAT(S, 13, acc_d);
    PTK_STAT(S, 14, acc_p);
    PTK_STAT(S, 15, acc_l);
    PTK_STAT(S, 17, acc_r);
    PTK_STAT(S, 19, acc_f);
    PTK_STAT(S, 37, acc_a); // (22 / 23 are the descend passes' counters)
    PTK_STAT(S, 38, acc_n);
    PTK_STAT(S, 20, PTK_CLOCK() - clk_begin);
    PTK_STAT(S, 21, 1);
#if defined(PTK_STATS) && defined(PTK_STATS_COARSE)
    { // per bounce (stats hold one launch chain): clocks of every wave summed, their number, the slowest wave
        const unsigned long long t = PTK_CLOCK() - clk_begin;
        if (lane == 0u) {
            atomicAdd(S.stats + 32 + bounce, t);
            atomicAdd(S.stats + 40 + bounce, 1ull);
            atomicMax(S.stats + 48 + bounce, t);
            atomicAdd(S.stats + 56, PTK_CLOCK() - clk_dry); // from "the lists are dry" to the wave's end
            atomicMax(S.stats + 57 + bounce, PTK_CLOCK() - clk_dry);

        }
    }
#endif
}

// Shades the parked rays of a segment once the walker has finished their hits: dense again (one lane
// per parked ray), survivors go behind the ones k_bounce_split left in the segment of the next queue.
__global__ __launch_bounds__(kBlock) void k_shade_parked(const DScene S, const BounceArgs A, const SplitBufs Q)
{
    __shared__ uint32_t alive_count[kBlock / 64];
    const uint32_t seg = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t cnt = Q.seg_pend[seg];
    if (cnt == 0u) return;
    const bool on = tid < cnt;
    const bool last = A.bounce + 1 >= PTK_MAX_DEPTH;
    const uint64_t cap = A.cap;
    const float* __restrict__ qin = Q.q[A.bounce & 1u];
    float* __restrict__ qout = Q.q[(A.bounce + 1u) & 1u];
    const uint32_t p = seg * kBlock + (on ? tid : 0u);
    const uint32_t slot = Q.pend_slot[p];
    const float* __restrict__ pb = Q.pend_best;
    Hit h;
    h.d0 = pb[0 * cap + p];
    h.node = __float_as_int(pb[1 * cap + p]);
    h.prim = __float_as_int(pb[2 * cap + p]);
    h.a0 = pb[3 * cap + p];
    h.a1 = pb[4 * cap + p];
    h.a2 = pb[5 * cap + p];
    h.pw = mk(pb[6 * cap + p], pb[7 * cap + p], pb[8 * cap + p]);
    V3 o = mk(qin[0 * cap + slot], qin[1 * cap + slot], qin[2 * cap + slot]);
    V3 d = mk(qin[3 * cap + slot], qin[4 * cap + slot], qin[5 * cap + slot]);
    V3 T = mk(qin[6 * cap + slot], qin[7 * cap + slot], qin[8 * cap + slot]);
    V3 L = mk(qin[9 * cap + slot], qin[10 * cap + slot], qin[11 * cap + slot]);
    const uint32_t path = __float_as_uint(qin[12 * cap + slot]);
    bool alive = false;
    if (on) alive = shade_ray(S, A, A.bounce, h, o, d, T, L, path);
    if (last) return;
    uint32_t rank, off, total;
    block_rank_fill(alive, alive_count, rank);
    __syncthreads();
    block_rank_read(alive_count, off, total);
    uint32_t* __restrict__ next = Q.seg_count + (size_t)(A.bounce + 1) * Q.num_segments + seg;
    const uint32_t base = *next; // what k_bounce_split left; only this workgroup touches the segment now
    __syncthreads();
    if (tid == 0) *next = base + total;
    if (alive) queue_store(qout, cap, seg * kBlock + base + off + rank, o, d, T, L, path);
}

// parity-entry twin of k_shade_parked
__global__ __launch_bounds__(kBlock) void k_batch_parked(const DScene S, const BatchIO B, const SplitBufs Q)
{
    const uint32_t seg = blockIdx.x;
    if (threadIdx.x >= Q.seg_pend[seg]) return;
    const uint64_t cap = Q.cap;
    const uint32_t p = seg * kBlock + threadIdx.x;
    const float* __restrict__ pb = Q.pend_best;
    Hit h;
    h.d0 = pb[0 * cap + p];
    h.node = __float_as_int(pb[1 * cap + p]);
    h.prim = __float_as_int(pb[2 * cap + p]);
    h.a0 = pb[3 * cap + p];
    h.a1 = pb[4 * cap + p];
    h.a2 = pb[5 * cap + p];
    h.pw = mk(pb[6 * cap + p], pb[7 * cap + p], pb[8 * cap + p]);
    write_batch_hit(S, B, Q.pend_slot[p], h);
}

// abstract_pathtracer.cpp:70,85: color = sum over the pixel's samples.  One wave per
// pixel: lane l adds samples l, l+64, ... in order, then a fixed butterfly combines
// the 64 partial sums, so the result is deterministic.
__global__ __launch_bounds__(kBlock) void k_accumulate(const float* __restrict__ sample_L, uint64_t cap,
                                                       uint32_t num_pixels, uint32_t samples, uint64_t pixel_begin,
                                                       float* __restrict__ sum_rgb, int accumulate)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (wave >= num_pixels) return;
    const uint64_t base = wave * samples;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t s = lane; s < samples; s += 64u) {
        r += sample_L[0 * cap + base + s];
        g += sample_L[1 * cap + base + s];
        b += sample_L[2 * cap + base + s];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        r += __shfl_xor(r, off);
        g += __shfl_xor(g, off);
        b += __shfl_xor(b, off);
    }
    if (lane == 0) {
        float* o = sum_rgb + (pixel_begin + wave) * 3;
        if (accumulate) {
            o[0] += r; o[1] += g; o[2] += b;
        } else {
            o[0] = r; o[1] = g; o[2] = b;
        }
    }
}

// rays of a chunk = sum over bounces of the live-ray counts; one atomic per block.
__global__ __launch_bounds__(kBlock) void k_tally(const uint32_t* __restrict__ seg_count, uint32_t entries,
                                                  uint32_t n_paths, unsigned long long* __restrict__ counters)
{
    __shared__ unsigned long long part[kBlock / 64];
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < entries; i += gridDim.x * kBlock) acc += seg_count[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < kBlock / 64; ++w) tot += part[w];
        if (blockIdx.x == 0) {
            tot += n_paths; // bounce 0 casts one ray per path
            atomicAdd(&counters[0], (unsigned long long)n_paths);
        }
        atomicAdd(&counters[1], tot);
    }
}

// small counter arrays are cleared by a kernel, not by a memset node: the launch chain must replay inside a HIP graph
// (ptk_render_samples_noalloc_dev under capture), where a captured 40-byte memset of the static scratch did not run
__global__ __launch_bounds__(kBlock) void k_clear_u32(uint32_t* __restrict__ p, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// abstract_pathtracer.cpp:66,88-90
__global__ __launch_bounds__(kBlock) void k_resolve(const float* __restrict__ sum_rgb, float* __restrict__ image,
                                                    uint64_t n, float spi, float t)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float c = sum_rgb[i] / spi;
    image[i] = c * (1.0f - t) + image[i] * t;
}

// CLI/main.cpp:91-103
__global__ __launch_bounds__(kBlock) void k_tonemap(const float* __restrict__ image, uint32_t* __restrict__ rgba,
                                                    uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t px = 255u << 24;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = 1.0f - expf(-image[i * 3 + c]);
        v = powf(v, 1.0f / 2.2f);
        px |= ((uint32_t)(v * 255) & 0xFFu) << (8 * c);
    }
    rgba[i] = px;
}

// ---- probes: the device functions of the samplers, the hemisphere mapping and one shading step, run on caller
// data so that tests can hold each of them against the oracle directly (ptk_probe_*_dev) --------------------------
// image::sample (sampler.cpp:36-61) on texture `tex` of the uploaded scene; uv / rgb are SoA arrays of n
__global__ __launch_bounds__(kBlock) void k_probe_image(const DScene S, uint32_t tex, uint32_t mode, uint64_t n,
                                                        const float* __restrict__ uv, float* __restrict__ rgb)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const V3 c = image_sample(S.textures[tex], mode, uv[i], uv[n + i]);
    rgb[i] = c.x; rgb[n + i] = c.y; rgb[2 * n + i] = c.z;
}

// scene.background->sample(direction): color_sampler or equirectangular_map::sample (sampler.cpp:116-124)
__global__ __launch_bounds__(kBlock) void k_probe_background(const DScene S, uint64_t n, const float* __restrict__ dir,
                                                             float* __restrict__ rgb)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const V3 c = background_sample(S, mk(dir[i], dir[n + i], dir[2 * n + i]));
    rgb[i] = c.x; rgb[n + i] = c.y; rgb[2 * n + i] = c.z;
}

// rng::hemisphere (rng.cpp:12-36) with the two draws supplied
__global__ __launch_bounds__(kBlock) void k_probe_hemisphere(uint64_t n, const float* __restrict__ normal, const float* __restrict__ u12,
                                                             float* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const V3 h = hemisphere_dir(mk(normal[i], normal[n + i], normal[2 * n + i]), u12[i], u12[n + i]);
    out[i] = h.x; out[n + i] = h.y; out[2 * n + i] = h.z;
}

// One step of pathtracer::trace_recursive (pathtracer.cpp:29-68) for caller rays and caller draws: cast, then
// shade_hit (hit) or the background (miss).
struct ShadeProbeIO {
    uint64_t n;
    const float* o;
    const float* d;
    const float* u12;
    int32_t* node;
    float* emission;
    float* weight;
    float* next_o;
    float* next_d;
};
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_probe_shade(const DScene S, const ShadeProbeIO P)
{
    __shared__ WaveLds wave_lds[(MODE & 8) ? kBlock / 64 : 1];
    const uint64_t n = P.n;
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool on = i < n;
    const uint64_t j = on ? i : 0;
    const V3 ro = mk(P.o[j], P.o[n + j], P.o[2 * n + j]);
    const V3 rd = mk(P.d[j], P.d[n + j], P.d[2 * n + j]);
    Hit h;
    intersect_scene_mode<MODE>(S, ro, rd, on, h, &wave_lds[(MODE & 8) ? threadIdx.x >> 6 : 0]);
    if (!on) return;
    V3 e, w = mk(0.0f, 0.0f, 0.0f), no = ro, nd = rd;
    if (h.node < 0) e = background_sample(S, rd); // pathtracer.cpp:67
    else shade_hit(S, h, rd, P.u12[i], P.u12[n + i], e, w, no, nd);
    if (P.node) P.node[i] = h.node;
    if (P.emission) { P.emission[i] = e.x; P.emission[n + i] = e.y; P.emission[2 * n + i] = e.z; }
    if (P.weight) { P.weight[i] = w.x; P.weight[n + i] = w.y; P.weight[2 * n + i] = w.z; }
    if (P.next_o) { P.next_o[i] = no.x; P.next_o[n + i] = no.y; P.next_o[2 * n + i] = no.z; }
    if (P.next_d) { P.next_d[i] = nd.x; P.next_d[n + i] = nd.y; P.next_d[2 * n + i] = nd.z; }
}

// ---- debug AOVs: rt::utility::debug_pathtracer::trace (debug_pathtracer.cpp:8-32) for the jittered
// primary ray of every (pixel, sample) of a chunk: cast, then the chosen material channel at the hit, the shading
// normal mapped to [0, 1], or the background on a miss.  Same ray generation, cast and sample_L / k_accumulate path
// as bounce 0 of the integrator.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_aov(const DScene S, const DView V, const BounceArgs A, const uint32_t aov)
{
    __shared__ WaveLds wave_lds[(MODE & 8) ? kBlock / 64 : 1];
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const bool on = slot < A.n_paths;
    const uint32_t path = on ? slot : 0u;
    V3 o, d;
    primary_ray(S, V, A, path, o, d);
    Hit h;
    intersect_scene_mode<MODE, true>(S, o, d, on, h, &wave_lds[(MODE & 8) ? threadIdx.x >> 6 : 0]);
    if (!on) return;
    V3 c;
    if (h.node < 0) {
        c = background_sample(S, d); // debug_pathtracer.cpp:29
    } else {
        V3 n;
        float uvx, uvy, uu, vv;
        int prim_id;
        hit_attributes<true>(S, h, n, uvx, uvy, prim_id, uu, vv);
        if (aov == PTK_AOV_NORMAL) c = n * 0.5f + mk(0.5f, 0.5f, 0.5f); // debug_pathtracer.cpp:25
        else c = sampler_sample(S, S.node_samplers[(size_t)h.node * 4 + (aov == PTK_AOV_ALBEDO ? 0u : aov == PTK_AOV_EMISSION ? 1u : aov == PTK_AOV_ROUGHNESS ? 2u : 3u)], uvx, uvy);
    }
    A.sample_L[0 * A.cap + path] = c.x;
    A.sample_L[1 * A.cap + path] = c.y;
    A.sample_L[2 * A.cap + path] = c.z;
}

std::string g_create_error;

} // namespace

// ------------------------------------------------------------------ context

struct ptk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    std::atomic<int> interrupt{ 0 }; // ptk_request_interrupt: looked at by ptk_render_image between launch chains

    // scene tables in HBM
    std::vector<void*> allocs;      // scene allocations
    DScene S{};
    bool has_scene = false;
    uint32_t num_meshes = 0, num_samplers = 0, num_textures = 0;
    std::vector<ptk_kd_info> kd_infos;

    // render state
    float cam_dir[3] = { 0.0f, 0.0f, -1.0f }; // rt::camera::m_direction (scene.h:36)
    DView V{};
    float fov_y = 0.0f;
    bool has_view = false;
    uint64_t cap = 0;
    float* d_state = nullptr;
    float* d_sample_L = nullptr;
    uint32_t* d_seg_count = nullptr;
    uint32_t* d_bucket = nullptr;  // [PTK_MAX_DEPTH + 1][kBuckets], PTK_OPT_COMPACT across segments
    float* d_state_other = nullptr; // its second queue (allocated on first use)
    uint64_t other_cap = 0;
    unsigned long long* d_counters = nullptr;
    // PTK_OPT_COMPACT = 2: rays per path of the chunks rendered so far, from snapshots of d_counters that arrive in
    // pinned host memory behind the launches (never waited for: a snapshot that is not there yet is looked at later)
    unsigned long long* h_snapshot = nullptr; // pinned, 2 words
    hipEvent_t snapshot_ev = nullptr;
    bool snapshot_pending = false;
    unsigned long long seen_paths = 0, seen_rays = 0;
    double rays_per_path = -1.0; // < 0: not known yet

    int traversal = 2; // PTK_OPT_TRAVERSAL
    int kd_prune = 1;  // PTK_OPT_KD_PRUNE
    int node_cull = 2; // PTK_OPT_NODE_CULL
    int balance = 2;   // PTK_OPT_BALANCE
    int split = 2;     // PTK_OPT_SPLIT
    int device_kd = 2; // PTK_OPT_DEVICE_KD
    int walk_stack = 64; // PTK_OPT_WALK_STACK (clamped to the kernels' own capacity)
    int compact = 2;     // PTK_OPT_COMPACT
    int fuse = 2;        // PTK_OPT_FUSE
    int overlap = 1;     // PTK_OPT_OVERLAP
    hipStream_t side_stream = nullptr; // second launch chain of a split-mode call (PTK_OPT_OVERLAP)
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int exp_stage = 0;   // PTK_EXP builds: stage limit of bounce 2 (environment PTK_EXP_STAGE)
    uint32_t num_trees = 0; // nodes of class kClassTree

    // split mode (scenes with KD-trees): second ray queue, parked-ray list, counters
    float* d_state2 = nullptr;
    uint32_t* d_split_counters = nullptr; // walk_head[PTK_MAX_DEPTH]
    uint32_t* d_seg_pend = nullptr;       // [cap / 256]
    uint32_t* d_pend_slot = nullptr;
    unsigned long long* d_pend_mask = nullptr;
    float* d_pend_best = nullptr;
    uint64_t split_cap = 0;
    // the same for ptk_intersect_batch_dev (sized by the largest batch so far)
    uint32_t* d_bsplit_counters = nullptr; // walk_head
    uint32_t* d_bseg_pend = nullptr;
    uint32_t* d_bpend_slot = nullptr;
    unsigned long long* d_bpend_mask = nullptr;
    float* d_bpend_best = nullptr;
    uint64_t bsplit_cap = 0;
    bool deep_kd = false; // some mesh has more than one KD node (the pruned walk only pays off then)
    bool all_balls = false; // every node is a sphere whose record carries its bounding ball (kernel MODE bit 4)

    // optional per-kernel HIP-event timing (ptk_ctx_enable_timing)
    bool timing = false;
    bool timing_hold = false; // set while the no-alloc entry point runs (HIP events are allocations) or a call is not sampled
    uint32_t timing_every = 1, timing_tick = 0; // ptk_ctx_enable_timing(n): every n-th ptk_render_samples_dev call is timed
    struct Stamp {
        int kind;
        hipEvent_t a, b;
    };
    std::vector<Stamp> stamps;
};

namespace {

int fail(ptk_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define PTK_HIP(ctx, call)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail((ctx), PTK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)

void free_scene(ptk_ctx* c)
{
    for (void* p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    c->has_scene = false;
    c->kd_infos.clear();
}

void free_split(ptk_ctx* c)
{
    if (c->d_state2) (void)hipFree(c->d_state2);
    if (c->d_split_counters) (void)hipFree(c->d_split_counters);
    if (c->d_seg_pend) (void)hipFree(c->d_seg_pend);
    c->d_seg_pend = nullptr;
    if (c->d_pend_slot) (void)hipFree(c->d_pend_slot);
    if (c->d_pend_mask) (void)hipFree(c->d_pend_mask);
    if (c->d_pend_best) (void)hipFree(c->d_pend_best);
    c->d_state2 = c->d_pend_best = nullptr;
    c->d_split_counters = c->d_pend_slot = nullptr;
    c->d_pend_mask = nullptr;
    c->split_cap = 0;
}

void free_batch_split(ptk_ctx* c)
{
    if (c->d_bsplit_counters) (void)hipFree(c->d_bsplit_counters);
    if (c->d_bseg_pend) (void)hipFree(c->d_bseg_pend);
    c->d_bseg_pend = nullptr;
    if (c->d_bpend_slot) (void)hipFree(c->d_bpend_slot);
    if (c->d_bpend_mask) (void)hipFree(c->d_bpend_mask);
    if (c->d_bpend_best) (void)hipFree(c->d_bpend_best);
    c->d_bsplit_counters = c->d_bpend_slot = nullptr;
    c->d_bpend_mask = nullptr;
    c->d_bpend_best = nullptr;
    c->bsplit_cap = 0;
}

void free_render(ptk_ctx* c)
{
    free_split(c);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_sample_L) (void)hipFree(c->d_sample_L);
    if (c->d_seg_count) (void)hipFree(c->d_seg_count);
    if (c->d_bucket) (void)hipFree(c->d_bucket);
    if (c->d_state_other) (void)hipFree(c->d_state_other);
    c->d_state = c->d_sample_L = c->d_state_other = nullptr;
    c->d_seg_count = c->d_bucket = nullptr;
    c->other_cap = 0;
    c->cap = 0;
    c->has_view = false;
}

template <typename T>
int upload(ptk_ctx* c, const std::vector<T>& host, const T** out)
{
    const size_t bytes = (host.empty() ? 1 : host.size()) * sizeof(T);
    void* p = nullptr;
    PTK_HIP(c, hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    if (!host.empty()) PTK_HIP(c, hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return PTK_OK;
}

struct HV3 {
    float x, y, z;
};
// host and device: the build flags make both sides round every operation alike (no contraction,
// correctly rounded division and square root)
__host__ __device__ inline HV3 hsub(HV3 a, HV3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__host__ __device__ inline float hdot(HV3 a, HV3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline HV3 hcross(HV3 a, HV3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
__host__ __device__ inline HV3 hnormalize(HV3 a)
{
    const float s = 1.0f / sqrtf(hdot(a, a));
    return { a.x * s, a.y * s, a.z * s };
}

// abstract_pathtracer.cpp:20-25 (host fp32, same operand order as the oracle)
DView make_view(const float cam_dir[3], uint32_t width, uint32_t height, float fov_y)
{
    const HV3 fwd = hnormalize({ cam_dir[0], cam_dir[1], cam_dir[2] });
    const HV3 right = hnormalize(hcross(fwd, { 0.0f, 1.0f, 0.0f }));
    const HV3 up = hcross(right, fwd);
    DView V{};
    V.fx = fwd.x; V.fy = fwd.y; V.fz = fwd.z;
    V.rx = right.x; V.ry = right.y; V.rz = right.z;
    V.ux = up.x; V.uy = up.y; V.uz = up.z;
    V.h2 = std::atan(fov_y / 2.0f);
    V.w2 = V.h2 * (float)width / (float)height;
    V.width = width;
    V.height = height;
    return V;
}

// triangle::update, scene.cpp:91-107 -> 64-B hit record
__host__ __device__ void make_hit_record(const ptk_triangle& t, float4* q)
{
    const HV3 p0 = { t.v[0].position[0], t.v[0].position[1], t.v[0].position[2] };
    const HV3 p1 = { t.v[1].position[0], t.v[1].position[1], t.v[1].position[2] };
    const HV3 p2 = { t.v[2].position[0], t.v[2].position[1], t.v[2].position[2] };
    const HV3 N = hnormalize(hcross(hsub(p1, p0), hsub(p2, p1)));
    const HV3 e0 = hsub(p1, p0), e1 = hsub(p2, p0);
    const float d00 = hdot(e0, e0), d01 = hdot(e0, e1), d11 = hdot(e1, e1);
    const float inv_den = 1.0f / (d00 * d11 - d01 * d01);
    q[0] = make_float4(p0.x, p0.y, p0.z, N.x);
    q[1] = make_float4(N.y, N.z, e0.x, e0.y);
    q[2] = make_float4(e0.z, e1.x, e1.y, e1.z);
    q[3] = make_float4(d00, d01, d11, inv_den);
}

__host__ __device__ void make_shade_record(const ptk_triangle& t, float4* s)
{
    s[0] = make_float4(t.v[0].normal[0], t.v[0].normal[1], t.v[0].normal[2], t.v[1].normal[0]);
    s[1] = make_float4(t.v[1].normal[1], t.v[1].normal[2], t.v[2].normal[0], t.v[2].normal[1]);
    s[2] = make_float4(t.v[2].normal[2], t.v[0].uv[0], t.v[0].uv[1], t.v[1].uv[0]);
    s[3] = make_float4(t.v[1].uv[1], t.v[2].uv[0], t.v[2].uv[1], 0.0f);
}

// Children of interior node `n` (preorder index i) whose first child is `c` (index i + 1).  The builder makes a
// child only for a non-empty side (scene.cpp:280-284), so the next preorder slot is the LEFT child when there is
// one and the RIGHT child otherwise (every vertex above the split: the fp32 mean of scene.cpp:240-249 can round
// below all of them, e.g. a planar cluster at x = 0.1f).  Child boxes are the parent's box with one plane moved to
// the split (scene.cpp:31-56): the left child keeps the parent's min on the axis, the right child gets min = split.
// So the slot holds a right-only child exactly when its min differs from the parent's (compared as bits; if they
// are equal the two readings describe the same box and the same six quotients).  A right-only node is marked by
// right == i + 1, which walk_descend_step reads as "no left child".
__host__ __device__ inline uint32_t kd_f2u(float f)
{
    union { float f; uint32_t u; } v;
    v.f = f;
    return v.u;
}
__host__ __device__ inline void kd_children(const ptk_kd_node& n, const ptk_kd_node& c, uint32_t i, uint32_t axis, DKdNode& k)
{
    const bool right_only = c.skip >= n.skip && kd_f2u(c.bmin[axis]) != kd_f2u(n.bmin[axis]);
    if (right_only) {
        k.right = i + 1u;
        k.split = c.bmin[axis];
    } else {
        k.right = c.skip; // == n.skip (= k.a) when there is no right child
        k.split = c.bmax[axis];
    }
}

// ---- flattening of a tree that the GPU builder left in device memory (ptk_scene_upload) ----------
// Same records as the host loops of ptk_scene_upload write for a host-built tree.

// ptk_kd_node (preorder, skip links) -> DKdNode; max_leaf collects the largest leaf
__global__ __launch_bounds__(256) void k_flat_kd(const ptk_kd_node* __restrict__ nodes, uint32_t count, DKdNode* __restrict__ out,
                                                 uint32_t* __restrict__ max_leaf)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const ptk_kd_node n = nodes[i];
    DKdNode k;
    for (int a = 0; a < 3; ++a) {
        k.cmin[a] = n.bmin[a];
        k.cmax[a] = n.bmax[a];
        k.tmin[a] = n.tmin[a];
        k.tmax[a] = n.tmax[a];
    }
    const uint32_t axis = n.depth % 3u; // scene.cpp:230
    k.split = 0.0f;
    if (n.num_prims) {
        k.a = n.first_prim;
        k.b_axis = (n.num_prims & kKdCountMask) | (axis << 30);
        k.right = n.skip;
        if (n.num_prims > kKdCountMask) atomicMax(max_leaf, n.num_prims);
    } else {
        k.a = n.skip;
        k.b_axis = axis << 30;
        k.right = n.skip;
        if (i + 1 < count && i + 1 < n.skip) kd_children(n, nodes[i + 1], i, axis, k);
    }
    out[i] = k;
}

// triangle::update (scene.cpp:91-107) and the shading record of every triangle, original order
__global__ __launch_bounds__(256) void k_flat_records(const ptk_triangle* __restrict__ tris, uint32_t n, float4* __restrict__ rec,
                                                      float4* __restrict__ shade)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const ptk_triangle tri = tris[t];
    float4 q[4];
    make_hit_record(tri, q);
    for (int j = 0; j < 4; ++j) rec[(size_t)t * 4 + j] = q[j];
    make_shade_record(tri, q);
    for (int j = 0; j < 4; ++j) shade[(size_t)t * 4 + j] = q[j];
}

// leaf-ordered copies: hit record, triangle index, shading record index per leaf slot
__global__ __launch_bounds__(256) void k_flat_leaves(const uint32_t* __restrict__ prims, uint32_t count, const float4* __restrict__ rec,
                                                     uint32_t shade_base, float4* __restrict__ tri_hit, uint32_t* __restrict__ leaf_prim,
                                                     uint32_t* __restrict__ leaf_shade)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; // one thread per 16 bytes
    if (i >= (uint64_t)count * 4) return;
    const uint32_t k = (uint32_t)(i >> 2), j = (uint32_t)(i & 3u);
    const uint32_t t = prims[k];
    tri_hit[i] = rec[(size_t)t * 4 + j];
    if (j == 0u) {
        leaf_prim[k] = t;
        leaf_shade[k] = shade_base + t;
    }
}

// planar mesh: every triangle::m_face_normal within 1e-6 of the first one (and that one finite)
__global__ __launch_bounds__(256) void k_flat_planar(const ptk_triangle* __restrict__ tris, uint32_t n, uint32_t* __restrict__ differs,
                                                     float* __restrict__ n0_out)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float4 q0[4], q[4];
    make_hit_record(tris[0], q0);
    make_hit_record(tris[t], q);
    const float n0[3] = { q0[0].w, q0[1].x, q0[1].y }, nt[3] = { q[0].w, q[1].x, q[1].y };
    bool same = isfinite(n0[0]) && isfinite(n0[1]) && isfinite(n0[2]);
    for (int a = 0; a < 3; ++a) same = same && fabsf(nt[a] - n0[a]) <= 1e-6f; // false on NaN
    if (!same) atomicOr(differs, 1u);
    if (t == 0u)
        for (int a = 0; a < 3; ++a) n0_out[a] = n0[a];
}

// a table with room for `extra` more elements behind the host part
template <typename T>
int upload_extra(ptk_ctx* c, const std::vector<T>& host, size_t extra, const T** out)
{
    const size_t count = host.size() + extra;
    void* p = nullptr;
    PTK_HIP(c, hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    c->allocs.push_back(p);
    if (!host.empty()) PTK_HIP(c, hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return PTK_OK;
}

// Conservative world-space bounds of a node for the per-lane candidate lists
// (intersect_scene_lanelist, ptk_device.h).  lo/hi: the node's local box (mesh::m_bounds or the
// unit ball's box).  All in double from the fp32 matrices the caller supplied.  The margin
// coefficient is 256 x (fp32 epsilon x condition number) + 4 x the measured inconsistency of
// transform * inv_transform; matrices that are ill-conditioned, inconsistent or not finite
// switch the test off for the node (infinite box), which is always correct.
// Returns the condition estimate ||M||_F ||M^-1||_F the margins were built from, or a negative
// value when the test was switched off.
double make_cull_bounds(const ptk_node_desc& n, const float lo[3], const float hi[3], DNode& d)
{
    const float big = 3.402823466e+38f;
    auto disable = [&]() -> double {
        for (int a = 0; a < 3; ++a) {
            d.wmin[a] = -big;
            d.wmax[a] = big;
        }
        d.cull_k = 0.0f;
        d.cull_c = 0.0f;
        return -1.0;
    };
    const float* M = n.transform;      // column-major: M[c*4+r]
    const float* I = n.inv_transform;
    double fa = 0.0, fi = 0.0, incons = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            fa += (double)M[c * 4 + r] * M[c * 4 + r];
            fi += (double)I[c * 4 + r] * I[c * 4 + r];
            double e = (r == c) ? -1.0 : 0.0;
            for (int k = 0; k < 3; ++k) e += (double)M[k * 4 + r] * I[c * 4 + k];
            incons = std::fmax(incons, std::fabs(e));
        }
    double tmag = 0.0, tres = 0.0;
    for (int r = 0; r < 3; ++r) {
        double e = M[12 + r];
        for (int k = 0; k < 3; ++k) e += (double)M[k * 4 + r] * I[12 + k];
        tres = std::fmax(tres, std::fabs(e));
        tmag = std::fmax(tmag, std::fabs((double)M[12 + r]));
    }
    const double cond = std::sqrt(fa) * std::sqrt(fi);
    const double k = 256.0 * 1.1920928955078125e-7 * cond + 4.0 * incons;
    if (!(cond == cond) || !std::isfinite(cond) || !(k < 0.02) || !(tres <= 1e-3 * (tmag + 1.0))) return disable();
    double wlo[3] = { 1e300, 1e300, 1e300 }, whi[3] = { -1e300, -1e300, -1e300 };
    for (int corner = 0; corner < 8; ++corner) {
        const double p[3] = { (corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2] };
        for (int r = 0; r < 3; ++r) {
            const double w = M[0 * 4 + r] * p[0] + M[1 * 4 + r] * p[1] + M[2 * 4 + r] * p[2] + M[12 + r];
            wlo[r] = std::fmin(wlo[r], w);
            whi[r] = std::fmax(whi[r], w);
        }
    }
    double mag = 0.0;
    for (int r = 0; r < 3; ++r) {
        if (!std::isfinite(wlo[r]) || !std::isfinite(whi[r])) return disable();
        mag = std::fmax(mag, std::fmax(std::fabs(wlo[r]), std::fabs(whi[r])) + (whi[r] - wlo[r]));
    }
    if (!(mag < 1e30)) return disable();
    for (int r = 0; r < 3; ++r) {
        d.wmin[r] = std::nextafter((float)wlo[r], -big); // round outwards
        d.wmax[r] = std::nextafter((float)whi[r], big);
    }
    d.cull_k = (float)k;
    d.cull_c = (float)(k * (mag + tres) + 4.0 * tres) + 1e-30f;
    return cond;
}

// DCull record of a node (intersect_scene_balanced).  `planar`: the face normal every triangle of
// the node's mesh shares (local space), or NULL.  The reference rejects a triangle when
// cosine = dot(normalize(M^-1 d), N) >= 0 (scene.cpp:18-23,158).  In exact arithmetic that cosine
// is c_w |M^-T N| / |M^-1 d| with c_w = dot(d, fn), fn = normalize(M^-T N), so c_w >= ftau implies
// cosine >= ftau / cond.  The fp32 chain that produces the reference's cosine is off by less than
// 1e-6 * cond, and the triangles' own normals differ from N by less than 2e-6 (checked by the
// caller), so ftau = 1e-5 * cond^2 (at least 1e-4) leaves a factor of ten; above 0.5 the test is
// switched off.
DCull make_cull_record(const ptk_node_desc& n, const DNode& d, double cond, const float* planar)
{
    DCull c{};
    c.bx[0] = d.wmin[0]; c.bx[1] = d.wmax[0];
    c.by[0] = d.wmin[1]; c.by[1] = d.wmax[1];
    c.bz[0] = d.wmin[2]; c.bz[1] = d.wmax[2];
    c.cull_k = d.cull_k;
    c.cull_c = d.cull_c;
    c.cls = d.cls;
    c.ftau = 2.0f;
    if (planar && cond > 0.0) {
        const float* I = n.inv_transform; // column-major I[c*4+r]
        double w[3];
        for (int r = 0; r < 3; ++r) w[r] = (double)I[r * 4 + 0] * planar[0] + (double)I[r * 4 + 1] * planar[1] + (double)I[r * 4 + 2] * planar[2];
        const double len = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double tau = std::fmax(1e-4, 1e-5 * cond * cond);
        if (std::isfinite(len) && len > 1e-30 && tau <= 0.5) {
            for (int r = 0; r < 3; ++r) c.fn[r] = (float)(w[r] / len);
            c.ftau = (float)tau;
        }
    }
    return c;
}

constexpr size_t kMaxStamps = 8192;

// Brackets one kernel launch with HIP events on the context's stream when timing is on.
struct LaunchTimer {
    ptk_ctx* c;
    ptk_ctx::Stamp st{};
    bool on = false;
    LaunchTimer(ptk_ctx* ctx, int kind) : c(ctx)
    {
        if (!c->timing || c->timing_hold || c->stamps.size() >= kMaxStamps) return;
        if (hipEventCreate(&st.a) != hipSuccess) return;
        if (hipEventCreate(&st.b) != hipSuccess) {
            (void)hipEventDestroy(st.a);
            return;
        }
        st.kind = kind;
        on = hipEventRecord(st.a, c->stream) == hipSuccess;
    }
    ~LaunchTimer()
    {
        if (!on) return;
        (void)hipEventRecord(st.b, c->stream);
        c->stamps.push_back(st);
    }
};

void drop_stamps(ptk_ctx* c)
{
    for (auto& s : c->stamps) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    c->stamps.clear();
}

// Template MODE of the cast kernels (ptk_device.h, intersect_scene_mode) for this context / scene.
// `primary`: the coherent camera rays of bounce 0.
constexpr uint32_t kBalanceMinNodes = 2; // PTK_OPT_BALANCE = 2: pooled work lists from this many nodes on
constexpr uint32_t kCullMinNodes = 16; // PTK_OPT_NODE_CULL = 2: candidate lists from this many nodes on
int kernel_mode(const ptk_ctx* c, bool primary)
{
    // "auto": the pooled work lists, except for the incoherent bounces of a scene of many analytic / one-leaf
    // nodes, where the wave-uniform cull over every node costs more than per-lane candidate lists
    // (materials.json, 25 spheres: k_bounce<false> 182 -> 171 us)
#ifdef PTK_BALLS_PRIMARY_POOLED // bounce 0 of a spheres-only scene through the pooled work lists, for A/B runs
    const bool lists_primary = false;
#else
    // (a spheres-only scene of at most 64 nodes: the candidate lists take a bundle pre-cull of their own for the primary rays)
    const bool lists_primary = c->all_balls && c->S.num_nodes <= 64u;
#endif
    const bool many_flat_nodes = (!primary || lists_primary) && !c->deep_kd && c->S.num_nodes >= kCullMinNodes && c->traversal == 2 && c->node_cull == 2;
    if (c->balance == 1 || (c->balance == 2 && c->S.num_nodes >= kBalanceMinNodes && !many_flat_nodes))
        return 8 | ((c->kd_prune && c->deep_kd) ? 2 : 0);
    const bool cull = c->node_cull == 1 || (c->node_cull == 2 && c->S.num_nodes >= kCullMinNodes);
    // PTK_OPT_TRAVERSAL = 2: primary rays walk every node together unless the cull pays off
    const bool lists = c->traversal == 1 || (c->traversal == 2 && (!primary || cull));
    const int mode = (lists ? 1 : 0) | ((c->kd_prune && c->deep_kd) ? 2 : 0) | ((lists && cull) ? 4 : 0);
    return (mode == 5 && c->all_balls) ? 21 : mode; // candidate lists of a spheres-only scene: culled against the bounding balls
}

// PTK_OPT_SPLIT: the two-kernel bounce (k_bounce_split + k_walk) needs the pruned walk and a tree mask of 64 bits
bool split_mode(const ptk_ctx* c)
{
    const bool possible = c->kd_prune && c->deep_kd && c->num_trees >= 1u && c->num_trees <= 64u && c->balance != 0;
    return possible && c->split != 0;
}

constexpr uint32_t kWalkBlocks = 1024; // persistent walker grid: 4 resident waves on each of the 1024 SIMDs

// Walker work units are quarter segments; small launches take them one at a time (short tail), large
// ones in chunks so that the head counters see about 96 atomics per resident wave (with one head counter 24 was
// the optimum on the 5 M-triangle scene: 6 -> 712, 12 -> 795, 24 -> 832, 48 -> 823 Mpaths/s; with 32 heads
// 24 -> 928, 48 -> 964, 96 -> 971).
uint32_t walk_chunk_for(uint32_t num_segments)
{
    const uint64_t units = (uint64_t)num_segments * kWalkUnitsPerSegment;
    const uint64_t chunk = units / (uint64_t(kWalkBlocks) * (kBlock / 64) * PTK_WALK_GRABS);
    if (chunk < 1u) return 1u;
    return (uint32_t)(chunk >= kWalkUnitsPerSegment ? (chunk & ~uint64_t(kWalkUnitsPerSegment - 1u)) : chunk); // whole segments once a grab spans one
}

constexpr uint64_t kOverlapMinPaths = PTK_OVERLAP_MIN_PATHS; // PTK_OPT_OVERLAP: smaller calls stay one launch chain
constexpr uint64_t kOverlapFirstPct = PTK_OVERLAP_FIRST_PCT; // share of the pixels (and of the buffers) in the first chain
constexpr uint64_t kDeviceKdMinTriangles = 32768; // PTK_OPT_DEVICE_KD = 2: meshes at least this large build on the GPU

// The launch chain of ptk_render_samples_dev / ptk_render_samples_noalloc_dev over the scratch buffers `R`
// (the context's own allocations or the static array): chunks of at most R.cap paths, five bounces each.
struct RenderBufs {
    float* state;                 // queue 0: kStateFields arrays of `cap` floats
    float* state2;                // queue 1 (split mode)
    float* sample_L;              // 3 arrays of `cap` floats
    uint32_t* seg_count;          // [PTK_MAX_DEPTH][cap / 256]
    uint32_t* split_counters;     // [kSplitCounters]
    uint32_t* seg_pend;           // [cap / 256]
    uint32_t* pend_slot;          // [cap]
    unsigned long long* pend_mask; // [cap]
    float* pend_best;             // 9 arrays of `cap` floats
    float* state_other;           // PTK_OPT_COMPACT across segments: second queue (null = compact inside the segment)
    uint32_t* bucket;             //   and its bucket counters
    bool fused;                   // PTK_OPT_FUSE: all bounces in one kernel (k_bounce_fused), no queue
    uint32_t walk_blocks;         // persistent workgroups of k_walk (0 = kWalkBlocks)
    uint64_t cap;
};

// Device address of `scratch` on each GPU (every device holds its own copy at its own address), looked up
// once per device by ptk_ctx_create with that device current.
constexpr int kMaxDevices = 64;
std::mutex g_noalloc_mutex;
void* g_noalloc_ptr[kMaxDevices] = {};
hipError_t g_noalloc_err[kMaxDevices] = {};
bool g_noalloc_done[kMaxDevices] = {};

void noalloc_scratch_init(int device) // the caller has made `device` current
{
    if (device < 0 || device >= kMaxDevices) return;
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_done[device]) return;
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    g_noalloc_err[device] = e;
    g_noalloc_ptr[device] = e == hipSuccess ? ptr : nullptr;
    g_noalloc_done[device] = true;
}

// Carves the static array (base = its device address) into the buffers of one launch chain.
RenderBufs noalloc_layout(void* base)
{
    unsigned char* p = static_cast<unsigned char*>(base);
    auto take = [&](uint64_t bytes) {
        unsigned char* r = p;
        p += (bytes + 255u) & ~uint64_t(255);
        return r;
    };
    RenderBufs R{};
    R.cap = kNoallocPaths;
    R.pend_mask = reinterpret_cast<unsigned long long*>(take(kNoallocPaths * 8));
    R.state = reinterpret_cast<float*>(take(kNoallocPaths * kStateFields * 4));
    R.state2 = reinterpret_cast<float*>(take(kNoallocPaths * kStateFields * 4));
    R.sample_L = reinterpret_cast<float*>(take(kNoallocPaths * 3 * 4));
    R.pend_slot = reinterpret_cast<uint32_t*>(take(kNoallocPaths * 4));
    R.pend_best = reinterpret_cast<float*>(take(kNoallocPaths * 9 * 4));
    R.seg_count = reinterpret_cast<uint32_t*>(take(kNoallocSegments * PTK_MAX_DEPTH * 4));
    R.seg_pend = reinterpret_cast<uint32_t*>(take(kNoallocSegments * 4));
    R.split_counters = reinterpret_cast<uint32_t*>(take(kSplitCounters * 4));
    if ((uint64_t)(p - static_cast<unsigned char*>(base)) > kNoallocBytes) R.cap = 0; // cannot happen: see kNoallocBytes
    return R;
}

int set_device(ptk_ctx* c)
{
    PTK_HIP(c, hipSetDevice(c->device));
    return PTK_OK;
}

// (RenderBufs above) the launch chain itself
int render_samples_impl(ptk_ctx* c, const RenderBufs& R, uint64_t seed, uint64_t pixel_begin, uint64_t pixel_count,
                        uint64_t sample_begin, uint32_t sample_count, float* sum_rgb, int accumulate)
{
    const uint64_t cap = R.cap;
    const bool split = split_mode(c);
    const uint32_t s_chunk = (uint32_t)((uint64_t)sample_count < cap ? sample_count : cap);
    const uint64_t p_chunk = cap / s_chunk;
    for (uint32_t s0 = 0; s0 < sample_count; s0 += s_chunk) {
        const uint32_t ns = (sample_count - s0 < s_chunk) ? sample_count - s0 : s_chunk;
        const int acc = (accumulate || s0 > 0) ? 1 : 0;
        for (uint64_t p0 = 0; p0 < pixel_count; p0 += p_chunk) {
            const uint64_t np = (pixel_count - p0 < p_chunk) ? pixel_count - p0 : p_chunk;
            const uint64_t n_paths = np * ns; // <= cap < 2^31
            BounceArgs A{};
            A.state = R.state;
            A.sample_L = R.sample_L;
            A.seg_count = R.seg_count;
            A.cap = cap;
            A.num_segments = (uint32_t)((n_paths + kBlock - 1) / kBlock);
            A.n_paths = (uint32_t)n_paths;
            A.samples = ns;
            A.pixel_begin = pixel_begin + p0;
            A.sample_begin = sample_begin + s0;
            A.seed = seed;
            A.bounce = 0;
            // (the variants that walk KD-trees inside k_bounce keep the in-place form: fewer kernels to build)
            A.dense = (!split && R.state_other && R.bucket && !(kernel_mode(c, true) & 2) && !(kernel_mode(c, false) & 2)) ? 1u : 0u;
            A.state_other = R.state_other;
            A.bucket = R.bucket;
            if (A.dense) k_clear_u32<<<dim3(((PTK_MAX_DEPTH + 1) * kBuckets + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream>>>(R.bucket, (PTK_MAX_DEPTH + 1) * kBuckets);
            const dim3 grid(A.num_segments), block(kBlock);
            if (split) {
                k_clear_u32<<<dim3((kSplitCounters + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream>>>(R.split_counters, kSplitCounters);
                SplitBufs Q{};
                Q.q[0] = R.state;
                Q.q[1] = R.state2;
                Q.seg_count = R.seg_count;
                Q.seg_pend = R.seg_pend;
                Q.walk_head = R.split_counters;
                Q.num_segments = A.num_segments;
                Q.walk_chunk = walk_chunk_for(A.num_segments);
                Q.pend_slot = R.pend_slot;
                Q.pend_mask = R.pend_mask;
                Q.pend_best = R.pend_best;
                Q.cap = cap;
                const uint32_t wb_max = R.walk_blocks ? R.walk_blocks : kWalkBlocks;
                const uint32_t wb = A.num_segments < wb_max ? A.num_segments : wb_max;
                for (uint32_t b = 0; b < PTK_MAX_DEPTH; ++b) {
                    A.bounce = b;
                    {
                        LaunchTimer lt(c, b == 0 ? PTK_KERNEL_BOUNCE_FIRST : PTK_KERNEL_BOUNCE_NEXT);
                        if (b == 0) k_bounce_split<true><<<grid, block, 0, c->stream>>>(c->S, c->V, A, Q);
                        else k_bounce_split<false><<<grid, block, 0, c->stream>>>(c->S, c->V, A, Q);
                    }
                    k_walk_plan<<<1, 1024, 0, c->stream>>>(Q.seg_pend, Q.num_segments, Q.walk_chunk, wb * (kBlock / 64),
                                                            Q.walk_head + kWalkPlanBase + b);
                    {
                        LaunchTimer lt(c, PTK_KERNEL_WALK);
                        k_walk<false><<<dim3(wb * (kBlock / kWalkBlock)), dim3(kWalkBlock), 0, c->stream>>>(c->S, b, Q, BatchIO{});
                    }
                    LaunchTimer lt(c, PTK_KERNEL_SHADE_PARKED);
                    k_shade_parked<<<grid, block, 0, c->stream>>>(c->S, A, Q);
                }
                const uint32_t acc_blocks = (uint32_t)((np * 64 + kBlock - 1) / kBlock);
                {
                    LaunchTimer lt(c, PTK_KERNEL_ACCUMULATE);
                    hipLaunchKernelGGL(k_accumulate, dim3(acc_blocks), dim3(kBlock), 0, c->stream, R.sample_L, cap,
                                       (uint32_t)np, ns, pixel_begin + p0, sum_rgb, acc);
                }
                const uint32_t entries = A.num_segments * (PTK_MAX_DEPTH - 1);
                const uint32_t tally_blocks = entries / (kBlock * 8) + 1 > 256u ? 256u : entries / (kBlock * 8) + 1;
                hipLaunchKernelGGL(k_tally, dim3(tally_blocks), dim3(kBlock), 0, c->stream,
                                   R.seg_count + A.num_segments, entries, (uint32_t)n_paths, c->d_counters);
                PTK_HIP(c, hipGetLastError());
                continue;
            }
#ifdef PTK_EXP
            DScene Sx = c->S; // stage limit on bounce 2 only: its input rays are those of a normal run
#define PTK_BOUNCE(F, M) (Sx.exp_stage = (A.bounce == 2u) ? (uint32_t)c->exp_stage : 0u, k_bounce<F, M><<<grid, block, 0, c->stream>>>(Sx, c->V, A))
#define PTK_BOUNCE_D(F, M) PTK_BOUNCE(F, M)
#else
#define PTK_BOUNCE(F, M) k_bounce<F, M><<<grid, block, 0, c->stream>>>(c->S, c->V, A)
#define PTK_BOUNCE_D(F, M)                                                      \
    do {                                                                        \
        if (A.dense) k_bounce<F, M, true><<<grid, block, 0, c->stream>>>(c->S, c->V, A); \
        else k_bounce<F, M><<<grid, block, 0, c->stream>>>(c->S, c->V, A);      \
    } while (0)
#endif
            if (R.fused && !A.dense && kernel_mode(c, true) == 8 && kernel_mode(c, false) == 8) {
                {
                    LaunchTimer lt(c, PTK_KERNEL_BOUNCE_FUSED);
                    if (c->num_trees == 0u) k_bounce_fused<40><<<grid, block, 0, c->stream>>>(c->S, c->V, A); // no KD walk in the kernel
                    else k_bounce_fused<8><<<grid, block, 0, c->stream>>>(c->S, c->V, A);
                }
                const uint32_t acc_blocks = (uint32_t)((np * 64 + kBlock - 1) / kBlock);
                {
                    LaunchTimer lt(c, PTK_KERNEL_ACCUMULATE);
                    hipLaunchKernelGGL(k_accumulate, dim3(acc_blocks), dim3(kBlock), 0, c->stream, R.sample_L, cap,
                                       (uint32_t)np, ns, pixel_begin + p0, sum_rgb, acc);
                }
                const uint32_t entries = A.num_segments * (PTK_MAX_DEPTH - 1);
                const uint32_t tally_blocks = entries / (kBlock * 8) + 1 > 256u ? 256u : entries / (kBlock * 8) + 1;
                hipLaunchKernelGGL(k_tally, dim3(tally_blocks), dim3(kBlock), 0, c->stream,
                                   R.seg_count + A.num_segments, entries, (uint32_t)n_paths, c->d_counters);
                PTK_HIP(c, hipGetLastError());
                continue;
            }
            {
                LaunchTimer lt(c, PTK_KERNEL_BOUNCE_FIRST);
                switch (kernel_mode(c, true)) {
                case 0: PTK_BOUNCE_D(true, 0); break;
                case 1: PTK_BOUNCE_D(true, 1); break;
                case 2: PTK_BOUNCE(true, 2); break;
                case 3: PTK_BOUNCE(true, 3); break;
                case 5: PTK_BOUNCE_D(true, 5); break;
                case 21: PTK_BOUNCE_D(true, 21); break;
                case 7: PTK_BOUNCE(true, 7); break;
                case 8: PTK_BOUNCE_D(true, 8); break;
                default: PTK_BOUNCE(true, 10); break;
                }
            }
            for (uint32_t b = 1; b < PTK_MAX_DEPTH; ++b) {
                A.bounce = b;
                LaunchTimer lt(c, PTK_KERNEL_BOUNCE_NEXT);
                switch (kernel_mode(c, false)) {
                case 0: PTK_BOUNCE_D(false, 0); break;
                case 1: PTK_BOUNCE_D(false, 1); break;
                case 2: PTK_BOUNCE(false, 2); break;
                case 3: PTK_BOUNCE(false, 3); break;
                case 5: PTK_BOUNCE_D(false, 5); break;
                case 21: PTK_BOUNCE_D(false, 21); break;
                case 7: PTK_BOUNCE(false, 7); break;
                case 8: PTK_BOUNCE_D(false, 8); break;
                default: PTK_BOUNCE(false, 10); break;
                }
            }
#undef PTK_BOUNCE
#undef PTK_BOUNCE_D
            const uint32_t acc_blocks = (uint32_t)((np * 64 + kBlock - 1) / kBlock);
            {
                LaunchTimer lt(c, PTK_KERNEL_ACCUMULATE);
                hipLaunchKernelGGL(k_accumulate, dim3(acc_blocks), dim3(kBlock), 0, c->stream, R.sample_L, cap,
                                   (uint32_t)np, ns, pixel_begin + p0, sum_rgb, acc);
            }
            const uint32_t entries = A.num_segments * (PTK_MAX_DEPTH - 1);
            const uint32_t tally_blocks = entries / (kBlock * 8) + 1 > 256u ? 256u : entries / (kBlock * 8) + 1;
            hipLaunchKernelGGL(k_tally, dim3(tally_blocks), dim3(kBlock), 0, c->stream,
                               R.seg_count + A.num_segments, entries, (uint32_t)n_paths, c->d_counters);
            PTK_HIP(c, hipGetLastError());
        }
    }
    return PTK_OK;
}

} // namespace

extern "C" {

int ptk_ctx_create(int device, ptk_ctx** out)
{
    if (!out) return fail(nullptr, PTK_ERR_INVALID, "ptk_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return fail(nullptr, PTK_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    if (device < 0 || device >= count)
        return fail(nullptr, PTK_ERR_INVALID, "ptk_ctx_create: device " + std::to_string(device) + " out of range (" +
                                                  std::to_string(count) + " visible)");
    ptk_ctx* c = new ptk_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&c->own_stream)) != hipSuccess ||
        (e = hipMalloc((void**)&c->d_counters, 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(c->d_counters, 0, 2 * sizeof(unsigned long long))) != hipSuccess) {
        g_create_error = std::string("ptk_ctx_create: ") + hipGetErrorString(e);
        delete c;
        return PTK_ERR_HIP;
    }
    c->stream = c->own_stream;
    if (hipHostMalloc((void**)&c->h_snapshot, 2 * sizeof(unsigned long long)) != hipSuccess) c->h_snapshot = nullptr;
    if (c->h_snapshot && hipEventCreateWithFlags(&c->snapshot_ev, hipEventDisableTiming) != hipSuccess) c->snapshot_ev = nullptr;
    (void)hipGetLastError(); // without the snapshot PTK_OPT_COMPACT = 2 simply stays in place
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) c->side_stream = nullptr;
    if (c->side_stream && (hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming) != hipSuccess)) {
        (void)hipStreamDestroy(c->side_stream);
        c->side_stream = nullptr;
    }
    (void)hipGetLastError(); // without the side stream PTK_OPT_OVERLAP does nothing
    if (const char* e = std::getenv("PTK_EXP_STAGE")) c->exp_stage = std::atoi(e);
    noalloc_scratch_init(device); // `device` is current here
    *out = c;
    return PTK_OK;
}

void ptk_ctx_destroy(ptk_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    free_render(c);
    free_batch_split(c);
    drop_stamps(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->snapshot_ev) (void)hipEventDestroy(c->snapshot_ev);
    if (c->h_snapshot) (void)hipHostFree(c->h_snapshot);
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* ptk_last_error(const ptk_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ptk_ctx_set_stream(ptk_ctx* c, void* hip_stream)
{
    if (!c) return PTK_ERR_INVALID;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return PTK_OK;
}

void* ptk_ctx_get_stream(ptk_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

int ptk_request_interrupt(ptk_ctx* c)
{
    if (!c) return PTK_ERR_INVALID;
    c->interrupt = 1;
    return PTK_OK;
}

int ptk_ctx_synchronize(ptk_ctx* c)
{
    if (!c) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_scene_upload(ptk_ctx* c, const ptk_scene_desc* sc)
{
    if (!c || !sc) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    // Validate every index the kernels will follow before anything reaches the GPU.
    if ((sc->num_nodes && !sc->nodes) || (sc->num_meshes && !sc->meshes) || (sc->num_samplers && !sc->samplers) ||
        (sc->num_textures && !sc->textures))
        return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: NULL table with non-zero count");
    for (uint32_t i = 0; i < sc->num_textures; ++i)
        if (!sc->textures[i].rgb || sc->textures[i].width == 0 || sc->textures[i].height == 0)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: texture " + std::to_string(i) + " is empty");
    for (uint32_t i = 0; i < sc->num_samplers; ++i) {
        const ptk_sampler_desc& s = sc->samplers[i];
        if (s.kind != PTK_SAMPLER_COLOR && s.kind != PTK_SAMPLER_IMAGE)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: sampler " + std::to_string(i) + " has unknown kind");
        if (s.kind == PTK_SAMPLER_IMAGE && (s.texture >= sc->num_textures || s.mode > PTK_MODE_NEAREST))
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: sampler " + std::to_string(i) + " texture/mode out of range");
    }
    for (uint32_t i = 0; i < sc->num_nodes; ++i) {
        const ptk_node_desc& n = sc->nodes[i];
        if (n.shape_kind != PTK_SHAPE_SPHERE && n.shape_kind != PTK_SHAPE_MESH)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: node " + std::to_string(i) + " has no shape");
        if (n.shape_kind == PTK_SHAPE_MESH && n.mesh >= sc->num_meshes)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: node " + std::to_string(i) + " mesh index out of range");
        if (n.albedo >= sc->num_samplers || n.emission >= sc->num_samplers || n.roughness >= sc->num_samplers ||
            n.metallic >= sc->num_samplers)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: node " + std::to_string(i) + " sampler index out of range");
    }
    if (sc->background.kind != PTK_BACKGROUND_COLOR && sc->background.kind != PTK_BACKGROUND_EQUIRECT)
        return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: unknown background kind");
    if (sc->background.kind == PTK_BACKGROUND_EQUIRECT &&
        (sc->background.texture >= sc->num_textures || sc->background.mode > PTK_MODE_NEAREST))
        return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: background texture/mode out of range");
    for (uint32_t i = 0; i < sc->num_meshes; ++i)
        if (sc->meshes[i].num_triangles && !sc->meshes[i].triangles)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: mesh " + std::to_string(i) + " has NULL triangles");

    PTK_HIP(c, hipStreamSynchronize(c->stream));
    free_scene(c);

    // Host flattening: KD build per mesh (reference topology), leaf-ordered hit
    // records, original-order shading records.
    std::vector<DMesh> meshes(sc->num_meshes);
    std::vector<DKdNode> kd;
    std::vector<float4> tri_hit, tri_shade;
    std::vector<uint32_t> leaf_prim, leaf_shade;
    c->kd_infos.resize(sc->num_meshes);
    std::vector<float> planar((size_t)sc->num_meshes * 4, 0.0f); // per mesh: shared face normal, flag
    bool deep_kd = false;
    // One-leaf meshes (plane, cube: the 1.5x rule keeps them whole) are laid out first and on even
    // leaf slots, so that the pair table below can be indexed by slot / 2.
    std::vector<std::vector<ptk_kd_node>> built_nodes(sc->num_meshes);
    std::vector<std::vector<uint32_t>> built_prims(sc->num_meshes);
    std::vector<uint32_t> order;
    // Large meshes are built on the GPU and stay there: their tables are written by kernels behind the
    // host-built part of each table (k_flat_*), nothing of them comes back to the host.
    struct DeviceTrees {
        std::vector<PtkDeviceTree> t;
        ~DeviceTrees() { for (PtkDeviceTree& x : t) x.release(); }
    } dev;
    dev.t.resize(sc->num_meshes);
    for (uint32_t m = 0; m < sc->num_meshes; ++m) {
        const uint64_t nt = sc->meshes[m].num_triangles;
        // small meshes: the host builder is done before a kernel would have launched
        const bool on_device = nt != 0 && (c->device_kd == 1 || (c->device_kd == 2 && nt >= kDeviceKdMinTriangles));
        if (on_device) {
            std::string msg;
            if (int r = ptk_kd_build_device_keep(c->stream, sc->meshes[m].triangles, nt, nullptr, 0u, &dev.t[m], &msg)) {
                dev.t[m].release();
                // "auto" falls back to the host builder when the GPU build ran out of memory (its level buffers
                // take ~60 B per (node, triangle) entry on top of the tables); an explicit 1 reports the error
                if (r != PTK_ERR_NOMEM || c->device_kd != 2) return fail(c, r, msg);
                (void)hipGetLastError();
                ptk_kd_build_vectors(sc->meshes[m].triangles, nt, &c->kd_infos[m], built_nodes[m], built_prims[m]);
                continue;
            }
            c->kd_infos[m] = dev.t[m].info;
            if (dev.t[m].info.num_nodes <= 1u) { // a one-leaf mesh goes through the pair table: host path
                if (int r = ptk_kd_tree_download(c->stream, dev.t[m], built_nodes[m], built_prims[m], &msg)) return fail(c, r, msg);
                dev.t[m].release();
            }
        } else {
            ptk_kd_build_vectors(sc->meshes[m].triangles, nt, &c->kd_infos[m], built_nodes[m], built_prims[m]);
        }
    }
    for (int pass = 0; pass < 3; ++pass)
        for (uint32_t m = 0; m < sc->num_meshes; ++m) {
            const int group = dev.t[m].nodes ? 2 : (built_nodes[m].size() > 1 ? 1 : 0);
            if (group == pass) order.push_back(m);
        }
    size_t kd_extra = 0, leaf_extra = 0, shade_extra = 0; // device-built meshes, behind the host-built ones
    std::vector<uint32_t> dev_planar; // per device-built mesh: differs flag + 3 floats, filled by k_flat_planar
    size_t one_leaf_slots = 0;
    for (uint32_t m : order) {
        const ptk_mesh_desc& md = sc->meshes[m];
        std::vector<ptk_kd_node> nodes;
        std::vector<uint32_t> prims;
        nodes.swap(built_nodes[m]);
        prims.swap(built_prims[m]);
        if (!dev.t[m].nodes && nodes.size() <= 1) {
            if (leaf_prim.size() & 1u) { // pad to an even slot (never referenced by a leaf)
                leaf_prim.push_back(0u);
                tri_hit.resize(tri_hit.size() + 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            }
            one_leaf_slots = leaf_prim.size() + prims.size();
        }
        if (kd.size() + nodes.size() > 0xFFFFFFF0ull || leaf_prim.size() + prims.size() > 0x7FFFFFF0ull)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: scene exceeds 32-bit table indices");
        DMesh& dm = meshes[m];
        if (dev.t[m].nodes) {
            // (order: every host-built mesh has been laid out by now)
            const ptk_kd_info& ki = dev.t[m].info;
            if (kd.size() + kd_extra + ki.num_nodes > 0xFFFFFFF0ull || leaf_prim.size() + leaf_extra + ki.num_leaf_prims > 0x7FFFFFF0ull)
                return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: scene exceeds 32-bit table indices");
            if (ki.max_depth > PTK_KD_MAX_DEPTH) return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: KD-tree deeper than PTK_KD_MAX_DEPTH");
            dm.kd_base = (uint32_t)(kd.size() + kd_extra);
            dm.kd_count = ki.num_nodes;
            dm.leaf_base = (uint32_t)(leaf_prim.size() + leaf_extra);
            dm.shade_base = (uint32_t)(tri_shade.size() / 4 + shade_extra);
            kd_extra += ki.num_nodes;
            leaf_extra += ki.num_leaf_prims;
            shade_extra += md.num_triangles;
            deep_kd = true;
            continue;
        }
        dm.kd_base = (uint32_t)kd.size();
        dm.kd_count = (uint32_t)nodes.size();
        dm.leaf_base = (uint32_t)leaf_prim.size();
        dm.shade_base = (uint32_t)(tri_shade.size() / 4);
        if (c->kd_infos[m].max_depth > PTK_KD_MAX_DEPTH)
            return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: KD-tree deeper than PTK_KD_MAX_DEPTH");
        if (nodes.size() > 1) deep_kd = true;
        for (size_t i = 0; i < nodes.size(); ++i) {
            const ptk_kd_node& n = nodes[i];
            DKdNode k;
            std::memcpy(k.cmin, n.bmin, 12);
            std::memcpy(k.cmax, n.bmax, 12);
            std::memcpy(k.tmin, n.tmin, 12);
            std::memcpy(k.tmax, n.tmax, 12);
            const uint32_t axis = n.depth % 3u; // scene.cpp:230
            k.split = 0.0f;
            if (n.num_prims) {
                if (n.num_prims > kKdCountMask) return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: leaf too large");
                k.a = n.first_prim;
                k.b_axis = n.num_prims | (axis << 30);
                k.right = n.skip;
            } else {
                k.a = n.skip;
                k.b_axis = axis << 30;
                k.right = n.skip;
                if (i + 1 < nodes.size() && i + 1 < n.skip) kd_children(n, nodes[i + 1], (uint32_t)i, axis, k);
            }
            kd.push_back(k);
        }
        std::vector<float4> rec((size_t)md.num_triangles * 4);
        for (uint64_t t = 0; t < md.num_triangles; ++t) make_hit_record(md.triangles[t], &rec[t * 4]);
        // planar mesh: every triangle::m_face_normal (scene.cpp:93) within 1e-6 of the first one
        if (md.num_triangles) {
            const float n0[3] = { rec[0].w, rec[1].x, rec[1].y };
            bool same = std::isfinite(n0[0]) && std::isfinite(n0[1]) && std::isfinite(n0[2]);
            for (uint64_t t = 1; t < md.num_triangles && same; ++t) {
                const float nt[3] = { rec[t * 4].w, rec[t * 4 + 1].x, rec[t * 4 + 1].y };
                for (int a = 0; a < 3; ++a) same = same && std::fabs(nt[a] - n0[a]) <= 1e-6f; // false on NaN
            }
            if (same) {
                planar[m * 4 + 0] = n0[0];
                planar[m * 4 + 1] = n0[1];
                planar[m * 4 + 2] = n0[2];
                planar[m * 4 + 3] = 1.0f;
            }
        }
        const size_t h0 = tri_hit.size();
        tri_hit.resize(h0 + prims.size() * 4);
        for (size_t k = 0; k < prims.size(); ++k) std::memcpy(&tri_hit[h0 + k * 4], &rec[(size_t)prims[k] * 4], 64);
        leaf_prim.insert(leaf_prim.end(), prims.begin(), prims.end());
        leaf_shade.resize(leaf_prim.size(), 0u); // padding slots included
        for (size_t k = 0; k < prims.size(); ++k) leaf_shade[dm.leaf_base + k] = dm.shade_base + prims[k];
        const size_t s0 = tri_shade.size();
        tri_shade.resize(s0 + (size_t)md.num_triangles * 4);
        for (uint64_t t = 0; t < md.num_triangles; ++t) make_shade_record(md.triangles[t], &tri_shade[s0 + t * 4]);
    }
    // planar test of the device-built meshes (make_cull_record below wants it)
    {
        uint32_t nd = 0;
        for (uint32_t m = 0; m < sc->num_meshes; ++m) nd += dev.t[m].nodes ? 1u : 0u;
        if (nd) {
            uint32_t* d_flags = nullptr;
            PTK_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_flags), (size_t)nd * 4 * sizeof(uint32_t)));
            hipError_t e = hipMemsetAsync(d_flags, 0, (size_t)nd * 4 * sizeof(uint32_t), c->stream);
            uint32_t k = 0;
            for (uint32_t m = 0; m < sc->num_meshes && e == hipSuccess; ++m)
                if (dev.t[m].nodes) {
                    k_flat_planar<<<(dev.t[m].num_triangles + 255u) / 256u, 256, 0, c->stream>>>(
                        dev.t[m].tris, dev.t[m].num_triangles, d_flags + (size_t)k * 4, reinterpret_cast<float*>(d_flags + (size_t)k * 4 + 1));
                    ++k;
                }
            dev_planar.resize((size_t)nd * 4);
            if (e == hipSuccess) e = hipMemcpyAsync(dev_planar.data(), d_flags, dev_planar.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            (void)hipFree(d_flags);
            PTK_HIP(c, e);
            k = 0;
            for (uint32_t m = 0; m < sc->num_meshes; ++m)
                if (dev.t[m].nodes) {
                    if (dev_planar[(size_t)k * 4] == 0u) {
                        std::memcpy(&planar[m * 4], &dev_planar[(size_t)k * 4 + 1], 12);
                        planar[m * 4 + 3] = 1.0f;
                    }
                    ++k;
                }
        }
    }
    // Pair table of the one-leaf meshes (intersect_scene_balanced): slots 2k and 2k+1 interleaved
    // word by word, so one 16-B load yields two packed-fp32 operands.  An odd tail repeats itself.
    if (one_leaf_slots >= (size_t(1) << 26)) // the triangle stage packs a pair record's index in 25 bits
        return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: more than 2^26 triangles in one-leaf meshes");
    std::vector<float4> tri_pair(((one_leaf_slots + 1) / 2) * 8);
    for (size_t k = 0; k * 2 < one_leaf_slots; ++k) {
        const float* A = reinterpret_cast<const float*>(&tri_hit[(2 * k) * 4]);
        const float* B = (2 * k + 1 < one_leaf_slots) ? reinterpret_cast<const float*>(&tri_hit[(2 * k + 1) * 4]) : A;
        float* out = reinterpret_cast<float*>(&tri_pair[k * 8]);
        for (int w = 0; w < 16; ++w) {
            out[2 * w] = A[w];
            out[2 * w + 1] = B[w];
        }
    }

    std::vector<DNode> nodes(sc->num_nodes);
    std::vector<DCull> culls(sc->num_nodes);
    for (uint32_t i = 0; i < sc->num_nodes; ++i) {
        const ptk_node_desc& n = sc->nodes[i];
        DNode& d = nodes[i];
        std::memset(&d, 0, sizeof(d));
        std::memcpy(d.inv, n.inv_transform, 64);
        std::memcpy(d.xf, n.transform, 64);
        std::memcpy(d.nrm, n.normal_transform, 64);
        d.kind = n.shape_kind;
        d.mesh = n.shape_kind == PTK_SHAPE_MESH ? n.mesh : 0u;
        d.albedo = n.albedo;
        d.emission = n.emission;
        d.roughness = n.roughness;
        d.metallic = n.metallic;
        d.uses_uv = (sc->samplers[n.albedo].kind == PTK_SAMPLER_IMAGE || sc->samplers[n.emission].kind == PTK_SAMPLER_IMAGE ||
                     sc->samplers[n.roughness].kind == PTK_SAMPLER_IMAGE || sc->samplers[n.metallic].kind == PTK_SAMPLER_IMAGE)
                        ? 1u
                        : 0u;
        const float unit_lo[3] = { -1.0f, -1.0f, -1.0f }, unit_hi[3] = { 1.0f, 1.0f, 1.0f };
        if (n.shape_kind == PTK_SHAPE_MESH) {
            const ptk_kd_info& ki = c->kd_infos[n.mesh];
            const double cond = make_cull_bounds(n, ki.bmin, ki.bmax, d);
            d.cls = (ki.num_nodes > 1 || ki.num_leaf_prims > kLeafItemMax)
                        ? kClassTree
                        : (ki.num_leaf_prims <= kSmallLeafMax ? kClassSmallLeaf : kClassLeaf);
            culls[i] = make_cull_record(n, d, cond, planar[n.mesh * 4 + 3] != 0.0f ? &planar[n.mesh * 4] : nullptr);
        } else {
            const double cond = make_cull_bounds(n, unit_lo, unit_hi, d);
            d.cls = kClassSphere;
            culls[i] = make_cull_record(n, d, cond, nullptr);
        }
    }
    // Bounding balls of the sphere nodes (DBall): centre = translation, radius = largest singular value of the 3x3 (power
    // iteration on M^T M in double), a hair up.  Nodes whose box test is switched off (ill-conditioned matrices) get none.
    struct DBall {
        float cx, cy, cz, r;
    };
    std::vector<DBall> balls(sc->num_nodes, DBall{ 0.0f, 0.0f, 0.0f, -1.0f });
    for (uint32_t i = 0; i < sc->num_nodes; ++i) {
        const ptk_node_desc& n = sc->nodes[i];
        if (n.shape_kind != PTK_SHAPE_SPHERE || !(std::fabs((double)nodes[i].wmax[0]) < 1e30)) continue;
        const float* M = n.transform; // column-major M[c*4+r]
        double A[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                A[a][b] = 0.0;
                for (int r = 0; r < 3; ++r) A[a][b] += (double)M[a * 4 + r] * (double)M[b * 4 + r];
            }
        double x[3] = { 1.0, 0.9, 0.8 }, lambda = 0.0;
        for (int it = 0; it < 200; ++it) {
            double y[3];
            for (int a = 0; a < 3; ++a) y[a] = A[a][0] * x[0] + A[a][1] * x[1] + A[a][2] * x[2];
            lambda = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
            if (!(lambda > 0.0)) break;
            for (int a = 0; a < 3; ++a) x[a] = y[a] / lambda;
        }
        // power iteration approaches the largest eigenvalue from below: bound it from above with the trace as well
        const double trace = A[0][0] + A[1][1] + A[2][2];
        double r2 = lambda * 1.0001;
        if (!(r2 <= trace) || !(r2 >= trace / 3.0 * 0.9999)) r2 = trace; // not converged / degenerate: the trace is always an upper bound
        const double r = std::sqrt(r2) * 1.00001;
        if (!std::isfinite(r) || !std::isfinite((double)M[12]) || !std::isfinite((double)M[13]) || !std::isfinite((double)M[14])) continue;
        balls[i] = DBall{ M[12], M[13], M[14], (float)r * 1.000001f };

    }
    // The cull loop's margins, applied once: every ray that starts inside the scene's boxes (all rays after bounce 0: they
    // start a hair off a surface) has |o|_1 <= o1max, so its margin is at most cull_k * o1max + cull_c; the table holds the
    // boxes grown by that, rounded outwards.  Rays from further out are exempted from the per-ray cull in the kernel.
    double o1max = 0.0;
    for (uint32_t i = 0; i < sc->num_nodes; ++i) {
        double s1 = 0.0;
        bool finite = true;
        for (int a = 0; a < 3; ++a) {
            const double lo = nodes[i].wmin[a], hi = nodes[i].wmax[a];
            finite = finite && std::fabs(lo) < 1e30 && std::fabs(hi) < 1e30;
            s1 += std::fmax(std::fabs(lo), std::fabs(hi));
        }
        if (finite) o1max = std::fmax(o1max, s1);
    }
    o1max = o1max * 1.001 + 0.01; // the ray's start offset (pathtracer.h:19) and then some
    for (uint32_t i = 0; i < sc->num_nodes; ++i) {
        DCull& cu = culls[i];
        float* box[3] = { cu.bx, cu.by, cu.bz };
        const double m = (double)cu.cull_k * o1max + (double)cu.cull_c;
        for (int a = 0; a < 3; ++a) {
            if (!(std::fabs((double)box[a][0]) < 1e30) || !(std::fabs((double)box[a][1]) < 1e30)) continue; // switched off: stays infinite
            box[a][0] = std::nextafter((float)((double)box[a][0] - m), -3.402823466e+38f);
            box[a][1] = std::nextafter((float)((double)box[a][1] + m), 3.402823466e+38f);
        }
    }
    // A scene made of spheres only: the per-lane candidate lists cull against the balls, which then take the place of the
    // world boxes in the node records (intersect_scene_lanelist<.., BALLS>; DNode::uses_uv bit 1 marks them)
    bool all_balls = sc->num_nodes > 0;
    for (uint32_t i = 0; i < sc->num_nodes; ++i) all_balls = all_balls && balls[i].r >= 0.0f;
    if (all_balls)
        for (uint32_t i = 0; i < sc->num_nodes; ++i) {
            nodes[i].wmin[0] = balls[i].cx; nodes[i].wmin[1] = balls[i].cy; nodes[i].wmin[2] = balls[i].cz;
            nodes[i].wmax[0] = balls[i].r;
            nodes[i].uses_uv |= 2u;
        }
    c->all_balls = all_balls;
    std::vector<DLeaf> leaf_info(sc->num_nodes);
    for (uint32_t i = 0; i < sc->num_nodes; ++i) {
        DLeaf& lf = leaf_info[i];
        std::memset(&lf, 0, sizeof(lf));
        const ptk_node_desc& n = sc->nodes[i];
        if (n.shape_kind == PTK_SHAPE_MESH && meshes[n.mesh].kd_count == 1u) {
            const DKdNode& root = kd[meshes[n.mesh].kd_base];
            std::memcpy(lf.cmin, root.cmin, 12);
            std::memcpy(lf.cmax, root.cmax, 12);
            lf.first = meshes[n.mesh].leaf_base + root.a;
            lf.count = root.b_axis & kKdCountMask;
        }
    }
    std::vector<uint32_t> tree_nodes;
    for (uint32_t i = 0; i < sc->num_nodes; ++i)
        if (culls[i].cls == kClassTree) {
            culls[i].tree_slot = (uint32_t)tree_nodes.size();
            tree_nodes.push_back(i);
        }
    std::vector<DSampler> samplers(sc->num_samplers);
    for (uint32_t i = 0; i < sc->num_samplers; ++i) {
        const ptk_sampler_desc& s = sc->samplers[i];
        samplers[i] = DSampler{ s.kind, s.color[0], s.color[1], s.color[2], s.kind == PTK_SAMPLER_IMAGE ? s.texture : 0u,
                                s.mode, 0u, 0u };
    }
    std::vector<DTexture> textures(sc->num_textures);
    for (uint32_t i = 0; i < sc->num_textures; ++i) {
        const ptk_texture_desc& t = sc->textures[i];
        const size_t bytes = (size_t)t.width * t.height * 3 * sizeof(float);
        void* p = nullptr;
        PTK_HIP(c, hipMalloc(&p, bytes));
        c->allocs.push_back(p);
        PTK_HIP(c, hipMemcpy(p, t.rgb, bytes, hipMemcpyHostToDevice));
        textures[i] = DTexture{ static_cast<const float*>(p), t.width, t.height };
    }

    DScene S{};
    if (int r = upload(c, nodes, &S.nodes)) return r;
    if (int r = upload(c, culls, &S.cull)) return r;
    if (int r = upload(c, tree_nodes, &S.tree_nodes)) return r;
    S.num_trees = (uint32_t)tree_nodes.size();
    if (int r = upload(c, meshes, &S.meshes)) return r;
    if (int r = upload_extra(c, kd, kd_extra, &S.kd)) return r;
    if (int r = upload_extra(c, tri_hit, leaf_extra * 4, &S.tri_hit)) return r;
    if (int r = upload(c, tri_pair, &S.tri_pair)) return r;
    if (int r = upload_extra(c, leaf_prim, leaf_extra, &S.leaf_prim)) return r;
    if (int r = upload_extra(c, leaf_shade, leaf_extra, &S.leaf_shade)) return r;
    if (int r = upload(c, leaf_info, &S.leaf_info)) return r;
    if (int r = upload_extra(c, tri_shade, shade_extra * 4, &S.tri_shade)) return r;
    for (uint32_t m = 0; m < sc->num_meshes; ++m)
        if (dev.t[m].nodes) {
            const PtkDeviceTree& t = dev.t[m];
            const DMesh& dm = meshes[m];
            float4* rec = nullptr;
            uint32_t* d_max_leaf = nullptr;
            PTK_HIP(c, hipMalloc(reinterpret_cast<void**>(&rec), (size_t)t.num_triangles * 64));
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_max_leaf), sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemsetAsync(d_max_leaf, 0, sizeof(uint32_t), c->stream);
            uint32_t max_leaf = 0;
            if (e == hipSuccess) {
                k_flat_kd<<<(t.info.num_nodes + 255u) / 256u, 256, 0, c->stream>>>(t.nodes, t.info.num_nodes,
                                                                                   const_cast<DKdNode*>(S.kd) + dm.kd_base, d_max_leaf);
                k_flat_records<<<(t.num_triangles + 255u) / 256u, 256, 0, c->stream>>>(
                    t.tris, t.num_triangles, rec, const_cast<float4*>(S.tri_shade) + (size_t)dm.shade_base * 4);
                const uint64_t quads = t.info.num_leaf_prims * 4;
                if (quads)
                    k_flat_leaves<<<(uint32_t)((quads + 255u) / 256u), 256, 0, c->stream>>>(
                        t.prims, (uint32_t)t.info.num_leaf_prims, rec, dm.shade_base, const_cast<float4*>(S.tri_hit) + (size_t)dm.leaf_base * 4,
                        const_cast<uint32_t*>(S.leaf_prim) + dm.leaf_base, const_cast<uint32_t*>(S.leaf_shade) + dm.leaf_base);
                e = hipMemcpyAsync(&max_leaf, d_max_leaf, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipGetLastError();
            (void)hipFree(rec);
            if (d_max_leaf) (void)hipFree(d_max_leaf);
            PTK_HIP(c, e);
            if (max_leaf) return fail(c, PTK_ERR_INVALID, "ptk_scene_upload: leaf too large");
            dev.t[m].release();
        }
    if (int r = upload(c, samplers, &S.samplers)) return r;
    {
        std::vector<DSampler> node_samplers((size_t)sc->num_nodes * 4);
        for (uint32_t i = 0; i < sc->num_nodes; ++i) {
            const ptk_node_desc& n = sc->nodes[i];
            node_samplers[(size_t)i * 4 + 0] = samplers[n.albedo];
            node_samplers[(size_t)i * 4 + 1] = samplers[n.emission];
            node_samplers[(size_t)i * 4 + 2] = samplers[n.roughness];
            node_samplers[(size_t)i * 4 + 3] = samplers[n.metallic];
        }
        if (int r = upload(c, node_samplers, &S.node_samplers)) return r;
    }
    if (int r = upload(c, textures, &S.textures)) return r;
    S.num_nodes = sc->num_nodes;
    S.bg_kind = sc->background.kind;
    S.bg_r = sc->background.color[0];
    S.bg_g = sc->background.color[1];
    S.bg_b = sc->background.color[2];
    S.bg_texture = sc->background.kind == PTK_BACKGROUND_EQUIRECT ? sc->background.texture : 0u;
    S.bg_mode = sc->background.mode;
    S.cam_x = sc->camera.position[0];
    S.cam_y = sc->camera.position[1];
    S.cam_z = sc->camera.position[2];
    S.walk_stack = (uint32_t)c->walk_stack;
    S.cull_o1max = (float)o1max;
#ifdef PTK_STATS
    {
        void* sp = nullptr;
        PTK_HIP(c, hipMalloc(&sp, 64 * sizeof(unsigned long long)));
        PTK_HIP(c, hipMemset(sp, 0, 64 * sizeof(unsigned long long)));
        c->allocs.push_back(sp);
        S.stats = static_cast<unsigned long long*>(sp);
    }
#endif
    c->S = S;
    c->num_meshes = sc->num_meshes;
    c->num_samplers = sc->num_samplers;
    c->num_textures = sc->num_textures;
    c->has_scene = true;
    c->rays_per_path = -1.0; // PTK_OPT_COMPACT = 2 measures the new scene afresh
    c->deep_kd = deep_kd;
    c->num_trees = (uint32_t)tree_nodes.size();
    // camera direction feeds the view basis built by ptk_render_begin
    std::memcpy(c->cam_dir, sc->camera.direction, sizeof(c->cam_dir));
    c->has_view = false;
    return PTK_OK;
}

int ptk_scene_set_camera(ptk_ctx* c, const ptk_camera_desc* cam)
{
    if (!c || !cam) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_scene_set_camera: no scene uploaded");
    c->S.cam_x = cam->position[0];
    c->S.cam_y = cam->position[1];
    c->S.cam_z = cam->position[2];
    std::memcpy(c->cam_dir, cam->direction, sizeof(c->cam_dir));
    // the view basis follows the camera (abstract_pathtracer.cpp:20-22); viewport, fov and queues stay
    if (c->has_view) c->V = make_view(c->cam_dir, c->V.width, c->V.height, c->fov_y);
    return PTK_OK;
}

int ptk_scene_kd_info(ptk_ctx* c, uint32_t mesh, ptk_kd_info* info)
{
    if (!c || !info) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_scene_kd_info: no scene uploaded");
    if (mesh >= c->kd_infos.size()) return fail(c, PTK_ERR_INVALID, "ptk_scene_kd_info: mesh index out of range");
    *info = c->kd_infos[mesh];
    return PTK_OK;
}

int ptk_intersect_batch_dev(ptk_ctx* c, uint64_t n, const float* o, const float* d, int32_t* node, int32_t* prim,
                            float* t, float* u, float* v, float* position, float* normal, float* uv)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_intersect_batch: no scene uploaded");
    if (n == 0) return PTK_OK;
    if (!o || !d) return fail(c, PTK_ERR_INVALID, "ptk_intersect_batch: NULL ray arrays");
    if (n > 0x7FFFFFFFull * kBlock) return fail(c, PTK_ERR_INVALID, "ptk_intersect_batch: batch too large");
    if (int r = set_device(c)) return r;
    const uint32_t blocks = (uint32_t)((n + kBlock - 1) / kBlock);
    const BatchIO B{ n, o, d, node, prim, t, u, v, position, normal, uv };
    const dim3 g(blocks), b(kBlock);
    if (split_mode(c) && n < 0xFFFFFFFFull) {
        // two kernels: everything but the KD-trees, then the walker over the parked rays
        const uint64_t need = (uint64_t)blocks * kBlock;
        if (need > c->bsplit_cap) {
            PTK_HIP(c, hipStreamSynchronize(c->stream));
            free_batch_split(c);
            PTK_HIP(c, hipMalloc((void**)&c->d_bsplit_counters, kSplitCounters * sizeof(uint32_t)));
            PTK_HIP(c, hipMalloc((void**)&c->d_bseg_pend, (need / kBlock) * sizeof(uint32_t)));
            PTK_HIP(c, hipMalloc((void**)&c->d_bpend_slot, need * sizeof(uint32_t)));
            PTK_HIP(c, hipMalloc((void**)&c->d_bpend_mask, need * sizeof(unsigned long long)));
            PTK_HIP(c, hipMalloc((void**)&c->d_bpend_best, need * 9 * sizeof(float)));
            c->bsplit_cap = need;
        }
        PTK_HIP(c, hipMemsetAsync(c->d_bsplit_counters, 0, kSplitCounters * sizeof(uint32_t), c->stream));
        SplitBufs Q{};
        Q.seg_pend = c->d_bseg_pend;
        Q.walk_head = c->d_bsplit_counters;
        Q.pend_slot = c->d_bpend_slot;
        Q.pend_mask = c->d_bpend_mask;
        Q.pend_best = c->d_bpend_best;
        Q.cap = c->bsplit_cap;
        Q.num_segments = blocks;
        Q.walk_chunk = walk_chunk_for(blocks);
        LaunchTimer lt(c, PTK_KERNEL_INTERSECT_BATCH);
        k_intersect_batch_split<<<g, b, 0, c->stream>>>(c->S, B, Q);
        const uint32_t wb = blocks < kWalkBlocks ? blocks : kWalkBlocks;
        k_walk_plan<<<1, 1024, 0, c->stream>>>(Q.seg_pend, Q.num_segments, Q.walk_chunk, wb * (kBlock / 64), Q.walk_head + kWalkPlanBase);
        k_walk<true><<<dim3(wb * (kBlock / kWalkBlock)), dim3(kWalkBlock), 0, c->stream>>>(c->S, 0u, Q, B);
        k_batch_parked<<<g, b, 0, c->stream>>>(c->S, B, Q);
    } else {
        LaunchTimer lt(c, PTK_KERNEL_INTERSECT_BATCH);
#define PTK_BATCH(M) k_intersect_batch<M><<<g, b, 0, c->stream>>>(c->S, B)
        switch (kernel_mode(c, false)) {
        case 0: PTK_BATCH(0); break;
        case 1: PTK_BATCH(1); break;
        case 2: PTK_BATCH(2); break;
        case 3: PTK_BATCH(3); break;
        case 5: PTK_BATCH(5); break;
        case 21: PTK_BATCH(21); break;
        case 7: PTK_BATCH(7); break;
        case 8: PTK_BATCH(8); break;
        default: PTK_BATCH(10); break;
        }
#undef PTK_BATCH
    }
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_intersect_batch(ptk_ctx* c, uint64_t n, const float* o, const float* d, int32_t* node, int32_t* prim, float* t,
                        float* u, float* v, float* position, float* normal, float* uv)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_intersect_batch: no scene uploaded");
    if (n == 0) return PTK_OK;
    if (!o || !d) return fail(c, PTK_ERR_INVALID, "ptk_intersect_batch: NULL ray arrays");
    if (int r = set_device(c)) return r;
    // one staging allocation: o(3n) d(3n) node prim t u v pos(3n) nrm(3n) uv(2n) = 19n words
    float* buf = nullptr;
    PTK_HIP(c, hipMalloc((void**)&buf, (size_t)n * 19 * sizeof(float)));
    float* d_o = buf;
    float* d_d = buf + 3 * n;
    int32_t* d_node = reinterpret_cast<int32_t*>(buf + 6 * n);
    int32_t* d_prim = reinterpret_cast<int32_t*>(buf + 7 * n);
    float* d_t = buf + 8 * n;
    float* d_u = buf + 9 * n;
    float* d_v = buf + 10 * n;
    float* d_pos = buf + 11 * n;
    float* d_nrm = buf + 14 * n;
    float* d_uv = buf + 17 * n;
    int rc = PTK_OK;
    hipError_t e;
    if ((e = hipMemcpyAsync(d_o, o, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d_d, d, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream)) != hipSuccess) {
        rc = fail(c, PTK_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    if (rc == PTK_OK) rc = ptk_intersect_batch_dev(c, n, d_o, d_d, d_node, d_prim, d_t, d_u, d_v, d_pos, d_nrm, d_uv);
    auto back = [&](void* dst, const void* src, size_t words) {
        if (rc == PTK_OK && dst) {
            e = hipMemcpyAsync(dst, src, words * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, PTK_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
        }
    };
    back(node, d_node, n);
    back(prim, d_prim, n);
    back(t, d_t, n);
    back(u, d_u, n);
    back(v, d_v, n);
    back(position, d_pos, 3 * n);
    back(normal, d_nrm, 3 * n);
    back(uv, d_uv, 2 * n);
    e = hipStreamSynchronize(c->stream);
    if (rc == PTK_OK && e != hipSuccess) rc = fail(c, PTK_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    (void)hipFree(buf);
    return rc;
}

int ptk_render_begin(ptk_ctx* c, uint32_t width, uint32_t height, float fov_y, uint64_t max_paths)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_render_begin: no scene uploaded");
    if (width == 0 || height == 0) return fail(c, PTK_ERR_INVALID, "ptk_render_begin: empty viewport");
    if ((uint64_t)width * height > 0xFFFFFFFFull) return fail(c, PTK_ERR_INVALID, "ptk_render_begin: viewport too large");
    if (int r = set_device(c)) return r;
    if (max_paths == 0) max_paths = kDefaultMaxPaths;
    if (max_paths > 0x7FFFFF00ull) max_paths = 0x7FFFFF00ull;
    const uint64_t cap = (max_paths + kBlock - 1) / kBlock * kBlock;

    const DView V = make_view(c->cam_dir, width, height, fov_y);

    if (cap != c->cap) {
        PTK_HIP(c, hipStreamSynchronize(c->stream));
        free_render(c);
        PTK_HIP(c, hipMalloc((void**)&c->d_state, cap * kStateFields * sizeof(float)));
        PTK_HIP(c, hipMalloc((void**)&c->d_sample_L, cap * 3 * sizeof(float)));
        PTK_HIP(c, hipMalloc((void**)&c->d_seg_count, (cap / kBlock) * PTK_MAX_DEPTH * sizeof(uint32_t)));
        c->cap = cap;
    }
    c->V = V;
    c->fov_y = fov_y;
    c->has_view = true;
    return PTK_OK;
}

int ptk_render_samples_dev(ptk_ctx* c, uint64_t seed, uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                           uint32_t sample_count, float* sum_rgb, int accumulate)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene || !c->has_view) return fail(c, PTK_ERR_STATE, "ptk_render_samples: call ptk_render_begin first");
    if (!sum_rgb) return fail(c, PTK_ERR_INVALID, "ptk_render_samples: NULL sum buffer");
    const uint64_t num_pixels = (uint64_t)c->V.width * c->V.height;
    if (pixel_begin > num_pixels || pixel_count > num_pixels - pixel_begin)
        return fail(c, PTK_ERR_INVALID, "ptk_render_samples: pixel range outside the viewport");
    if (pixel_count == 0) return PTK_OK;
    if (int r = set_device(c)) return r;
    if (sample_count == 0) {
        if (!accumulate) PTK_HIP(c, hipMemsetAsync(sum_rgb + pixel_begin * 3, 0, pixel_count * 3 * sizeof(float), c->stream));
        return PTK_OK;
    }
    const bool split = split_mode(c);
    const uint64_t cap = c->cap;
    if (split && c->split_cap != cap) {
        PTK_HIP(c, hipStreamSynchronize(c->stream));
        free_split(c);
        PTK_HIP(c, hipMalloc((void**)&c->d_state2, cap * kStateFields * sizeof(float)));
        PTK_HIP(c, hipMalloc((void**)&c->d_split_counters, 2 * kSplitCounters * sizeof(uint32_t))); // one set per launch chain (PTK_OPT_OVERLAP)
        PTK_HIP(c, hipMalloc((void**)&c->d_seg_pend, (cap / kBlock) * sizeof(uint32_t)));
        PTK_HIP(c, hipMalloc((void**)&c->d_pend_slot, cap * sizeof(uint32_t)));
        PTK_HIP(c, hipMalloc((void**)&c->d_pend_mask, cap * sizeof(unsigned long long)));
        PTK_HIP(c, hipMalloc((void**)&c->d_pend_best, cap * 9 * sizeof(float)));
        c->split_cap = cap;
    }
    // PTK_OPT_COMPACT: survivors packed across segments (second queue); split mode has its own queues.  At 2 (auto) the
    // choice follows the rays per path of the chunks rendered so far: packing costs a closed scene (all five bounces
    // alive: Room) 2.5 %, and gains 25 % where most paths leave after a bounce or two (materials.json: 1.44 rays per path)
    if (c->snapshot_pending && hipEventQuery(c->snapshot_ev) == hipSuccess) {
        c->snapshot_pending = false;
        const unsigned long long paths = c->h_snapshot[0], rays = c->h_snapshot[1];
        if (paths > c->seen_paths && rays >= c->seen_rays)
            c->rays_per_path = double(rays - c->seen_rays) / double(paths - c->seen_paths);
        c->seen_paths = paths; // (a counter reset makes one snapshot useless, no more)
        c->seen_rays = rays;
    }
    (void)hipGetLastError(); // hipEventQuery's "not ready" is not an error of ours
    const bool dense = !split && (c->compact == 1 || (c->compact == 2 && c->rays_per_path >= 0.0 && c->rays_per_path < 4.0));
    if (dense && c->other_cap != cap) {
        PTK_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_state_other) (void)hipFree(c->d_state_other);
        if (c->d_bucket) (void)hipFree(c->d_bucket);
        c->d_state_other = nullptr;
        c->d_bucket = nullptr;
        c->other_cap = 0;
        PTK_HIP(c, hipMalloc((void**)&c->d_state_other, cap * kStateFields * sizeof(float)));
        PTK_HIP(c, hipMalloc((void**)&c->d_bucket, (PTK_MAX_DEPTH + 1) * kBuckets * sizeof(uint32_t)));
        c->other_cap = cap;
    }
    RenderBufs R{};
    R.state_other = dense ? c->d_state_other : nullptr;
    R.bucket = dense ? c->d_bucket : nullptr;
    // PTK_OPT_FUSE: a closed scene (nearly five rays per path in the chunks rendered so far) keeps its rays in registers
    R.fused = !split && !dense && (c->fuse == 1 || (c->fuse == 2 && c->rays_per_path >= 4.75));
    R.state = c->d_state;
    R.state2 = c->d_state2;
    R.sample_L = c->d_sample_L;
    R.seg_count = c->d_seg_count;
    R.split_counters = c->d_split_counters;
    R.seg_pend = c->d_seg_pend;
    R.pend_slot = c->d_pend_slot;
    R.pend_mask = c->d_pend_mask;
    R.pend_best = c->d_pend_best;
    R.cap = cap;
    // per-launch events cost a few microseconds of GPU pipeline each (5 % on a chain of short kernels): with
    // ptk_ctx_enable_timing(n > 1) only every n-th call carries them
    c->timing_hold = c->timing && (c->timing_tick++ % c->timing_every) != 0u;
    // PTK_OPT_OVERLAP: a walker launch ends with its slowest wave (a quarter of its time is waves that have run out of rays
    // waiting for that one), and so does every other kernel of the chain in its own small way.  A large split-mode call
    // therefore goes out as TWO chains over the two halves of its pixel range, each on half of the buffers, the second
    // on a stream of its own between two events: the GPU fills one chain's tails with the other's work.  Same samples,
    // same sums (a pixel's samples stay in one chain, in the same order).  Calls whose launches are being timed
    // (ptk_ctx_enable_timing) run as one chain, so that the durations are those of kernels running alone.
    const uint64_t cap1 = (cap * kOverlapFirstPct / 100u) / kBlock * kBlock, cap2 = (cap - cap1) / kBlock * kBlock;
    const bool timed = c->timing && !c->timing_hold;
    int rc;
    if (split && c->overlap && c->side_stream && !timed && pixel_count >= 2 && cap1 >= kBlock && cap2 >= kBlock &&
        pixel_count * (uint64_t)sample_count >= kOverlapMinPaths) {
        RenderBufs R1 = R, R2 = R;
        R1.cap = cap1;
        R2.cap = cap2;
        R1.walk_blocks = R2.walk_blocks = PTK_OVERLAP_WALK_BLOCKS;
        R2.state = R.state + kStateFields * cap1;
        R2.state2 = R.state2 + kStateFields * cap1;
        R2.sample_L = R.sample_L + 3 * cap1;
        R2.seg_count = R.seg_count + (cap1 / kBlock) * PTK_MAX_DEPTH;
        R2.split_counters = R.split_counters + kSplitCounters;
        R2.seg_pend = R.seg_pend + cap1 / kBlock;
        R2.pend_slot = R.pend_slot + cap1;
        R2.pend_mask = R.pend_mask + cap1;
        R2.pend_best = R.pend_best + 9 * cap1;
        uint64_t first = pixel_count * kOverlapFirstPct / 100u;
        if (first < 1) first = 1;
        hipStream_t main_stream = c->stream;
        PTK_HIP(c, hipEventRecord(c->fork_ev, main_stream));
        PTK_HIP(c, hipStreamWaitEvent(c->side_stream, c->fork_ev, 0));
        rc = render_samples_impl(c, R1, seed, pixel_begin, first, sample_begin, sample_count, sum_rgb, accumulate);
        c->stream = c->side_stream; // (the chain launches on the context's stream)
        const int rc2 = render_samples_impl(c, R2, seed, pixel_begin + first, pixel_count - first, sample_begin, sample_count, sum_rgb, accumulate);
        c->stream = main_stream;
        // the join is made whatever the chains returned: nothing may be left running beside the caller's stream
        const hipError_t e1 = hipEventRecord(c->join_ev, c->side_stream);
        const hipError_t e2 = hipStreamWaitEvent(main_stream, c->join_ev, 0);
        if (rc == PTK_OK) rc = rc2;
        if (rc == PTK_OK && (e1 != hipSuccess || e2 != hipSuccess))
            rc = fail(c, PTK_ERR_HIP, std::string("ptk_render_samples: joining the second launch chain: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    } else {
        rc = render_samples_impl(c, R, seed, pixel_begin, pixel_count, sample_begin, sample_count, sum_rgb, accumulate);
    }
    c->timing_hold = false;
    if (rc == PTK_OK && !split && (c->compact == 2 || c->fuse == 2) && c->snapshot_ev && !c->snapshot_pending) {
        PTK_HIP(c, hipMemcpyAsync(c->h_snapshot, c->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        PTK_HIP(c, hipEventRecord(c->snapshot_ev, c->stream));
        c->snapshot_pending = true;
    }
    return rc;
}

int ptk_render_samples_noalloc_dev(ptk_ctx* c, uint64_t seed, uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                                   uint32_t sample_count, float* sum_rgb, int accumulate)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene || !c->has_view) return fail(c, PTK_ERR_STATE, "ptk_render_samples_noalloc: call ptk_render_begin first");
    if (!sum_rgb) return fail(c, PTK_ERR_INVALID, "ptk_render_samples_noalloc: NULL sum buffer");
    const uint64_t num_pixels = (uint64_t)c->V.width * c->V.height;
    if (pixel_begin > num_pixels || pixel_count > num_pixels - pixel_begin)
        return fail(c, PTK_ERR_INVALID, "ptk_render_samples_noalloc: pixel range outside the viewport");
    if (pixel_count * (uint64_t)sample_count > kNoallocPaths) // pixel_count < 2^32, sample_count < 2^32: no overflow
        return fail(c, PTK_ERR_INVALID, "ptk_render_samples_noalloc: pixel_count * sample_count = " +
                                            std::to_string(pixel_count * (uint64_t)sample_count) + " exceeds PTK_NOALLOC_MAX_PATHS = " +
                                            std::to_string(kNoallocPaths) + " (split the pixel or sample range over several calls)");
    if (pixel_count == 0) return PTK_OK;
    if (int r = set_device(c)) return r;
    if (sample_count == 0) {
        if (!accumulate) PTK_HIP(c, hipMemsetAsync(sum_rgb + pixel_begin * 3, 0, pixel_count * 3 * sizeof(float), c->stream));
        return PTK_OK;
    }
    void* base = nullptr;
    if (c->device >= 0 && c->device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_noalloc_mutex);
        if (g_noalloc_done[c->device] && g_noalloc_err[c->device] != hipSuccess)
            return fail(c, PTK_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(g_noalloc_err[c->device]));
        base = g_noalloc_ptr[c->device];
    }
    if (!base) return fail(c, PTK_ERR_STATE, "ptk_render_samples_noalloc: no static scratch on this device");
    const RenderBufs R = noalloc_layout(base);
    if (R.cap != kNoallocPaths) return fail(c, PTK_ERR_STATE, "ptk_render_samples_noalloc: scratch layout exceeds the static array");
    // one chunk by construction (n_paths <= R.cap); every counter the chain reads is written by the chain
    // itself first (seg_count / seg_pend by the bounce kernels, the split counters by the memset in the
    // chain), so nothing of the previous call's contents is ever read
    c->timing_hold = true;
    const int rc = render_samples_impl(c, R, seed, pixel_begin, pixel_count, sample_begin, sample_count, sum_rgb, accumulate);
    c->timing_hold = false;
    return rc;
}

int ptk_render_aov_dev(ptk_ctx* c, uint32_t aov, uint64_t seed, uint64_t pixel_begin, uint64_t pixel_count, uint64_t sample_begin,
                       uint32_t sample_count, float* sum_rgb, int accumulate)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene || !c->has_view) return fail(c, PTK_ERR_STATE, "ptk_render_aov: call ptk_render_begin first");
    if (aov > PTK_AOV_NORMAL) return fail(c, PTK_ERR_INVALID, "ptk_render_aov: unknown AOV");
    if (!sum_rgb) return fail(c, PTK_ERR_INVALID, "ptk_render_aov: NULL sum buffer");
    const uint64_t num_pixels = (uint64_t)c->V.width * c->V.height;
    if (pixel_begin > num_pixels || pixel_count > num_pixels - pixel_begin)
        return fail(c, PTK_ERR_INVALID, "ptk_render_aov: pixel range outside the viewport");
    if (pixel_count == 0) return PTK_OK;
    if (int r = set_device(c)) return r;
    if (sample_count == 0) {
        if (!accumulate) PTK_HIP(c, hipMemsetAsync(sum_rgb + pixel_begin * 3, 0, pixel_count * 3 * sizeof(float), c->stream));
        return PTK_OK;
    }
    const uint64_t cap = c->cap;
    const uint32_t s_chunk = (uint32_t)((uint64_t)sample_count < cap ? sample_count : cap);
    const uint64_t p_chunk = cap / s_chunk;
    for (uint32_t s0 = 0; s0 < sample_count; s0 += s_chunk) {
        const uint32_t ns = (sample_count - s0 < s_chunk) ? sample_count - s0 : s_chunk;
        const int acc = (accumulate || s0 > 0) ? 1 : 0;
        for (uint64_t p0 = 0; p0 < pixel_count; p0 += p_chunk) {
            const uint64_t np = (pixel_count - p0 < p_chunk) ? pixel_count - p0 : p_chunk;
            BounceArgs A{};
            A.sample_L = c->d_sample_L;
            A.cap = cap;
            A.n_paths = (uint32_t)(np * ns);
            A.num_segments = (A.n_paths + kBlock - 1) / kBlock;
            A.samples = ns;
            A.pixel_begin = pixel_begin + p0;
            A.sample_begin = sample_begin + s0;
            A.seed = seed;
            const dim3 grid(A.num_segments), block(kBlock);
#define PTK_AOV(M) k_aov<M><<<grid, block, 0, c->stream>>>(c->S, c->V, A, aov)
            switch (kernel_mode(c, true)) {
            case 0: PTK_AOV(0); break;
            case 1: PTK_AOV(1); break;
            case 2: PTK_AOV(2); break;
            case 3: PTK_AOV(3); break;
            case 5: PTK_AOV(5); break;
            case 21: PTK_AOV(21); break;
            case 7: PTK_AOV(7); break;
            case 8: PTK_AOV(8); break;
            default: PTK_AOV(10); break;
            }
#undef PTK_AOV
            hipLaunchKernelGGL(k_accumulate, dim3((uint32_t)((np * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->d_sample_L, cap,
                               (uint32_t)np, ns, pixel_begin + p0, sum_rgb, acc);
            PTK_HIP(c, hipGetLastError());
        }
    }
    return PTK_OK;
}

int ptk_resolve_range_dev(ptk_ctx* c, const float* sum_rgb, float* image_rgb, uint64_t pixel_begin, uint64_t pixel_count,
                          uint64_t spi, uint64_t iteration)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_view) return fail(c, PTK_ERR_STATE, "ptk_resolve_iteration: call ptk_render_begin first");
    if (!sum_rgb || !image_rgb || spi == 0) return fail(c, PTK_ERR_INVALID, "ptk_resolve_iteration: bad arguments");
    const uint64_t num_pixels = (uint64_t)c->V.width * c->V.height;
    if (pixel_begin > num_pixels || pixel_count > num_pixels - pixel_begin)
        return fail(c, PTK_ERR_INVALID, "ptk_resolve_iteration: pixel range outside the viewport");
    if (pixel_count == 0) return PTK_OK;
    if (int r = set_device(c)) return r;
    const uint64_t n = pixel_count * 3;
    const float t = (float)iteration / (float)(iteration + 1);
    hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, sum_rgb + pixel_begin * 3,
                       image_rgb + pixel_begin * 3, n, (float)spi, t);
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_resolve_iteration_dev(ptk_ctx* c, const float* sum_rgb, float* image_rgb, uint64_t spi, uint64_t iteration)
{
    if (!c) return PTK_ERR_INVALID;
    return ptk_resolve_range_dev(c, sum_rgb, image_rgb, 0, (uint64_t)c->V.width * c->V.height, spi, iteration);
}

int ptk_render_image(ptk_ctx* c, uint64_t seed, uint64_t iterations, uint32_t spi, float* image_rgb)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene || !c->has_view) return fail(c, PTK_ERR_STATE, "ptk_render_image: call ptk_render_begin first");
    if (!image_rgb || spi == 0) return fail(c, PTK_ERR_INVALID, "ptk_render_image: bad arguments");
    if (int r = set_device(c)) return r;
    const uint64_t np = (uint64_t)c->V.width * c->V.height;
    float* buf = nullptr;
    PTK_HIP(c, hipMalloc((void**)&buf, np * 6 * sizeof(float)));
    float* d_sum = buf;
    float* d_img = buf + np * 3;
    int rc = PTK_OK;
    hipError_t e = hipMemsetAsync(d_img, 0, np * 3 * sizeof(float), c->stream);
    if (e != hipSuccess) rc = fail(c, PTK_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    // One launch chain (at most max_paths_per_launch paths) at a time is handed to the GPU, two in flight: between
    // chains the interrupt flag is looked at, as the reference looks per scan line (abstract_pathtracer.cpp:63-72).
    // An interrupted iteration folds the pixels it finished into the image (the other pixels keep the previous
    // iterations' mean, like the reference's unfinished scan lines) and the call returns PTK_INTERRUPTED.
    hipEvent_t ev[2] = { nullptr, nullptr };
    for (int i = 0; i < 2 && rc == PTK_OK; ++i)
        if ((e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)) != hipSuccess)
            rc = fail(c, PTK_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
    const uint64_t per_pixel = (uint64_t)spi < c->cap ? spi : c->cap;
    const uint64_t budget = c->cap / per_pixel; // pixels per chain
    bool interrupted = false;
    uint64_t chains = 0;
    for (uint64_t it = 0; it < iterations && rc == PTK_OK && !interrupted; ++it) {
        uint64_t done = 0;
        for (uint64_t p0 = 0; p0 < np && rc == PTK_OK; p0 += budget) {
            if (c->interrupt.load()) {
                interrupted = true;
                break;
            }
            const uint64_t cnt = np - p0 < budget ? np - p0 : budget;
            if (chains >= 2 && (e = hipEventSynchronize(ev[chains & 1])) != hipSuccess)
                rc = fail(c, PTK_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
            if (rc == PTK_OK) rc = ptk_render_samples_dev(c, seed, p0, cnt, it * spi, spi, d_sum, 0);
            if (rc == PTK_OK && (e = hipEventRecord(ev[chains & 1], c->stream)) != hipSuccess)
                rc = fail(c, PTK_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e));
            ++chains;
            done = p0 + cnt;
        }
        if (rc == PTK_OK) rc = ptk_resolve_range_dev(c, d_sum, d_img, 0, done, spi, it);
    }
    if (rc == PTK_OK) {
        e = hipMemcpyAsync(image_rgb, d_img, np * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(c, PTK_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    e = hipStreamSynchronize(c->stream);
    if (rc == PTK_OK && e != hipSuccess) rc = fail(c, PTK_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    for (int i = 0; i < 2; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    (void)hipFree(buf);
    c->interrupt = 0; // the request has been served (or came too late for this call)
    return (rc == PTK_OK && interrupted) ? PTK_INTERRUPTED : rc;
}

int ptk_tonemap_dev(ptk_ctx* c, const float* image_rgb, uint32_t* rgba8, uint64_t num_pixels)
{
    if (!c) return PTK_ERR_INVALID;
    if (!image_rgb || !rgba8) return fail(c, PTK_ERR_INVALID, "ptk_tonemap: NULL buffer");
    if (num_pixels == 0) return PTK_OK;
    if (int r = set_device(c)) return r;
    hipLaunchKernelGGL(k_tonemap, dim3((uint32_t)((num_pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                       image_rgb, rgba8, num_pixels);
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_probe_image_sample_dev(ptk_ctx* c, uint32_t texture, uint32_t mode, uint64_t n, const float* uv, float* rgb)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_probe_image_sample: no scene uploaded");
    if (texture >= c->num_textures || mode > PTK_MODE_NEAREST) return fail(c, PTK_ERR_INVALID, "ptk_probe_image_sample: texture / mode out of range");
    if (n == 0) return PTK_OK;
    if (!uv || !rgb) return fail(c, PTK_ERR_INVALID, "ptk_probe_image_sample: NULL array");
    if (n > 0x7FFFFFFFull * kBlock) return fail(c, PTK_ERR_INVALID, "ptk_probe_image_sample: too many probes");
    if (int r = set_device(c)) return r;
    k_probe_image<<<dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream>>>(c->S, texture, mode, n, uv, rgb);
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_probe_background_dev(ptk_ctx* c, uint64_t n, const float* directions, float* rgb)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_probe_background: no scene uploaded");
    if (n == 0) return PTK_OK;
    if (!directions || !rgb) return fail(c, PTK_ERR_INVALID, "ptk_probe_background: NULL array");
    if (n > 0x7FFFFFFFull * kBlock) return fail(c, PTK_ERR_INVALID, "ptk_probe_background: too many probes");
    if (int r = set_device(c)) return r;
    k_probe_background<<<dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream>>>(c->S, n, directions, rgb);
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_probe_hemisphere_dev(ptk_ctx* c, uint64_t n, const float* normals, const float* u12, float* directions)
{
    if (!c) return PTK_ERR_INVALID;
    if (n == 0) return PTK_OK;
    if (!normals || !u12 || !directions) return fail(c, PTK_ERR_INVALID, "ptk_probe_hemisphere: NULL array");
    if (n > 0x7FFFFFFFull * kBlock) return fail(c, PTK_ERR_INVALID, "ptk_probe_hemisphere: too many probes");
    if (int r = set_device(c)) return r;
    k_probe_hemisphere<<<dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream>>>(n, normals, u12, directions);
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_probe_shade_dev(ptk_ctx* c, uint64_t n, const float* origins, const float* directions, const float* u12, int32_t* node,
                        float* emission, float* weight, float* next_origin, float* next_direction)
{
    if (!c) return PTK_ERR_INVALID;
    if (!c->has_scene) return fail(c, PTK_ERR_STATE, "ptk_probe_shade: no scene uploaded");
    if (n == 0) return PTK_OK;
    if (!origins || !directions || !u12) return fail(c, PTK_ERR_INVALID, "ptk_probe_shade: NULL input array");
    if (n > 0x7FFFFFFFull * kBlock) return fail(c, PTK_ERR_INVALID, "ptk_probe_shade: too many probes");
    if (int r = set_device(c)) return r;
    const ShadeProbeIO P{ n, origins, directions, u12, node, emission, weight, next_origin, next_direction };
    const dim3 g((uint32_t)((n + kBlock - 1) / kBlock)), b(kBlock);
#define PTK_PROBE(M) k_probe_shade<M><<<g, b, 0, c->stream>>>(c->S, P)
    switch (kernel_mode(c, false)) {
    case 0: PTK_PROBE(0); break;
    case 1: PTK_PROBE(1); break;
    case 2: PTK_PROBE(2); break;
    case 3: PTK_PROBE(3); break;
    case 5: PTK_PROBE(5); break;
    case 21: PTK_PROBE(21); break;
    case 7: PTK_PROBE(7); break;
    case 8: PTK_PROBE(8); break;
    default: PTK_PROBE(10); break;
    }
#undef PTK_PROBE
    PTK_HIP(c, hipGetLastError());
    return PTK_OK;
}

int ptk_counters_download_async(ptk_ctx* c, ptk_counters* host_out)
{
    if (!c || !host_out) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    static_assert(sizeof(ptk_counters) == 2 * sizeof(unsigned long long), "ptk_counters = the two device counters");
    PTK_HIP(c, hipMemcpyAsync(host_out, c->d_counters, sizeof(ptk_counters), hipMemcpyDeviceToHost, c->stream));
    return PTK_OK;
}

int ptk_get_counters(ptk_ctx* c, ptk_counters* out)
{
    if (!c || !out) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    unsigned long long h[2] = { 0, 0 };
    PTK_HIP(c, hipMemcpyAsync(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    out->paths = h[0];
    out->rays = h[1];
    return PTK_OK;
}

int ptk_ctx_set_option(ptk_ctx* c, int option, int value)
{
    if (!c) return PTK_ERR_INVALID;
    if (option == PTK_OPT_TRAVERSAL && value >= 0 && value <= 2) {
        c->traversal = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_NODE_CULL && value >= 0 && value <= 2) {
        c->node_cull = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_SPLIT && value >= 0 && value <= 2) {
        c->split = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_BALANCE && value >= 0 && value <= 2) {
        c->balance = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_DEVICE_KD && value >= 0 && value <= 2) {
        c->device_kd = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_WALK_STACK && value >= 0 && value <= 64) {
        c->walk_stack = value;
        c->S.walk_stack = (uint32_t)value;
        return PTK_OK;
    }
    if (option == PTK_OPT_COMPACT && value >= 0 && value <= 2) {
        c->compact = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_FUSE && value >= 0 && value <= 2) {
        c->fuse = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_KD_PRUNE && (value == 0 || value == 1)) {
        c->kd_prune = value;
        return PTK_OK;
    }
    if (option == PTK_OPT_OVERLAP && (value == 0 || value == 1)) {
        c->overlap = value;
        return PTK_OK;
    }
    return fail(c, PTK_ERR_INVALID, "ptk_ctx_set_option: unknown option or value");
}

int ptk_kd_build_dev(ptk_ctx* c, const ptk_triangle* triangles, uint64_t num_triangles, const float* bounds, uint32_t root_depth,
                     ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap, uint32_t* prims, uint64_t prims_cap)
{
    if (!c) return PTK_ERR_INVALID;
    if (!triangles || num_triangles == 0)
        return fail(c, PTK_ERR_INVALID, "ptk_kd_build_dev: no triangles (an empty mesh is one empty leaf: use ptk_kd_build)");
    if (int r = set_device(c)) return r;
    std::vector<ptk_kd_node> n;
    std::vector<uint32_t> p;
    ptk_kd_info local;
    std::string msg;
    if (int r = ptk_kd_build_device(c->stream, triangles, num_triangles, bounds, root_depth, &local, n, p, &msg)) return fail(c, r, msg);
    if (info) *info = local;
    if (nodes) {
        if (nodes_cap < n.size()) return fail(c, PTK_ERR_INVALID, "ptk_kd_build_dev: nodes_cap too small");
        std::memcpy(nodes, n.data(), n.size() * sizeof(ptk_kd_node));
    }
    if (prims) {
        if (prims_cap < p.size()) return fail(c, PTK_ERR_INVALID, "ptk_kd_build_dev: prims_cap too small");
        std::memcpy(prims, p.data(), p.size() * sizeof(uint32_t));
    }
    return PTK_OK;
}

int ptk_ctx_enable_timing(ptk_ctx* c, int enable)
{
    if (!c) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    drop_stamps(c);
    c->timing = enable != 0;
    c->timing_every = enable > 1 ? (uint32_t)enable : 1u;
    c->timing_tick = 0;
    return PTK_OK;
}

int ptk_get_timing(ptk_ctx* c, ptk_kernel_timing* out)
{
    if (!c || !out) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    std::memset(out, 0, sizeof(ptk_kernel_timing) * PTK_KERNEL_KINDS);
    for (auto& s : c->stamps) {
        float ms = 0.0f;
        PTK_HIP(c, hipEventElapsedTime(&ms, s.a, s.b));
        out[s.kind].launches += 1;
        out[s.kind].total_ms += ms;
    }
    drop_stamps(c);
    return PTK_OK;
}

int ptk_dev_alloc(ptk_ctx* c, uint64_t bytes, void** out)
{
    if (!c || !out) return PTK_ERR_INVALID;
    *out = nullptr;
    if (int r = set_device(c)) return r;
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return fail(c, PTK_ERR_NOMEM, "ptk_dev_alloc: out of device memory");
    if (e != hipSuccess) return fail(c, PTK_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    *out = p;
    return PTK_OK;
}

int ptk_dev_free(ptk_ctx* c, void* p)
{
    if (!c) return PTK_ERR_INVALID;
    if (!p) return PTK_OK;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    PTK_HIP(c, hipFree(p));
    return PTK_OK;
}

int ptk_dev_memset(ptk_ctx* c, void* p, int value, uint64_t bytes)
{
    if (!c || (!p && bytes)) return PTK_ERR_INVALID;
    if (bytes == 0) return PTK_OK;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipMemsetAsync(p, value, bytes, c->stream));
    return PTK_OK;
}

int ptk_dev_download(ptk_ctx* c, void* dst, const void* src, uint64_t bytes)
{
    if (!c || ((!dst || !src) && bytes)) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    if (bytes) PTK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

int ptk_dev_upload(ptk_ctx* c, void* dst, const void* src, uint64_t bytes)
{
    if (!c || ((!dst || !src) && bytes)) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    if (bytes) PTK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    return PTK_OK;
}

#ifdef PTK_STATS
/* experiment builds only (-DPTK_STATS): reads and clears the work counters */
int ptk_debug_stats(ptk_ctx* c, unsigned long long* out16)
{
    if (!c || !out16 || !c->has_scene) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipStreamSynchronize(c->stream));
    PTK_HIP(c, hipMemcpy(out16, c->S.stats, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    PTK_HIP(c, hipMemset(c->S.stats, 0, 64 * sizeof(unsigned long long)));
    return PTK_OK;
}
#endif

int ptk_reset_counters(ptk_ctx* c)
{
    if (!c) return PTK_ERR_INVALID;
    if (int r = set_device(c)) return r;
    PTK_HIP(c, hipMemsetAsync(c->d_counters, 0, 2 * sizeof(unsigned long long), c->stream));
    return PTK_OK;
}

} // extern "C"
