// ptk_group.hip — one host driver, N contexts on N devices (include/ptk.h, ptk_group_*).
//
// paths are independent and every GPU holds the whole scene, so an iteration is split
// by sample range (by pixel range when it has fewer samples than members), each member accumulates an fp32
// SUM image, and the sums meet in one reduce to member 0, which applies the running mean of
// abstract_pathtracer.cpp:88-90.  The exchange step:
//   * members that share a device are added into the first member of that device by a kernel;
//   * the per-device partial sums go to member 0 with ONE ncclReduce(sum, float) over RCCL / xGMI
//     (single-process communicators from ncclCommInitAll, the calls fused by ncclGroupStart / End).
// RCCL is loaded with dlopen on first need, so a one-GPU process never touches it and libptk.so carries no
// load-time dependency on it.  Built only on the public entry points of ptk.h plus group-owned streams.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptk.h"

namespace {

__global__ __launch_bounds__(256) void k_group_add(float* __restrict__ dst, const float* __restrict__ src, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err)
    {
        if (handle) return true;
        for (const char* name : { "librccl.so.1", "librccl.so" })
            if ((handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!handle) {
            err = std::string("cannot load RCCL: ") + dlerror();
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(handle, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Reduce = reinterpret_cast<decltype(Reduce)>(sym("ncclReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !GetErrorString) {
            err = "RCCL library lacks a required symbol";
            return false;
        }
        return true;
    }
};

// [begin, begin + count) of `total` items for member m of n (same arithmetic as pathtracing_amd/parallel.py)
void split_range(uint64_t total, uint64_t m, uint64_t n, uint64_t& begin, uint64_t& count)
{
    begin = total * m / n;
    count = total * (m + 1) / n - begin;
}

} // namespace

struct ptk_group {
    std::vector<ptk_ctx*> ctx;
    std::vector<int> device;
    std::vector<int> leader;        // index of the first member on the same device
    std::vector<float*> sum;        // per member: W*H*3 floats on its device
    float* image = nullptr;         // member 0: the running-mean image
    std::vector<int> leaders;       // members that lead a device, member 0 first
    std::vector<hipStream_t> stream; // per leader (same order as `leaders`): exchange step
    std::vector<ncclComm_t> comm;   // per leader, when there is more than one
    Rccl rccl;
    bool use_rccl = false;          // more than one device, or PTK_GROUP_RCCL=1 (a one-device group then runs a one-rank reduce)
    uint32_t width = 0, height = 0;
    std::string err;
};

namespace {

int gfail(ptk_group* g, int code, const std::string& msg)
{
    if (g) g->err = msg;
    return code;
}
int from_ctx(ptk_group* g, int member, int rc)
{
    if (rc != PTK_OK) g->err = "member " + std::to_string(member) + ": " + ptk_last_error(g->ctx[member]);
    return rc;
}
#define PTKG_HIP(g, call)                                                                                \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return gfail((g), PTK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define PTKG_NCCL(g, call)                                                                               \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess) return gfail((g), PTK_ERR_HIP, std::string(#call) + ": " + (g)->rccl.GetErrorString(r_)); \
    } while (0)

void free_frames(ptk_group* g)
{
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (g->sum[m]) {
            ptk_dev_free(g->ctx[m], g->sum[m]);
            g->sum[m] = nullptr;
        }
    if (g->image) ptk_dev_free(g->ctx[0], g->image);
    g->image = nullptr;
    g->width = g->height = 0;
}

std::string g_group_create_error;

} // namespace

extern "C" {

int ptk_group_create(const int* device_ids, int n, ptk_group** out)
{
    if (!out) return PTK_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n < 1 || n > 64) {
        g_group_create_error = "ptk_group_create: 1 to 64 device ids expected";
        return PTK_ERR_INVALID;
    }
    ptk_group* g = new ptk_group();
    for (int m = 0; m < n; ++m) {
        ptk_ctx* c = nullptr;
        const int rc = ptk_ctx_create(device_ids[m], &c);
        if (rc != PTK_OK) {
            g_group_create_error = std::string("ptk_group_create: member ") + std::to_string(m) + ": " + ptk_last_error(nullptr);
            ptk_group_destroy(g);
            return rc;
        }
        g->ctx.push_back(c);
        g->device.push_back(device_ids[m]);
        g->sum.push_back(nullptr);
        int lead = m;
        for (int k = 0; k < m; ++k)
            if (device_ids[k] == device_ids[m]) {
                lead = k;
                break;
            }
        g->leader.push_back(lead);
        if (lead == m) g->leaders.push_back(m);
    }
    for (int l : g->leaders) {
        hipStream_t s = nullptr;
        if (hipSetDevice(g->device[l]) != hipSuccess || hipStreamCreate(&s) != hipSuccess) {
            g_group_create_error = "ptk_group_create: cannot create the exchange stream";
            ptk_group_destroy(g);
            return PTK_ERR_HIP;
        }
        g->stream.push_back(s);
    }
    // More than one device: RCCL communicators, one per device.  PTK_GROUP_RCCL=1 asks for them with a single device
    // too, so that a one-GPU machine can run the whole exchange path (library load, communicator, grouped reduce on
    // the exchange stream) with one rank.
    const char* force = std::getenv("PTK_GROUP_RCCL");
    g->use_rccl = g->leaders.size() > 1 || (force && force[0] == '1');
    if (g->use_rccl) {
        std::string err;
        if (!g->rccl.load(err)) {
            g_group_create_error = "ptk_group_create: " + err;
            ptk_group_destroy(g);
            return PTK_ERR_HIP;
        }
        std::vector<int> devs;
        for (int l : g->leaders) devs.push_back(g->device[l]);
        g->comm.resize(devs.size());
        const ncclResult_t r = g->rccl.CommInitAll(g->comm.data(), (int)devs.size(), devs.data());
        if (r != ncclSuccess) {
            g_group_create_error = std::string("ptk_group_create: ncclCommInitAll: ") + g->rccl.GetErrorString(r);
            g->comm.clear();
            ptk_group_destroy(g);
            return PTK_ERR_HIP;
        }
    }
    *out = g;
    return PTK_OK;
}

void ptk_group_destroy(ptk_group* g)
{
    if (!g) return;
    for (ptk_ctx* c : g->ctx) (void)ptk_ctx_synchronize(c);
    free_frames(g);
    for (ncclComm_t c : g->comm) (void)g->rccl.CommDestroy(c);
    for (size_t i = 0; i < g->stream.size(); ++i) {
        (void)hipSetDevice(g->device[g->leaders[i]]);
        (void)hipStreamDestroy(g->stream[i]);
    }
    for (ptk_ctx* c : g->ctx) ptk_ctx_destroy(c);
    delete g;
}

const char* ptk_group_last_error(const ptk_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int ptk_group_size(const ptk_group* g) { return g ? (int)g->ctx.size() : 0; }

ptk_ctx* ptk_group_ctx(ptk_group* g, int member)
{
    return (g && member >= 0 && member < (int)g->ctx.size()) ? g->ctx[member] : nullptr;
}

int ptk_group_scene_upload(ptk_group* g, const ptk_scene_desc* scene)
{
    if (!g || !scene) return PTK_ERR_INVALID;
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (int rc = from_ctx(g, (int)m, ptk_scene_upload(g->ctx[m], scene))) return rc;
    return PTK_OK;
}

int ptk_group_scene_set_camera(ptk_group* g, const ptk_camera_desc* camera)
{
    if (!g || !camera) return PTK_ERR_INVALID;
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (int rc = from_ctx(g, (int)m, ptk_scene_set_camera(g->ctx[m], camera))) return rc;
    return PTK_OK;
}

int ptk_group_render_begin(ptk_group* g, uint32_t width, uint32_t height, float fov_y, uint64_t max_paths_per_launch)
{
    if (!g) return PTK_ERR_INVALID;
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (int rc = from_ctx(g, (int)m, ptk_render_begin(g->ctx[m], width, height, fov_y, max_paths_per_launch))) return rc;
    const uint64_t bytes = (uint64_t)width * height * 3 * sizeof(float);
    if (width != g->width || height != g->height) {
        for (ptk_ctx* c : g->ctx) (void)ptk_ctx_synchronize(c);
        free_frames(g);
        for (size_t m = 0; m < g->ctx.size(); ++m)
            if (int rc = from_ctx(g, (int)m, ptk_dev_alloc(g->ctx[m], bytes, reinterpret_cast<void**>(&g->sum[m])))) return rc;
        if (int rc = from_ctx(g, 0, ptk_dev_alloc(g->ctx[0], bytes, reinterpret_cast<void**>(&g->image)))) return rc;
        g->width = width;
        g->height = height;
    }
    // a new run starts from a black image (abstract_pathtracer.cpp:18)
    if (int rc = from_ctx(g, 0, ptk_dev_memset(g->ctx[0], g->image, 0, bytes))) return rc;
    return PTK_OK;
}

static int group_iteration(ptk_group* g, int aov, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration);

int ptk_group_render_iteration(ptk_group* g, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration)
{
    return group_iteration(g, -1, seed, iteration, samples_per_iteration);
}

int ptk_group_render_aov_iteration(ptk_group* g, uint32_t aov, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration)
{
    if (aov > PTK_AOV_NORMAL) return gfail(g, PTK_ERR_INVALID, "ptk_group_render_aov_iteration: unknown AOV");
    return group_iteration(g, (int)aov, seed, iteration, samples_per_iteration);
}

// aov < 0: the integrator (ptk_render_samples_dev); otherwise ptk_render_aov_dev
static int group_iteration(ptk_group* g, int aov, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration)
{
    if (!g) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_render_iteration: call ptk_group_render_begin first");
    if (samples_per_iteration == 0 || samples_per_iteration > 0xFFFFFFFFull)
        return gfail(g, PTK_ERR_INVALID, "ptk_group_render_iteration: samples_per_iteration must be in [1, 2^32)");
    const uint64_t n = g->ctx.size();
    const uint64_t spi = samples_per_iteration;
    const uint64_t num_pixels = (uint64_t)g->width * g->height;
    const uint64_t floats = num_pixels * 3;
    // every member's share, enqueued on its own context (they run concurrently)
    for (uint64_t m = 0; m < n; ++m) {
        uint64_t p0 = 0, np = num_pixels, s0 = 0, ns = spi;
        if (spi >= n) split_range(spi, m, n, s0, ns);
        else split_range(num_pixels, m, n, p0, np);
        if (n > 1 && (np != num_pixels || ns == 0)) // pixels outside the share contribute 0 to the reduce
            if (int rc = from_ctx(g, (int)m, ptk_dev_memset(g->ctx[m], g->sum[m], 0, floats * sizeof(float)))) return rc;
        const int rc = aov < 0 ? ptk_render_samples_dev(g->ctx[m], seed, p0, np, iteration * spi + s0, (uint32_t)ns, g->sum[m], 0)
                               : ptk_render_aov_dev(g->ctx[m], (uint32_t)aov, seed, p0, np, iteration * spi + s0, (uint32_t)ns, g->sum[m], 0);
        if (from_ctx(g, (int)m, rc)) return rc;
    }
    for (uint64_t m = 0; m < n; ++m)
        if (int rc = from_ctx(g, (int)m, ptk_ctx_synchronize(g->ctx[m]))) return rc;
    // exchange, step 1: members sharing a device fold into that device's first member
    for (size_t i = 0; i < g->leaders.size(); ++i) {
        const int l = g->leaders[i];
        bool any = false;
        for (uint64_t m = 0; m < n; ++m)
            if (g->leader[m] == l && (int)m != l) {
                if (!any) PTKG_HIP(g, hipSetDevice(g->device[l]));
                any = true;
                k_group_add<<<dim3((uint32_t)((floats + 255) / 256)), dim3(256), 0, g->stream[i]>>>(g->sum[l], g->sum[m], floats);
            }
        if (any) {
            PTKG_HIP(g, hipGetLastError());
            PTKG_HIP(g, hipStreamSynchronize(g->stream[i]));
        }
    }
    // step 2: one reduce(sum) of the per-device partial sums to member 0 over RCCL
    if (g->use_rccl) {
        PTKG_NCCL(g, g->rccl.GroupStart());
        for (size_t i = 0; i < g->leaders.size(); ++i) {
            const int l = g->leaders[i];
            PTKG_HIP(g, hipSetDevice(g->device[l]));
            PTKG_NCCL(g, g->rccl.Reduce(g->sum[l], g->sum[l], floats, ncclFloat, ncclSum, 0, g->comm[i], g->stream[i]));
        }
        PTKG_NCCL(g, g->rccl.GroupEnd());
        for (size_t i = 0; i < g->leaders.size(); ++i) {
            PTKG_HIP(g, hipSetDevice(g->device[g->leaders[i]]));
            PTKG_HIP(g, hipStreamSynchronize(g->stream[i]));
        }
    }
    // abstract_pathtracer.cpp:88-90 on member 0
    return from_ctx(g, 0, ptk_resolve_iteration_dev(g->ctx[0], g->sum[0], g->image, spi, iteration));
}

int ptk_group_read_image(ptk_group* g, float* image_rgb)
{
    if (!g || !image_rgb) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_read_image: call ptk_group_render_begin first");
    return from_ctx(g, 0, ptk_dev_download(g->ctx[0], image_rgb, g->image, (uint64_t)g->width * g->height * 3 * sizeof(float)));
}

int ptk_group_get_counters(ptk_group* g, ptk_counters* out)
{
    if (!g || !out) return PTK_ERR_INVALID;
    out->paths = out->rays = 0;
    for (size_t m = 0; m < g->ctx.size(); ++m) {
        ptk_counters c{};
        if (int rc = from_ctx(g, (int)m, ptk_get_counters(g->ctx[m], &c))) return rc;
        out->paths += c.paths;
        out->rays += c.rays;
    }
    return PTK_OK;
}

} // extern "C"
