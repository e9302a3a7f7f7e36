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
//
// Nothing here waits for the GPU except the two wait calls.  An iteration is a list of SLICES (pixel ranges sized
// so that every member's part of a slice is one launch chain); the driver submits slices, waits for the one
// submitted two before (at most a few in flight: that is what makes a long iteration interruptible and gives it
// a progress read-out), and finishes the iteration with one asynchronous exchange:
//   member streams --share event--> exchange stream of the device's first member: add, ncclReduce -->
//   member 0's exchange stream: running mean over the finished pixel range, image and counters to pinned memory.
// The sum images are double-buffered by iteration parity, so the exchange of iteration i runs under the render
// of iteration i + 1 (which starts on the other buffer); iteration i + 2 waits, on the device, for the event that
// says the exchange of iteration i has let go of its buffer.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptk.h"

namespace {

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

} // namespace

constexpr int kRing = 8; // slice events kept: a driver may have this many slices in flight

struct ptk_group {
    std::vector<ptk_ctx*> ctx;
    std::vector<int> device;
    std::vector<int> leader;        // index of the first member on the same device
    std::vector<int> leader_slot;   // per member: position of its leader in `leaders`
    std::vector<float*> sum[2];     // per member, per iteration parity: W*H*3 floats on its device
    float* image = nullptr;         // member 0: the running-mean image
    float* host_image[2] = { nullptr, nullptr };           // pinned: the image after the iteration of that parity
    unsigned long long* host_counters[2] = { nullptr, nullptr }; // pinned: [member][paths, rays] behind that iteration
    std::vector<int> leaders;       // members that lead a device, member 0 first
    std::vector<hipStream_t> rstream; // per member: the stream its context renders on
    std::vector<hipStream_t> stream;  // per leader (same order as `leaders`): exchange step
    std::vector<ncclComm_t> comm;   // per leader, when there is more than one
    std::vector<hipEvent_t> slice_ev[kRing]; // per member: the slice submitted with ticket t (slot t % kRing) has finished
    std::vector<hipEvent_t> share_ev[2];     // per member: its share of the iteration of that parity has finished
    std::vector<hipEvent_t> free_ev[2];      // per leader: its exchange stream has let go of that parity's sum images
    hipEvent_t done_ev[2] = { nullptr, nullptr }; // member 0's exchange stream: iteration resolved (and copied out)
    bool free_recorded[2] = { false, false };
    bool done_recorded[2] = { false, false };
    uint64_t done_iteration[2] = { 0, 0 };
    uint64_t tickets = 0;           // slices submitted so far
    Rccl rccl;
    bool use_rccl = false;          // more than one device, or PTK_GROUP_RCCL=1 (a one-device group then runs a one-rank reduce)
    uint32_t width = 0, height = 0;
    uint64_t cap = 0;               // paths per launch chain of a member (ptk_render_begin's max_paths_per_launch)
    std::atomic<int> interrupt{ 0 };
    std::atomic<float> progress{ 0.0f };
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

// Every group call leaves the caller's current HIP device as it found it.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

void free_frames(ptk_group* g)
{
    for (int par = 0; par < 2; ++par) {
        for (size_t m = 0; m < g->ctx.size(); ++m)
            if (m < g->sum[par].size() && g->sum[par][m]) {
                ptk_dev_free(g->ctx[m], g->sum[par][m]);
                g->sum[par][m] = nullptr;
            }
        if (g->host_image[par]) (void)hipHostFree(g->host_image[par]);
        g->host_image[par] = nullptr;
    }
    if (g->image) ptk_dev_free(g->ctx[0], g->image);
    g->image = nullptr;
    g->width = g->height = 0;
}

// every stream of the group idle (render streams and exchange streams)
void drain(ptk_group* g)
{
    for (ptk_ctx* c : g->ctx) (void)ptk_ctx_synchronize(c);
    for (size_t i = 0; i < g->stream.size(); ++i) {
        (void)hipSetDevice(g->device[g->leaders[i]]);
        (void)hipStreamSynchronize(g->stream[i]);
    }
}

// [begin, begin + count) of `total` items for part m of n (same arithmetic as pathtracing_amd/parallel.py)
void split_range(uint64_t total, uint64_t m, uint64_t n, uint64_t& begin, uint64_t& count)
{
    begin = total * m / n;
    count = total * (m + 1) / n - begin;
}

// Member m's share of one iteration: a sample range of every pixel, or (fewer samples than members) a pixel range.
struct Share {
    uint64_t p0, np, s0, ns;
};
Share share_of(const ptk_group* g, uint64_t spi, uint64_t m)
{
    const uint64_t n = g->ctx.size(), num_pixels = (uint64_t)g->width * g->height;
    Share s{ 0, num_pixels, 0, spi };
    if (spi >= n) split_range(spi, m, n, s.s0, s.ns);
    else split_range(num_pixels, m, n, s.p0, s.np);
    return s;
}

// Slices per iteration: every member's part of a slice (its pixels x its samples) fits one launch chain of
// g->cap paths, so a slice is the unit the GPU works through without coming back to the host.
uint32_t slice_count(const ptk_group* g, uint64_t spi)
{
    const uint64_t n = g->ctx.size();
    uint64_t k = 1;
    for (uint64_t m = 0; m < n; ++m) {
        const Share s = share_of(g, spi, m);
        if (s.np == 0 || s.ns == 0) continue;
        const uint64_t per_pixel = s.ns < g->cap ? s.ns : g->cap;
        const uint64_t budget = g->cap / per_pixel; // pixels per slice, >= 1
        uint64_t need = (s.np + budget - 1) / budget;
        if (need > s.np) need = s.np;
        if (need > k) k = need;
    }
    return (uint32_t)(k > 0xFFFFFFFFull ? 0xFFFFFFFFull : k);
}

std::string g_group_create_error;

__global__ __launch_bounds__(256) void k_group_add(float* __restrict__ dst, const float* __restrict__ src, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

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
    DeviceGuard guard;
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
        g->rstream.push_back(static_cast<hipStream_t>(ptk_ctx_get_stream(c)));
        int lead = m;
        for (int k = 0; k < m; ++k)
            if (device_ids[k] == device_ids[m]) {
                lead = k;
                break;
            }
        g->leader.push_back(lead);
        if (lead == m) g->leaders.push_back(m);
    }
    g->sum[0].assign(n, nullptr);
    g->sum[1].assign(n, nullptr);
    for (int m = 0; m < n; ++m)
        for (size_t i = 0; i < g->leaders.size(); ++i)
            if (g->leaders[i] == g->leader[m]) g->leader_slot.push_back((int)i);
    auto fail_create = [&](const char* what) {
        g_group_create_error = std::string("ptk_group_create: ") + what;
        ptk_group_destroy(g);
        return PTK_ERR_HIP;
    };
    for (int l : g->leaders) {
        hipStream_t s = nullptr;
        if (hipSetDevice(g->device[l]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
            return fail_create("cannot create the exchange stream");
        g->stream.push_back(s);
    }
    // events live on the device of the stream that records them
    auto make_event = [&](int device, hipEvent_t* ev) {
        return hipSetDevice(device) == hipSuccess && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
    };
    for (int r = 0; r < kRing; ++r) g->slice_ev[r].assign(n, nullptr);
    for (int par = 0; par < 2; ++par) {
        g->share_ev[par].assign(n, nullptr);
        g->free_ev[par].assign(g->leaders.size(), nullptr);
    }
    for (int m = 0; m < n; ++m) {
        for (int r = 0; r < kRing; ++r)
            if (!make_event(g->device[m], &g->slice_ev[r][m])) return fail_create("cannot create an event");
        for (int par = 0; par < 2; ++par)
            if (!make_event(g->device[m], &g->share_ev[par][m])) return fail_create("cannot create an event");
    }
    for (int par = 0; par < 2; ++par) {
        for (size_t i = 0; i < g->leaders.size(); ++i)
            if (!make_event(g->device[g->leaders[i]], &g->free_ev[par][i])) return fail_create("cannot create an event");
        if (!make_event(g->device[0], &g->done_ev[par])) return fail_create("cannot create an event");
        if (hipHostMalloc((void**)&g->host_counters[par], (size_t)n * 2 * sizeof(unsigned long long)) != hipSuccess)
            return fail_create("cannot allocate pinned host memory");
        std::memset(g->host_counters[par], 0, (size_t)n * 2 * sizeof(unsigned long long));
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
    DeviceGuard guard;
    drain(g);
    free_frames(g);
    for (ncclComm_t c : g->comm) (void)g->rccl.CommDestroy(c);
    for (size_t i = 0; i < g->stream.size(); ++i) {
        (void)hipSetDevice(g->device[g->leaders[i]]);
        (void)hipStreamDestroy(g->stream[i]);
    }
    auto drop = [](std::vector<hipEvent_t>& v) {
        for (hipEvent_t e : v)
            if (e) (void)hipEventDestroy(e);
        v.clear();
    };
    for (int r = 0; r < kRing; ++r) drop(g->slice_ev[r]);
    for (int par = 0; par < 2; ++par) {
        drop(g->share_ev[par]);
        drop(g->free_ev[par]);
        if (g->done_ev[par]) (void)hipEventDestroy(g->done_ev[par]);
        if (g->host_counters[par]) (void)hipHostFree(g->host_counters[par]);
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
    DeviceGuard guard;
    drain(g);
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (int rc = from_ctx(g, (int)m, ptk_scene_upload(g->ctx[m], scene))) return rc;
    return PTK_OK;
}

int ptk_group_scene_set_camera(ptk_group* g, const ptk_camera_desc* camera)
{
    if (!g || !camera) return PTK_ERR_INVALID;
    DeviceGuard guard;
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (int rc = from_ctx(g, (int)m, ptk_scene_set_camera(g->ctx[m], camera))) return rc;
    return PTK_OK;
}

int ptk_group_render_begin(ptk_group* g, uint32_t width, uint32_t height, float fov_y, uint64_t max_paths_per_launch)
{
    if (!g) return PTK_ERR_INVALID;
    DeviceGuard guard;
    drain(g); // a new run: nothing of the previous one is in flight any more
    for (size_t m = 0; m < g->ctx.size(); ++m)
        if (int rc = from_ctx(g, (int)m, ptk_render_begin(g->ctx[m], width, height, fov_y, max_paths_per_launch))) return rc;
    uint64_t cap = max_paths_per_launch ? max_paths_per_launch : (1ull << 24); // as ptk_render_begin sizes its queues
    if (cap > 0x7FFFFF00ull) cap = 0x7FFFFF00ull;
    g->cap = (cap + 255) / 256 * 256;
    const uint64_t bytes = (uint64_t)width * height * 3 * sizeof(float);
    if (width != g->width || height != g->height) {
        free_frames(g);
        for (int par = 0; par < 2; ++par) {
            for (size_t m = 0; m < g->ctx.size(); ++m)
                if (int rc = from_ctx(g, (int)m, ptk_dev_alloc(g->ctx[m], bytes, reinterpret_cast<void**>(&g->sum[par][m])))) return rc;
            PTKG_HIP(g, hipHostMalloc((void**)&g->host_image[par], bytes));
        }
        if (int rc = from_ctx(g, 0, ptk_dev_alloc(g->ctx[0], bytes, reinterpret_cast<void**>(&g->image)))) return rc;
        g->width = width;
        g->height = height;
    }
    // a new run starts from a black image (abstract_pathtracer.cpp:18)
    if (int rc = from_ctx(g, 0, ptk_dev_memset(g->ctx[0], g->image, 0, bytes))) return rc;
    if (int rc = from_ctx(g, 0, ptk_ctx_synchronize(g->ctx[0]))) return rc; // the exchange stream resolves into it
    g->free_recorded[0] = g->free_recorded[1] = false;
    g->done_recorded[0] = g->done_recorded[1] = false;
    g->interrupt = 0;
    g->progress = 0.0f;
    return PTK_OK;
}

int ptk_group_iteration_slices(ptk_group* g, uint64_t samples_per_iteration, uint32_t* num_slices)
{
    if (!g || !num_slices) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_iteration_slices: call ptk_group_render_begin first");
    *num_slices = samples_per_iteration ? slice_count(g, samples_per_iteration) : 1u;
    return PTK_OK;
}

int ptk_group_submit_slice(ptk_group* g, int aov, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration,
                           uint32_t slice, uint32_t num_slices, uint64_t* ticket)
{
    if (!g) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_submit_slice: call ptk_group_render_begin first");
    if (samples_per_iteration == 0 || num_slices == 0 || slice >= num_slices || aov > (int)PTK_AOV_NORMAL)
        return gfail(g, PTK_ERR_INVALID, "ptk_group_submit_slice: bad arguments");
    DeviceGuard guard;
    const uint64_t n = g->ctx.size(), spi = samples_per_iteration;
    const uint64_t floats = (uint64_t)g->width * g->height * 3;
    const int par = (int)(iteration & 1u);
    for (uint64_t m = 0; m < n; ++m) {
        const Share s = share_of(g, spi, m);
        float* sum = g->sum[par][m];
        PTKG_HIP(g, hipSetDevice(g->device[m]));
        if (slice == 0u) {
            // the exchange of iteration - 2 read (and, on a leader, wrote) this buffer: wait for it on the device
            if (g->free_recorded[par]) PTKG_HIP(g, hipStreamWaitEvent(g->rstream[m], g->free_ev[par][g->leader_slot[m]], 0));
            if (n > 1 && (s.np != (uint64_t)g->width * g->height || s.ns == 0)) // pixels outside the share add 0 to the reduce
                if (int rc = from_ctx(g, (int)m, ptk_dev_memset(g->ctx[m], sum, 0, floats * sizeof(float)))) return rc;
        }
        uint64_t b0, bn;
        split_range(s.np, slice, num_slices, b0, bn);
        // (a sample range of 2^32 or more goes out in parts that add up in the sum image)
        for (uint64_t done = 0; done < s.ns || done == 0; ) {
            const uint64_t part = s.ns - done < 0x40000000ull ? s.ns - done : 0x40000000ull;
            const int acc = done > 0 ? 1 : 0;
            const int rc = aov < 0 ? ptk_render_samples_dev(g->ctx[m], seed, s.p0 + b0, bn, iteration * spi + s.s0 + done, (uint32_t)part, sum, acc)
                                   : ptk_render_aov_dev(g->ctx[m], (uint32_t)aov, seed, s.p0 + b0, bn, iteration * spi + s.s0 + done, (uint32_t)part, sum, acc);
            if (from_ctx(g, (int)m, rc)) return rc;
            done += part;
            if (part == 0) break;
        }
        PTKG_HIP(g, hipEventRecord(g->slice_ev[g->tickets % kRing][m], g->rstream[m]));
    }
    if (ticket) *ticket = g->tickets;
    ++g->tickets;
    return PTK_OK;
}

int ptk_group_wait_slice(ptk_group* g, uint64_t ticket)
{
    if (!g) return PTK_ERR_INVALID;
    if (ticket >= g->tickets || g->tickets - ticket > (uint64_t)kRing)
        return gfail(g, PTK_ERR_INVALID, "ptk_group_wait_slice: only the last " + std::to_string(kRing) + " submitted slices can be waited for");
    for (size_t m = 0; m < g->ctx.size(); ++m) PTKG_HIP(g, hipEventSynchronize(g->slice_ev[ticket % kRing][m]));
    return PTK_OK;
}

int ptk_group_finish_iteration(ptk_group* g, uint64_t iteration, uint64_t samples_per_iteration, uint32_t slices_done,
                               uint32_t num_slices, int download)
{
    if (!g) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_finish_iteration: call ptk_group_render_begin first");
    if (samples_per_iteration == 0 || num_slices == 0 || slices_done > num_slices)
        return gfail(g, PTK_ERR_INVALID, "ptk_group_finish_iteration: bad arguments");
    DeviceGuard guard;
    const uint64_t n = g->ctx.size(), spi = samples_per_iteration;
    const uint64_t num_pixels = (uint64_t)g->width * g->height, floats = num_pixels * 3;
    const int par = (int)(iteration & 1u);
    // every member: counters behind its share (pinned memory, read after the iteration's done event), then the share event
    for (uint64_t m = 0; m < n; ++m) {
        PTKG_HIP(g, hipSetDevice(g->device[m]));
        if (int rc = from_ctx(g, (int)m, ptk_counters_download_async(g->ctx[m], reinterpret_cast<ptk_counters*>(g->host_counters[par] + 2 * m)))) return rc;
        PTKG_HIP(g, hipEventRecord(g->share_ev[par][m], g->rstream[m]));
    }
    if (slices_done > 0u) {
        // exchange, step 1: members sharing a device fold into that device's first member, on its exchange stream
        for (size_t i = 0; i < g->leaders.size(); ++i) {
            const int l = g->leaders[i];
            PTKG_HIP(g, hipSetDevice(g->device[l]));
            for (uint64_t m = 0; m < n; ++m)
                if (g->leader[m] == l) PTKG_HIP(g, hipStreamWaitEvent(g->stream[i], g->share_ev[par][m], 0));
            for (uint64_t m = 0; m < n; ++m)
                if (g->leader[m] == l && (int)m != l)
                    k_group_add<<<dim3((uint32_t)((floats + 255) / 256)), dim3(256), 0, g->stream[i]>>>(g->sum[par][l], g->sum[par][m], floats);
            PTKG_HIP(g, hipGetLastError());
        }
        // step 2: one reduce(sum) of the per-device partial sums to member 0 over RCCL
        if (g->use_rccl) {
            ncclResult_t r = g->rccl.GroupStart();
            hipError_t he = hipSuccess;
            if (r == ncclSuccess) {
                for (size_t i = 0; i < g->leaders.size() && r == ncclSuccess && he == hipSuccess; ++i) {
                    const int l = g->leaders[i];
                    he = hipSetDevice(g->device[l]);
                    if (he == hipSuccess)
                        r = g->rccl.Reduce(g->sum[par][l], g->sum[par][l], floats, ncclFloat, ncclSum, 0, g->comm[i], g->stream[i]);
                }
                // the bracket is closed whatever happened inside it: a group left open would swallow every later RCCL call
                const ncclResult_t re = g->rccl.GroupEnd();
                if (r == ncclSuccess) r = re;
            }
            if (he != hipSuccess) return gfail(g, PTK_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
            if (r != ncclSuccess) return gfail(g, PTK_ERR_HIP, std::string("ncclReduce: ") + g->rccl.GetErrorString(r));
        }
        // abstract_pathtracer.cpp:88-90 on member 0's exchange stream, over the pixels whose slices were rendered
        // (a partial iteration leaves the other pixels as they were, like the reference's unfinished scan lines)
        if (int rc = from_ctx(g, 0, ptk_ctx_set_stream(g->ctx[0], g->stream[0]))) return rc;
        int rc = PTK_OK;
        for (uint64_t m = 0; m < n && rc == PTK_OK; ++m) {
            const Share s = share_of(g, spi, m);
            if (s.np == num_pixels && m > 0) break; // sample split: every member covers the same pixels
            uint64_t b0, bn;
            split_range(s.np, slices_done, num_slices, b0, bn); // pixels before slice `slices_done` of the share
            rc = ptk_resolve_range_dev(g->ctx[0], g->sum[par][0], g->image, s.p0, b0, spi, iteration);
        }
        const int rc2 = ptk_ctx_set_stream(g->ctx[0], g->rstream[0]);
        if (from_ctx(g, 0, rc)) return rc;
        if (from_ctx(g, 0, rc2)) return rc2;
    } else {
        PTKG_HIP(g, hipSetDevice(g->device[0]));
        PTKG_HIP(g, hipStreamWaitEvent(g->stream[0], g->share_ev[par][0], 0));
    }
    PTKG_HIP(g, hipSetDevice(g->device[0]));
    if (download) PTKG_HIP(g, hipMemcpyAsync(g->host_image[par], g->image, floats * sizeof(float), hipMemcpyDeviceToHost, g->stream[0]));
    PTKG_HIP(g, hipEventRecord(g->done_ev[par], g->stream[0]));
    for (size_t i = 0; i < g->leaders.size(); ++i) {
        PTKG_HIP(g, hipSetDevice(g->device[g->leaders[i]]));
        PTKG_HIP(g, hipEventRecord(g->free_ev[par][i], g->stream[i]));
    }
    g->free_recorded[par] = true;
    g->done_recorded[par] = true;
    g->done_iteration[par] = iteration;
    return PTK_OK;
}

int ptk_group_wait_iteration(ptk_group* g, uint64_t iteration, float* image_rgb, ptk_counters* counters)
{
    if (!g) return PTK_ERR_INVALID;
    const int par = (int)(iteration & 1u);
    if (!g->done_recorded[par] || g->done_iteration[par] != iteration)
        return gfail(g, PTK_ERR_STATE, "ptk_group_wait_iteration: that iteration is not the last one finished on its buffer");
    PTKG_HIP(g, hipEventSynchronize(g->done_ev[par]));
    if (image_rgb) std::memcpy(image_rgb, g->host_image[par], (size_t)g->width * g->height * 3 * sizeof(float));
    if (counters) {
        counters->paths = counters->rays = 0;
        for (size_t m = 0; m < g->ctx.size(); ++m) {
            counters->paths += g->host_counters[par][2 * m];
            counters->rays += g->host_counters[par][2 * m + 1];
        }
    }
    return PTK_OK;
}

int ptk_group_request_interrupt(ptk_group* g)
{
    if (!g) return PTK_ERR_INVALID;
    g->interrupt = 1;
    return PTK_OK;
}

float ptk_group_progress(const ptk_group* g) { return g ? g->progress.load() : 0.0f; }

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

// The blocking form: the slices of one iteration with two in flight, the interrupt flag looked at before each
// (abstract_pathtracer.cpp:63-72 looks per scan line, pixel and sample), progress after each (abstract_pathtracer.cpp:51).
// aov < 0: the integrator (ptk_render_samples_dev); otherwise ptk_render_aov_dev.
static int group_iteration(ptk_group* g, int aov, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration)
{
    if (!g) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_render_iteration: call ptk_group_render_begin first");
    if (samples_per_iteration == 0) return gfail(g, PTK_ERR_INVALID, "ptk_group_render_iteration: samples_per_iteration must be at least 1");
    constexpr uint32_t kWindow = 2;
    const uint32_t K = slice_count(g, samples_per_iteration);
    uint64_t tickets[kWindow] = {};
    uint32_t submitted = 0, waited = 0;
    g->progress = 0.0f;
    bool interrupted = false;
    while (waited < submitted || (submitted < K && !interrupted)) {
        if (submitted < K && !interrupted && submitted - waited < kWindow) {
            if (g->interrupt.load()) {
                interrupted = true;
                continue;
            }
            if (int rc = ptk_group_submit_slice(g, aov, seed, iteration, samples_per_iteration, submitted, K, &tickets[submitted % kWindow])) return rc;
            ++submitted;
            continue;
        }
        if (int rc = ptk_group_wait_slice(g, tickets[waited % kWindow])) return rc;
        ++waited;
        g->progress = (float)waited / (float)K;
    }
    if (int rc = ptk_group_finish_iteration(g, iteration, samples_per_iteration, submitted, K, 0)) return rc;
    if (int rc = ptk_group_wait_iteration(g, iteration, nullptr, nullptr)) return rc;
    return interrupted ? PTK_INTERRUPTED : PTK_OK;
}

int ptk_group_read_image(ptk_group* g, float* image_rgb)
{
    if (!g || !image_rgb) return PTK_ERR_INVALID;
    if (!g->image) return gfail(g, PTK_ERR_STATE, "ptk_group_read_image: call ptk_group_render_begin first");
    DeviceGuard guard;
    // the image is written on member 0's exchange stream
    for (int par = 0; par < 2; ++par)
        if (g->done_recorded[par]) PTKG_HIP(g, hipEventSynchronize(g->done_ev[par]));
    return from_ctx(g, 0, ptk_dev_download(g->ctx[0], image_rgb, g->image, (uint64_t)g->width * g->height * 3 * sizeof(float)));
}

int ptk_group_get_counters(ptk_group* g, ptk_counters* out)
{
    if (!g || !out) return PTK_ERR_INVALID;
    DeviceGuard guard;
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
