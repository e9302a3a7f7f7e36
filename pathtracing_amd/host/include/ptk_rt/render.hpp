// ptk_rt/render.hpp — host mirror of the reference's renderer API
// (reference src/Pathtracing/abstract_pathtracer.h:27-200, pathtracer.h:13-22,
// rng.h:12-41): event_emitter, view_parameters, trace_parameters, pathtracer_result,
// abstract_pathtracer, pathtracer, rng.
//
// run() keeps the reference's contract (abstract_pathtracer.cpp:11-119): scene.compile()
// on the caller's thread, then a driver thread that fires on_iteration_start /
// on_iteration_end / on_end.  When the object is exactly an rt::pathtracer, the three
// hot loops (abstract_pathtracer.cpp:63-90) run on the GPU through the C ABI of
// include/ptk.h; there is no CPU path for it (a missing/failed libptk is an error,
// reported through pathtracer_result::error()).  Any other subclass (e.g. the
// reference's debug_pathtracer) is driven on host threads through its virtual trace().
#pragma once
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <functional>
#include <list>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <glm/glm.hpp>
#include <glm/ext.hpp>

#include "samplers.hpp"
#include "scene_graph.hpp"

namespace rt {

namespace detail {
struct device_session;
}

template <typename... Args>
class event_emitter { // abstract_pathtracer.h:27-85
public:
    using handler_fn = std::function<void(Args...)>;
    template <typename T>
    using member_handler_fn = void (T::*)(Args...);

    void subscribe(const handler_fn& fn)
    {
        std::lock_guard<std::mutex> g(m_lock);
        m_handlers.push_back(fn);
    }

    template <typename T>
    void subscribe(T* instance, member_handler_fn<T> fn)
    {
        subscribe([instance, fn](Args... args) { (instance->*fn)(std::forward<Args>(args)...); });
    }

    template <typename... EvtArgs>
    void emit(EvtArgs&&... evt)
    {
        // Both reference callers subscribe after the driver thread has started
        // (CLI/main.cpp:72-83, rtsb.cpp:355-356); the list is walked under a lock so a
        // late subscribe cannot race the traversal.
        std::unique_lock<std::mutex> g(m_lock);
        const size_t n = m_handlers.size();
        auto it = m_handlers.begin();
        for (size_t i = 0; i < n; ++i, ++it) {
            handler_fn& h = *it; // std::list: stable across push_back
            g.unlock();
            h(std::forward<EvtArgs>(evt)...);
            g.lock();
        }
    }

    template <typename... EvtArgs>
    void operator()(EvtArgs&&... evt)
    {
        emit(std::forward<EvtArgs>(evt)...);
    }

private:
    std::list<handler_fn> m_handlers;
    std::mutex m_lock;
};

struct view_parameters { // abstract_pathtracer.h:90-95
    uint32_t width = 800;
    uint32_t height = 600;
    float fov_y = glm::pi<float>() / 4.0f;
};

struct trace_parameters { // abstract_pathtracer.h:100-105
    uint32_t num_threads = 4; // host subclasses only; the GPU integrator ignores it
    uint64_t iterations = 1;  // 0 = until interrupt()
    uint64_t samples_per_iteration = 1;
};

class pathtracer_result { // abstract_pathtracer.h:111-174
public:
    using fn = std::function<void(pathtracer_result&)>;

    std::atomic<float> progress = 0.0f;
    std::atomic_uint64_t iteration = 0;
    std::atomic_uint64_t samples_per_pixel = 0;

    pathtracer_result(const fn& fn);
    ~pathtracer_result();

    void wait()
    {
        if (m_thread.joinable()) m_thread.join();
    }
    void interrupt() { m_interrupted = true; }
    bool is_interrupted() const { return m_interrupted; }
    float get_elapsed_time() const;

    event_emitter<const uint64_t&> on_iteration_start;
    event_emitter<const image&, const uint64_t&> on_iteration_end;
    event_emitter<const image&> on_end;

    // Additive (the reference has no error channel): empty unless the GPU driver failed.
    std::string error() const;
    void set_error(const std::string& msg);
    // Additive: device counters of the finished part of the run (0 for host subclasses).
    std::atomic_uint64_t paths_traced = 0;
    std::atomic_uint64_t rays_cast = 0;

private:
    std::thread m_thread;
    std::atomic_bool m_interrupted = false;
    std::chrono::system_clock::time_point m_start_time;
    mutable std::mutex m_error_lock;
    std::string m_error;
};

class abstract_pathtracer { // abstract_pathtracer.h:179-200
public:
    virtual ~abstract_pathtracer() = default;
    std::shared_ptr<pathtracer_result> run(const view_parameters& view_params, const trace_parameters& trace_params,
                                           scene& scene);
    virtual glm::vec3 trace(const view_parameters& params, const ray& ray, const scene& scene) = 0;
};

class pathtracer : public abstract_pathtracer { // pathtracer.h:13-22
public:
    glm::vec3 trace(const view_parameters& params, const ray& r, const scene& scene) override;

    // Additive GPU controls (defaults reproduce the stock behaviour on device 0).
    struct device_options {
        int device = 0;                    // HIP device ordinal
        std::vector<int> devices;          // non-empty: render on these devices together (ids may repeat); overrides `device`
        uint64_t seed = 0;                 // Philox key; same seed => same image
        uint64_t max_paths_per_launch = 0; // ray-queue capacity, 0 = library default (2^24)
        bool keep_scene = true;            // keep the uploaded tables between run() calls (camera-only changes skip the upload)
    };
    device_options device;
    // Additive: what the last run() did before its driver thread started.
    struct run_info {
        bool scene_reused = false; // the previous run()'s device tables were kept (same scene, maybe another camera)
        double prepare_ms = 0.0;   // scene hashing + KD build + upload, or just the camera update
    };
    run_info last_run;

    pathtracer();
    ~pathtracer() override;

private:
    friend class abstract_pathtracer; // run() dispatches the built-in integrator to the GPU driver
    std::shared_ptr<detail::device_session> m_session;
    static constexpr float s_epsilon = 1e-3f;
    glm::vec3 trace_recursive(const view_parameters& params, const ray& r, const scene& scene, uint32_t recursion);
};

class rng { // rng.h:12-41
private:
    static thread_local std::mt19937 m_e;
    static thread_local std::uniform_real_distribution<float> m_01;

public:
    rng() = delete;
    static float next() { return m_01(m_e); }
    template <typename T>
    static T next(const T min, const T max)
    {
        return static_cast<T>(min + (max - min) * next());
    }
    static glm::vec3 hemisphere(const glm::vec3& n);
    static void seed(const std::uint32_t s) { m_e.seed(s); }
};

} // namespace rt
