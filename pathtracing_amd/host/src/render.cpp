// render.cpp — host side of ptk_rt/render.hpp: pathtracer_result, the run() dispatcher,
// the host-thread loop for user subclasses, rt::rng and the single-ray trace() API
// (reference src/Pathtracing/abstract_pathtracer.cpp, pathtracer.cpp, rng.cpp).
#include "ptk_rt/render.hpp"

#include <cmath>
#include <limits>
#include <typeinfo>
#include <vector>

#include "device_render.hpp"
#include "ptk_rt/utility.hpp"

namespace rt {

// ---------------------------------------------------------------- pathtracer_result

// abstract_pathtracer.cpp:121-125: the driver thread starts in the constructor.  The
// start time is taken first so get_elapsed_time() is valid inside the first event.
pathtracer_result::pathtracer_result(const fn& body) : m_start_time(std::chrono::system_clock::now())
{
    m_thread = std::thread([this, body] { body(*this); });
}

pathtracer_result::~pathtracer_result() // abstract_pathtracer.cpp:127-131
{
    if (m_thread.joinable()) m_thread.join();
}

float pathtracer_result::get_elapsed_time() const // abstract_pathtracer.cpp:133-137
{
    return std::chrono::duration<float>(std::chrono::system_clock::now() - m_start_time).count();
}

std::string pathtracer_result::error() const
{
    std::lock_guard<std::mutex> g(m_error_lock);
    return m_error;
}

void pathtracer_result::set_error(const std::string& msg)
{
    std::lock_guard<std::mutex> g(m_error_lock);
    m_error = msg;
}

// ---------------------------------------------------------------- rng (rng.cpp)

thread_local std::mt19937 rng::m_e;
thread_local std::uniform_real_distribution<float> rng::m_01(0.0f, 1.0f);

glm::vec3 rng::hemisphere(const glm::vec3& n) // rng.cpp:12-36: uniform, pdf 1/2pi, not re-normalised
{
    const glm::vec3 tangent = glm::abs(n.x) > glm::abs(n.y) ? glm::normalize(glm::vec3(n.z, 0.0f, -n.x))
                                                            : glm::normalize(glm::vec3(0.0f, -n.z, n.y));
    const glm::vec3 bitangent = glm::cross(n, tangent);
    const float z = next();
    const float ring = glm::sqrt(1.0f - z * z);
    const float phi = glm::pi<float>() * 2.0f * next();
    return ring * glm::cos(phi) * tangent + ring * glm::sin(phi) * bitangent + z * n;
}

// ---------------------------------------------------------------- pathtracer::trace
// The reference's per-ray virtual (pathtracer.cpp:14-69), kept because it is public API
// that callers and subclasses may invoke for individual rays.  run() never calls it for
// an rt::pathtracer: images are produced by k_bounce on the GPU.

pathtracer::pathtracer() = default;
pathtracer::~pathtracer() = default;

glm::vec3 pathtracer::trace(const view_parameters& params, const ray& r, const scene& scene)
{
    return trace_recursive(params, r, scene, 5);
}

glm::vec3 pathtracer::trace_recursive(const view_parameters& params, const ray& r, const scene& scene, uint32_t recursion)
{
    if (recursion == 0) return glm::vec3(0.0f);
    const auto [hit, node] = scene.cast_ray(r);
    if (!hit.hit) return scene.background->sample(r.direction);
    const rt::material& m = node->material;
    const glm::vec3 albedo = m.albedo->sample(hit.uv);
    const glm::vec3 emission = m.emission->sample(hit.uv);
    const float roughness = m.roughness->sample(hit.uv).r;
    const float metallic = m.metallic->sample(hit.uv).r;
    const glm::vec3 diffuse_dir = rng::hemisphere(hit.normal);
    const glm::vec3 mirror_dir = glm::reflect(r.direction, hit.normal);
    ray next;
    next.direction = glm::normalize(glm::mix(mirror_dir, diffuse_dir, roughness));
    next.origin = hit.position + next.direction * s_epsilon;
    const float cos_theta = glm::max(0.0f, glm::dot(next.direction, hit.normal));
    const glm::vec3 incoming = trace_recursive(params, next, scene, recursion - 1);
    return emission + glm::mix(albedo, glm::vec3(1.0f), metallic) * incoming * cos_theta * 2.0f;
}

// ---------------------------------------------------------------- run

namespace {

// Host-thread driver for user subclasses of abstract_pathtracer (e.g. the reference's
// debug_pathtracer, PathtracingUtility/debug_pathtracer.cpp): same loop structure and event
// order as abstract_pathtracer.cpp:14-117, calling the virtual trace() per sample.
void host_driver(abstract_pathtracer& tracer, const view_parameters view, const trace_parameters params, scene& sc,
                 pathtracer_result& self)
{
    image frame(view.width, view.height);
    const glm::vec3 forward = glm::normalize(sc.camera.get_direction());
    const glm::vec3 right = glm::normalize(glm::cross(forward, glm::vec3{ 0.0f, 1.0f, 0.0f }));
    const glm::vec3 up = glm::cross(right, forward);
    const float h2 = std::atan(view.fov_y / 2.0f); // sic: atan, abstract_pathtracer.cpp:24
    const float w2 = h2 * (float)view.width / view.height;

    for (uint64_t it = 0; (params.iterations == 0 || it < params.iterations) && !self.is_interrupted(); ++it) {
        self.on_iteration_start(it);
        std::atomic<uint32_t> next_line{ 0 };
        const float keep = it / float(it + 1);
        auto worker = [&](uint32_t seed) {
            rng::seed(seed);
            for (;;) {
                if (self.is_interrupted()) return;
                const uint32_t y = next_line.fetch_add(1);
                if (y >= view.height) return;
                self.progress = y / float(view.height);
                for (uint32_t x = 0; x < view.width && !self.is_interrupted(); ++x) {
                    glm::vec3 sum(0.0f);
                    for (uint64_t s = 0; s < params.samples_per_iteration && !self.is_interrupted(); ++s) {
                        const float fx = rng::next() - 0.5f + x;
                        const float fy = rng::next() - 0.5f + y;
                        const float xf = fx / view.width * 2.0f - 1.0f;
                        const float yf = 1.0f - fy / view.height * 2.0f;
                        ray r;
                        r.origin = sc.camera.position;
                        r.direction = glm::normalize(forward + right * xf * w2 + up * yf * h2);
                        sum += tracer.trace(view, r, sc);
                    }
                    frame.set_pixel(x, y, glm::mix(sum / float(params.samples_per_iteration), frame.get_pixel(x, y), keep));
                }
            }
        };
        std::vector<std::thread> pool;
        pool.reserve(params.num_threads);
        for (uint32_t i = 0; i < params.num_threads; ++i)
            pool.emplace_back(worker, rng::next<std::uint32_t>(0u, std::numeric_limits<std::uint32_t>::max()));
        for (std::thread& t : pool) t.join();
        self.iteration = it;
        self.samples_per_pixel += params.samples_per_iteration;
        self.on_iteration_end(frame, it);
    }
    self.on_end(frame);
}

} // namespace

std::shared_ptr<pathtracer_result> abstract_pathtracer::run(const view_parameters& view_params,
                                                            const trace_parameters& trace_params, scene& scene)
{
    scene.compile(); // on the caller's thread, abstract_pathtracer.cpp:13
    if (typeid(*this) == typeid(pathtracer)) {
        // The built-in integrator: flatten + upload here (the work scene.compile() stands
        // for), render on the GPU from a driver thread.  No host rendering path exists.
        pathtracer& self = static_cast<pathtracer&>(*this);
        const std::vector<int> devices = self.device.devices.empty() ? std::vector<int>{ self.device.device } : self.device.devices;
        if (self.device.keep_scene && !self.m_session) self.m_session = detail::make_device_session();
        detail::device_run_info info;
        auto result = detail::run_on_device(view_params, trace_params, scene, devices, self.device.seed, self.device.max_paths_per_launch,
                                            self.device.keep_scene ? self.m_session : nullptr, &info);
        self.last_run.scene_reused = info.scene_reused;
        self.last_run.prepare_ms = info.prepare_ms;
        return result;
    }
    if (typeid(*this) == typeid(utility::debug_pathtracer) && static_cast<utility::debug_pathtracer&>(*this).use_device) {
        // the reference's debug tracer (debug_pathtracer.cpp:8-32): its five modes are kernels of the library too
        utility::debug_pathtracer& self = static_cast<utility::debug_pathtracer&>(*this);
        if (!self.m_session) self.m_session = detail::make_device_session();
        return detail::run_on_device(view_params, trace_params, scene, { self.device }, self.seed, 0, self.m_session, nullptr,
                                     int(self.current_mode));
    }
    return std::make_shared<pathtracer_result>([this, view_params, trace_params, &scene](pathtracer_result& self) {
        host_driver(*this, view_params, trace_params, scene, self);
    });
}

} // namespace rt
