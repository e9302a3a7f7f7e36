// device_render.hpp — internal: the GPU driver behind rt::pathtracer::run.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "ptk_rt/render.hpp"

namespace rt {
namespace detail {

struct device_session; // contexts + the uploaded scene's key, kept by an rt::pathtracer across run() calls
std::shared_ptr<device_session> make_device_session();

struct device_run_info {
    bool scene_reused = false; // the tables of the previous run() were kept, only the camera was sent
    double prepare_ms = 0.0;   // time run() spent before the driver thread started (hashing, KD build, upload)
};

// Flattens `scene` to a ptk_scene_desc and uploads it through the C ABI (include/ptk.h) on the caller's
// thread -- or, when `session` already holds this scene, only sends the camera -- then starts the driver
// thread that renders iteration by iteration on `devices` (one context each; ids may repeat).
// aov >= 0 renders that debug AOV (PTK_AOV_*, include/ptk.h) instead of the integrator.
// Never renders on the host: failures end the run and are reported by result->error().
std::shared_ptr<pathtracer_result> run_on_device(const view_parameters& view, const trace_parameters& params, scene& scene,
                                                 const std::vector<int>& devices, uint64_t seed, uint64_t max_paths,
                                                 std::shared_ptr<device_session> session, device_run_info* info, int aov = -1);

} // namespace detail
} // namespace rt
