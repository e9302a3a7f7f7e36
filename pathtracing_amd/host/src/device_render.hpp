// device_render.hpp — internal: the GPU driver behind rt::pathtracer::run.
#pragma once
#include <cstdint>
#include <memory>

#include "ptk_rt/render.hpp"

namespace rt {
namespace detail {

// Flattens `scene` to a ptk_scene_desc, uploads it through the C ABI (include/ptk.h) on the
// caller's thread, then starts the driver thread that renders iteration by iteration.
// Never renders on the host: failures end the run and are reported by result->error().
std::shared_ptr<pathtracer_result> run_on_device(const view_parameters& view, const trace_parameters& params,
                                                 scene& scene, int device, uint64_t seed, uint64_t max_paths);

} // namespace detail
} // namespace rt
