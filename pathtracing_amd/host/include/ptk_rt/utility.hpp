// ptk_rt/utility.hpp — host mirror of the reference's PathtracingUtility library
// (reference src/PathtracingUtility/mesh_loader.h:19, scene_loader.h:14,
// debug_pathtracer.h:11-29): the callers' ingest on either side of the hot path
// (SURVEY.md section 8f rank 3).  Same names and signatures.
//
// load_meshes_from_wavefront: a streaming reader that accepts exactly the grammar of the
// reference's regex cascade (mesh_loader.cpp:40-150), quirks included — see src/mesh_loader.cpp.
// load_scene: the JSON schema of scene_loader.cpp:60-191, read by a small built-in JSON reader.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <string_view>

#include "render.hpp"

namespace rt {
namespace utility {

std::map<std::string, std::shared_ptr<rt::mesh>> load_meshes_from_wavefront(std::string_view file_name);

scene load_scene(std::string_view file_name);

class debug_pathtracer : public abstract_pathtracer { // debug_pathtracer.h:11-29
public:
    enum class mode { albedo, emission, roughness, metallic, normal };
    mode current_mode = mode::albedo;
    glm::vec3 trace(const view_parameters& params, const ray& ray, const scene& scene) override;

    // Additive: run() renders this tracer's AOVs on the GPU (ptk_render_aov_dev) when use_device is set (default);
    // cleared, it is driven on host threads through trace() like any other subclass, as in the reference.
    bool use_device = true;
    int device = 0;
    uint64_t seed = 0;

private:
    friend class rt::abstract_pathtracer;
    std::shared_ptr<detail::device_session> m_session;
};

} // namespace utility
} // namespace rt
