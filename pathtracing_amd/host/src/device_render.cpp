// device_render.cpp — the GPU driver behind rt::pathtracer::run: flattens an rt::scene
// into the tables of include/ptk.h and replays the reference's driver lambda
// (abstract_pathtracer.cpp:14-117) with the three hot loops replaced by
// ptk_render_samples_dev / ptk_resolve_iteration_dev.  Plain host C++: all device work,
// including buffer management, goes through the C ABI.
#include "device_render.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ptk.h"

namespace rt {
namespace detail {

namespace {

struct flat_scene {
    std::vector<ptk_node_desc> nodes;
    std::vector<std::vector<ptk_triangle>> mesh_triangles;
    std::vector<ptk_mesh_desc> meshes;
    std::vector<ptk_sampler_desc> samplers;
    std::vector<ptk_texture_desc> textures;
    ptk_scene_desc desc{};
};

uint32_t mode_of(const image& img) { return img.sample_mode == sample_mode::nearest ? PTK_MODE_NEAREST : PTK_MODE_LINEAR; }

// rt::scene -> ptk_scene_desc.  Shared objects (a mesh or sampler referenced by several
// nodes, scene_loader.cpp:165) are emitted once.  Only the reference's own shape and
// sampler classes can be flattened; anything else is an error, not a host fallback.
bool flatten(const scene& sc, flat_scene& out, std::string& err)
{
    std::map<const mesh*, uint32_t> mesh_ids;
    std::map<const sampler_2d*, uint32_t> sampler_ids;
    std::map<const image*, uint32_t> texture_ids;

    auto texture_id = [&](const image* img) -> uint32_t {
        auto it = texture_ids.find(img);
        if (it != texture_ids.end()) return it->second;
        ptk_texture_desc t{};
        t.width = uint32_t(img->get_width());
        t.height = uint32_t(img->get_height());
        t.rgb = reinterpret_cast<const float*>(img->data()); // glm::vec3 is 3 packed floats
        const uint32_t id = uint32_t(out.textures.size());
        out.textures.push_back(t);
        texture_ids.emplace(img, id);
        return id;
    };
    auto sampler_id = [&](const std::shared_ptr<sampler_2d>& s, uint32_t& id) -> bool {
        if (!s) return false;
        auto it = sampler_ids.find(s.get());
        if (it != sampler_ids.end()) {
            id = it->second;
            return true;
        }
        ptk_sampler_desc d{};
        if (const auto* c = dynamic_cast<const color_sampler*>(s.get())) {
            const glm::vec3 rgb = c->average();
            d.kind = PTK_SAMPLER_COLOR;
            d.color[0] = rgb.r;
            d.color[1] = rgb.g;
            d.color[2] = rgb.b;
        } else if (const auto* img = dynamic_cast<const image*>(s.get())) {
            if (img->get_width() == 0 || img->get_height() == 0) return false;
            d.kind = PTK_SAMPLER_IMAGE;
            d.texture = texture_id(img);
            d.mode = mode_of(*img);
        } else {
            return false;
        }
        id = uint32_t(out.samplers.size());
        out.samplers.push_back(d);
        sampler_ids.emplace(s.get(), id);
        return true;
    };

    out.nodes.resize(sc.nodes.size());
    for (size_t i = 0; i < sc.nodes.size(); ++i) {
        const scene_node& n = *sc.nodes[i];
        ptk_node_desc& d = out.nodes[i];
        std::memset(&d, 0, sizeof(d));
        std::memcpy(d.transform, glm::value_ptr(n.get_transform()), 64);
        std::memcpy(d.inv_transform, glm::value_ptr(n.get_inverse_transform()), 64);
        std::memcpy(d.normal_transform, glm::value_ptr(n.get_normal_transform()), 64);
        if (dynamic_cast<const sphere*>(n.shape.get())) {
            d.shape_kind = PTK_SHAPE_SPHERE;
        } else if (const auto* m = dynamic_cast<const mesh*>(n.shape.get())) {
            d.shape_kind = PTK_SHAPE_MESH;
            auto it = mesh_ids.find(m);
            if (it == mesh_ids.end()) {
                const std::vector<triangle>& tris = m->get_triangles();
                std::vector<ptk_triangle> packed(tris.size());
                for (size_t t = 0; t < tris.size(); ++t) std::memcpy(&packed[t], tris[t].vertices.data(), sizeof(ptk_triangle));
                it = mesh_ids.emplace(m, uint32_t(out.mesh_triangles.size())).first;
                out.mesh_triangles.push_back(std::move(packed));
            }
            d.mesh = it->second;
        } else {
            err = "node " + std::to_string(i) + ": shape is neither rt::sphere nor rt::mesh (not renderable on the GPU)";
            return false;
        }
        if (!sampler_id(n.material.albedo, d.albedo) || !sampler_id(n.material.emission, d.emission) ||
            !sampler_id(n.material.roughness, d.roughness) || !sampler_id(n.material.metallic, d.metallic)) {
            err = "node " + std::to_string(i) + ": material sampler is null, empty or of an unknown class";
            return false;
        }
    }
    out.meshes.resize(out.mesh_triangles.size());
    for (size_t m = 0; m < out.meshes.size(); ++m) {
        out.meshes[m].triangles = out.mesh_triangles[m].data();
        out.meshes[m].num_triangles = out.mesh_triangles[m].size();
    }

    ptk_background_desc& bg = out.desc.background;
    if (const auto* c = dynamic_cast<const color_sampler*>(sc.background.get())) {
        const glm::vec3 rgb = c->average();
        bg.kind = PTK_BACKGROUND_COLOR;
        bg.color[0] = rgb.r;
        bg.color[1] = rgb.g;
        bg.color[2] = rgb.b;
    } else if (const auto* eq = dynamic_cast<const equirectangular_map*>(sc.background.get())) {
        const image* img = eq->get_image().get();
        if (!img || img->get_width() == 0 || img->get_height() == 0) {
            err = "background: equirectangular map has no image";
            return false;
        }
        bg.kind = PTK_BACKGROUND_EQUIRECT;
        bg.texture = texture_id(img);
        bg.mode = mode_of(*img);
    } else {
        err = "background is neither rt::color_sampler nor rt::equirectangular_map";
        return false;
    }

    const glm::vec3& dir = sc.camera.get_direction();
    ptk_camera_desc& cam = out.desc.camera;
    cam.position[0] = sc.camera.position.x;
    cam.position[1] = sc.camera.position.y;
    cam.position[2] = sc.camera.position.z;
    cam.direction[0] = dir.x;
    cam.direction[1] = dir.y;
    cam.direction[2] = dir.z;

    out.desc.nodes = out.nodes.data();
    out.desc.num_nodes = uint32_t(out.nodes.size());
    out.desc.meshes = out.meshes.data();
    out.desc.num_meshes = uint32_t(out.meshes.size());
    out.desc.samplers = out.samplers.data();
    out.desc.num_samplers = uint32_t(out.samplers.size());
    out.desc.textures = out.textures.data();
    out.desc.num_textures = uint32_t(out.textures.size());
    return true;
}

struct device_run {
    ptk_ctx* ctx = nullptr;
    float* d_sum = nullptr;
    float* d_image = nullptr;
    ~device_run()
    {
        if (!ctx) return;
        ptk_dev_free(ctx, d_sum);
        ptk_dev_free(ctx, d_image);
        ptk_ctx_destroy(ctx);
    }
};

std::shared_ptr<pathtracer_result> failed_run(const view_parameters& view, const std::string& why)
{
    std::fprintf(stderr, "[rt] GPU render failed: %s\n", why.c_str());
    return std::make_shared<pathtracer_result>([view, why](pathtracer_result& self) {
        self.set_error(why);
        const image black(view.width, view.height);
        self.on_end(black); // waiters are released; error() says why the image is empty
    });
}

} // namespace

std::shared_ptr<pathtracer_result> run_on_device(const view_parameters& view, const trace_parameters& params,
                                                 scene& sc, int device, uint64_t seed, uint64_t max_paths)
{
    auto run = std::make_shared<device_run>();
    {
        flat_scene flat;
        std::string err;
        if (!flatten(sc, flat, err)) return failed_run(view, err);
        if (ptk_ctx_create(device, &run->ctx) != PTK_OK) return failed_run(view, ptk_last_error(nullptr));
        if (ptk_scene_upload(run->ctx, &flat.desc) != PTK_OK) return failed_run(view, ptk_last_error(run->ctx));
    } // host copies are released: the library keeps its own tables in HBM

    return std::make_shared<pathtracer_result>([run, view, params, seed, max_paths](pathtracer_result& self) {
        ptk_ctx* ctx = run->ctx;
        image frame(view.width, view.height);
        const uint64_t num_pixels = uint64_t(view.width) * view.height;
        const uint64_t frame_bytes = num_pixels * 3 * sizeof(float);
        auto ok = [&](int rc) {
            if (rc == PTK_OK) return true;
            self.set_error(ptk_last_error(ctx));
            std::fprintf(stderr, "[rt] GPU render failed: %s\n", ptk_last_error(ctx));
            return false;
        };
        bool healthy = num_pixels > 0 && ok(ptk_render_begin(ctx, view.width, view.height, view.fov_y, max_paths)) &&
                       ok(ptk_dev_alloc(ctx, frame_bytes, reinterpret_cast<void**>(&run->d_sum))) &&
                       ok(ptk_dev_alloc(ctx, frame_bytes, reinterpret_cast<void**>(&run->d_image))) &&
                       ok(ptk_dev_memset(ctx, run->d_sum, 0, frame_bytes)) &&
                       ok(ptk_dev_memset(ctx, run->d_image, 0, frame_bytes));

        const uint64_t spi = params.samples_per_iteration;
        const uint64_t queue = max_paths ? max_paths : (1ull << 24);
        // One slice = the pixels whose samples fill the ray queue once; progress and the
        // interrupt flag are looked at between slices (the reference does so per scan line).
        const uint64_t slice_pixels = std::max<uint64_t>(1, spi ? queue / std::min<uint64_t>(spi, queue) : num_pixels);

        for (uint64_t it = 0; healthy && (params.iterations == 0 || it < params.iterations) && !self.is_interrupted(); ++it) {
            self.on_iteration_start(it);
            bool complete = true;
            for (uint64_t p0 = 0; healthy && p0 < num_pixels; p0 += slice_pixels) {
                if (self.is_interrupted()) {
                    complete = false;
                    break;
                }
                self.progress = float(p0) / float(num_pixels);
                const uint64_t np = std::min(slice_pixels, num_pixels - p0);
                // sample indices are global: iteration `it` owns [it*spi, (it+1)*spi)
                for (uint64_t s0 = 0; healthy && s0 < spi; s0 += 0x40000000ull) {
                    const uint32_t ns = uint32_t(std::min<uint64_t>(0x40000000ull, spi - s0));
                    healthy = ok(ptk_render_samples_dev(ctx, seed, p0, np, it * spi + s0, ns, run->d_sum, s0 ? 1 : 0));
                }
                healthy = healthy && ok(ptk_ctx_synchronize(ctx));
            }
            if (!healthy) break;
            // An iteration cut short by interrupt() leaves the image as it was; the events
            // below still fire, as in abstract_pathtracer.cpp:105-109.
            if (complete && spi > 0) healthy = ok(ptk_resolve_iteration_dev(ctx, run->d_sum, run->d_image, spi, it));
            healthy = healthy && ok(ptk_dev_download(ctx, frame.data(), run->d_image, frame_bytes));
            ptk_counters counters{};
            if (healthy && ptk_get_counters(ctx, &counters) == PTK_OK) {
                self.paths_traced = counters.paths;
                self.rays_cast = counters.rays;
            }
            if (!healthy) break;
            self.progress = complete ? 1.0f : self.progress.load();
            self.iteration = it;
            self.samples_per_pixel += spi;
            self.on_iteration_end(frame, it);
        }
        self.on_end(frame);
    });
}

} // namespace detail
} // namespace rt
