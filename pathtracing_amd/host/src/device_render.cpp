// device_render.cpp — the GPU driver behind rt::pathtracer::run: flattens an rt::scene
// into the tables of include/ptk.h and replays the reference's driver lambda
// (abstract_pathtracer.cpp:14-117) with the three hot loops replaced by
// ptk_render_samples_dev / ptk_resolve_iteration_dev.  Plain host C++: all device work,
// including buffer management, goes through the C ABI.
#include "device_render.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <atomic>
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

// 64-bit content hash (4 interleaved multiply-xor lanes over 8-byte words): decides whether the tables a
// previous run() uploaded can be kept.  Not cryptographic; a collision would reuse stale tables.
uint64_t hash_bytes(const void* data, size_t bytes, uint64_t seed)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint64_t h[4] = { seed ^ 0x9E3779B97F4A7C15ull, seed ^ 0xC2B2AE3D27D4EB4Full, seed ^ 0x165667B19E3779F9ull, seed ^ 0x27D4EB2F165667C5ull };
    auto mix = [](uint64_t a, uint64_t w) {
        a = (a ^ w) * 0xFF51AFD7ED558CCDull;
        return a ^ (a >> 32);
    };
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) h[k] = mix(h[k], w[k]);
    }
    for (; i < bytes; ++i) h[i & 3] = mix(h[i & 3], p[i]);
    uint64_t r = bytes;
    for (int k = 0; k < 4; ++k) r = mix(r, h[k]);
    return r;
}

// Everything of an rt::scene that ends up in the device tables, except the camera: node matrices and
// material wiring, each mesh's identity (object_id, scene.h:194-201), triangle count and vertex data, sampler
// colours, texture pixels, background.  Walks the scene graph directly (no packed copies are made).
bool scene_key(const scene& sc, uint64_t& key)
{
    uint64_t h = 0x5CE11E;
    auto add = [&](const void* d, size_t n) { h = hash_bytes(d, n, h); };
    auto add_image = [&](const image* img) {
        const uint64_t dims[3] = { img->get_width(), img->get_height(), uint64_t(img->sample_mode) };
        add(dims, sizeof(dims));
        add(img->data(), img->get_width() * img->get_height() * sizeof(glm::vec3));
    };
    std::map<const void*, int> seen;
    for (const std::shared_ptr<scene_node>& np : sc.nodes) {
        const scene_node& n = *np;
        add(glm::value_ptr(n.get_transform()), 64);
        add(glm::value_ptr(n.get_inverse_transform()), 64);
        add(glm::value_ptr(n.get_normal_transform()), 64);
        if (const auto* m = dynamic_cast<const mesh*>(n.shape.get())) {
            const uint64_t head[3] = { 1, m->id, m->get_triangles().size() };
            add(head, sizeof(head));
            if (seen.emplace(m, 0).second)
                for (const triangle& t : m->get_triangles()) add(t.vertices.data(), sizeof(ptk_triangle));
        } else if (dynamic_cast<const sphere*>(n.shape.get())) {
            const uint64_t head = 2;
            add(&head, sizeof(head));
        } else {
            return false;
        }
        for (const std::shared_ptr<sampler_2d>* sp : { &n.material.albedo, &n.material.emission, &n.material.roughness, &n.material.metallic }) {
            const sampler_2d* s = sp->get();
            if (!s) return false;
            if (const auto* c = dynamic_cast<const color_sampler*>(s)) {
                const glm::vec3 rgb = c->average();
                add(&rgb, sizeof(rgb));
            } else if (const auto* img = dynamic_cast<const image*>(s)) {
                const uint64_t tag = 3;
                add(&tag, sizeof(tag));
                const bool first = seen.emplace(img, 0).second;
                const void* id = img; // which nodes share an image is part of the wiring
                add(&id, sizeof(id));
                if (first) add_image(img);
            } else {
                return false;
            }
        }
    }
    if (const auto* c = dynamic_cast<const color_sampler*>(sc.background.get())) {
        const glm::vec3 rgb = c->average();
        add(&rgb, sizeof(rgb));
    } else if (const auto* eq = dynamic_cast<const equirectangular_map*>(sc.background.get())) {
        const image* img = eq->get_image().get();
        if (!img) return false;
        const uint64_t tag = 4;
        add(&tag, sizeof(tag));
        add_image(img);
    } else {
        return false;
    }
    key = h;
    return true;
}

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

// The device side of one rt::pathtracer, kept across run() calls: the group of contexts (one per device) and
// the key of the scene whose tables they hold.  Sandbox calls run() again on every camera move
// (rtsb.cpp:295-299,355); the reference recompiles the scene each time (abstract_pathtracer.cpp:13), here a
// run whose scene key is unchanged only sends the camera (ptk_group_scene_set_camera): no KD build, no upload.
struct device_session {
    std::atomic<bool> busy{ false }; // set by the run that uses the session, from run() until its driver thread ends
    ptk_group* group = nullptr;
    std::vector<int> devices;
    bool has_scene = false;
    uint64_t key = 0;
    ~device_session()
    {
        if (group) ptk_group_destroy(group);
    }
};

std::shared_ptr<device_session> make_device_session() { return std::make_shared<device_session>(); }

std::shared_ptr<pathtracer_result> run_on_device(const view_parameters& view, const trace_parameters& params, scene& sc,
                                                 const std::vector<int>& devices, uint64_t seed, uint64_t max_paths,
                                                 std::shared_ptr<device_session> session, device_run_info* info, int aov)
{
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    // a session that is still rendering (run() called again without wait()) is left alone: this run gets its own
    if (!session || session->busy.exchange(true)) {
        session = std::make_shared<device_session>();
        session->busy = true;
    }
    // clears the flag when the last holder (this function on an early return, else the driver thread) lets go
    std::shared_ptr<void> held(nullptr, [session](void*) { session->busy = false; });
    if (info) *info = device_run_info{};

    uint64_t key = 0;
    if (!scene_key(sc, key)) {
        flat_scene probe;
        std::string err;
        flatten(sc, probe, err); // for its message
        return failed_run(view, err.empty() ? "scene cannot be rendered on the GPU" : err);
    }
    if (session->group && session->devices != devices) {
        ptk_group_destroy(session->group);
        session->group = nullptr;
        session->has_scene = false;
    }
    if (!session->group) {
        if (ptk_group_create(devices.data(), int(devices.size()), &session->group) != PTK_OK) {
            session->group = nullptr;
            return failed_run(view, ptk_group_last_error(nullptr));
        }
        session->devices = devices;
    }
    ptk_group* group = session->group;
    if (session->has_scene && session->key == key) {
        ptk_camera_desc cam{};
        const glm::vec3& dir = sc.camera.get_direction();
        cam.position[0] = sc.camera.position.x; cam.position[1] = sc.camera.position.y; cam.position[2] = sc.camera.position.z;
        cam.direction[0] = dir.x; cam.direction[1] = dir.y; cam.direction[2] = dir.z;
        if (ptk_group_scene_set_camera(group, &cam) != PTK_OK) return failed_run(view, ptk_group_last_error(group));
        if (info) info->scene_reused = true;
    } else {
        flat_scene flat;
        std::string err;
        if (!flatten(sc, flat, err)) return failed_run(view, err);
        session->has_scene = false;
        if (ptk_group_scene_upload(group, &flat.desc) != PTK_OK) return failed_run(view, ptk_group_last_error(group));
        session->has_scene = true;
        session->key = key;
    } // host copies are released: the library keeps its own tables in HBM
    if (info) info->prepare_ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();

    // `held` travels with the driver thread's closure: the session is free again when the run is over
    return std::make_shared<pathtracer_result>([session, held, group, view, params, seed, max_paths, aov](pathtracer_result& self) mutable {
        image frame(view.width, view.height);
        const uint64_t num_pixels = uint64_t(view.width) * view.height;
        auto ok = [&](int rc) {
            if (rc == PTK_OK) return true;
            self.set_error(ptk_group_last_error(group));
            std::fprintf(stderr, "[rt] GPU render failed: %s\n", ptk_group_last_error(group));
            return false;
        };
        bool healthy = num_pixels > 0 && ok(ptk_group_render_begin(group, view.width, view.height, view.fov_y, max_paths));
        const uint64_t spi = params.samples_per_iteration;
        uint32_t K = 1; // slices per iteration (pixel ranges of one launch chain each)
        if (healthy && spi > 0) healthy = ok(ptk_group_iteration_slices(group, spi, &K));

        // The reference's driver (abstract_pathtracer.cpp:27-112) as a pipeline.  Slices are submitted ahead of the
        // GPU (kWindow in flight, across iteration boundaries), so the frame download and the handlers of iteration i
        // run while the GPU renders iteration i + 1; the handlers still see start(i), end(i), start(i + 1), ... in
        // order.  The interrupt flag is looked at before every submission (the reference: per scan line, pixel and
        // sample, :63-72), progress follows the finished slices (:51).  After an interrupt nothing more is submitted;
        // the iteration that is open ends with the slices it got (at most kWindow launch chains are still in flight):
        // one that got only part of them is resolved over those pixels, the others keep the previous mean, and
        // on_iteration_end / on_end fire as the reference fires them (:109,114).  No further iteration opens (:38).
        constexpr uint32_t kWindow = 2;
        const auto wanted = [&](uint64_t it) { return params.iterations == 0 || it < params.iterations; };
        uint64_t sub_it = 0, done_it = 0;   // iteration of the next slice to submit / to wait for
        uint32_t sub_k = 0, done_k = 0;     // that slice
        uint64_t tickets[kWindow] = {};
        uint64_t submitted = 0, waited = 0; // slices so far
        bool stop = false;                  // nothing more is submitted
        bool open = false;                  // on_iteration_start fired for done_it, its on_iteration_end not yet
        auto complete = [&](uint64_t it, bool rendered) { // iteration `it` has been finished (ptk_group_finish_iteration)
            ptk_counters counters{};
            if (rendered) healthy = ok(ptk_group_wait_iteration(group, it, reinterpret_cast<float*>(frame.data()), &counters));
            if (!healthy) return;
            if (rendered) {
                self.paths_traced = counters.paths;
                self.rays_cast = counters.rays;
            }
            self.progress = 1.0f;
            self.iteration = it;
            self.samples_per_pixel += spi;
            self.on_iteration_end(frame, it);
            open = false;
        };
        if (healthy && wanted(0) && !self.is_interrupted()) {
            self.on_iteration_start(0);
            open = true;
        }
        while (healthy && open) {
            if (spi == 0) { // traces nothing, like the reference's empty sample loop, and leaves the image as it is
                complete(done_it, false);
                ++done_it;
                if (healthy && wanted(done_it) && !self.is_interrupted()) {
                    self.on_iteration_start(done_it);
                    open = true;
                }
                continue;
            }
            // fill the window
            while (healthy && !stop && submitted - waited < kWindow && wanted(sub_it)) {
                if (self.is_interrupted()) {
                    stop = true;
                    break;
                }
                healthy = ok(ptk_group_submit_slice(group, aov, seed, sub_it, spi, sub_k, K, &tickets[submitted % kWindow]));
                if (!healthy) break;
                ++submitted;
                if (++sub_k == K) {
                    healthy = ok(ptk_group_finish_iteration(group, sub_it, spi, K, K, 1));
                    sub_k = 0;
                    ++sub_it;
                }
            }
            if (!healthy) break;
            if (waited == submitted) {
                // nothing in flight and nothing more to come: the open iteration ends with the slices it got
                if (sub_it == done_it) { // (else it was finished whole and completed below)
                    if (sub_k > 0) healthy = ok(ptk_group_finish_iteration(group, done_it, spi, sub_k, K, 1));
                    if (healthy) complete(done_it, sub_k > 0);
                }
                break;
            }
            healthy = ok(ptk_group_wait_slice(group, tickets[waited % kWindow]));
            if (!healthy) break;
            ++waited;
            ++done_k;
            self.progress = float(done_k) / float(K);
            if (done_k == K) {
                complete(done_it, true);
                done_k = 0;
                ++done_it;
                // the reference's loop condition (abstract_pathtracer.cpp:38).  After an interrupt the slices of later
                // iterations that were already submitted are left to run out unreported: the frame handed to on_end is
                // the one of the last iteration that ended.
                if (healthy && wanted(done_it) && !self.is_interrupted()) {
                    self.on_iteration_start(done_it);
                    open = true;
                }
            }
        }
        self.on_end(frame);
        held.reset();
    });
}

} // namespace detail
} // namespace rt
