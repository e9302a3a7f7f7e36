// rt_render — a caller written purely against the reference-shaped rt:: API, used to show
// (and test) that the API drives the GPU path end to end.  It plays the role of the
// reference's CLI (src/CLI/main.cpp): build an rt::scene the way scene_loader.cpp does
// (translate -> multiply(Rz*Ry*Rx) -> scale; shared meshes and samplers), call
// rt::pathtracer::run, subscribe to the events, wait(), save the result.
//
// Input is either the reference's own scene JSON (a file name ending in .json: read by
// rt::utility::load_scene with its OBJ meshes and stb-decoded textures, asset paths relative to
// --asset-root as they are relative to the working directory in the reference) or a small
// token stream ("ptkscene 1", written by tests/host_scene_text.py from the same scene objects
// the Python tests use).
//
//   rt_render --scene S --out image.bin [--resolution W H]
//             [--iterations N] [--spp-per-iteration K] [--seed X] [--device D | --devices D0,D1,..]
//             [--png-out image.png]   (what the reference CLI writes: CLI/main.cpp:85-107; needs stb_image_write.h on the include path at build time)
//             [--threads T] [--ldr-out rgba8.bin] [--interrupt-after N] [--asset-root DIR]
//             [--max-paths N] [--interrupt-at I F]
//             [--aov albedo|emission|roughness|metallic|normal]   (rt::utility::debug_pathtracer instead of rt::pathtracer)
//             [--then-camera PX PY PZ DX DY DZ --then-out image2.bin]   (a second run() on the same
//              rt::pathtracer with only the camera moved: the Sandbox pattern, rtsb.cpp:295-299,355)
//
// image.bin: W*H*3 little-endian floats, row 0 = top (linear HDR, before the CLI tone map).
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include <pathtracer.h>
#include <scene_loader.h>
#include <debug_pathtracer.h>
#include <unistd.h>
#if __has_include(<stb_image_write.h>)
// the reference vendors stb; the header is taken from the include path at build time (Makefile: STB_INC), never copied
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include <stb_image_write.h>
#define PTK_HAVE_STB_WRITE 1
#endif

namespace {

bool fail(const std::string& msg)
{
    std::fprintf(stderr, "rt_render: %s\n", msg.c_str());
    return false;
}

float next_float(std::istream& in)
{
    std::string tok;
    in >> tok;
    return std::strtof(tok.c_str(), nullptr); // accepts %a hex floats: values round-trip exactly
}

glm::vec3 next_vec3(std::istream& in)
{
    const float x = next_float(in), y = next_float(in), z = next_float(in);
    return { x, y, z };
}

bool load_scene_text(const std::string& path, rt::scene& scene)
{
    std::ifstream in(path);
    if (!in) return fail("cannot open " + path);
    std::string tok;
    int version = 0;
    in >> tok >> version;
    if (tok != "ptkscene" || version != 1) return fail("not a ptkscene 1 file");

    std::map<std::string, std::shared_ptr<rt::image>> textures;
    std::map<std::string, std::shared_ptr<rt::sampler_2d>> samplers;
    std::map<std::string, std::shared_ptr<rt::color_sampler>> colors;
    std::map<std::string, std::shared_ptr<rt::mesh>> meshes;

    while (in >> tok) {
        if (tok == "end") return true;
        if (tok == "camera") {
            scene.camera.position = next_vec3(in);
            scene.camera.set_direction(next_vec3(in));
        } else if (tok == "texture") {
            std::string id, file, mode;
            int ldr = 0;
            in >> id >> file >> mode >> ldr;
            auto img = std::make_shared<rt::image>();
            img->load(file);
            if (img->get_width() == 0) return fail("texture " + file + " did not load");
            if (ldr) img->to_ldr();
            img->sample_mode = mode == "nearest" ? rt::sample_mode::nearest : rt::sample_mode::linear;
            textures[id] = img;
        } else if (tok == "sampler") {
            std::string id, kind;
            in >> id >> kind;
            if (kind == "color") {
                auto c = std::make_shared<rt::color_sampler>(next_vec3(in));
                samplers[id] = c;
                colors[id] = c;
            } else {
                std::string tex;
                in >> tex;
                samplers[id] = textures.at(tex);
            }
        } else if (tok == "background") {
            std::string kind, id;
            in >> kind >> id;
            if (kind == "color") scene.background = colors.at(id);
            else scene.background = std::make_shared<rt::equirectangular_map>(textures.at(id));
        } else if (tok == "mesh") {
            std::string id;
            size_t count = 0;
            in >> id >> count;
            auto m = std::make_shared<rt::mesh>();
            for (size_t t = 0; t < count; ++t) {
                rt::triangle& tri = m->add_triangle();
                for (rt::vertex& v : tri.vertices) {
                    v.position = next_vec3(in);
                    v.normal = next_vec3(in);
                    v.uv.x = next_float(in);
                    v.uv.y = next_float(in);
                }
            }
            meshes[id] = m;
        } else if (tok == "node") {
            auto node = std::make_shared<rt::scene_node>();
            std::string kind;
            in >> kind;
            if (kind == "sphere") node->shape = std::make_shared<rt::sphere>();
            else {
                std::string id;
                in >> id;
                node->shape = meshes.at(id);
            }
            for (int k = 0; k < 3; ++k) { // T, R, S in the loader's order (scene_loader.cpp:149-162)
                std::string tag;
                int present = 0;
                in >> tag >> present;
                if (!present) continue;
                const glm::vec3 v = next_vec3(in);
                if (tag == "T") node->translate(v);
                else if (tag == "R")
                    node->multiply(glm::rotate(glm::radians(v.z), glm::vec3(0.0f, 0.0f, 1.0f)) *
                                   glm::rotate(glm::radians(v.y), glm::vec3(0.0f, 1.0f, 0.0f)) *
                                   glm::rotate(glm::radians(v.x), glm::vec3(1.0f, 0.0f, 0.0f)));
                else node->scale(v);
            }
            std::string tag, a, e, r, m;
            in >> tag >> a >> e >> r >> m;
            node->material.albedo = samplers.at(a);
            node->material.emission = samplers.at(e);
            node->material.roughness = samplers.at(r);
            node->material.metallic = samplers.at(m);
            scene.nodes.push_back(node);
        } else {
            return fail("unknown token '" + tok + "'");
        }
    }
    return fail("missing 'end'");
}

} // namespace

int main(int argc, char** argv)
{
    uint32_t width = 512, height = 512;
    uint64_t iterations = 10, spi = 256, seed = 0; // CLI/main.cpp:14-17,70
    uint32_t threads = 4;
    uint64_t interrupt_after = 0; // > 0: call interrupt() from the on_iteration_end handler (Sandbox-style stop)
    // --interrupt-at I F: a second thread polls pathtracer_result::progress the way Sandbox's GUI thread does
    // (rtsb.cpp:264,434) and calls interrupt() once iteration I (0-based) has reached progress F: a stop inside an iteration
    bool watch = false;
    uint64_t watch_iteration = 0;
    float watch_progress = 0.0f;
    uint64_t max_paths = 0; // ray-queue capacity per launch chain (0 = library default); small values cut an iteration into many slices
    int device = 0;
    std::vector<int> devices;
    std::string scene_file, out_file = "result.bin", ldr_file, png_file, asset_root, then_out, aov;
    bool out_is_png = false;
    bool then_camera = false;
    glm::vec3 then_pos(0.0f), then_dir(0.0f, 0.0f, -1.0f);
    for (int i = 1; i < argc; ++i) {
        const std::string p = argv[i];
        auto need = [&](int n) { return i + n < argc; };
        if (p == "--scene" && need(1)) scene_file = argv[++i];
        else if (p == "--out" && need(1)) out_file = argv[++i];
        else if (p == "--ldr-out" && need(1)) ldr_file = argv[++i];
        else if (p == "--png-out" && need(1)) png_file = argv[++i];
        else if (p == "--threads" && need(1)) threads = uint32_t(std::stoul(argv[++i]));
        else if (p == "--iterations" && need(1)) iterations = std::stoull(argv[++i]);
        else if (p == "--spp-per-iteration" && need(1)) spi = std::stoull(argv[++i]);
        else if (p == "--seed" && need(1)) seed = std::stoull(argv[++i]);
        else if (p == "--device" && need(1)) device = std::stoi(argv[++i]);
        else if (p == "--interrupt-after" && need(1)) interrupt_after = std::stoull(argv[++i]);
        else if (p == "--asset-root" && need(1)) asset_root = argv[++i];
        else if (p == "--max-paths" && need(1)) max_paths = std::stoull(argv[++i]);
        else if (p == "--interrupt-at" && need(2)) {
            watch = true;
            watch_iteration = std::stoull(argv[++i]);
            watch_progress = std::strtof(argv[++i], nullptr);
        }
        else if (p == "--aov" && need(1)) aov = argv[++i];
        else if (p == "--then-out" && need(1)) then_out = argv[++i];
        else if (p == "--devices" && need(1)) {
            const std::string list = argv[++i];
            for (size_t a = 0; a <= list.size();) {
                const size_t b = std::min(list.find(',', a), list.size());
                if (b > a) devices.push_back(std::stoi(list.substr(a, b - a)));
                a = b + 1;
            }
        } else if (p == "--then-camera" && need(6)) {
            then_camera = true;
            for (int k = 0; k < 3; ++k) then_pos[k] = std::strtof(argv[++i], nullptr);
            for (int k = 0; k < 3; ++k) then_dir[k] = std::strtof(argv[++i], nullptr);
        }
        else if (p == "--resolution" && need(2)) {
            width = uint32_t(std::stoul(argv[++i]));
            height = uint32_t(std::stoul(argv[++i]));
        } else {
            std::fprintf(stderr, "Unknown parameter: %s\n", p.c_str());
            return -1; // CLI/main.cpp:46-50
        }
    }

    // `--out something.png` is the reference CLI's usage (CLI/main.cpp:19,29-32): the tone-mapped PNG, no raw image
    out_is_png = out_file.size() > 4 && out_file.compare(out_file.size() - 4, 4, ".png") == 0;
    if (out_is_png && png_file.empty()) png_file = out_file;

    rt::scene scene;
    if (scene_file.size() > 5 && scene_file.substr(scene_file.size() - 5) == ".json") {
        // the reference resolves asset paths against the working directory (premake5.lua:170-172)
        char cwd[4096];
        if (!getcwd(cwd, sizeof(cwd))) return 1;
        auto absolute = [&](const std::string& f) { return f.empty() || f[0] == '/' ? f : std::string(cwd) + "/" + f; };
        scene_file = absolute(scene_file);
        out_file = absolute(out_file);
        ldr_file = absolute(ldr_file);
        png_file = absolute(png_file);
        then_out = absolute(then_out);
        if (!asset_root.empty() && chdir(asset_root.c_str()) != 0) return fail("cannot enter " + asset_root) ? 1 : 1;
        scene = rt::utility::load_scene(scene_file);
        if (scene.nodes.empty()) return fail("no nodes in " + scene_file) ? 1 : 1;
    } else if (!load_scene_text(scene_file, scene)) {
        return 1;
    }

    rt::pathtracer pathtracer;
    pathtracer.device.device = device;
    pathtracer.device.devices = devices;
    pathtracer.device.seed = seed;
    pathtracer.device.max_paths_per_launch = max_paths;
    rt::view_parameters view_params;
    rt::trace_parameters trace_params;
    view_params.width = width;
    view_params.height = height;
    view_params.fov_y = glm::pi<float>() / 4.0f;
    trace_params.num_threads = threads;
    trace_params.iterations = iterations;
    trace_params.samples_per_iteration = spi;

    rt::utility::debug_pathtracer debug;
    debug.device = device;
    debug.seed = seed;
    const char* const aov_names[5] = { "albedo", "emission", "roughness", "metallic", "normal" };
    bool aov_known = aov.empty();
    for (int k = 0; k < 5; ++k)
        if (aov == aov_names[k]) {
            debug.current_mode = static_cast<rt::utility::debug_pathtracer::mode>(k);
            aov_known = true;
        }
    if (!aov_known) return fail("unknown --aov " + aov) ? 1 : 1;
    rt::abstract_pathtracer& tracer = aov.empty() ? static_cast<rt::abstract_pathtracer&>(pathtracer) : debug;

    auto result = tracer.run(view_params, trace_params, scene);

    std::atomic<uint64_t> starts{ 0 }, ends{ 0 };
    std::atomic<bool> finished{ false };
    result->on_iteration_start.subscribe([&starts](const uint64_t&) { ++starts; });
    std::thread watcher;
    if (watch)
        watcher = std::thread([&, result] {
            // the running iteration = iterations completed so far (samples_per_pixel is bumped once per iteration, before
            // its on_iteration_end, abstract_pathtracer.cpp:105-109).  progress keeps the finished iteration's 1.0 until
            // the next one reports its first slice, as the reference keeps its last scan line's value: one drop from 1.0
            // per iteration is the restart, any other decrease is an error
            uint64_t samples = 0, at_end = 0;
            float last = 0.0f, lo = 2.0f, hi = 0.0f, fired_at = -1.0f;
            bool monotone = true, dropped = false;
            while (!finished.load()) {
                const uint64_t e = spi ? result->samples_per_pixel.load() / spi : 0;
                const float pr = result->progress.load();
                if ((spi ? result->samples_per_pixel.load() / spi : 0) == e) { // both reads belong to the same iteration
                    if (e != at_end) {
                        at_end = e;
                        dropped = false;
                    }
                    if (pr < last) {
                        if (dropped || last != 1.0f) monotone = false;
                        dropped = true;
                    }
                    last = pr;
                    if (pr > 0.0f) {
                        lo = std::min(lo, pr);
                        hi = std::max(hi, pr);
                    }
                    ++samples;
                    if (fired_at < 0.0f && e == watch_iteration && pr >= watch_progress && pr < 1.0f) {
                        result->interrupt();
                        fired_at = pr;
                    }
                }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            std::printf("watch: samples=%" PRIu64 " monotone=%d min=%.4f max=%.4f interrupted_at_progress=%.4f\n", samples, monotone ? 1 : 0,
                        lo, hi, fired_at);
        });
    result->on_iteration_end.subscribe([&, result](const rt::image&, const uint64_t& iteration) {
        ++ends;
        const float elapsed = result->get_elapsed_time();
        std::printf("iteration %" PRIu64 " / %" PRIu64 ", %.1f spp/sec\n", iteration + 1, iterations,
                    result->samples_per_pixel.load() / elapsed);
        if (interrupt_after && iteration + 1 >= interrupt_after) result->interrupt();
    });
    result->on_end.subscribe([&](const rt::image& image) {
        if (!out_is_png) {
            std::ofstream out(out_file, std::ios::binary);
            out.write(reinterpret_cast<const char*>(image.data()), std::streamsize(image.get_width() * image.get_height() * 12));
        }
        if (ldr_file.empty() && png_file.empty()) return;
        // CLI/main.cpp:85-104: tone map, gamma, truncating 8-bit pack
        std::vector<uint32_t> pixels(image.get_height() * image.get_width());
        for (uint32_t x = 0; x < image.get_width(); ++x)
            for (uint32_t y = 0; y < image.get_height(); ++y) {
                glm::vec3 color = image.get_pixel(x, y);
                color = glm::vec3(1.0f) - glm::exp(-color);
                color = glm::pow(color, glm::vec3(1.0f / 2.2f));
                pixels[y * image.get_width() + x] = (uint32_t(color.r * 255) << 0) | (uint32_t(color.g * 255) << 8) |
                                                    (uint32_t(color.b * 255) << 16) | (uint32_t(255) << 24);
            }
        if (!ldr_file.empty()) {
            std::ofstream ldr(ldr_file, std::ios::binary);
            ldr.write(reinterpret_cast<const char*>(pixels.data()), std::streamsize(pixels.size() * 4));
        }
        if (!png_file.empty()) {
#ifdef PTK_HAVE_STB_WRITE
            if (!stbi_write_png(png_file.c_str(), int(image.get_width()), int(image.get_height()), 4, pixels.data(), 0)) // CLI/main.cpp:107
                std::fprintf(stderr, "rt_render: cannot write %s\n", png_file.c_str());
#else
            std::fprintf(stderr, "rt_render: built without stb_image_write.h, --png-out ignored\n");
#endif
        }
    });
    result->wait();
    finished = true;
    if (watcher.joinable()) watcher.join();

    if (!result->error().empty()) {
        std::fprintf(stderr, "rt_render: %s\n", result->error().c_str());
        return 2;
    }
    const double first_prepare_ms = pathtracer.last_run.prepare_ms;
    if (then_camera) {
        // Sandbox: the camera moved, run() again on the same pathtracer and scene objects
        scene.camera.position = then_pos;
        scene.camera.set_direction(then_dir);
        auto again = pathtracer.run(view_params, trace_params, scene);
        again->on_end.subscribe([&](const rt::image& image) {
            if (then_out.empty()) return;
            std::ofstream out(then_out, std::ios::binary);
            out.write(reinterpret_cast<const char*>(image.data()), std::streamsize(image.get_width() * image.get_height() * 12));
        });
        again->wait();
        if (!again->error().empty()) {
            std::fprintf(stderr, "rt_render: %s\n", again->error().c_str());
            return 2;
        }
        std::printf("second run: reused=%d prepare_ms=%.3f first_prepare_ms=%.3f\n", pathtracer.last_run.scene_reused ? 1 : 0,
                    pathtracer.last_run.prepare_ms, first_prepare_ms);
    }
    std::printf("done: iteration=%" PRIu64 " spp=%" PRIu64 " paths=%" PRIu64 " rays=%" PRIu64 " events=%" PRIu64 "/%" PRIu64 "\n",
                result->iteration.load(), result->samples_per_pixel.load(), result->paths_traced.load(),
                result->rays_cast.load(), starts.load(), ends.load());
    return 0;
}
