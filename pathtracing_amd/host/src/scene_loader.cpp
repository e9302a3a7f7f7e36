// scene_loader.cpp — rt::utility::load_scene (reference src/PathtracingUtility/scene_loader.cpp:39-195)
// and rt::utility::debug_pathtracer (debug_pathtracer.cpp:8-32).
//
// The reference parses with nlohmann::json (vendored there, not here); the schema is small, so this file
// carries its own JSON reader (objects, arrays, strings with escapes, numbers, true / false / null).
// Numbers are read as double and narrowed to float where the reference's get<float>() does.  Processing
// order and error behaviour follow the reference: meshes -> camera -> samplers -> background -> nodes;
// a file that cannot be opened logs and returns an empty scene (scene_loader.cpp:47-51); an unknown
// mesh / sampler id throws std::out_of_range like std::map::at (scene_loader.cpp:139,165,175-184);
// malformed JSON throws (nlohmann would throw parse_error).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <utility>
#include <vector>

#include "ptk_rt/utility.hpp"

namespace rt {
namespace utility {

namespace {

struct json {
    enum kind_t { null_v, bool_v, number_v, string_v, array_v, object_v } kind = null_v;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<json> items;                           // array
    std::vector<std::pair<std::string, json>> members; // object, file order

    bool contains(const std::string& key) const
    {
        for (const auto& m : members)
            if (m.first == key) return true;
        return false;
    }
    const json& operator[](const std::string& key) const
    {
        const json* found = nullptr;
        for (const auto& m : members)
            if (m.first == key) found = &m.second; // duplicate keys: the last one wins, as in nlohmann
        if (!found) throw std::out_of_range("json: key '" + key + "' not found");
        return *found;
    }
    const json& operator[](size_t i) const { return items.at(i); }
    float as_float() const
    {
        if (kind != number_v) throw std::runtime_error("json: number expected");
        return static_cast<float>(num);
    }
    const std::string& as_string() const
    {
        if (kind != string_v) throw std::runtime_error("json: string expected");
        return str;
    }
    bool as_bool() const
    {
        if (kind != bool_v) throw std::runtime_error("json: boolean expected");
        return b;
    }
    glm::vec3 as_vec3() const { return { (*this)[0].as_float(), (*this)[1].as_float(), (*this)[2].as_float() }; } // scene_loader.cpp:17-19
};

class json_reader {
public:
    explicit json_reader(const std::string& text) : p(text.data()), end(text.data() + text.size()) {}
    json parse()
    {
        json v = value();
        ws();
        if (p != end) fail("trailing characters");
        return v;
    }

private:
    const char* p;
    const char* end;
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("json: ") + what); }
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool eat(char c)
    {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    json value()
    {
        ws();
        if (p >= end) fail("unexpected end");
        json v;
        if (*p == '{') {
            ++p;
            v.kind = json::object_v;
            if (eat('}')) return v;
            do {
                ws();
                json key = string();
                if (!eat(':')) fail("':' expected");
                v.members.emplace_back(key.str, value());
            } while (eat(','));
            if (!eat('}')) fail("'}' expected");
        } else if (*p == '[') {
            ++p;
            v.kind = json::array_v;
            if (eat(']')) return v;
            do v.items.push_back(value());
            while (eat(','));
            if (!eat(']')) fail("']' expected");
        } else if (*p == '"') {
            v = string();
        } else if (end - p >= 4 && std::string(p, 4) == "true") {
            p += 4;
            v.kind = json::bool_v;
            v.b = true;
        } else if (end - p >= 5 && std::string(p, 5) == "false") {
            p += 5;
            v.kind = json::bool_v;
        } else if (end - p >= 4 && std::string(p, 4) == "null") {
            p += 4;
        } else {
            const char* q = p;
            if (q < end && *q == '-') ++q;
            while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) ++q;
            if (q == p) fail("value expected");
            const std::string text(p, q);
            char* stop = nullptr;
            v.num = std::strtod(text.c_str(), &stop);
            if (stop != text.c_str() + text.size()) fail("bad number");
            v.kind = json::number_v;
            p = q;
        }
        return v;
    }
    json string()
    {
        if (p >= end || *p != '"') fail("string expected");
        ++p;
        json v;
        v.kind = json::string_v;
        while (p < end && *p != '"') {
            if (*p != '\\') {
                v.str.push_back(*p++);
                continue;
            }
            if (++p >= end) fail("bad escape");
            switch (*p++) {
            case 'n': v.str.push_back('\n'); break;
            case 't': v.str.push_back('\t'); break;
            case 'r': v.str.push_back('\r'); break;
            case 'b': v.str.push_back('\b'); break;
            case 'f': v.str.push_back('\f'); break;
            case 'u': {
                if (end - p < 4) fail("bad \\u escape");
                const unsigned cp = unsigned(std::strtoul(std::string(p, 4).c_str(), nullptr, 16));
                p += 4;
                if (cp < 0x80) v.str.push_back(char(cp));
                else if (cp < 0x800) {
                    v.str.push_back(char(0xC0 | (cp >> 6)));
                    v.str.push_back(char(0x80 | (cp & 0x3F)));
                } else {
                    v.str.push_back(char(0xE0 | (cp >> 12)));
                    v.str.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
                    v.str.push_back(char(0x80 | (cp & 0x3F)));
                }
                break;
            }
            default: v.str.push_back(p[-1]); break; // \" \\ \/
            }
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return v;
    }
};

} // namespace

scene load_scene(std::string_view file_name)
{
    scene result;
    std::ifstream is{ std::string(file_name) };
    if (!is.good()) {
        std::fprintf(stderr, "[rt] Can't open file: %s\n", std::string(file_name).c_str());
        return result;
    }
    std::stringstream buffer;
    buffer << is.rdbuf();
    const json scene_def = json_reader(buffer.str()).parse();

    std::map<std::string, std::shared_ptr<rt::mesh>> meshes;
    std::map<std::string, std::shared_ptr<rt::sampler_2d>> samplers_2d;
    std::map<std::string, std::shared_ptr<rt::sampler_3d>> samplers_3d;

    if (scene_def.contains("meshes")) { // scene_loader.cpp:60-74
        for (const json& mesh_def : scene_def["meshes"].items) {
            std::vector<std::string> ids;
            for (const json& id : mesh_def["ids"].items) ids.push_back(id.as_string());
            for (auto& [name, mesh] : load_meshes_from_wavefront(mesh_def["file"].as_string()))
                for (const std::string& id : ids)
                    if (id == name) {
                        meshes[name] = mesh;
                        break;
                    }
        }
    }

    if (scene_def.contains("camera")) { // scene_loader.cpp:76-89
        const json& camera = scene_def["camera"];
        if (camera.contains("position")) result.camera.position = camera["position"].as_vec3();
        if (camera.contains("direction")) result.camera.set_direction(camera["direction"].as_vec3());
    }

    if (scene_def.contains("samplers")) { // scene_loader.cpp:91-132
        for (const json& sampler_def : scene_def["samplers"].items) {
            if (sampler_def.contains("file")) {
                const std::string id = sampler_def["id"].as_string();
                auto image = std::make_shared<rt::image>();
                image->load(sampler_def["file"].as_string());
                if (sampler_def.contains("ldr") && sampler_def["ldr"].as_bool()) image->to_ldr();
                if (sampler_def.contains("mode")) {
                    const std::string& mode = sampler_def["mode"].as_string(); // scene_loader.cpp:30-34: other strings leave it alone
                    if (mode == "linear") image->sample_mode = rt::sample_mode::linear;
                    else if (mode == "nearest") image->sample_mode = rt::sample_mode::nearest;
                }
                const std::string type = sampler_def.contains("type") ? sampler_def["type"].as_string() : "image";
                if (type == "image") samplers_2d[id] = std::move(image);
                else if (type == "equirectangular") samplers_3d[id] = std::make_shared<rt::equirectangular_map>(image);
                else std::fprintf(stderr, "[rt] Unkwnown sampler type: %s\n", type.c_str());
            } else if (sampler_def.contains("color")) {
                const std::string id = sampler_def["id"].as_string();
                auto sampler = std::make_shared<rt::color_sampler>(sampler_def["color"].as_vec3());
                samplers_2d[id] = sampler;
                samplers_3d[id] = sampler;
            }
        }
    }

    if (scene_def.contains("background")) { // scene_loader.cpp:134-141
        const json& background = scene_def["background"];
        if (background.contains("color")) result.background = samplers_3d.at(background["color"].as_string());
    }

    if (scene_def.contains("nodes")) { // scene_loader.cpp:143-191
        for (const json& node_def : scene_def["nodes"].items) {
            auto node = std::make_shared<rt::scene_node>();
            if (node_def.contains("translate")) node->translate(node_def["translate"].as_vec3());
            if (node_def.contains("rotate")) {
                const glm::vec3 angles = node_def["rotate"].as_vec3();
                node->multiply(glm::rotate(glm::radians(angles.z), glm::vec3(0.0f, 0.0f, 1.0f)) *
                               glm::rotate(glm::radians(angles.y), glm::vec3(0.0f, 1.0f, 0.0f)) *
                               glm::rotate(glm::radians(angles.x), glm::vec3(1.0f, 0.0f, 0.0f)));
            }
            if (node_def.contains("scale")) node->scale(node_def["scale"].as_vec3());
            if (node_def.contains("mesh")) node->shape = meshes.at(node_def["mesh"].as_string());
            else if (node_def.contains("shape") && node_def["shape"].as_string() == "sphere") node->shape = std::make_shared<rt::sphere>();
            if (node_def.contains("material")) {
                const json& mat_def = node_def["material"];
                if (mat_def.contains("albedo")) node->material.albedo = samplers_2d.at(mat_def["albedo"].as_string());
                if (mat_def.contains("emission")) node->material.emission = samplers_2d.at(mat_def["emission"].as_string());
                if (mat_def.contains("roughness")) node->material.roughness = samplers_2d.at(mat_def["roughness"].as_string());
                if (mat_def.contains("metallic")) node->material.metallic = samplers_2d.at(mat_def["metallic"].as_string());
            }
            result.nodes.push_back(std::move(node));
        }
    }
    return result;
}

// debug_pathtracer.cpp:8-32.  A user-style subclass: abstract_pathtracer::run drives it on host threads
// through this virtual, as the reference does (rtsb.cpp:378).  The reference falls off the end of the
// function for an out-of-range mode (undefined); black is returned here.
glm::vec3 debug_pathtracer::trace(const view_parameters&, const ray& ray, const scene& scene)
{
    auto [result, node] = scene.cast_ray(ray);
    if (!result.hit) return scene.background ? scene.background->sample(ray.direction) : glm::vec3(0.0f);
    switch (current_mode) {
    case mode::albedo: return node->material.albedo->sample(result.uv);
    case mode::emission: return node->material.emission->sample(result.uv);
    case mode::metallic: return node->material.metallic->sample(result.uv);
    case mode::roughness: return node->material.roughness->sample(result.uv);
    case mode::normal: return result.normal * 0.5f + 0.5f;
    }
    return glm::vec3(0.0f);
}

} // namespace utility
} // namespace rt
