// mesh_loader.cpp — rt::utility::load_meshes_from_wavefront (reference
// src/PathtracingUtility/mesh_loader.cpp:12-158) as a streaming reader.
//
// The reference runs up to ten std::regex_search calls per line (minutes at millions of faces).  This
// reader scans each line by hand but accepts EXACTLY the reference's grammar, because what it accepts
// decides which triangles exist:
//   * the tests run in the reference's order — v, vn, vt, f (4 index styles), o|g, # — and the first one
//     that matches anywhere in the line wins (regex_search, not regex_match: "usemtl foo bar" starts a
//     mesh called "bar", because "o bar" matches [og]\s+(.+));
//   * a number is -?[\d.]+ : no exponents (a trailing "3e-5" reads as 3; an exponent in an earlier number
//     makes the whole line match nothing), no leading '+', "1.2.3" reads as 1.2 (std::stof stops at the second
//     dot), a lone "." throws std::invalid_argument as std::stof does;
//   * faces keep their first three corners (quads are truncated), indices are global and 1-based;
//   * 'o' / 'g' close the current mesh if it has triangles (compile() + store under the current name,
//     later meshes of the same name replace earlier ones); meshes come back in a std::map, i.e. sorted
//     by name; a CR at the end of a line is never part of a name ('.' does not match it).
// Index 0 or an index beyond the data read so far is undefined behaviour in the reference
// (vertices[-1]); here it throws std::out_of_range.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <vector>

#include "ptk_rt/utility.hpp"

namespace rt {
namespace utility {

namespace {

// std::regex (ECMAScript, "C" locale) character classes
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_line_terminator(char c) { return c == '\n' || c == '\r'; } // what '.' refuses (U+2028/9 aside)

struct span {
    const char* b;
    const char* e;
};

inline bool skip_spaces(const char*& p, const char* end) // \s+
{
    const char* q = p;
    while (q < end && is_space(*q)) ++q;
    if (q == p) return false;
    p = q;
    return true;
}

inline bool real(const char*& p, const char* end, span& out) // (-?[\d.]+)
{
    const char* q = p;
    if (q < end && *q == '-') ++q;
    const char* d = q;
    while (q < end && (is_digit(*q) || *q == '.')) ++q;
    if (q == d) return false;
    out = { p, q };
    p = q;
    return true;
}

inline bool integer(const char*& p, const char* end, span& out) // ([0-9]+)
{
    const char* q = p;
    while (q < end && is_digit(*q)) ++q;
    if (q == p) return false;
    out = { p, q };
    p = q;
    return true;
}

inline bool literal(const char*& p, const char* end, const char* text)
{
    const char* q = p;
    for (; *text; ++text, ++q)
        if (q >= end || *q != *text) return false;
    p = q;
    return true;
}

// regex_search: the leftmost position from which `body` matches.  Every pattern here starts with a
// literal, and none of them can match in a second way from the same start (a greedy class is always
// followed by a character outside it), so one attempt per start position is the whole search.
template <typename Body>
inline bool search(const char* begin, const char* end, Body body)
{
    for (const char* s = begin; s < end; ++s)
        if (body(s)) return true;
    return false;
}

float to_float(span s) // std::stof(matches.str(k))
{
    const std::string text(s.b, s.e);
    char* stop = nullptr;
    const float v = std::strtof(text.c_str(), &stop);
    if (stop == text.c_str()) throw std::invalid_argument("stof");
    return v;
}

size_t to_index(span s, size_t limit, const char* what) // container[std::stoi(matches.str(k)) - 1]
{
    const std::string text(s.b, s.e);
    const long v = std::strtol(text.c_str(), nullptr, 10);
    if (v < 1 || size_t(v) > limit) throw std::out_of_range(std::string("wavefront ") + what + " index " + text + " out of range");
    return size_t(v - 1);
}

} // namespace

std::map<std::string, std::shared_ptr<rt::mesh>> load_meshes_from_wavefront(std::string_view file_name)
{
    std::map<std::string, std::shared_ptr<rt::mesh>> result;
    std::ifstream is{ std::string(file_name) };
    if (!is.is_open()) {
        std::fprintf(stderr, "[rt] Can't open file: %s\n", std::string(file_name).c_str()); // mesh_loader.cpp:19-23
        return result;
    }

    std::string line;
    std::string current_mesh_name = "default";
    auto current_mesh = std::make_shared<rt::mesh>();
    std::vector<glm::vec3> vertices, normals;
    std::vector<glm::vec2> uvs;

    const auto check_new_mesh = [&] { // mesh_loader.cpp:31-37
        if (current_mesh->get_triangles().size() > 0) {
            current_mesh->compile();
            result[current_mesh_name] = std::move(current_mesh);
            current_mesh = std::make_shared<rt::mesh>();
        }
    };

    while (is.good()) { // mesh_loader.cpp:62: one more pass after the last line, on an empty string
        std::getline(is, line);
        const char* const b = line.data();
        const char* const e = b + line.size();
        span g[9];

        // r_vertex / r_normal / r_uv (mesh_loader.cpp:41-43,67-89)
        auto reals = [&](const char* tag, int count) {
            return search(b, e, [&](const char* p) {
                if (!literal(p, e, tag)) return false;
                for (int k = 0; k < count; ++k)
                    if (!skip_spaces(p, e) || !real(p, e, g[k])) return false;
                return true;
            });
        };
        // r_face0..3 (mesh_loader.cpp:45-60): three corners of `per_corner` indices joined by `sep`
        auto face = [&](int per_corner, const char* sep) {
            return search(b, e, [&](const char* p) {
                if (!literal(p, e, "f")) return false;
                for (int c = 0; c < 3; ++c) {
                    if (!skip_spaces(p, e)) return false;
                    for (int k = 0; k < per_corner; ++k) {
                        if (k && !literal(p, e, sep)) return false;
                        if (!integer(p, e, g[c * per_corner + k])) return false;
                    }
                }
                return true;
            });
        };
        // r_object "[og]\s+(.+)" / r_comment "#\s+(.+)": \s+ gives characters back until (.+) can take one
        auto named = [&](bool (*head)(char), span& name) {
            return search(b, e, [&](const char* p) {
                if (!head(*p)) return false;
                const char* ws = ++p;
                if (!skip_spaces(p, e)) return false;
                for (const char* start = p; start > ws; --start) {
                    const char* q = start;
                    while (q < e && !is_line_terminator(*q)) ++q;
                    if (q > start) {
                        name = { start, q };
                        return true;
                    }
                }
                return false;
            });
        };

        if (reals("v", 3)) {
            vertices.push_back({ to_float(g[0]), to_float(g[1]), to_float(g[2]) });
        } else if (reals("vn", 3)) {
            normals.push_back({ to_float(g[0]), to_float(g[1]), to_float(g[2]) });
        } else if (reals("vt", 2)) {
            uvs.push_back({ to_float(g[0]), to_float(g[1]) });
        } else if (face(1, "")) { // f a b c
            triangle& t = current_mesh->add_triangle();
            for (size_t i = 0; i < 3; ++i) t.vertices[i].position = vertices[to_index(g[i], vertices.size(), "vertex")];
        } else if (face(2, "//")) { // f a//n ...
            triangle& t = current_mesh->add_triangle();
            for (size_t i = 0; i < 3; ++i) {
                t.vertices[i].position = vertices[to_index(g[i * 2], vertices.size(), "vertex")];
                t.vertices[i].normal = normals[to_index(g[i * 2 + 1], normals.size(), "normal")];
            }
        } else if (face(2, "/")) { // f a/t ...
            triangle& t = current_mesh->add_triangle();
            for (size_t i = 0; i < 3; ++i) {
                t.vertices[i].position = vertices[to_index(g[i * 2], vertices.size(), "vertex")];
                t.vertices[i].uv = uvs[to_index(g[i * 2 + 1], uvs.size(), "uv")];
            }
        } else if (face(3, "/")) { // f a/t/n ...
            triangle& t = current_mesh->add_triangle();
            for (size_t i = 0; i < 3; ++i) {
                t.vertices[i].position = vertices[to_index(g[i * 3], vertices.size(), "vertex")];
                t.vertices[i].uv = uvs[to_index(g[i * 3 + 1], uvs.size(), "uv")];
                t.vertices[i].normal = normals[to_index(g[i * 3 + 2], normals.size(), "normal")];
            }
        } else {
            span name{ nullptr, nullptr };
            if (named([](char c) { return c == 'o' || c == 'g'; }, name)) {
                check_new_mesh();
                current_mesh_name.assign(name.b, name.e);
            }
            // comments and unparsable lines are only logged by the reference (mesh_loader.cpp:140-147)
        }
    }

    check_new_mesh();
    return result;
}

} // namespace utility
} // namespace rt
