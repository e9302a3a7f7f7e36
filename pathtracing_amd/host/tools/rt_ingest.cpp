// rt_ingest — dumps what the host library's loaders produce, so tests can hold them against the Python ingest and
// the harness can decode textures exactly as the reference does (stb_image through rt::image::load).
//   rt_ingest obj   <file.obj> <out.bin>   per mesh (sorted by name): u32 name length, name, u64 triangle count,
//                                          count * 24 floats (3 x position3 normal3 uv2)
//   rt_ingest image <file>     <out.pfm>   the image as image::load stores it (rows bottom-up), little-endian PFM
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include <mesh_loader.h>
#include <sampler.h>

int main(int argc, char** argv)
{
    if (argc != 4) {
        std::fprintf(stderr, "usage: rt_ingest obj|image <in> <out>\n");
        return 2;
    }
    const std::string what = argv[1];
    if (what == "obj") {
        const auto meshes = rt::utility::load_meshes_from_wavefront(argv[2]);
        std::ofstream out(argv[3], std::ios::binary);
        for (const auto& [name, mesh] : meshes) {
            const uint32_t len = uint32_t(name.size());
            const uint64_t count = mesh->get_triangles().size();
            out.write(reinterpret_cast<const char*>(&len), 4);
            out.write(name.data(), len);
            out.write(reinterpret_cast<const char*>(&count), 8);
            for (const rt::triangle& t : mesh->get_triangles()) out.write(reinterpret_cast<const char*>(t.vertices.data()), 96);
        }
        std::printf("%zu meshes\n", meshes.size());
        return out ? 0 : 1;
    }
    if (what == "image") {
        rt::image img;
        img.load(argv[2]);
        if (img.get_width() == 0 || img.get_height() == 0) return 1;
        std::ofstream out(argv[3], std::ios::binary);
        out << "PF\n" << img.get_width() << " " << img.get_height() << "\n-1.0\n";
        out.write(reinterpret_cast<const char*>(img.data()), std::streamsize(img.get_width() * img.get_height() * 12));
        std::printf("%zu x %zu\n", img.get_width(), img.get_height());
        return out ? 0 : 1;
    }
    std::fprintf(stderr, "rt_ingest: unknown mode %s\n", what.c_str());
    return 2;
}
