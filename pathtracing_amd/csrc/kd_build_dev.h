// kd_build_dev.h — C++ interface between the GPU KD builder (kd_build_dev.hip) and ptk.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ptk.h"

// A mesh's KD-tree left in device memory by the builder, with the triangles it was built from.
// Same content as ptk_kd_build's output: nodes in DFS preorder, prims in leaf order.
struct PtkDeviceTree {
    ptk_triangle* tris = nullptr;
    ptk_kd_node* nodes = nullptr;
    uint32_t* prims = nullptr;
    uint32_t num_triangles = 0;
    ptk_kd_info info{};
    void release()
    {
        if (tris) (void)hipFree(tris);
        if (nodes) (void)hipFree(nodes);
        if (prims) (void)hipFree(prims);
        tris = nullptr;
        nodes = nullptr;
        prims = nullptr;
    }
};

// Builds on the current device, on `stream`; `triangles` is a host pointer, `bounds` (min xyz, max xyz)
// may be null = mesh::compile's bounds (scene.cpp:123-137).  The caller releases `out`.
int ptk_kd_build_device_keep(hipStream_t stream, const ptk_triangle* triangles, uint64_t num_triangles, const float* bounds,
                             uint32_t root_depth, PtkDeviceTree* out, std::string* err);
int ptk_kd_tree_download(hipStream_t stream, const PtkDeviceTree& tree, std::vector<ptk_kd_node>& nodes_out,
                         std::vector<uint32_t>& prims_out, std::string* err);
// build + download + release
int ptk_kd_build_device(hipStream_t stream, const ptk_triangle* triangles, uint64_t num_triangles, const float* bounds,
                        uint32_t root_depth, ptk_kd_info* info, std::vector<ptk_kd_node>& nodes_out,
                        std::vector<uint32_t>& prims_out, std::string* err);
