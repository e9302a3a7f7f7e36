/*
 * ptk.h — C ABI of the MI355X path-tracing kernel library (libptk.so).
 *
 * This is the drop-in boundary between host C++ (the re-authored rt:: API in
 * pathtracing_amd/host, which mirrors the reference's Pathtracing library) and
 * the hand-written gfx950 HIP kernels.  POD only, plain pointers and sizes, int
 * status (0 = ok), no exceptions cross it.  Every entry point cites the
 * reference interface (file:line under /root/reference) whose work it takes
 * over.  The reference has no FFI of its own (it is one C++17 static library,
 * premake5.lua:106-124); these are the calls its `abstract_pathtracer::run`
 * driver thread makes once the per-pixel loop is moved to the GPU.
 *
 * Conventions
 *   - all matrices are column-major float[16] exactly as glm::mat4 stores them
 *     (m[col*4+row]);
 *   - all arithmetic is IEEE fp32, kernels are built with -ffp-contract=off so
 *     hit decisions replay the reference's x86-64 -O2 arithmetic bit for bit;
 *   - pointers named *_dev are device (HBM) pointers on the context's GPU,
 *     everything else is host memory that the library copies during the call;
 *   - one context per driver thread; calls on one context are serialised by the
 *     caller.  Work is enqueued on the context's stream (ptk_ctx_set_stream);
 *     calls that return host data synchronise that stream, the *_dev calls do not.
 */
#ifndef PTK_H
#define PTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTK_OK 0
#define PTK_ERR_INVALID 1   /* bad argument / shape mismatch                   */
#define PTK_ERR_HIP 2       /* a HIP runtime call failed (see ptk_last_error)  */
#define PTK_ERR_STATE 3     /* call made in the wrong order (no scene, ...)     */
#define PTK_ERR_NOMEM 4
#define PTK_INTERRUPTED 5   /* not an error: the call stopped early on ptk_request_interrupt / ptk_group_request_interrupt */

#define PTK_MAX_DEPTH 5u        /* pathtracer.cpp:16  trace_recursive(..., 5)    */
#define PTK_EPSILON 1e-3f       /* pathtracer.h:19    s_epsilon                  */
#define PTK_KD_MAX_DEPTH 100u   /* scene.cpp:220                                 */

/* rt::vertex, scene.h:40-45 (32 B, same member order). */
typedef struct ptk_vertex {
    float position[3];
    float normal[3];
    float uv[2];
} ptk_vertex;

/* rt::triangle::vertices, scene.h:101 (the private precomputed members of
 * scene.h:123-126 are rebuilt by the library, scene.cpp:91-107). */
typedef struct ptk_triangle {
    ptk_vertex v[3];
} ptk_triangle;

/* rt::mesh as the caller sees it: mesh::add_triangle order, scene.h:262-268. */
typedef struct ptk_mesh_desc {
    const ptk_triangle* triangles;
    uint64_t num_triangles;
} ptk_mesh_desc;

enum { PTK_SAMPLER_COLOR = 0, PTK_SAMPLER_IMAGE = 1 };          /* sampler.h:48,73 */
enum { PTK_MODE_LINEAR = 0, PTK_MODE_NEAREST = 1 };             /* sampler.h:65-68 */
enum { PTK_SHAPE_SPHERE = 0, PTK_SHAPE_MESH = 1 };              /* scene.h:231,244 */
enum { PTK_BACKGROUND_COLOR = 0, PTK_BACKGROUND_EQUIRECT = 1 }; /* sampler.h:48,145 */

/* rt::image pixel store, sampler.h:138-139: row-major float RGB, row 0 is the
 * row image::load put first (bottom of the file, sampler.cpp:81). */
typedef struct ptk_texture_desc {
    uint32_t width, height;
    const float* rgb; /* width*height*3 floats */
} ptk_texture_desc;

/* rt::sampler_2d: color_sampler (sampler.h:48-63) or image (sampler.h:73-140). */
typedef struct ptk_sampler_desc {
    uint32_t kind;    /* PTK_SAMPLER_*                       */
    float color[3];   /* PTK_SAMPLER_COLOR                   */
    uint32_t texture; /* PTK_SAMPLER_IMAGE: index in textures */
    uint32_t mode;    /* PTK_MODE_*  (image::sample_mode)     */
} ptk_sampler_desc;

/* rt::scene_node, scene.h:284-350: the three matrices update_matrices keeps
 * (scene.cpp:295-299) + shape + the four material samplers (scene.h:133-140). */
typedef struct ptk_node_desc {
    float transform[16];
    float inv_transform[16];
    float normal_transform[16];
    uint32_t shape_kind; /* PTK_SHAPE_*                */
    uint32_t mesh;       /* PTK_SHAPE_MESH: mesh index */
    uint32_t albedo, emission, roughness, metallic; /* sampler indices */
} ptk_node_desc;

/* rt::scene::background (scene.h:369): color_sampler or equirectangular_map. */
typedef struct ptk_background_desc {
    uint32_t kind;    /* PTK_BACKGROUND_* */
    float color[3];
    uint32_t texture; /* PTK_BACKGROUND_EQUIRECT */
    uint32_t mode;
} ptk_background_desc;

/* rt::camera, scene.h:27-37 (direction already normalised by set_direction). */
typedef struct ptk_camera_desc {
    float position[3];
    float direction[3];
} ptk_camera_desc;

/* rt::scene, scene.h:355-394, flattened to tables. */
typedef struct ptk_scene_desc {
    const ptk_node_desc* nodes;
    uint32_t num_nodes;
    const ptk_mesh_desc* meshes;
    uint32_t num_meshes;
    const ptk_sampler_desc* samplers;
    uint32_t num_samplers;
    const ptk_texture_desc* textures;
    uint32_t num_textures;
    ptk_background_desc background;
    ptk_camera_desc camera;
} ptk_scene_desc;

/* One node of the flattened KD-tree, DFS preorder (left subtree, then right),
 * exactly the order mesh::intersect_internal visits (scene.cpp:188-212).
 * `skip` = index of the first node after this node's subtree, so the
 * reference's recursion becomes: hit box -> i+1, miss box -> skip. */
typedef struct ptk_kd_node {
    float bmin[3];
    float bmax[3];
    uint32_t skip;
    uint32_t first_prim; /* offset into the mesh's leaf-ordered prim list */
    uint32_t num_prims;  /* 0 for interior nodes (scene.cpp:280-284)      */
    uint32_t depth;      /* split axis of an interior node = depth % 3 (scene.cpp:230) */
    /* Not in the reference: the box of the triangles stored below this node (whole
     * triangles, slightly inflated).  bmin/bmax are the reference's cut boxes, which
     * decide whether a leaf's triangles are tested at all; tmin/tmax only let the GPU
     * skip subtrees that cannot hold a closer hit ("pruned traversal"). */
    float tmin[3];
    float tmax[3];
} ptk_kd_node;

typedef struct ptk_kd_info {
    uint32_t num_nodes;
    uint32_t num_leaves;
    uint64_t num_leaf_prims; /* sum of leaf sizes (triangles are duplicated) */
    uint32_t max_depth;
    float bmin[3], bmax[3];  /* mesh::m_bounds, scene.cpp:123-137 */
} ptk_kd_info;

typedef struct ptk_counters {
    uint64_t paths; /* trace() calls (pathtracer.cpp:14)                      */
    uint64_t rays;  /* scene::cast_ray calls (scene.cpp:355), <= 5 per path   */
} ptk_counters;

typedef struct ptk_ctx ptk_ctx;

/* ---- context ------------------------------------------------------------ */

/* Creates a context on HIP device `device`.  No reference counterpart (the
 * reference has no device); stands where pathtracer_result's driver thread is
 * created, abstract_pathtracer.cpp:121-125. */
int ptk_ctx_create(int device, ptk_ctx** out);
void ptk_ctx_destroy(ptk_ctx* ctx);
/* Last error text of `ctx` (or of the last failed ptk_ctx_create if NULL). */
const char* ptk_last_error(const ptk_ctx* ctx);
/* Use `hip_stream` (a hipStream_t) for all later work; NULL = the context's own. */
int ptk_ctx_set_stream(ptk_ctx* ctx, void* hip_stream);
/* The hipStream_t later work is enqueued on (the context's own unless ptk_ctx_set_stream changed it). */
void* ptk_ctx_get_stream(ptk_ctx* ctx);
int ptk_ctx_synchronize(ptk_ctx* ctx);
/* pathtracer_result::interrupt (abstract_pathtracer.h:150; the flag the driver's loops poll,
 * abstract_pathtracer.cpp:38,63,68,72).  May be called from any thread while another thread is inside
 * ptk_render_image on the same context: that call stops before its next launch chain (it hands the GPU at most
 * max_paths_per_launch paths at a time, two chains in flight), folds the pixels finished so far into the image
 * and returns PTK_INTERRUPTED.  A request with no ptk_render_image running is served by the next one. */
int ptk_request_interrupt(ptk_ctx* ctx);

/* ---- scene ---------------------------------------------------------------
 * Replaces scene::compile (scene.cpp:331-348) -> mesh::compile
 * (scene.cpp:123-140) -> kd_tree_node ctor (scene.cpp:214-286) and
 * triangle::update (scene.cpp:91-107): builds each mesh's KD-tree with the
 * reference's topology (on the host for small meshes, on the GPU from 32768
 * triangles on: PTK_OPT_DEVICE_KD; both give the same tree byte for byte),
 * flattens it and uploads every table to HBM.  Invalidates the view: call
 * ptk_render_begin again before rendering. */
int ptk_scene_upload(ptk_ctx* ctx, const ptk_scene_desc* scene);
/* Only the camera changed (Sandbox orbit camera, rtsb.cpp:295-299): position and
 * direction replace rt::scene::camera; scene tables, viewport, fov and ray queues
 * stay, and the view basis of abstract_pathtracer.cpp:20-22 is rebuilt from the new
 * direction, so the next ptk_render_samples_dev renders the new view (no
 * ptk_render_begin needed if one was made before; work already enqueued on the
 * stream keeps the old camera). */
int ptk_scene_set_camera(ptk_ctx* ctx, const ptk_camera_desc* camera);

/* Host-side KD build alone (no GPU needed): kd_tree_node ctor, scene.cpp:214-286.
 * Two-call pattern: nodes/prims may be NULL to query sizes via `info`.
 * prims[k] = index into `triangles` of the k-th leaf-ordered primitive. */
int ptk_kd_build(const ptk_triangle* triangles, uint64_t num_triangles,
                 ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap,
                 uint32_t* prims, uint64_t prims_cap);
/* The public kd_tree_node constructor (scene.h:159, scene.cpp:214): the caller supplies
 * the root box (bounds = {min xyz, max xyz}) and the depth of the root, which selects the
 * first split axis (depth % 3) and counts towards PTK_KD_MAX_DEPTH. */
int ptk_kd_build_bounded(const ptk_triangle* triangles, uint64_t num_triangles,
                         const float bounds[6], uint32_t root_depth,
                         ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap,
                         uint32_t* prims, uint64_t prims_cap);

/* The same build on the GPU of `ctx` (kd_tree_node ctor scene.cpp:214-286, bounds
 * of mesh::compile scene.cpp:123-137 when `bounds` is NULL).  `triangles` is a host pointer; nodes and
 * prims come back to host memory, byte-identical to ptk_kd_build / ptk_kd_build_bounded.  Two-call
 * pattern as above (each call builds).  num_triangles must be at least 1. */
int ptk_kd_build_dev(ptk_ctx* ctx, const ptk_triangle* triangles, uint64_t num_triangles,
                     const float* bounds /* 6 floats or NULL */, uint32_t root_depth,
                     ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap,
                     uint32_t* prims, uint64_t prims_cap);

/* ---- closest hit for caller-supplied rays (parity entry) -----------------
 * scene::cast_ray (scene.cpp:355-399) for n rays.  SoA device arrays:
 * origins/directions are 3 arrays of n floats each laid out [x0..xn-1|y..|z..].
 * Outputs (any may be NULL): node = index in scene.nodes or -1; prim = index in
 * the mesh's triangle list or -1 (sphere / miss); t = length(pos_world-origin)
 * (scene.cpp:383 before squaring); (u,v) = (bar.y, bar.z) of
 * triangle::baricentric (scene.cpp:78-89) for triangles, raycast_result::uv
 * for spheres (scene.cpp:434-437); position/normal/uv = raycast_result in world
 * space (3n,3n,2n floats SoA). */
int ptk_intersect_batch_dev(ptk_ctx* ctx, uint64_t n,
                            const float* origins_dev, const float* directions_dev,
                            int32_t* node_dev, int32_t* prim_dev,
                            float* t_dev, float* u_dev, float* v_dev,
                            float* position_dev, float* normal_dev, float* uv_dev);
/* Same with host pointers (copies in and out, synchronises). */
int ptk_intersect_batch(ptk_ctx* ctx, uint64_t n,
                        const float* origins, const float* directions,
                        int32_t* node, int32_t* prim, float* t, float* u, float* v,
                        float* position, float* normal, float* uv);

/* ---- render loop ---------------------------------------------------------
 * ptk_render_begin: the per-run set-up of the driver lambda,
 * abstract_pathtracer.cpp:18-25 (camera basis, h2 = atan(fov_y/2), w2).
 * max_paths_per_launch bounds the ray-queue allocation (0 = default 1<<24). */
int ptk_render_begin(ptk_ctx* ctx, uint32_t width, uint32_t height, float fov_y,
                     uint64_t max_paths_per_launch);

/* The three hot loops of abstract_pathtracer.cpp:63-86 + pathtracer::trace
 * (pathtracer.cpp:14-69): for every pixel in [pixel_begin, pixel_begin+pixel_count)
 * (row-major index y*width+x, row 0 = top) trace the samples with global sample
 * indices [sample_begin, sample_begin+sample_count) and write (accumulate=0) or
 * add (accumulate=1) the SUM of their radiance to sum_rgb_dev[pixel*3+c]
 * (width*height*3 floats).  Random numbers are a pure function of
 * (seed, pixel, global sample index, dimension), so any split of the sample or
 * pixel range over calls / GPUs produces the same set of samples. */
int ptk_render_samples_dev(ptk_ctx* ctx, uint64_t seed,
                           uint64_t pixel_begin, uint64_t pixel_count,
                           uint64_t sample_begin, uint32_t sample_count,
                           float* sum_rgb_dev, int accumulate);

/* ptk_render_samples_dev for callers that cannot allocate device memory: code under HIP graph
 * capture, tight display loops (Sandbox re-renders on every camera move, rtsb.cpp:295-299,355),
 * harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same kernels
 * and the same sums, but the ray queues, per-sample radiance and parked-ray records live in a static
 * device array of the library (about 41 MiB per GPU, held until the process ends) instead of the
 * buffers ptk_render_begin / ptk_render_samples_dev allocate; the call itself makes no hipMalloc /
 * hipFree / event creation (ptk_ctx_enable_timing records nothing for it).
 * Limit: pixel_count * sample_count <= PTK_NOALLOC_MAX_PATHS, else PTK_ERR_INVALID (split the pixel
 * or sample range over several calls; with accumulate = 1 the sums of sample sub-ranges add up).
 * ptk_scene_upload and ptk_render_begin (which do allocate) must have been called before, outside
 * the no-allocation region; pass ptk_render_begin a small max_paths_per_launch if only this entry
 * point is used.  All contexts on one GPU share the array: calls that overlap on one GPU race, so
 * serialise them (one stream, or events between streams). */
#define PTK_NOALLOC_MAX_PATHS (1u << 18)
int ptk_render_samples_noalloc_dev(ptk_ctx* ctx, uint64_t seed,
                                   uint64_t pixel_begin, uint64_t pixel_count,
                                   uint64_t sample_begin, uint32_t sample_count,
                                   float* sum_rgb_dev, int accumulate);

/* Debug AOVs: rt::utility::debug_pathtracer::trace
 * (PathtracingUtility/debug_pathtracer.cpp:8-32, modes of debug_pathtracer.h:16-23) in place of
 * pathtracer::trace for the same pixel / sample ranges as ptk_render_samples_dev: the jittered primary
 * ray is cast, a hit yields the node's albedo / emission / roughness / metallic sampler at the hit's
 * uv (all three channels) or normal * 0.5 + 0.5, a miss the background.  Sums go to sum_rgb_dev like
 * ptk_render_samples_dev's; ptk_resolve_iteration_dev applies the running mean. */
enum { PTK_AOV_ALBEDO = 0, PTK_AOV_EMISSION = 1, PTK_AOV_ROUGHNESS = 2, PTK_AOV_METALLIC = 3, PTK_AOV_NORMAL = 4 };
int ptk_render_aov_dev(ptk_ctx* ctx, uint32_t aov, uint64_t seed,
                       uint64_t pixel_begin, uint64_t pixel_count,
                       uint64_t sample_begin, uint32_t sample_count,
                       float* sum_rgb_dev, int accumulate);

/* abstract_pathtracer.cpp:66,88-90: image = mix(sum/spi, image, it/float(it+1)). */
int ptk_resolve_iteration_dev(ptk_ctx* ctx, const float* sum_rgb_dev, float* image_rgb_dev,
                              uint64_t samples_per_iteration, uint64_t iteration);

/* The same over the pixels [pixel_begin, pixel_begin + pixel_count) only: what the driver does with an
 * interrupted iteration, whose unfinished scan lines keep the previous mean (abstract_pathtracer.cpp:63-72,109). */
int ptk_resolve_range_dev(ptk_ctx* ctx, const float* sum_rgb_dev, float* image_rgb_dev,
                          uint64_t pixel_begin, uint64_t pixel_count,
                          uint64_t samples_per_iteration, uint64_t iteration);

/* Convenience for single-GPU hosts: runs `iterations` full iterations
 * (abstract_pathtracer.cpp:38-112, without events) into a host image.  Returns PTK_INTERRUPTED (image valid,
 * last iteration partial) after ptk_request_interrupt. */
int ptk_render_image(ptk_ctx* ctx, uint64_t seed, uint64_t iterations,
                     uint32_t samples_per_iteration, float* image_rgb);

/* CLI tone map + gamma + 8-bit pack, CLI/main.cpp:85-104.
 * rgba8_dev: width*height uint32, r | g<<8 | b<<16 | 255<<24. */
int ptk_tonemap_dev(ptk_ctx* ctx, const float* image_rgb_dev, uint32_t* rgba8_dev,
                    uint64_t num_pixels);

/* ---- probes of single device functions (parity entries, like ptk_intersect_batch) ----------
 * Each runs one function of the path on caller-supplied inputs so that it can be held against
 * the reference function it restates directly, edge cases included.  All arrays are device
 * pointers, SoA like ptk_intersect_batch_dev ([x0..xn-1 | y.. | z..]).
 *
 * image::sample (sampler.cpp:36-61) of texture `texture` of the uploaded scene in `mode`
 * (PTK_MODE_*) at uv (2n floats) -> rgb (3n floats). */
int ptk_probe_image_sample_dev(ptk_ctx* ctx, uint32_t texture, uint32_t mode, uint64_t n,
                               const float* uv_dev, float* rgb_dev);
/* scene.background->sample(direction) (pathtracer.cpp:67): color_sampler (sampler.h:58) or
 * equirectangular_map::sample (sampler.cpp:116-124); directions 3n floats -> rgb 3n floats. */
int ptk_probe_background_dev(ptk_ctx* ctx, uint64_t n, const float* directions_dev, float* rgb_dev);
/* rng::hemisphere (rng.cpp:12-36) around normals (3n floats) with the two uniform draws of
 * rng.cpp:27,29 supplied in u12 (2n floats: z, then the phi draw) -> directions (3n floats). */
int ptk_probe_hemisphere_dev(ptk_ctx* ctx, uint64_t n, const float* normals_dev, const float* u12_dev,
                             float* directions_dev);
/* One level of pathtracer::trace_recursive (pathtracer.cpp:29-68) for n caller rays, with the
 * hemisphere draws supplied in u12: scene::cast_ray, then for a hit the material lookups, the
 * bounce direction and the next ray.  Outputs (any may be NULL): node = index of the hit node or
 * -1; emission = material emission at the hit, or the background's colour for a miss (3n);
 * weight = mix(albedo, 1, metallic) * cos_theta * 2, the factor of pathtracer.cpp:60 on the
 * next ray's radiance (3n; 0 for a miss); next_origin / next_direction = reflected_ray of
 * pathtracer.cpp:47-51 (3n each; the input ray for a miss). */
int ptk_probe_shade_dev(ptk_ctx* ctx, uint64_t n, const float* origins_dev, const float* directions_dev,
                        const float* u12_dev, int32_t* node_dev, float* emission_dev, float* weight_dev,
                        float* next_origin_dev, float* next_direction_dev);

/* Device ray / path counters since the last reset (synchronises). */
int ptk_get_counters(ptk_ctx* ctx, ptk_counters* out);
int ptk_reset_counters(ptk_ctx* ctx);
/* The same without waiting: the counters are copied to host_out (pinned host memory, for a copy that really is
 * asynchronous) behind the work already enqueued on the context's stream; host_out is valid once that stream, or
 * an event recorded on it after this call, has completed. */
int ptk_counters_download_async(ptk_ctx* ctx, ptk_counters* host_out);

/* Tuning switches (results are identical for every value; tests run both).
 * PTK_OPT_TRAVERSAL: how scene::cast_ray's node loop (scene.cpp:361-395) is mapped to a
 * wave: 0 = every lane walks every mesh node's triangles together; 1 = meshes are root-box
 * tested in the uniform loop and each lane then visits only the nodes whose box it pierced;
 * 2 (default) = 0 for the coherent primary rays of bounce 0 (1 if PTK_OPT_NODE_CULL applies), 1 for
 * later bounces and for ptk_intersect_batch.  No reference counterpart. */
enum { PTK_OPT_TRAVERSAL = 1, PTK_OPT_KD_PRUNE = 2, PTK_OPT_NODE_CULL = 3, PTK_OPT_BALANCE = 4, PTK_OPT_SPLIT = 5, PTK_OPT_DEVICE_KD = 6, PTK_OPT_WALK_STACK = 7, PTK_OPT_COMPACT = 8, PTK_OPT_FUSE = 9, PTK_OPT_OVERLAP = 10 };
/* PTK_OPT_KD_PRUNE: how a multi-node KD-tree is walked (mesh::intersect_internal,
 * scene.cpp:188-212).  0 = the reference's order, every box-hit node is visited;
 * 1 (default) = near child first (by the sign of the ray direction on the split axis),
 * subtrees whose triangle box lies beyond the best hit so far are skipped, exact ties are
 * resolved towards the reference's visiting order.  Same ids, same values.
 * PTK_OPT_NODE_CULL: with per-lane lists (PTK_OPT_TRAVERSAL 1/2), whether the lists are built by a
 * conservative world-space box test per node (1) or by the reference's own local-space root test
 * run by the whole wave (0); 2 (default) = 1 for scenes with at least 16 nodes.  The test only
 * discards nodes the reference cannot report a hit on, so results do not change.
 * PTK_OPT_BALANCE: 1 = the (ray, node) pairs that survive the conservative cull, and the triangles
 * of the one-leaf meshes among them, are pooled per wave in LDS and dealt evenly to its 64 lanes
 * (overrides PTK_OPT_TRAVERSAL / PTK_OPT_NODE_CULL); 0 = off; 2 (default) = on for scenes with at
 * least two nodes, except that bounces 1..4 of a scene with 16 or more nodes and no KD-tree use the per-lane
 * candidate lists (PTK_OPT_TRAVERSAL / PTK_OPT_NODE_CULL at their defaults).  Winners are chosen by the reference's own keys, so results do not change.
 * PTK_OPT_SPLIT: scenes with KD-trees only (and PTK_OPT_KD_PRUNE, PTK_OPT_BALANCE on, at most 64 tree
 * nodes).  1 / 2 (default) = a bounce runs as two kernels: one casts every ray against the spheres and
 * one-leaf meshes and parks the rays that may hit a tree; a persistent walker kernel then walks the
 * trees of the parked rays, refilling lanes whose walk has ended; a third kernel shades the parked rays.
 * 0 = the KD walk stays inside the bounce kernel.  Same results either way.
 * PTK_OPT_DEVICE_KD: where ptk_scene_upload builds a mesh's KD-tree.  0 = host (ptk_kd_build's code),
 * 1 = GPU (ptk_kd_build_dev's code), 2 (default) = GPU for meshes of at least 32768 triangles; with 2 a
 * GPU build that runs out of device memory is redone by the host builder (1 reports the error).  The two
 * builders give byte-identical trees; ray casting and shading never run on the host.
 * PTK_OPT_WALK_STACK: how many pending far children the pruned KD walk may hold per ray before it sweeps the
 * current subtree without a stack (0..64; the kernels clamp it to their own short stack: 12 entries in LDS in the
 * walker kernel, 16 in the per-lane loops).  Default 64 = the kernels' capacity.  Small values only force the
 * stackless sweep (tests); results are identical for every value.
 * PTK_OPT_COMPACT: where the rays that survive a bounce are packed (scenes without KD-trees; split mode has its own
 * queues).  0 = inside their 256-slot segment, in place; 1 = across segments, into a second queue (one more
 * max_paths * 52 B of device memory, allocated on first use): after a bounce that ends most paths the next one then
 * runs on full wavefronts; 2 (default) = decided per context from the rays per path of the chunks rendered so far
 * (a snapshot of the counters behind each call): below 4 rays per path (an open scene) it packs across segments,
 * otherwise in place.  The image does not depend on it (radiance is kept per path).
 * ptk_render_samples_noalloc_dev always packs in place.
 * PTK_OPT_FUSE: scenes without KD-trees, pooled work lists.  1 = the five bounces of a path run in ONE kernel that keeps the
 * ray in registers from bounce to bounce: no ray queue traffic, no compaction, one launch per chain instead of five; a path
 * that ends early idles its lane, so this only pays where nearly all paths live five bounces (a closed scene).  0 = a kernel
 * per bounce; 2 (default) = 1 once the counters of the chunks rendered so far show at least 4.75 rays per path, else 0.
 * Same image, same counters.
 * PTK_OPT_OVERLAP: scenes with KD-trees (PTK_OPT_SPLIT).  1 (default) = a ptk_render_samples_dev call of at least 2^20 paths
 * is issued as two launch chains over the two halves of its pixel range, the second on a stream the context owns, forked
 * from and joined to the context's stream with events: one chain's kernel tails (a persistent walker launch waits for its
 * slowest wave) are filled with the other's work.  Stream order as seen by the caller does not change; the sums are the
 * same bit for bit.  0 = one chain.  Calls whose launches are timed (ptk_ctx_enable_timing) always run as one chain. */
int ptk_ctx_set_option(ptk_ctx* ctx, int option, int value);

/* Per-kernel launch timing with HIP events on the context's stream (bench.py's
 * roofline leg).  No reference counterpart (the reference only has wall-clock
 * get_elapsed_time, abstract_pathtracer.cpp:133-137).  After enable(1) every
 * kernel launch is bracketed by two events (at most 8192 launches are kept);
 * enable(n), n > 1, brackets the launches of every n-th ptk_render_samples_dev call only
 * (the events themselves cost GPU time: 1 % on Room's long kernels, 5 % on a chain of
 * short ones; a sample of the calls gives the same averages);
 * ptk_get_timing synchronises, fills out[PTK_KERNEL_KINDS] and clears the log. */
enum {
    PTK_KERNEL_BOUNCE_FIRST = 0,    /* k_bounce<true>:  generate + cast + shade, bounce 0 */
    PTK_KERNEL_BOUNCE_NEXT = 1,     /* k_bounce<false>: cast + shade, bounces 1..4        */
    PTK_KERNEL_ACCUMULATE = 2,      /* k_accumulate                                       */
    PTK_KERNEL_INTERSECT_BATCH = 3, /* k_intersect_batch (+ its walker in split mode)     */
    PTK_KERNEL_WALK = 4,            /* k_walk: KD walks of the parked rays (split mode)   */
    PTK_KERNEL_SHADE_PARKED = 5,    /* k_shade_parked: their shading                      */
    PTK_KERNEL_BOUNCE_FUSED = 6,    /* k_bounce_fused: generate + five casts + shading    */
    PTK_KERNEL_KINDS = 7
};
typedef struct ptk_kernel_timing {
    uint64_t launches;
    double total_ms;
} ptk_kernel_timing;
int ptk_ctx_enable_timing(ptk_ctx* ctx, int enable);
int ptk_get_timing(ptk_ctx* ctx, ptk_kernel_timing* out /* [PTK_KERNEL_KINDS] */);

/* ---- device buffers for hosts without a HIP runtime of their own --------------
 * The C++ rt:: mirror (pathtracing_amd/host) is plain g++ code: it gets the sum / image
 * buffers that ptk_render_samples_dev and ptk_resolve_iteration_dev work on from here.
 * No reference counterpart (the reference keeps its image in host memory,
 * abstract_pathtracer.cpp:18).  memset is enqueued on the context's stream; download
 * and upload synchronise it. */
int ptk_dev_alloc(ptk_ctx* ctx, uint64_t bytes, void** out_dev);
int ptk_dev_free(ptk_ctx* ctx, void* dev);
int ptk_dev_memset(ptk_ctx* ctx, void* dev, int value, uint64_t bytes);
int ptk_dev_download(ptk_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes);
int ptk_dev_upload(ptk_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes);

/* Flattened KD-tree the context holds for mesh `mesh` (tests, diagnostics). */
int ptk_scene_kd_info(ptk_ctx* ctx, uint32_t mesh, ptk_kd_info* info);

/* ---- several devices behind one host driver --------------------------
 * The reference parallelises one run over host threads (abstract_pathtracer.cpp:97-103: num_threads
 * workers share the scan lines of an iteration); a group does the same over GPUs.  Every member is a
 * context on device_ids[m] (an id may repeat: two members on one GPU) holding the whole scene.  An
 * iteration's samples are split over the members by sample range (by pixel range when there are fewer
 * samples than members); sample indices are global, so the set of samples does not depend on the group
 * size.  Each member accumulates a float SUM image; members on the same device are added on that device,
 * the per-device sums are combined with one ncclReduce(sum) to member 0 over RCCL / xGMI (RCCL is loaded
 * on first need; a group on one device never touches it), and member 0 applies
 * image = mix(sum / spi, image, it / float(it + 1)) (abstract_pathtracer.cpp:88-90).
 * An N-member image differs from the 1-member image only in the order of the per-pixel fp32 sum.
 * Calls on a group are serialised by the caller, like calls on a context. */
typedef struct ptk_group ptk_group;
int ptk_group_create(const int* device_ids, int n, ptk_group** out);
void ptk_group_destroy(ptk_group* group);
/* Last error text of `group` (or of the last failed ptk_group_create if NULL). */
const char* ptk_group_last_error(const ptk_group* group);
int ptk_group_size(const ptk_group* group);
/* Member context (options, per-member calls); NULL if out of range.  Owned by the group. */
ptk_ctx* ptk_group_ctx(ptk_group* group, int member);
/* ptk_scene_upload / ptk_scene_set_camera / ptk_render_begin on every member; render_begin also
 * (re)allocates the per-member sum images and clears member 0's running-mean image. */
int ptk_group_scene_upload(ptk_group* group, const ptk_scene_desc* scene);
int ptk_group_scene_set_camera(ptk_group* group, const ptk_camera_desc* camera);
int ptk_group_render_begin(ptk_group* group, uint32_t width, uint32_t height, float fov_y,
                           uint64_t max_paths_per_launch);
/* One iteration of the driver loop (abstract_pathtracer.cpp:38-112): global samples
 * [iteration * spi, (iteration + 1) * spi) of every pixel, split over the members, reduced to
 * member 0 and folded into its running-mean image.  Synchronises.  Inside, the iteration goes to the GPUs in
 * slices (below) with two in flight; ptk_group_request_interrupt from another thread stops it before the next
 * slice: the pixels of the finished slices are folded into the image, the others keep the previous mean (the
 * reference's unfinished scan lines, abstract_pathtracer.cpp:63-72,109) and the call returns PTK_INTERRUPTED.
 * ptk_group_progress = finished slices / slices of the running iteration (pathtracer_result::progress,
 * abstract_pathtracer.cpp:51).  ptk_group_render_begin clears the interrupt request. */
int ptk_group_render_iteration(ptk_group* group, uint64_t seed, uint64_t iteration,
                               uint64_t samples_per_iteration);
/* The same with ptk_render_aov_dev in place of ptk_render_samples_dev (aov = PTK_AOV_*). */
int ptk_group_render_aov_iteration(ptk_group* group, uint32_t aov, uint64_t seed, uint64_t iteration,
                                   uint64_t samples_per_iteration);
int ptk_group_request_interrupt(ptk_group* group);
float ptk_group_progress(const ptk_group* group);

/* The iteration in asynchronous steps, for a driver that keeps the GPUs busy while it handles the previous
 * iteration (on_iteration_end handlers, abstract_pathtracer.cpp:109) and decides itself when to stop:
 *   ptk_group_iteration_slices   how many slices an iteration of spi samples has: pixel ranges sized so that every
 *                                member's part of a slice is one launch chain (<= max_paths_per_launch paths);
 *   ptk_group_submit_slice       enqueues one slice on every member (aov < 0: the integrator, else PTK_AOV_*) and
 *                                returns a ticket; never waits for the GPU;
 *   ptk_group_wait_slice         blocks until every member has finished the slice of that ticket (one of the last
 *                                8 submitted);
 *   ptk_group_finish_iteration   enqueues the exchange (add on shared devices, ncclReduce to member 0) behind the
 *                                members' slices, the running mean over the pixels of the first slices_done
 *                                slices (slices_done < num_slices: an interrupted iteration) and, if download is set,
 *                                the copy of the image to pinned host memory; never waits.  The sum images are
 *                                double-buffered by iteration parity: slices of iteration i + 1 may be submitted
 *                                before or after this call for iteration i, slices of iteration i + 2 only after it;
 *   ptk_group_wait_iteration     blocks until that iteration has been resolved; copies the image (if it was
 *                                downloaded and image_rgb is not NULL) and the members' summed counters out. */
int ptk_group_iteration_slices(ptk_group* group, uint64_t samples_per_iteration, uint32_t* num_slices);
int ptk_group_submit_slice(ptk_group* group, int aov, uint64_t seed, uint64_t iteration, uint64_t samples_per_iteration,
                           uint32_t slice, uint32_t num_slices, uint64_t* ticket);
int ptk_group_wait_slice(ptk_group* group, uint64_t ticket);
int ptk_group_finish_iteration(ptk_group* group, uint64_t iteration, uint64_t samples_per_iteration,
                               uint32_t slices_done, uint32_t num_slices, int download);
int ptk_group_wait_iteration(ptk_group* group, uint64_t iteration, float* image_rgb, ptk_counters* counters);
/* Member 0's running-mean image -> host (width*height*3 floats, row 0 = top).  Waits for the iterations finished so far. */
int ptk_group_read_image(ptk_group* group, float* image_rgb);
/* Sum of the members' counters. */
int ptk_group_get_counters(ptk_group* group, ptk_counters* out);

#ifdef __cplusplus
}
#endif
#endif /* PTK_H */
