/* rtx.h -- C ABI of the MI355X (gfx950) ray-trace hot path for the holoskii/Rendering scene API.
 *
 * The reference has no plugin / FFI layer: its seam is the public Scene API in
 * include/scene.h:68-100 of the reference --
 *     void Scene::launchWorkers(Vec3f* frameBuffer)            scene.cpp:470-506   (pass 1)
 *     void Scene::renderWorker(Vec3f*, const tileInfo&)        scene.cpp:444-468
 *     void Scene::launchSSAA(Vec3f* frameBuffer)               scene.cpp:542-593   (Sobel + pass 2)
 *     void Scene::SSAAworker(Vec3f*, bool*, const tileInfo&)   scene.cpp:508-540
 *     static Vec3f Render::castRay(const Ray&, const Scene&, int)   scene.cpp:758-946
 *     static bool  Render::trace(const Ray&, const ObjectVector&, IntersectInfo&)  scene.cpp:724-756
 * A maintainer drops this library in by flattening the already-loaded Scene into an rtx_scene_desc
 * (INTEGRATION.md shows the ~60-line binding) and replacing the bodies of launchWorkers / launchSSAA
 * with rtx_render_pass1 / rtx_sobel + rtx_render_ssaa.  rendering_amd/host/ is this repo's own C++17
 * host side (same Scene/Options/Object API, .scene/OBJ/BMP loaders, BVH builder) built that way.
 *
 * Conventions: plain C, opaque handle, every call returns 0 on success or a negative rtx_status; no
 * exceptions cross the boundary; one host thread drives one rtx_scene.  Pointers named *_dev are device
 * (HBM) pointers owned by the caller; everything in rtx_scene_desc is host memory, copied to the device
 * once by rtx_scene_create and never read again.  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Framebuffer layout is the reference's `Vec3f[H*W]`: fp32 RGB, row 0 = top,
 * index x + y*W (scene.cpp:463, 599).
 */
#ifndef RTX_H
#define RTX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	RTX_OK = 0,
	RTX_ERR_ARG = -1,        /* bad argument / inconsistent description */
	RTX_ERR_DEVICE = -2,     /* HIP runtime error (message in rtx_last_error) */
	RTX_ERR_NO_DEVICE = -3,  /* no gfx950 device visible */
	RTX_ERR_UNSUPPORTED = -4 /* description uses something outside the hot path */
} rtx_status;

/* enum values mirror the reference's (objects.h:18-19, lights.h:12) */
enum { RTX_OBJ_SPHERE = 1, RTX_OBJ_PLANE = 2, RTX_OBJ_MESH = 3 };
enum { RTX_MAT_DIFFUSE = 0, RTX_MAT_REFLECTIVE = 1, RTX_MAT_TRANSPARENT = 2, RTX_MAT_PHONG = 3 };
enum { RTX_LIGHT_DISTANT = 1, RTX_LIGHT_POINT = 2, RTX_LIGHT_AREA = 3 };

#define RTX_FLAG_BACKFACE_CULL 1u /* options::useBackfaceCulling (options.h:27) */
#define RTX_FLAG_SKYBOX 2u        /* options::useSkybox (options.h:32)          */

/* Options + Camera, everything the workers read (options.h:9-20; scene.h:52-66; scene.cpp:447-457). */
typedef struct rtx_view {
	uint32_t width, height;
	float bias;            /* Options::bias */
	int32_t max_ray_depth; /* Options::maxRayDepth */
	float background[3];   /* Options::backgroundColor */
	uint32_t flags;        /* RTX_FLAG_* */
	float cam_pos[3];      /* Camera::pos */
	float cam_matrix[16];  /* Camera::rMatrix, row-major x[i][j] (scene.cpp:48) -- built on the host */
	float scale;           /* tanf(fov*0.5f/180.0f*(float)M_PI) (scene.cpp:447) -- host libm */
	float aspect;          /* width / (float)height (scene.cpp:448) */
} rtx_view;

/* One Object (objects.h:24-47, 166-191). */
typedef struct rtx_object {
	int32_t type;     /* RTX_OBJ_* */
	int32_t material; /* RTX_MAT_* */
	float pos[3];
	float color[3];
	float ior, ambient, diffuse, specular, n_specular;
	float radius2;   /* Sphere::r2 */
	float normal[3]; /* Plane::normal exactly as stored (not re-normalised, scene.cpp:300) */
	int32_t mesh;    /* index into rtx_scene_desc::meshes, or -1 */
} rtx_object;

/* One Mesh + its AccelerationStructure (objects.h:69-164), flattened:
 * nodes in pre-order (left child first = the reference's visiting order, objects.cpp:601-616);
 * leaf references in that same order (objects.cpp:622-629), so "first hit wins ties" is index order. */
typedef struct rtx_mesh {
	uint32_t n_nodes, n_refs, n_tris;
	const float* node_bounds;   /* n_nodes x 6: bounds[0].xyz, bounds[1].xyz */
	const int32_t* node_skip;   /* n_nodes: pre-order index of the first node after this subtree */
	const int32_t* leaf_begin;  /* n_nodes: first leaf reference, -1 for inner nodes */
	const int32_t* leaf_count;  /* n_nodes: number of leaf references, -1 for inner nodes */
	const uint32_t* refs;       /* n_refs: triangle index of each leaf reference */
	const float* tri_pos;       /* n_tris x 9: a, b, c */
	const float* tri_nrm;       /* n_tris x 9: n_a, n_b, n_c */
	const float* tri_uv;        /* n_tris x 6: t_a, t_b, t_c */
	const float* tri_tb;        /* n_tris x 6: tangent, bitangent (may be NULL when no normal map) */
	uint32_t diffuse_w, diffuse_h;
	const float* diffuse_map;   /* w*h x 3 fp32 (byte/256, objects.cpp:405-412) or NULL */
	uint32_t normal_w, normal_h;
	const float* normal_map;    /* w*h x 3, as loaded (objects.cpp:426-434) or NULL */
	uint32_t specular_w, specular_h;
	const float* specular_map;  /* w*h fp32 (objects.cpp:448-455) or NULL */
} rtx_mesh;

/* One Light (lights.h:21-73). */
typedef struct rtx_light {
	int32_t type; /* RTX_LIGHT_* */
	float color[3];
	float intensity;
	float dir[3];       /* DistantLight::dir as stored (not re-normalised, scene.cpp:222) */
	float pos[3];       /* PointLight::pos / AreaLight::pos */
	uint32_t n_points;  /* AreaLight: sample points built on the host (lights.cpp:46-63) */
	const float* points; /* n_points x 3 */
} rtx_light;

typedef struct rtx_scene_desc {
	rtx_view view;
	uint32_t n_objects;
	const rtx_object* objects; /* scene-file order = Render::trace's iteration order (scene.cpp:731) */
	uint32_t n_meshes;
	const rtx_mesh* meshes;
	uint32_t n_lights;
	const rtx_light* lights;
	uint32_t sky_w, sky_h;
	const float* sky[6];       /* Scene::skyboxes[0..5] (left, front, right, back, top, bottom), w*h x 3 */
} rtx_scene_desc;

/* 64-bit statistics under reference traversal semantics (stats.h:11-16 without the int overflow). */
typedef struct rtx_counters {
	uint64_t rays;      /* Render::trace invocations (stats::raysCasted) */
	uint64_t box_tests; /* AccelerationStructure::intersectBox calls (stats::accelStructTests) */
	uint64_t tri_tests; /* Triangle::rayTriangleIntersect calls (stats::rayTriTests) */
	uint64_t moot_rays; /* of `rays`: shadow rays whose answer cannot influence the pixel (Diffuse material, N . -L <= 0:
	                       vis * max(0, N . -L) is +0 either way, scene.cpp:788); only the instrumented variant traces them */
} rtx_counters;

typedef struct rtx_scene rtx_scene;

const char* rtx_last_error(void);
int rtx_device_count(int* count);

/* Flatten-and-upload, once per scene (replaces nothing in the reference: its Scene is read in place). */
int rtx_scene_create(const rtx_scene_desc* desc, int device, rtx_scene** out);
void rtx_scene_destroy(rtx_scene* scene);
/* Bytes of scene data rtx_scene_create placed in HBM (nodes, leaf references, shading arrays, texture maps, skybox). */
int rtx_scene_bytes(rtx_scene* scene, size_t* bytes);
/* Change resolution / camera / flags without re-uploading geometry. */
int rtx_scene_set_view(rtx_scene* scene, const rtx_view* view);

/* Pass 1 = Scene::launchWorkers (scene.cpp:470-506) restricted to image rows [row_begin,row_end):
 * writes fb_dev[x + y*W] for x in [0,W-1), y in [row_begin, min(row_end,H-1)) -- the last row and column
 * are never written, as in the reference (scene.cpp:369-372).  Asynchronous on `stream`. */
int rtx_render_pass1(rtx_scene* scene, uint32_t row_begin, uint32_t row_end, float* fb_dev, void* stream);

/* The whole frame in one launch: the same pixels and the same mask as
 *   rtx_render_pass1(rows) ; rtx_sobel(rows) ; rtx_render_ssaa(rows)
 * (scene.cpp:444-568: renderWorker, the Sobel pass and SSAAworker of Scene::render), with the three stages
 * overlapped on the device through per-tile dependencies instead of separated by launch boundaries.
 * mask_dev rows [row_begin, row_end) are written completely (0 for rows owned by another part).
 * Asynchronous on `stream`.  The single launch can give up (its SSAA item queues overflow, or its watchdog fires): the
 * frame is then incomplete, which Scene::render (scene.cpp:595-606) can never deliver.  rtx_frame_status is the
 * host's synchronisation point for a frame: it waits for the device, and if the last single launch gave up it renders
 * that frame AGAIN through the three launches into the same buffers and keeps the view on three launches; status = 0
 * (frame complete as rendered) or error | 0x100 (frame complete after the re-render; error 1: queue entry never
 * written, 2: work never completed, 3: item queue overflow).  Callers that consume a frame call it first.
 * Whether the single launch or the three launches are faster depends on the view (slowest tile against total work);
 * rtx_render_frame measures both on the first warm frames of a view and keeps the faster (environment
 * RTX_FRAME_MODE=fused|split forces one).  rtx_frame_mode: what the last call used (0 three launches, 1 one launch)
 * and the measured durations of the current view (ms, -1 = not measured yet).
 * which = 3 in rtx_last_kernel_ms / rtx_kernel_time_stats: the frame, either way; 4: the single launch's kernel alone. */
int rtx_render_frame(rtx_scene* scene, uint32_t row_begin, uint32_t row_end, float* fb_dev, uint8_t* mask_dev, void* stream);
int rtx_frame_status(rtx_scene* scene, uint32_t* status);

/* Sobel edge mask of Scene::launchSSAA (scene.cpp:547-568) for rows [row_begin,row_end); reads the
 * 3x3 neighbourhood from fb_dev; border entries (row 0, H-1, column 0, W-1) are written as 0. */
int rtx_sobel(rtx_scene* scene, const float* fb_dev, uint32_t row_begin, uint32_t row_end,
              uint8_t* mask_dev, void* stream);

/* Pass 2 = the SSAAworker loop (scene.cpp:508-540) for rows [row_begin,row_end): every pixel with
 * mask_dev[y*W+x] != 0 is replaced by the mean of 4 castRay samples. */
int rtx_render_ssaa(rtx_scene* scene, const uint8_t* mask_dev, uint32_t row_begin, uint32_t row_end,
                    float* fb_dev, void* stream);

/* saveImage's quantiser (util.cpp:46-58): bottom-up rows, BGR, (uint8)(clamp(0,1,v)*255).
 * bgr_dev: H*W*3 bytes (W % 4 == 0), 4-byte aligned; fb_dev 16-byte aligned.  Under rtx_set_row_ownership only the rows this
 * device owns are converted; the others are not touched (rtx_gather fills them on the root). */
int rtx_quantize_bgr8(rtx_scene* scene, const float* fb_dev, uint8_t* bgr_dev, void* stream);

/* Convenience for hosts that hold a plain `Vec3f*`: pass 1 (+ optional Sobel/SSAA) into a host buffer;
 * allocates a device framebuffer internally and copies back (PCIe-inclusive). */
int rtx_render_frame_host(rtx_scene* scene, int with_ssaa, float* fb_host);

/* Statistics: when enabled the kernels count under reference semantics (no any-hit early-out). */
int rtx_counters_enable(rtx_scene* scene, int enable);
int rtx_counters_reset(rtx_scene* scene);
int rtx_counters_read(rtx_scene* scene, rtx_counters* out); /* synchronises the device */

/* Kernel timing from HIP events recorded on the launch stream around every launch.
 * which: 0 = pass 1, 1 = sobel, 2 = ssaa (mask compaction + 4-ray kernel), 3 / 4: see rtx_render_frame.
 * rtx_last_kernel_ms: the most recent launch (synchronises on its stop event).
 * rtx_kernel_time_stats: number of launches and their summed duration since rtx_kernel_time_reset
 * (synchronises on the recorded events; call it after the timed region). */
int rtx_last_kernel_ms(rtx_scene* scene, int which, float* ms);

/* Acceleration-structure build on the device.  Replaces Mesh::loadModel's
 * `ac = make_unique<AccelerationStructure>(...); ac->setup(...)` (objects.cpp:385-392) and the recursive builder
 * behind it (objects.cpp:470-526, 633-763): same topology, bounds and leaf-reference order, bit for bit.
 *   tri_pos   n_tris x 9 host floats (a, b, c of every Triangle, in `allTris` order)
 *   root_lo/hi  the root box Mesh::loadModel computes (objects.cpp:328-330)
 *   ac_penalty  options::acPenalty (leaf iff n <= depth * acPenalty, root depth 1)
 * The result stays in device memory; rtx_bvh_read copies it out in the rtx_mesh layout (any pointer may be NULL). */
typedef struct rtx_bvh rtx_bvh;
int rtx_bvh_build(const float* tri_pos, uint32_t n_tris, const float* root_lo, const float* root_hi, int32_t ac_penalty, int device,
                  rtx_bvh** out);
int rtx_bvh_info(const rtx_bvh* bvh, uint32_t* n_nodes, uint32_t* n_refs, uint32_t* max_depth, float* build_ms);
int rtx_bvh_read(const rtx_bvh* bvh, float* node_bounds, int32_t* node_skip, int32_t* leaf_begin, int32_t* leaf_count, uint32_t* refs);
void rtx_bvh_destroy(rtx_bvh* bvh);

/* Pixel sharding across the GPUs of a node: bands of band_height rows are dealt round-robin
 * to n_parts devices; this device renders / masks / re-renders only rows y with (y / band_height) % n_parts
 * == part.  With halo != 0 pass 1 additionally renders the one row above and below every owned band, so
 * rtx_sobel needs no exchange of pass-1 results.  band_height == 0 restores "all rows" (the default). */
int rtx_set_row_ownership(rtx_scene* scene, uint32_t band_height, uint32_t n_parts, uint32_t part, int halo);

/* Multi-GPU (SURVEY.md 8e, 8b `rtx_gather`): one process per GPU; the scene is replicated, the rows of a frame are dealt
 * out with rtx_set_row_ownership(band, n_ranks, rank, halo = 1), every rank renders / masks / re-renders its own rows, and
 * rtx_gather delivers all rows to ONE rank.  The reference's Scene::render (scene.cpp:595-606) runs in one address
 * space, so it has no counterpart of this step; a maintainer calls it between launchSSAA and saveImage.
 * Transport: RCCL (librccl.so.1, loaded on first use) -- every owned band is one ncclSend from its owner straight into
 * its final place in root's buffer (a band of rows is a contiguous slab), all inside one ncclGroup: xGMI links are
 * point-to-point, so the owners' transfers use different links in parallel.  Bootstrap: rank 0 obtains an id with
 * rtx_comm_unique_id and hands its RTX_COMM_ID_BYTES bytes to the other ranks by any means (pipe, file, MPI,
 * torch.distributed); every rank then calls rtx_comm_create.  One GPU per rank (RCCL refuses two ranks on one device). */
#define RTX_COMM_ID_BYTES 128
typedef struct rtx_comm rtx_comm;
int rtx_comm_unique_id(void* id128);
int rtx_comm_create(const void* id128, int n_ranks, int rank, int device, rtx_comm** out);
int rtx_comm_info(const rtx_comm* comm, int* n_ranks, int* rank);
void rtx_comm_destroy(rtx_comm* comm);
/* Every rank reports whether its part of the frame went well (ok != 0); *all_ok = 1 iff every rank said so (a 4-byte
 * ncclAllReduce).  Called before rtx_gather, so that a rank that failed does not leave the others waiting for its bands:
 * when *all_ok is 0 every rank skips the gather and reports the error.  Synchronises `stream`. */
int rtx_comm_agree(rtx_comm* comm, int ok, int* all_ok, void* stream);
/* img_dev: an image of `height` rows of row_bytes bytes each on this rank's device -- the fp32 framebuffer
 * (row_bytes = W*12, bottom_up = 0) or the BGR8 image of rtx_quantize_bgr8 (row_bytes = W*3, bottom_up = 1: image row y
 * is stored at row H-1-y, util.cpp:50).  On return (asynchronously, on `stream`) rank `root`'s buffer holds every
 * rank's owned rows; the other ranks' buffers are unchanged.  The scene's row ownership must be (band, n_ranks, rank). */
int rtx_gather(rtx_scene* scene, rtx_comm* comm, void* img_dev, size_t row_bytes, int bottom_up, int root, void* stream);
/* The transfer list rtx_gather executes (host-only, no GPU needed; used by the tests): band k = image rows
 * [k*band_height, min((k+1)*band_height, height)), owned by rank k % n_parts, stored at byte `offset[k]`, `bytes[k]` long.
 * Writes at most max_bands entries, returns the number of bands in *n_bands. */
int rtx_gather_plan(uint32_t height, uint32_t band_height, uint32_t n_parts, size_t row_bytes, int bottom_up, uint32_t max_bands,
                    uint32_t* owner, size_t* offset, size_t* bytes, uint32_t* n_bands);

/* Probe rays (host buffers, synchronous): for each of n rays {orig xyz, dir xyz} runs Render::trace
 * (scene.cpp:724-756) and Render::castRay at depth 0 (scene.cpp:758-946).
 * hits: n x 8 floats = [hit 0/1, object index, triangle index (-1 unless mesh), tNear, u, v, 0, 0];
 * colours: n x 3.  Used by the per-ray parity tests. */
int rtx_cast_rays(rtx_scene* scene, uint32_t n, const float* rays_host, float* hits_host, float* colours_host);

#ifdef __cplusplus
}
#endif
#endif
