/* Diagnostics, probes and tuning hooks of librtx_hip.so -- NOT part of the drop-in boundary (include/rtx.h is): nothing a
 * reference-side caller binds.  They exist for the parity tests (tests/), bench.py's reporting and the A/B
 * tools; every symbol here may change between rounds.  No entry point in this header changes a pixel. */
#ifndef RTX_DEBUG_H
#define RTX_DEBUG_H
#include "rtx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Which way rtx_render_frame rendered the last frame (0 three launches, 1 one) and what it measured; forcing either (tests, A/B runs). */
int rtx_frame_mode(rtx_scene* scene, int* mode, float* split_ms, float* fused_ms);
int rtx_set_frame_mode(rtx_scene* scene, int mode); /* -1 measure and choose (default), 0 always three launches, 1 always one */

/* Host only (no device is touched): what rtx_scene_create derives from a mesh before uploading it -- the tree with S = rtx_wide_node_slots()
 * descendants per node (log2 S binary levels per fetch; n_wide records of 32 S bytes: S slots of {lo.x hi.x lo.y hi.y lo.z hi.z, link,
 * first}; 0 when the boxes are not nested) and the prune blocks of its slots (n_wide records of 64 S bytes: S {c[3], P, h[3], Pgen} then S
 * {qc[3], wlo, qr[3], whi}; rtx_device.h, DESIGN_HISTORY.md 3.1c), plus the record of the whole mesh.  For the CPU tests of their
 * invariants (tests/test_host_cpu.py); cap_wide = records the output arrays hold. */
int rtx_mesh_flatten_probe(const rtx_mesh* mesh, uint32_t* n_wide, void* wide_out, void* prune_out, uint32_t cap_wide, float* root_rec8);
int rtx_wide_node_slots(void);      /* slots of a wide node in this build: 4 or 8 */

/* Host only: the P of the source copies of the prune records (rtx_device.h PruneRec, csrc/rtx_source.hip sourceP; DESIGN_HISTORY.md 3.1d)
 * for n triangles given as (v0, e1, e2) = 9 floats each, the source point S3, its radius sigma and cam != 0 when the rays start
 * at S (the camera) rather than pass through it (a point light).  The function the device kernels run, for the CPU tests of the
 * bound (tests/test_prune_bound_cpu.py). */
int rtx_source_p_probe(const float* tris9, uint32_t n, const double* S3, double sigma, int cam, float* out);

/* First-frame cost estimate (rtx_scene_create / rtx_scene_set_view project every leaf box of the meshes through the camera;
 * the reference renders one frame per process, main.cpp:15, so there is no previous frame to learn the tile costs from):
 * per cell of 2 x 2 tiles (16 x 16 pixels) the references and the leaves whose boxes cover it, interleaved (refs, leaves),
 * grid_w x grid_h cells.  out == NULL: only the dimensions.  Diagnostic (tools/cost_fit.py); no pixel depends on the estimate. */
int rtx_cost_grid_read(rtx_scene* scene, uint32_t* out, size_t n, uint32_t* grid_w, uint32_t* grid_h);

/* Experiment / test knobs of a live scene.  Their environment variables (RTX_STRIP_LIMIT, RTX_SSAA_HEAVY_TICKS,
 * RTX_SSAA_SPREAD_SLOTS, RTX_SPLIT_PERCENT, RTX_SSAA_LOCAL_BELOW, RTX_SSAA_SPARSE_BELOW, RTX_FRAME_QUEUE_CAP, RTX_DEBUG_ITEMS, ...) are read
 * once, by rtx_scene_create, and only when RTX_ALLOW_ENV_KNOBS=1 is set (the product ignores RTX_* variables otherwise); names here: strip_limit, ssaa_heavy_ticks, ssaa_spread_slots, split_percent,
 * ssaa_local_below, ssaa_sparse_below, frame_queue_cap, frame_rule_tiles, frame_rule_tiles_analytic, debug_items.  No knob changes a pixel. */
int rtx_set_knob(rtx_scene* scene, const char* name, double value);

/* Launch counts and summed durations of kernel `which` (rtx_last_kernel_ms in rtx.h) since rtx_kernel_time_reset; synchronises on the recorded events. */
int rtx_kernel_time_reset(rtx_scene* scene);
int rtx_kernel_time_stats(rtx_scene* scene, int which, uint32_t* launches, double* total_ms);

/* Per-tile cost of the most recent rtx_render_pass1 (profiling aid; also what orders the SSAA work list):
 * out[ty * ceil(width/8) + tx] = wall-clock ticks (100 MHz) one wave spent on the 8x8 pixel tile (tx, ty).
 * n must be ceil(width/8) * ceil(height/8) -- or twice that: the second half then holds, per tile, the slowest SSAA
 * work item of the most recent rtx_render_ssaa (ticks, scaled to a 16-pixel item).  Synchronises the device. */
int rtx_tile_cost_read(rtx_scene* scene, uint32_t* out, size_t n);

/* Self-check of the device math the parity contract depends on: evaluates powf / normalize / division /
 * sqrt on `n` inputs on the device so tests can compare with the host.  op: 0 powf(x,y), 1 1/x,
 * 2 sqrtf(x), 3 (float)(1/sqrt((double)x)), 4 x/y. */
int rtx_math_probe(int device, int op, uint32_t n, const float* x, const float* y, float* out);

/* The shading path's vector helpers on the device, n inputs at a time (host buffers, n x 3 floats; unit tests against the
 * reference's vectors).  op: 0 Render::reflect(a, b) (scene.cpp:672-675), 1 Render::refract(a, b, ior) (677-696),
 * 2 Render::fresnel(a, b, ior) in out[3 i] (698-722), 3 Vec3::normalize(a) (geometry.h:104-112; b may be NULL). */
int rtx_vec_probe(int device, int op, uint32_t n, const float* a, const float* b, float ior, float* out);

/* Host only: the description as one canonical byte string (every scalar and the contents of every array rtx_scene_create reads, in declaration
 * order; no pointers, no padding).  *need = its length; written to out when cap >= *need (out may be NULL to ask for the length).  The parity
 * check of the reference-side binding (oracle/ref_binding.cpp) against this repo's host: tests/test_ref_binding.py. */
int rtx_desc_serialize(const rtx_scene_desc* desc, void* out, size_t cap, size_t* need);

/* rtx_bvh_build builds in a handful of launches (two persistent ones walk the tree through a queue of nodes); the level-by-level build of rounds 1-4 remains
 * as its fallback (a pool ran out, its watchdog fired).  mode 1 forces the fallback (tests compare the two), mode 2 gives the persistent launches pools that are far
 * too small (tests: they must give up cleanly and the fallback must take over); process-wide. */
int rtx_bvh_build_mode(int mode);
/* Kernel launches (fills included) of a finished build, and whether the persistent launches did it (1) or the level-by-level build (0). */
int rtx_bvh_launches(const rtx_bvh* bvh, uint32_t* launches, int* queued);

#ifdef __cplusplus
}
#endif
#endif
