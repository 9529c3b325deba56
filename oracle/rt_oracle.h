/* TEST INFRASTRUCTURE -- CPU oracle for the holoskii/Rendering per-pixel ray-trace hot path.
 *
 * A from-scratch CPU restatement (own code, own structure) of the reference algorithm:
 * scene/OBJ/BMP loading, the spatial-split "SAH" BVH build, Render::trace / castRay recursion,
 * Phong/Fresnel/texture/skybox shading, the tile driver and the Sobel-adaptive SSAA pass.
 * Every function in rt_oracle.cpp cites the reference file:line it follows.
 *
 * PARITY PINNED: bit-identical (float framebuffers, BVH topology/bounds, per-ray hit records, BMP bytes)
 * to the real reference compiled from /root/reference (oracle/_ref, see oracle/Makefile) on every scene
 * under scenes/ -- checked by tests/test_oracle_vs_reference.py in the build container and by the
 * committed golden vectors under tests/golden/ everywhere else.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (rendering_amd/) never links, imports or calls it.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* cwd may be NULL; width/height <= 0 keep the scene file's values.  Returns NULL on load failure
 * (the reference would LOG_ERROR + exit(-1); the message is kept in orc_last_error()). */
orc_scene* orc_load(const char* cwd, const char* scene_path, int width, int height);
void orc_free(orc_scene*);
const char* orc_last_error(void);

void orc_dims(const orc_scene*, int* width, int* height, int* n_objects, int* n_lights);
void orc_set_workers(orc_scene*, int n);          /* default: hardware_concurrency (scene.cpp:68-70) */
void orc_set_flag(orc_scene*, const char* name, int value); /* useBackfaceCulling, collectStatistics */
void orc_camera(orc_scene*, float* scale, float* aspect, float* rmatrix16, float* pos3);

/* fb: H*W*3 floats, zero-initialised by the caller (scene.cpp:599).  Returns elapsed ms. */
double orc_pass1(orc_scene*, float* fb);
/* rows [y0,y1) only -- the sharding unit of the multi-GPU path; same pixels as the tile driver writes. */
double orc_pass1_rows(orc_scene*, float* fb, int y0, int y1);
/* Sobel mask (scene.cpp:547-568); border entries are defined as 0 (reference: uninitialised). */
void orc_sobel(const orc_scene*, const float* fb, uint8_t* mask);
double orc_ssaa(orc_scene*, float* fb, const uint8_t* mask);

/* 64-bit statistics counted under reference traversal semantics: rays (Render::trace calls),
 * box tests, triangle tests.  Collected only while the collectStatistics flag is set. */
void orc_stats_reset(orc_scene*);
void orc_stats(const orc_scene*, int64_t out3[3]);

/* rays: n x 6 (orig, dir). out: n x 8 = [hit, objIdx, triIdx, t, u, v, 0, 0]; colour n x 3 (castRay depth 0). */
void orc_probe(orc_scene*, int n, const float* rays, float* out, float* colour);

void orc_reflect(const float* d, const float* n, float* out);
void orc_refract(const float* d, const float* n, float ior, float* out);
float orc_fresnel(const float* d, const float* n, float ior);
void orc_skybox(const orc_scene*, const float* d, float* out);
void orc_normalize(const float* v, float* out);
void orc_illuminate(const orc_scene*, int light, const float* p, float* out8);
float orc_powf(float x, float y);   /* restated glibc-2.35 FMA-variant powf (see rt_oracle.cpp) */

/* BVH of the mesh at object index obj: counts = [nNodes, nLeaves, nRefs, maxDepth, nTris]; dump in
 * pre-order (left child first).  Same layout as oracle/ref_harness.cpp. */
int orc_bvh_counts(const orc_scene*, int obj, int64_t* counts);
int orc_bvh_dump(const orc_scene*, int obj, float* bounds, int32_t* skip, int32_t* leaf_begin,
                 int32_t* leaf_count, uint32_t* refs);
int orc_tris(const orc_scene*, int obj, float* out30);

/* BMP writer with the reference's header bytes and truncating quantiser (util.cpp:15-76), saturated
 * channel = 255 (the -O0 / MSVC behaviour).  Returns 0 on success. */
int orc_save_bmp(const float* fb, int width, int height, const char* path);
/* In-memory variant: out must hold 54 + 3*W*H bytes. */
void orc_encode_bmp(const float* fb, int width, int height, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
