/*
 * smallpt_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's path-tracing hot path
 * (/root/reference/smallpt.cpp:54-70,154-267,269-361 and scene.cpp:118-140) under the
 * decisions D1-D18 recorded in DESIGN.md.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path
 * (optix-test-smallpt_amd/) never links, includes or calls anything in oracle/.
 *
 * PARITY PIN STATUS: the reference ships no tests, fixtures or golden images
 * (SURVEY.md section 4), and it cannot be compiled in this image (it needs the NVIDIA
 * OptiX SDK headers, OptiX Prime and GLFW, none of which are present; no stand-ins are
 * written).  The oracle is therefore "parity unpinned by the reference's own tests"; it is
 * pinned instead against the known-answer values that SURVEY.md section 8(c) recorded from the
 * reference's own intersectAnalytic()/makeHit() (tests/golden/reference_kats.json).
 */
#ifndef SMALLPT_ORACLE_H
#define SMALLPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 48-byte POD as the product's spt_sphere (include/smallpt_mi355x.h); restated here so the
 * oracle does not include product headers.  Field order = Sphere ctor, scene.h:91. */
typedef struct {
    float   center[3];   /* scene.h:77  Sphere::center            */
    float   radius;      /* scene.h:76  Sphere::radius            */
    float   emission[3]; /* scene.h:68  Material::emission        */
    float   color[3];    /* scene.h:69  Material::color           */
    int32_t refl;        /* scene.h:64  Refl_t: DIFF=0 SPEC=1 REFR=2 */
    uint32_t pad;
} orc_sphere;

/* smallpt camera of cpuRender (smallpt.cpp:277-279,331-333). */
typedef struct {
    float origin[3];     /* cam.o                                  */
    float dir[3];        /* cam.d (already normalised)             */
    float cx[3];         /* horizontal image-plane vector          */
    float cy[3];         /* vertical image-plane vector            */
    float push;          /* 140: ray origin = cam.o + d*push       */
    uint32_t sampler;    /* 0: cpuRender tent filter (smallpt.cpp:327-332); 1: Renderer::render box-in-cell +
                            sampleRay pinhole (smallpt.cpp:745-760,626-641) */
} orc_camera;

/* Material, scene.h:66-73 (emission, color, refl) + padding: the tail of orc_sphere from `emission` on has this layout. */
typedef struct {
    float   emission[3];
    float   color[3];
    int32_t refl;
    uint32_t pad;
} orc_material;

/* TriMesh, scene.h:6-15: positionBuffer / normalBuffer (nverts x 3 floats), indexBuffer (ntris x 3). */
typedef struct {
    const float*    positions;
    const float*    normals;
    const uint32_t* indices;
    uint32_t nverts, ntris;
} orc_mesh;

typedef struct { float o[3], d[3]; } orc_ray;              /* Ray, scene.h:58-62 */
/* Hit, scene.h:31-43 (44 bytes): dist = 1e20 (inf, maths.h:16) on a miss */
typedef struct { float dist; uint32_t instId, triId; float x[3], n[3], uv[2]; } orc_hit;

typedef struct {
    uint64_t samples;    /* camera paths started                   */
    uint64_t bounces;    /* intersectGlobalSpheres() calls executed */
    uint64_t max_depth_kills; /* paths cut by ORC_MAX_DEPTH (D18)  */
} orc_stats;

enum { ORC_DIFF = 0, ORC_SPEC = 1, ORC_REFR = 2 };

#define ORC_FLAG_NORMALISE        1u  /* divide by spp like cpuRender :358-361 (else raw sum like render() :813) */
#define ORC_FLAG_NO_ZERO_WEIGHT_CUT 2u /* keep bouncing zero-weight paths (test of result-preservation) */
#define ORC_FLAG_SEQUENTIAL_CELLS 4u   /* D9 with NB = 1 whatever the sample count: one accumulator per jitter cell (round 1's spec = smallpt's
                                          per-subpixel accumulator); bounds what the block split of D9 changes (tests/test_oracle.py) */
#define ORC_FLAG_SEQUENTIAL_PIXEL 8u   /* one accumulator per PIXEL, every emission event added in (cell, sample, DFS) order: the closest a
                                          path-owning worker comes to `outColor[pixelIdx] +=` of smallpt.cpp:179, 358-361 */
#define ORC_MAX_DEPTH 4096u            /* D18 */

/* --- unit-level entry points (for known-answer tests) --- */
/* scene.cpp:129-140; returns dist (1e20f on miss) and writes hit point x */
float orc_intersect_analytic(const orc_sphere* s, const float o[3], const float d[3], float x[3]);
/* scene.cpp:118-127; n = normalize(x - center) */
void  orc_make_hit_normal(const orc_sphere* s, const float x[3], float n[3]);
/* smallpt.cpp:54-70; returns sphere index or -1; writes dist, x, n */
int   orc_intersect_global_spheres(const orc_sphere* s, uint32_t n, const float o[3], const float d[3],
                                   float* dist, float x[3], float nrm[3]);
/* scene.cpp:52-70 triIntersect */
void  orc_tri_intersect(const float ro[3], const float rd[3], const float v0[3], const float v1[3],
                        const float v2[3], float* t, float* u, float* v);
/* scene.cpp:3-48 makeSphereTriMesh: fills (L+1)(2L+1) positions/normals and 4L^2 triangles (L = subdiv_longitude);
 * returns the triangle count */
uint32_t orc_make_sphere_trimesh(const float origin[3], float radius, uint32_t subdiv_longitude,
                                 float* positions, float* normals, uint32_t* indices);
/* Intersector::traceRays (smallpt.cpp:427-473: CPUIntersector::intersect = scene.cpp:95-116 per mesh + makeHit :73-93) */
void  orc_trace_rays(const orc_mesh* meshes, uint32_t nmesh, const orc_ray* rays, uint64_t n, orc_hit* hits);
/* D7 counter-based RNG */
uint32_t orc_mix32(uint32_t x);
void  orc_sample_keys(uint64_t seed, uint32_t pixel_idx, uint32_t sample_idx, uint32_t* k0, uint32_t* k1);
uint32_t orc_rng_bits(uint32_t k0, uint32_t k1, uint32_t ctr);
float orc_rng_uniform(uint32_t k0, uint32_t k1, uint32_t ctr);
/* D17 sin/cos of 2*pi*u */
void  orc_sincos2pi(float u, float* s, float* c);
/* camera of smallpt.cpp:277-279 for a w x h image */
void  orc_camera_smallpt(uint32_t w, uint32_t h, orc_camera* cam);
/* Camera{vx,vy,vz,org,near} of smallpt.cpp:607-624 in the d = cx*ax + cy*ay + dir form */
void  orc_camera_pinhole(const float vx[3], const float vy[3], const float vz[3], const float org[3], float near_plane, orc_camera* cam);
/* smallpt.cpp:327-333: camera ray of sample (px,py,sx,sy) given the two uniforms */
void  orc_camera_ray(const orc_camera* cam, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                     uint32_t sx, uint32_t sy, float u1, float u2, float o[3], float d[3]);
/* smallpt.cpp:52 */
int   orc_to_int(float x);

/* D9: block layout of a jitter cell's samples in the accumulation order (nb = 1, 2, 4 or 8 blocks of sb samples) */
void  orc_sample_blocks(uint32_t samps_per_cell, uint32_t* nb, uint32_t* sb);

/* --- the render (cpuRender :269-361 restated) ---
 * Renders rows [row_begin, row_begin+row_count) of a w x h image into out (row_count*w*3 floats,
 * row 0 of the band first; row index 0 = bottom of the image, D14).
 * threads<=0: all cores (OpenMP, dynamic scheduling of 16-pixel chunks).  Returns 0 on success. */
int orc_render(const orc_sphere* spheres, uint32_t n, const orc_camera* cam,
               uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
               uint32_t samps_per_cell, uint64_t seed, uint32_t flags, int threads,
               float* out, orc_stats* stats);

/* The same render over a triangle-mesh scene (materials[i] belongs to mesh instance i, smallpt.cpp:170). */
int orc_render_meshes(const orc_mesh* meshes, uint32_t nmesh, const orc_material* materials, const orc_camera* cam,
                      uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
                      uint32_t samps_per_cell, uint64_t seed, uint32_t flags, int threads,
                      float* out, orc_stats* stats);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
