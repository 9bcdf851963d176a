/*
 * smallpt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See smallpt_oracle.h
 * for the role and the parity-pin status ("parity unpinned by the reference's own tests";
 * pinned to the SURVEY.md 8(c) known-answer values).
 *
 * Plain C restatement of the reference algorithm.  Every function cites the reference lines it
 * follows (paths relative to /root/reference).  Arithmetic rules (DESIGN.md "Arithmetic spec"):
 *   - every float operation is one IEEE-754 binary32 operation, round-to-nearest-even, NO
 *     contraction into FMA (the reference is host C++ compiled for x86-64: mul and add are
 *     separate SSE instructions).  Built with -ffp-contract=off, never -ffast-math.
 *   - where the reference's C++ promotes to double (double literals / size_t operands,
 *     smallpt.cpp:210,256,331-332) this file does the same in double.
 *   - sqrtf and '/' are the correctly rounded IEEE operations.
 *   - no libm transcendental is on the path: sin/cos come from orc_sincos2pi (D17), the RNG is
 *     the counter-based generator of D7.
 */
#include "smallpt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } f3;

#define ORC_INF 1e20f /* maths.h:16 */

static inline f3 mk(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 ld(const float* p) { return mk(p[0], p[1], p[2]); }
static inline void st(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
/* optixu_math semantics (SURVEY.md 8(c) "Semantics assumed"): componentwise ops,
 * dot = x*x' + y*y' + z*z' left to right, normalize = v * (1/sqrt(dot(v,v))). */
static inline f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scl(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross(f3 a, f3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 normalize(f3 v)
{
    float invLen = 1.0f / sqrtf(dot(v, v));
    return scl(v, invLen);
}

/* ---------------------------------------------------------------- D7: counter-based RNG ----
 * Replaces std::mt19937 + uniform_real_distribution<float> (smallpt.cpp:157,292,319).  The
 * mt19937 stream is consumed in wavefront order across a whole image row, which cannot be
 * reproduced by independent lanes; D7 keys a bijective 32-bit mixer by (seed, pixel, sample)
 * and addresses it by (branch, depth, dimension), so the value of every random decision is
 * independent of scheduling, of the GPU count and of how many numbers other paths drew. */
uint32_t orc_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

void orc_sample_keys(uint64_t seed, uint32_t pixel_idx, uint32_t sample_idx, uint32_t* k0, uint32_t* k1)
{
    uint32_t s0 = orc_mix32((uint32_t)seed + 0x243F6A88u);
    uint32_t s1 = orc_mix32((uint32_t)(seed >> 32) ^ s0 ^ 0x85A308D3u);
    uint32_t p0 = orc_mix32(pixel_idx + s0);
    uint32_t p1 = orc_mix32(pixel_idx ^ s1);
    *k0 = orc_mix32(p0 ^ (sample_idx * 0x9E3779B9u));
    *k1 = orc_mix32(p1 + sample_idx * 0x85EBCA6Bu);
}

uint32_t orc_rng_bits(uint32_t k0, uint32_t k1, uint32_t ctr)
{
    uint32_t x = k0 + ctr * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x21f0aaadu;
    x += k1;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

float orc_rng_uniform(uint32_t k0, uint32_t k1, uint32_t ctr)
{
    /* 24 random bits -> [0,1) exactly representable; never 1.0 */
    return (float)(orc_rng_bits(k0, k1, ctr) >> 8) * 0x1p-24f;
}

/* counter layout: [31:29] branch bits (bit 29+k set = transmitted child of the split at depth k),
 * [28] camera flag, [27:2] depth, [1:0] dimension j. */
#define CTR_CAM(j) ((1u << 28) | (uint32_t)(j))
static inline uint32_t ctr_of(uint32_t branch, uint32_t depth, uint32_t j)
{
    return (branch << 29) | (depth << 2) | j;
}
enum { J_RR = 0, J_R1 = 1, J_R2 = 2 };

/* --------------------------------------------------------------- D17: sin/cos(2*pi*u) ------
 * The reference calls libm cos()/sin() on r1 = 2*M_PI*u (smallpt.cpp:210,212); libm results are
 * not bit-reproducible across C libraries or on a GPU, so the spec fixes one polynomial:
 * quadrant q = floor(4u), f = 4u - q (both exact), S(z) = sin(pi/2 z) ~ odd degree-9 polynomial
 * in Horner form (|err| <= 2.1e-7 in binary32), cos(pi/2 f) = S(1 - f). */
static inline float sin_quarter(float z)
{
    const float c1 = 0x1.921fb4p+0f, c3 = -0x1.4abbb6p-1f, c5 = 0x1.46676ep-4f,
                c7 = -0x1.3232fap-8f, c9 = 0x1.3c4b2cp-13f;
    float z2 = z * z;
    float p = c9;
    p = p * z2 + c7;
    p = p * z2 + c5;
    p = p * z2 + c3;
    p = p * z2 + c1;
    return p * z;
}

void orc_sincos2pi(float u, float* s, float* c)
{
    float t = 4.0f * u;          /* exact */
    int q = (int)t;              /* 0..3 for u in [0,1) */
    float f = t - (float)q;      /* exact */
    float S = sin_quarter(f);
    float C = sin_quarter(1.0f - f);
    switch (q & 3) {
    case 0: *s = S;  *c = C;  break;
    case 1: *s = C;  *c = -S; break;
    case 2: *s = -S; *c = -C; break;
    default: *s = -C; *c = S; break;
    }
}

/* ---------------------------------------------------------------- geometry ---------------- */
/* scene.cpp:129-140 Sphere::intersectAnalytic (D1).  eps = 1e-4 stored in a float (:133). */
static inline float intersect_analytic(f3 center, float radius, f3 o, f3 d, f3* x)
{
    f3 op = sub(center, o);                                   /* :132 */
    float t, eps = 1e-4f, b = dot(op, d);                     /* :133 */
    float det = b * b - dot(op, op) + radius * radius;        /* :133 */
    if (det < 0) return ORC_INF; else det = sqrtf(det);       /* :134 */
    float dist = (t = b - det) > eps ? t : ((t = b + det) > eps ? t : 0); /* :135 */
    if (dist > 0) {                                           /* :136 */
        *x = add(o, scl(d, dist));                            /* :137 */
        return dist;
    }
    return ORC_INF;                                           /* :139, SphereHit{} dist = inf */
}

float orc_intersect_analytic(const orc_sphere* s, const float o[3], const float d[3], float x[3])
{
    f3 hx = mk(0, 0, 0);
    float t = intersect_analytic(ld(s->center), s->radius, ld(o), ld(d), &hx);
    st(x, hx);
    return t;
}

/* scene.cpp:118-127 Sphere::makeHit(SphereHit) */
void orc_make_hit_normal(const orc_sphere* s, const float x[3], float n[3])
{
    st(n, normalize(sub(ld(x), ld(s->center))));              /* :124 */
}

/* smallpt.cpp:54-70 intersectGlobalSpheres (D16: ascending index, strict '<', dist > 0). */
static inline int intersect_global_spheres(const orc_sphere* sph, uint32_t n, f3 o, f3 d,
                                           float* dist, f3* x, f3* nrm)
{
    float nearest = ORC_INF;                                  /* :57, SphereHit{} */
    f3 nx = mk(0, 0, 0);
    int inst = -1;
    for (uint32_t i = 0; i < n; ++i) {                        /* :59 */
        f3 cx = mk(0, 0, 0);
        float cur = intersect_analytic(ld(sph[i].center), sph[i].radius, o, d, &cx); /* :60 */
        if (cur > 0.f && cur < nearest) {                     /* :61 */
            nearest = cur; nx = cx; inst = (int)i;            /* :62-63 */
        }
    }
    if (nearest == ORC_INF) return -1;                        /* :66-67 */
    *dist = nearest;
    *x = nx;
    *nrm = normalize(sub(nx, ld(sph[inst].center)));          /* :69 -> scene.cpp:124 */
    return inst;
}

int orc_intersect_global_spheres(const orc_sphere* s, uint32_t n, const float o[3], const float d[3],
                                 float* dist, float x[3], float nrm[3])
{
    f3 hx = mk(0, 0, 0), hn = mk(0, 0, 0);
    float t = ORC_INF;
    int id = intersect_global_spheres(s, n, ld(o), ld(d), &t, &hx, &hn);
    *dist = t; st(x, hx); st(nrm, hn);
    return id;
}

/* scene.cpp:52-70 triIntersect.  The reference uses double literals here: d = 1.0 / dot(rd, n) is a double division
 * rounded to float by the assignment, and the comparisons u < 0.0 ... promote to double (same truth values). */
static inline float tri_intersect(f3 ro, f3 rd, f3 v0, f3 v1, f3 v2, float* u_, float* v_)
{
    f3 v1v0 = sub(v1, v0), v2v0 = sub(v2, v0), rov0 = sub(ro, v0);   /* :56-58 */
    f3 n = cross(v1v0, v2v0);                                        /* :60 */
    f3 q = cross(rov0, rd);                                          /* :61 */
    float d = (float)(1.0 / (double)dot(rd, n));                     /* :62 */
    float u = d * dot(scl(q, -1.0f), v2v0);                          /* :63 */
    float v = d * dot(q, v1v0);                                      /* :64 */
    float t = d * dot(scl(n, -1.0f), rov0);                          /* :65 */
    if (u < 0.0 || u > 1.0 || v < 0.0 || (u + v) > 1.0) t = ORC_INF; /* :67 */
    *u_ = u; *v_ = v;
    return t;
}

void orc_tri_intersect(const float ro_[3], const float rd_[3], const float v0_[3], const float v1_[3],
                       const float v2_[3], float* t_, float* u_, float* v_)
{
    *t_ = tri_intersect(ld(ro_), ld(rd_), ld(v0_), ld(v1_), ld(v2_), u_, v_);
}

/* scene.cpp:3-48 makeSphereTriMesh (subdivLongitude default 32, scene.h:17): (discLong+1)*(discLat+1) vertices with
 * discLat = 2*discLong, 2*discLong*discLat triangles.  cos/sin are the C++ float overloads (cosf/sinf of the C library:
 * host-side table generation, not on the device path).  Buffers are caller-allocated; returns the triangle count. */
uint32_t orc_make_sphere_trimesh(const float origin[3], float radius, uint32_t subdiv_longitude,
                                 float* positions, float* normals, uint32_t* indices)
{
    const uint32_t discLong = subdiv_longitude, discLat = 2 * discLong;        /* :5-6 */
    const float pi = 3.14159265358979323846f, half_pi = 1.57079632679489661923f; /* M_PIf, M_PI_2f (maths.h:14-15) */
    const float rcpLat = 1.f / discLat, rcpLong = 1.f / discLong;              /* :8 */
    const float dPhi = pi * 2.f * rcpLat, dTheta = pi * rcpLong;               /* :9 */
    uint32_t nv = 0;
    for (uint32_t j = 0; j <= discLong; ++j) {                                 /* :13 */
        const float cosTheta = cosf(-half_pi + j * dTheta);                    /* :15 */
        const float sinTheta = sinf(-half_pi + j * dTheta);                    /* :16 */
        for (uint32_t i = 0; i <= discLat; ++i) {                              /* :18 */
            const f3 coords = mk(sinf(i * dPhi) * cosTheta, sinTheta, cosf(i * dPhi) * cosTheta);   /* :19-23 */
            st(positions + 3 * nv, add(ld(origin), scl(coords, radius)));      /* :25 origin + radius * coords */
            st(normals + 3 * nv, coords);                                      /* :26 */
            ++nv;
        }
    }
    uint32_t ni = 0;
    for (uint32_t j = 0; j < discLong; ++j) {                                  /* :32 */
        const uint32_t offset = j * (discLat + 1);                             /* :34 */
        for (uint32_t i = 0; i < discLat; ++i) {                               /* :35 */
            indices[ni++] = offset + i;                                        /* :37-39 */
            indices[ni++] = offset + (i + 1);
            indices[ni++] = offset + discLat + 1 + (i + 1);
            indices[ni++] = offset + i;                                        /* :41-43 */
            indices[ni++] = offset + discLat + 1 + (i + 1);
            indices[ni++] = offset + i + discLat + 1;
        }
    }
    return ni / 3;
}

/* scene.cpp:95-116 intersect(ro, rd, mesh): brute force over the triangles, nearest dist > 0, strict '<' (lowest
 * triangle index wins ties); returns the triangle index or -1 and the barycentrics of the winner. */
static inline int intersect_mesh(const orc_mesh* m, f3 ro, f3 rd, float* dist, float* u, float* v)
{
    float minDistance = 3.402823466e+38f;                                      /* :97 numeric_limits<float>::max() */
    int minIdx = -1;
    float mu = 0, mv = 0;
    for (uint32_t i = 0; i < m->ntris; ++i) {                                  /* :100 */
        const uint32_t i1 = m->indices[3 * i], i2 = m->indices[3 * i + 1], i3 = m->indices[3 * i + 2];   /* :101-103 */
        float tu, tv;
        const float t = tri_intersect(ro, rd, ld(m->positions + 3 * i1), ld(m->positions + 3 * i2), ld(m->positions + 3 * i3), &tu, &tv);
        if (t > 0 && t < minDistance) {                                        /* :105 */
            minDistance = t; minIdx = (int)i; mu = tu; mv = tv;                /* :106-108 */
        }
    }
    if (minDistance <= 0.f || minDistance == 3.402823466e+38f) return -1;      /* :112-113: MeshHit{} (dist = inf) */
    *dist = minDistance; *u = mu; *v = mv;
    return minIdx;
}

/* CPUIntersector::intersect (smallpt.cpp:443-458) + makeHit(instId, mesh, meshHit) (scene.cpp:73-93): nearest mesh hit
 * over the instances in order, strict '<'; x and n interpolated as w*A + u*B + v*C with w = 1 - u - v (the reference's
 * barycentric convention, smallpt.cpp:544-546); n is NOT normalised (scene.cpp:90).  A triangle hit at dist >= inf
 * (1e20) is a miss like MeshHit{} (:455).  Returns the instance index or -1. */
static inline int intersect_meshes(const orc_mesh* meshes, uint32_t nmesh, f3 ro, f3 rd, orc_hit* hit)
{
    float nearest = ORC_INF;                                                   /* MeshHit{}: dist = inf */
    int inst = -1, tri = -1;
    float nu = 0, nv = 0;
    for (uint32_t i = 0; i < nmesh; ++i) {                                     /* :447 */
        float t, u, v;
        const int k = intersect_mesh(&meshes[i], ro, rd, &t, &u, &v);          /* :448 */
        if (k >= 0 && t > 0.f && t < nearest) {                                /* :449 */
            nearest = t; inst = (int)i; tri = k; nu = u; nv = v;               /* :450-451 */
        }
    }
    hit->dist = ORC_INF; hit->instId = 0; hit->triId = 0;
    st(hit->x, mk(0, 0, 0)); st(hit->n, mk(0, 0, 0)); hit->uv[0] = hit->uv[1] = 0;
    if (nearest == ORC_INF) return -1;                                         /* :454-455 */
    const orc_mesh* m = &meshes[inst];
    const float w = 1.f - nu - nv;                                             /* scene.cpp:82 */
    const uint32_t i1 = m->indices[3 * tri], i2 = m->indices[3 * tri + 1], i3 = m->indices[3 * tri + 2];   /* :84-86 */
    hit->dist = nearest; hit->instId = (uint32_t)inst; hit->triId = (uint32_t)tri;   /* :76-78 */
    st(hit->x, add(add(scl(ld(m->positions + 3 * i1), w), scl(ld(m->positions + 3 * i2), nu)), scl(ld(m->positions + 3 * i3), nv)));   /* :88 */
    st(hit->n, add(add(scl(ld(m->normals + 3 * i1), w), scl(ld(m->normals + 3 * i2), nu)), scl(ld(m->normals + 3 * i3), nv)));       /* :89 */
    hit->uv[0] = nu; hit->uv[1] = nv;                                          /* :90 */
    return inst;
}

/* Intersector::traceRays (smallpt.cpp:460-470): one Hit per ray. */
void orc_trace_rays(const orc_mesh* meshes, uint32_t nmesh, const orc_ray* rays, uint64_t n, orc_hit* hits)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
        intersect_meshes(meshes, nmesh, ld(rays[i].o), ld(rays[i].d), &hits[i]);
}

/* ---------------------------------------------------------------- camera ------------------ */
/* smallpt.cpp:277-279 with D10 (cx = (w*.5135/h, 0, 0)). */
void orc_camera_smallpt(uint32_t w, uint32_t h, orc_camera* cam)
{
    f3 o = mk(50, 52, 295.6f);
    f3 dir = normalize(mk(0, (float)-0.042612, -1));          /* :277 */
    f3 cx = mk((float)((int)w * .5135 / (int)h), 0, 0);       /* :278, double then float */
    f3 cy = scl(normalize(cross(cx, dir)), (float).5135);     /* :279 */
    st(cam->origin, o); st(cam->dir, dir); st(cam->cx, cx); st(cam->cy, cy);
    cam->push = 140.0f;                                       /* :333 */
    cam->sampler = 0;
}

/* Camera ctor smallpt.cpp:609-618: localToWorld columns (vx, vy, vz, org).  sampleRay (:635) multiplies it by
 * (clip.x, clip.y, near, 0): optix Matrix4x4 * float4 = ((m0*x + m1*y) + m2*z) + m3*w per row, so
 * direction = (vx*clip.x + vy*clip.y) + vz*near, the trailing + org*0 adds a signed zero. */
void orc_camera_pinhole(const float vx[3], const float vy[3], const float vz[3], const float org[3], float near_plane, orc_camera* cam)
{
    st(cam->cx, ld(vx)); st(cam->cy, ld(vy));
    st(cam->dir, scl(ld(vz), near_plane));
    st(cam->origin, ld(org));
    cam->push = 0.0f;
    cam->sampler = 1;
}

/* smallpt.cpp:327-333 (D8 tent filter, D10 camera).  jitterSize = 2 (:285). */
static inline void camera_ray(const orc_camera* cam, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                              uint32_t sx, uint32_t sy, float u1, float u2, f3* o, f3* d)
{
    float fax, fay;
    if (cam->sampler == 0) {
        const float r1 = 2 * u1;                                              /* :327 */
        const float dx = r1 < 1 ? sqrtf(r1) - 1 : 1 - sqrtf(2 - r1);          /* :328 */
        const float r2 = 2 * u2;                                              /* :329 */
        const float dy = r2 < 1 ? sqrtf(r2) - 1 : 1 - sqrtf(2 - r2);          /* :330 */
        /* :331-332: size_t + double literal => the bracket is evaluated in double, then converted to
         * float by operator*(float3, float). */
        const double ax = (((double)sx + .5 + (double)dx) / 2.0 + (double)px) / (double)(int)w - .5;
        const double ay = (((double)sy + .5 + (double)dy) / 2.0 + (double)py) / (double)(int)h - .5;
        fax = (float)ax; fay = (float)ay;
    } else {
        /* Renderer::render :745-760 */
        const float cellx = 1.f / 2, celly = 1.f / 2;                         /* :745 jitterCellSize */
        const float pixw = 1.f / (float)w, pixh = 1.f / (float)h;             /* :746 pixelSize */
        const float jx = ((float)sx + u1) * cellx, jy = ((float)sy + u2) * celly; /* :750 */
        const float fx = 0.5f * (2 * jx - 1), fy = 0.5f * (2 * jy - 1);       /* :753-758 box filter in [-0.5, 0.5] */
        /* sampleRay :626-633 */
        const float rx = ((float)px + 0.5f) + fx, ry = ((float)py + 0.5f) + fy; /* :628-630 */
        const float nx = rx * pixw, ny = ry * pixh;                           /* :631 */
        fax = 2.f * nx - 1.f; fay = 2.f * ny - 1.f;                           /* :633 */
    }
    const f3 dd = add(add(scl(ld(cam->cx), fax), scl(ld(cam->cy), fay)), ld(cam->dir));   /* :331-332 / :635 */
    *o = add(ld(cam->origin), scl(dd, cam->push));                         /* :333 */
    *d = normalize(dd);
}

void orc_camera_ray(const orc_camera* cam, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                    uint32_t sx, uint32_t sy, float u1, float u2, float o[3], float d[3])
{
    f3 ro, rd;
    camera_ray(cam, w, h, px, py, sx, sy, u1, u2, &ro, &rd);
    st(o, ro); st(d, rd);
}

/* smallpt.cpp:52 */
int orc_to_int(float x)
{
    float c = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);
    return (int)(pow((double)c, 1 / 2.2) * 255 + .5);
}

/* ---------------------------------------------------------------- one sample -------------- */
typedef struct {
    f3 o, d, w;          /* PathContrib::currentRay, ::weight (smallpt.cpp:106-111) */
    uint32_t depth;      /* PathContrib::depth */
    uint32_t branch;     /* D7 counter bits */
} path_t;

typedef struct {
    const orc_sphere* sph;       /* sphere scene (analytic primitives, D1) ... */
    uint32_t n;
    const orc_mesh* meshes;      /* ... or a triangle-mesh scene (the reference's Intersector seam): one material per instance */
    const orc_material* mats;
    uint32_t nmesh;
    int zero_cut;
    uint64_t bounces;
    uint64_t depth_kills;
} trace_ctx;

/* extend() smallpt.cpp:120-123 + D18 depth cap + zero-weight cut (SURVEY.md section 7,
 * "Zero-throughput waste": a path whose weight is exactly 0 can never contribute again). */
static inline int make_child(trace_ctx* tc, const path_t* p, f3 o, f3 d, f3 factor, uint32_t branch, path_t* c)
{
    c->o = o; c->d = d;
    c->w = mul(p->w, factor);                                  /* :122 */
    c->depth = p->depth + 1;
    c->branch = branch;
    if (c->depth >= ORC_MAX_DEPTH) { tc->depth_kills++; return 0; }
    if (tc->zero_cut && c->w.x == 0.f && c->w.y == 0.f && c->w.z == 0.f) return 0;
    return 1;
}

/* Traces the whole path tree of one camera ray; emission events are added to *acc in DFS
 * pre-order, reflected child before transmitted child (smallpt.cpp:251-252 order). */
static void trace_sample(trace_ctx* tc, path_t cam_path, uint32_t k0, uint32_t k1, f3* acc)
{
    path_t stack[4];
    int sp = 0;
    stack[sp++] = cam_path;
    while (sp > 0) {
        path_t p = stack[--sp];
        for (;;) {
            /* ---- intersectGlobalSpheres, smallpt.cpp:352 -> :144-152 -> :54-70; or Intersector::traceRays :782 ---- */
            float dist; f3 hx, n;
            tc->bounces++;
            int id;
            orc_material mesh_mat;
            const orc_material* m;
            if (tc->meshes) {
                orc_hit hit;
                id = intersect_meshes(tc->meshes, tc->nmesh, p.o, p.d, &hit);
                if (id < 0) break;                             /* :168 */
                hx = ld(hit.x); n = ld(hit.n);                 /* :172-173: the interpolated, un-normalised mesh normal */
                mesh_mat = tc->mats[id];                       /* :170 materials[hit.instId] */
                m = &mesh_mat;
            } else {
                id = intersect_global_spheres(tc->sph, tc->n, p.o, p.d, &dist, &hx, &n);
                if (id < 0) break;                             /* :168 miss => black (D13) */
                /* the material fields of orc_sphere from `emission` on have the layout of orc_material */
                m = (const orc_material*)tc->sph[id].emission;
            }
            /* ---- shadePaths body, smallpt.cpp:170-263 ---- */
            f3 nl = dot(n, p.d) < 0 ? n : scl(n, -1.0f);       /* :174 with the flip (D2) */
            f3 f = ld(m->color);                               /* :175 */
            const float pmax = fmaxf(fmaxf(f.x, f.y), f.z);    /* :177 optix::fmaxf(float3) */
            *acc = add(*acc, mul(p.w, ld(m->emission)));       /* :179 (D4) */
            const uint32_t depth = p.depth;                    /* :185 */
            if (depth > 5) {                                   /* :188 (D5) */
                if (orc_rng_uniform(k0, k1, ctr_of(p.branch, depth, J_RR)) < pmax)
                    f = scl(f, 1 / pmax);                      /* :192 */
                else
                    break;                                     /* :196 */
            }
            /* D3: the new origin is offset 0.02 along the side the outgoing ray leaves on */
            f3 off = scl(nl, 0.02f);                           /* :172 */
            f3 x_out = add(hx, off);
            path_t c;
            if (m->refl == ORC_DIFF) {                         /* :208 */
                float u1 = orc_rng_uniform(k0, k1, ctr_of(p.branch, depth, J_R1));
                float r2 = orc_rng_uniform(k0, k1, ctr_of(p.branch, depth, J_R2));
                float r2s = sqrtf(r2);                         /* :210 */
                float sn, cs;
                orc_sincos2pi(u1, &sn, &cs);                   /* :210,212 via D17 */
                f3 w = nl;                                     /* :211 */
                f3 u = normalize(cross(((double)fabsf(w.x) > .1 ? mk(0, 1, 0) : mk(1, 0, 0)), w));
                f3 v = cross(w, u);
                f3 d = normalize(add(add(scl(scl(u, cs), r2s), scl(scl(v, sn), r2s)),
                                     scl(w, sqrtf(1 - r2))));  /* :212 */
                if (!make_child(tc, &p, x_out, d, f, p.branch, &c)) break; /* :214 */
                p = c;
                continue;
            }
            /* :218 reflRay(x, r.d - n*2*dot(n, r.d)) */
            f3 rd = sub(p.d, scl(scl(n, 2.0f), dot(n, p.d)));
            if (m->refl == ORC_SPEC) {                         /* :219 */
                if (!make_child(tc, &p, x_out, rd, f, p.branch, &c)) break; /* :221 */
                p = c;
                continue;
            }
            const int into = dot(n, nl) > 0;                   /* :225 */
            const float nc = 1;                                /* :226 */
            const float nt = 1.5;                              /* :227 */
            const float nnt = into ? nc / nt : nt / nc;        /* :228 */
            const float ddn = dot(p.d, nl);                    /* :229 */
            const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn); /* :230 */
            if (cos2t < 0) {                                   /* :232 total internal reflection */
                if (!make_child(tc, &p, x_out, rd, f, p.branch, &c)) break; /* :234 */
                p = c;
                continue;
            }
            /* :238 */
            f3 tdir = normalize(sub(scl(p.d, nnt), scl(n, (float)(into ? 1 : -1) * (ddn * nnt + sqrtf(cos2t)))));
            const float a = nt - nc;                           /* :240 */
            const float b = nt + nc;                           /* :241 */
            const float R0 = a * a / (b * b);                  /* :242 */
            const float cc = 1 - (into ? -ddn : dot(tdir, n)); /* :243 */
            const float c2 = cc * cc;                          /* :244 */
            const float Re = R0 + (1 - R0) * c2 * c2 * cc;     /* :245 */
            const float Tr = 1 - Re;                           /* :246 */
            f3 x_in = sub(hx, off);                            /* D3: transmitted ray leaves on the -nl side */
            if (depth <= 2) {                                  /* :248 split (D6) */
                path_t ct;
                int has_t = make_child(tc, &p, x_in, tdir, scl(f, Tr), p.branch | (1u << depth), &ct); /* :252 */
                int has_r = make_child(tc, &p, x_out, rd, scl(f, Re), p.branch, &c);                    /* :251 */
                if (has_t) stack[sp++] = ct;                   /* processed after the reflected subtree */
                if (!has_r) break;
                p = c;
                continue;
            }
            const float P = (float)(.25 + .5 * Re);            /* :256 (double literals) */
            if (orc_rng_uniform(k0, k1, ctr_of(p.branch, depth, J_R1)) < P) { /* :257 */
                /* :259 f * Re / P ; optix operator/(float3,float) multiplies by 1.0f/P */
                if (!make_child(tc, &p, x_out, rd, scl(scl(f, Re), 1.0f / P), p.branch, &c)) break;
                p = c;
                continue;
            }
            /* :263 f * Tr / (1.f - P) */
            if (!make_child(tc, &p, x_in, tdir, scl(scl(f, Tr), 1.0f / (1.f - P)), p.branch, &c)) break;
            p = c;
        }
    }
}

/* ---------------------------------------------------------------- render ------------------ */
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* D9 block layout of a jitter cell's `samps` samples: NB = 1, 2, 4 or 8 blocks (>= 16 samples each), SB = ceil(samps/NB)
 * samples per block (the last block may be shorter, never empty). */
void orc_sample_blocks(uint32_t samps, uint32_t* nb, uint32_t* sb)
{
    const uint32_t n = samps >= 128u ? 8u : (samps >= 64u ? 4u : (samps >= 32u ? 2u : 1u));
    *nb = n;
    *sb = (samps + n - 1u) / n;
}

/* cpuRender smallpt.cpp:269-361: per pixel/cell/sample in the order of
 * foreachSampleInRow (:294-314).  D9 accumulation order: the reference adds weight*emission into the pixel in
 * wavefront (depth-major) order across all samples of a row (:179, :349-356), which no path-owning worker can
 * reproduce; the spec instead fixes: each jitter cell's samples are split into NB consecutive blocks
 * (orc_sample_blocks); inside a block every emission event is added to the block accumulator in (sample-ascending,
 * DFS pre-order) order; cell = ((B0 + B1) + B2) + ... in block order; pixel = ((c0 + c1) + c2) + c3.
 * (Up to 31 samples per cell there is one block: the classic per-subpixel accumulator of smallpt.) */
static int render_scene(const orc_sphere* spheres, uint32_t n, const orc_mesh* meshes, uint32_t nmesh, const orc_material* mats,
                        const orc_camera* cam, uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
                        uint32_t samps, uint64_t seed, uint32_t flags, int threads, float* out, orc_stats* stats);

int orc_render(const orc_sphere* spheres, uint32_t n, const orc_camera* cam,
               uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
               uint32_t samps, uint64_t seed, uint32_t flags, int threads,
               float* out, orc_stats* stats)
{
    if (!spheres && n) return 1;
    return render_scene(spheres, n, NULL, 0, NULL, cam, w, h, row_begin, row_count, samps, seed, flags, threads, out, stats);
}

/* The same render over a triangle-mesh scene: closest hit = CPUIntersector (smallpt.cpp:427-473), material = materials[instId]. */
int orc_render_meshes(const orc_mesh* meshes, uint32_t nmesh, const orc_material* materials, const orc_camera* cam,
                      uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
                      uint32_t samps, uint64_t seed, uint32_t flags, int threads, float* out, orc_stats* stats)
{
    static const orc_mesh none = {0, 0, 0, 0, 0};
    if (nmesh && (!meshes || !materials)) return 1;
    return render_scene(NULL, 0, meshes ? meshes : &none, nmesh, materials, cam, w, h, row_begin, row_count, samps, seed, flags, threads, out, stats);
}

static int render_scene(const orc_sphere* spheres, uint32_t n, const orc_mesh* meshes, uint32_t nmesh, const orc_material* mats,
                        const orc_camera* cam, uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
                        uint32_t samps, uint64_t seed, uint32_t flags, int threads, float* out, orc_stats* stats)
{
    if (!cam || !out || w == 0 || h == 0 || samps == 0) return 1;
    if ((uint64_t)w * h > 0xFFFFFFFFull) return 1;
    if ((uint64_t)row_begin + row_count > h) return 1;
    if ((uint64_t)samps * 4 > 0xFFFFFFFFull) return 1;
    uint64_t tot_b = 0, tot_k = 0;
    const uint32_t spp = 4 * samps;                                   /* :286 */
    uint32_t nb, sb;
    orc_sample_blocks(samps, &nb, &sb);
    if (flags & (ORC_FLAG_SEQUENTIAL_CELLS | ORC_FLAG_SEQUENTIAL_PIXEL)) { nb = 1; sb = samps; }
    const int one_sum = (flags & ORC_FLAG_SEQUENTIAL_PIXEL) != 0;
    /* The reference parallelises over rows (:317); here the unit is a chunk of 16 consecutive pixels so that a
     * band of a few rows still uses every core.  Pixels are independent, so the image does not depend on it. */
    const int64_t npix = (int64_t)row_count * w;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) reduction(+ : tot_b, tot_k)
#endif
    for (int64_t pi = 0; pi < npix; ++pi) {
        trace_ctx tc;
        tc.sph = spheres; tc.n = n; tc.meshes = meshes; tc.nmesh = nmesh; tc.mats = mats; tc.zero_cut = !(flags & ORC_FLAG_NO_ZERO_WEIGHT_CUT);
        tc.bounces = 0; tc.depth_kills = 0;
        const uint32_t r = (uint32_t)(pi / w);
        const uint32_t px = (uint32_t)(pi - (int64_t)r * w);                    /* :296 */
        const uint32_t py = row_begin + r;                                      /* :317 */
        {
            const uint32_t pixel_idx = py * w + px;                   /* :298 */
            f3 cell[4];
            f3 pixsum = mk(0, 0, 0);                                  /* ORC_FLAG_SEQUENTIAL_PIXEL: the pixel's only accumulator */
            for (uint32_t sy = 0; sy < 2; ++sy)                       /* :299 */
                for (uint32_t sx = 0; sx < 2; ++sx) {                 /* :301 */
                    const uint32_t g = sy * 2 + sx;                   /* :303 */
                    f3 cellsum = mk(0, 0, 0);
                    for (uint32_t blk = 0; blk < nb; ++blk) {         /* D9 blocks */
                        const uint32_t s_end = (blk + 1) * sb < samps ? (blk + 1) * sb : samps;
                        f3 acc = mk(0, 0, 0);
                        for (uint32_t s = blk * sb; s < s_end; ++s) { /* :304 */
                            const uint32_t index_in_pixel = g * samps + s; /* :306 */
                            uint32_t k0, k1;
                            orc_sample_keys(seed, pixel_idx, index_in_pixel, &k0, &k1);
                            const float u1 = orc_rng_uniform(k0, k1, CTR_CAM(0));
                            const float u2 = orc_rng_uniform(k0, k1, CTR_CAM(1));
                            path_t p;
                            camera_ray(cam, w, h, px, py, sx, sy, u1, u2, &p.o, &p.d);
                            p.w = mk(1, 1, 1); p.depth = 0; p.branch = 0; /* :338-339 */
                            trace_sample(&tc, p, k0, k1, one_sum ? &pixsum : &acc);
                        }
                        cellsum = blk == 0 ? acc : add(cellsum, acc);
                    }
                    cell[g] = cellsum;
                }
            f3 c = one_sum ? pixsum : add(add(add(cell[0], cell[1]), cell[2]), cell[3]);
            if (flags & ORC_FLAG_NORMALISE) c = scl(c, 1.0f / (float)spp); /* :360, operator/= */
            st(out + ((size_t)r * w + px) * 3, c);
        }
        tot_b += tc.bounces; tot_k += tc.depth_kills;
    }
    if (stats) {
        stats->samples = (uint64_t)row_count * w * spp;
        stats->bounces = tot_b;
        stats->max_depth_kills = tot_k;
    }
    return 0;
}
