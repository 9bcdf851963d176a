// scene.hpp -- host-side scene model kept from the reference (scene.h:58-92): Ray, Refl_t, Material, Sphere.
// The tessellated TriMesh members of the reference's Sphere (scene.h:89) are not carried: the MI355X path
// intersects analytic spheres (D1, scene.cpp:129-140).  float3 here is a plain POD (the reference uses
// optix::float3, maths.h:7).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/smallpt_mi355x.h"

namespace spt_host {

struct float3 { float x, y, z; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }

struct Ray {                       // scene.h:58-62
    float3 o, d;
};

enum Refl_t { DIFF, SPEC, REFR };  // scene.h:64

struct Material {                  // scene.h:66-73
    float3 emission;
    float3 color;
    Refl_t refl;
    Material(float3 e_, float3 c_, Refl_t refl_) : emission(e_), color(c_), refl(refl_) {}
};

struct Sphere {                    // scene.h:75-92 (ctor order: radius, position, emission, color, refl)
    float radius;
    float3 center;
    Material material;
    Sphere(float rad_, float3 p_, float3 e_, float3 c_, Refl_t refl_) : radius(rad_), center(p_), material{e_, c_, refl_} {}
};

// Camera description of the JSON scene file; defaults = cpuRender (smallpt.cpp:277-279,333).
struct CameraDesc {
    float3 origin{50, 52, 295.6f};
    float3 direction{0, -0.042612f, -1};
    double fov = .5135;
    float push = 140.0f;
    bool present = false;
};

struct Scene {
    std::vector<Sphere> spheres;
    CameraDesc camera;
};

// The 9-sphere Cornell box that is commented out at smallpt.cpp:36-48 (D11).
Scene cornell9();

// Conversion to the C-ABI records.
std::vector<spt_sphere> to_abi(const std::vector<Sphere>& spheres);
// Camera vectors as cpuRender builds them (smallpt.cpp:277-279) from a CameraDesc, for a w x h image.
spt_camera make_camera(const CameraDesc& c, uint32_t w, uint32_t h);

// JSON scene file: {"camera":{"origin":[3],"direction":[3],"fov":f,"push":f},
//                   "spheres":[{"radius":r,"center":[3],"emission":[3],"color":[3],"refl":"DIFF|SPEC|REFR"}]}
// Throws std::runtime_error with a position on malformed input.
Scene load_scene_json(const std::string& text);
Scene load_scene_file(const std::string& path);
std::string scene_to_json(const Scene& scene);

// One message of the viewer's request queue (smallpt.cpp:909-916,981-984), parsed with the same JSON reader:
// true + org for {"action":"update_camera","org":[x,y,z]}, false for other actions; throws on malformed text.
bool parse_update_camera_request(const std::string& text, float3* org);

}  // namespace spt_host
