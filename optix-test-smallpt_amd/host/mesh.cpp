// mesh.cpp -- TriMesh helpers of the host (scene.h:6-17, scene.cpp:3-48); kept apart from scene.cpp so that the scene /
// JSON reader builds without the device library (tests/sanitize).
#include "scene.hpp"

namespace spt_host {

TriMesh makeSphereTriMesh(const float3 origin, float radius, const uint32_t subdivLongitude)
{
    TriMesh m;
    const size_t n = subdivLongitude;
    m.positionBuffer.resize((n + 1) * (2 * n + 1));
    m.normalBuffer.resize(m.positionBuffer.size());
    m.indexBuffer.resize(12 * n * n);
    const float o[3] = {origin.x, origin.y, origin.z};
    spt_make_sphere_trimesh(o, radius, subdivLongitude, reinterpret_cast<float*>(m.positionBuffer.data()),
                            reinterpret_cast<float*>(m.normalBuffer.data()), m.indexBuffer.data());
    return m;
}

// Runs the tessellator for every "sphere" entry of a loaded scene (the loader itself is host-only and does not link the
// library); explicit-buffer entries are left as they are.
void realize_meshes(Scene& scene)
{
    for (MeshInstance& m : scene.meshes)
        if (m.generator == "sphere" && m.mesh.indexBuffer.empty()) m.mesh = makeSphereTriMesh(m.center, m.radius, m.subdiv);
}

Scene shipped_two_sphere_mesh_scene()
{
    Scene s;   // smallpt.cpp:32-33: Sphere(10, (50,40.8,81.6), 0, (.75,.25,.25), DIFF), Sphere(600, (50,681.6-.27,81.6), (1,1,1), 0, DIFF)
    MeshInstance a, b;
    a.generator = b.generator = "sphere";
    a.center = make_float3(50, 40.8f, 81.6f); a.radius = 10.f;
    a.material = Material(make_float3(0, 0, 0), make_float3(.75f, .25f, .25f), DIFF);
    b.center = make_float3(50, (float)(681.6 - .27), 81.6f); b.radius = 600.f;
    b.material = Material(make_float3(1, 1, 1), make_float3(0, 0, 0), DIFF);
    s.meshes = {a, b};
    realize_meshes(s);
    return s;
}

}  // namespace spt_host
