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

}  // namespace spt_host
