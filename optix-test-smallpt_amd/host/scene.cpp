// scene.cpp -- Cornell table, ABI conversion, camera constants and the JSON scene reader/writer.
// The reference only uses JSON for its GL-thread -> render-thread messages (smallpt.cpp:909-918,981-984);
// the scene file format is the one SURVEY.md 8(f).1 defines (vectors as 3-number arrays like those messages).
// A ~150-line recursive-descent reader is used instead of vendoring a 12-kLoC JSON library.
#include "scene.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace spt_host {

Scene cornell9()
{
    Scene s;
    const float3 z = make_float3(0, 0, 0);
    auto& v = s.spheres;   // Scene: radius, position, emission, color, material (smallpt.cpp:36)
    v.emplace_back(1e5f, make_float3(1e5f + 1, 40.8f, 81.6f), z, make_float3(.75f, .25f, .25f), DIFF);   // Left :38
    v.emplace_back(1e5f, make_float3(-1e5f + 99, 40.8f, 81.6f), z, make_float3(.25f, .25f, .75f), DIFF);  // Rght :39
    v.emplace_back(1e5f, make_float3(50, 40.8f, 1e5f), z, make_float3(.75f, .75f, .75f), DIFF);           // Back :40
    v.emplace_back(1e5f, make_float3(50, 40.8f, -1e5f + 170), z, z, DIFF);                                // Frnt :41
    v.emplace_back(1e5f, make_float3(50, 1e5f, 81.6f), z, make_float3(.75f, .75f, .75f), DIFF);           // Botm :42
    v.emplace_back(1e5f, make_float3(50, -1e5f + 81.6f, 81.6f), z, make_float3(.75f, .75f, .75f), DIFF);  // Top  :43
    v.emplace_back(16.5f, make_float3(27, 16.5f, 47), z, make_float3(.999f, .999f, .999f), SPEC);         // Mirr :44
    v.emplace_back(16.5f, make_float3(73, 16.5f, 78), z, make_float3(.999f, .999f, .999f), REFR);         // Glas :45
    v.emplace_back(600.f, make_float3(50, (float)(681.6 - .27), 81.6f), make_float3(1, 1, 1), z, DIFF);   // Lite :46
    return s;
}

std::vector<spt_sphere> to_abi(const std::vector<Sphere>& spheres)
{
    std::vector<spt_sphere> out(spheres.size());
    for (size_t i = 0; i < spheres.size(); ++i) {
        const Sphere& s = spheres[i];
        spt_sphere& o = out[i];
        std::memset(&o, 0, sizeof o);
        o.center[0] = s.center.x; o.center[1] = s.center.y; o.center[2] = s.center.z;
        o.radius = s.radius;
        o.emission[0] = s.material.emission.x; o.emission[1] = s.material.emission.y; o.emission[2] = s.material.emission.z;
        o.color[0] = s.material.color.x; o.color[1] = s.material.color.y; o.color[2] = s.material.color.z;
        o.refl = (int32_t)s.material.refl;
    }
    return out;
}

namespace {
inline float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross3(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float3 scale3(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 normalize3(float3 v) { const float inv = 1.0f / std::sqrt(dot3(v, v)); return scale3(v, inv); }
}  // namespace

spt_camera make_camera(const CameraDesc& c, uint32_t w, uint32_t h)
{
    spt_camera cam;
    const float3 dir = normalize3(c.direction);                                  // smallpt.cpp:277
    const float3 cx = make_float3((float)((int)w * c.fov / (int)h), 0, 0);       // :278 (D10)
    const float3 cy = scale3(normalize3(cross3(cx, dir)), (float)c.fov);         // :279
    cam.origin[0] = c.origin.x; cam.origin[1] = c.origin.y; cam.origin[2] = c.origin.z;
    cam.dir[0] = dir.x; cam.dir[1] = dir.y; cam.dir[2] = dir.z;
    cam.cx[0] = cx.x; cam.cx[1] = cx.y; cam.cx[2] = cx.z;
    cam.cy[0] = cy.x; cam.cy[1] = cy.y; cam.cy[2] = cy.z;
    cam.push = c.push;
    cam.sampler = SPT_SAMPLER_SMALLPT;
    return cam;
}

// ------------------------------------------------------------------------------------------ JSON reader
namespace {

struct JValue;
using JPtr = std::shared_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JPtr> arr;
    std::map<std::string, JPtr> obj;
};

class Parser {
public:
    explicit Parser(const std::string& t) : s_(t) {}
    JPtr parse()
    {
        JPtr v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s_;
    size_t i_ = 0;
    int depth_ = 0;                          // nesting of arrays/objects; bounded so that hostile input cannot overflow the stack
    static constexpr int kMaxDepth = 64;

    [[noreturn]] void fail(const char* what) const
    {
        std::ostringstream m;
        m << "scene JSON: " << what << " at offset " << i_;
        throw std::runtime_error(m.str());
    }
    void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) ++i_; }
    bool eat(char c) { ws(); if (i_ < s_.size() && s_[i_] == c) { ++i_; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail("unexpected character"); }

    JPtr value()
    {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        auto v = std::make_shared<JValue>();
        const char c = s_[i_];
        if ((c == '{' || c == '[') && ++depth_ > kMaxDepth) fail("nesting too deep");
        if (c == '{') {
            ++i_; v->kind = JValue::Object;
            if (eat('}')) { --depth_; return v; }
            do {
                ws();
                if (i_ >= s_.size() || s_[i_] != '"') fail("expected object key");
                std::string k = string();
                expect(':');
                v->obj[k] = value();
            } while (eat(','));
            expect('}');
            --depth_;
        } else if (c == '[') {
            ++i_; v->kind = JValue::Array;
            if (eat(']')) { --depth_; return v; }
            do { v->arr.push_back(value()); } while (eat(','));
            expect(']');
            --depth_;
        } else if (c == '"') {
            v->kind = JValue::String; v->str = string();
        } else if (!s_.compare(i_, 4, "true")) { i_ += 4; v->kind = JValue::Bool; v->b = true;
        } else if (!s_.compare(i_, 5, "false")) { i_ += 5; v->kind = JValue::Bool;
        } else if (!s_.compare(i_, 4, "null")) { i_ += 4;
        } else {
            const char* b = s_.c_str() + i_;
            char* e = nullptr;
            v->num = std::strtod(b, &e);
            if (e == b) fail("invalid value");
            v->kind = JValue::Number;
            i_ += (size_t)(e - b);
        }
        return v;
    }
    std::string string()
    {
        std::string out;
        ++i_;   // opening quote
        while (i_ < s_.size() && s_[i_] != '"') {
            char c = s_[i_++];
            if (c == '\\') {
                if (i_ >= s_.size()) fail("bad escape");
                const char e = s_[i_++];
                switch (e) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': if (i_ + 4 > s_.size()) fail("bad \\u escape"); out += '?'; i_ += 4; break;
                default: out += e;
                }
            } else out += c;
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return out;
    }
};

const JValue& member(const JValue& o, const char* key)
{
    auto it = o.obj.find(key);
    if (o.kind != JValue::Object || it == o.obj.end()) throw std::runtime_error(std::string("scene JSON: missing field \"") + key + "\"");
    return *it->second;
}
float3 vec3(const JValue& v, const char* what)
{
    if (v.kind != JValue::Array || v.arr.size() != 3) throw std::runtime_error(std::string("scene JSON: \"") + what + "\" must be an array of 3 numbers");
    for (auto& e : v.arr) if (e->kind != JValue::Number) throw std::runtime_error(std::string("scene JSON: \"") + what + "\" must be an array of 3 numbers");
    return make_float3((float)v.arr[0]->num, (float)v.arr[1]->num, (float)v.arr[2]->num);
}
double number(const JValue& v, const char* what)
{
    if (v.kind != JValue::Number) throw std::runtime_error(std::string("scene JSON: \"") + what + "\" must be a number");
    return v.num;
}

}  // namespace

namespace {
Refl_t parse_refl(const JValue& r)
{
    if (r.kind == JValue::String) {
        if (r.str == "DIFF") return DIFF;
        if (r.str == "SPEC") return SPEC;
        if (r.str == "REFR") return REFR;
        throw std::runtime_error("scene JSON: refl must be DIFF, SPEC or REFR");
    }
    const int k = (int)number(r, "refl");
    if (k < 0 || k > 2) throw std::runtime_error("scene JSON: refl must be 0, 1 or 2");
    return (Refl_t)k;
}
std::vector<float3> vec3_list(const JValue& v, const char* what)
{
    if (v.kind != JValue::Array) throw std::runtime_error(std::string("scene JSON: \"") + what + "\" must be an array of 3-number arrays");
    std::vector<float3> out;
    out.reserve(v.arr.size());
    for (auto& e : v.arr) out.push_back(vec3(*e, what));
    return out;
}
}  // namespace

// Host-only part of the mesh loader: everything except running the tessellator (mesh.cpp fills `mesh` of "sphere" entries).
static void load_meshes(const JValue& root, Scene& sc)
{
    auto it = root.obj.find("meshes");
    if (it == root.obj.end()) return;
    if (it->second->kind != JValue::Array) throw std::runtime_error("scene JSON: \"meshes\" must be an array");
    for (auto& e : it->second->arr) {
        MeshInstance mi;
        mi.material = Material(vec3(member(*e, "emission"), "emission"), vec3(member(*e, "color"), "color"), parse_refl(member(*e, "refl")));
        auto sp = e->obj.find("sphere");
        if (sp != e->obj.end()) {
            mi.generator = "sphere";
            mi.center = vec3(member(*sp->second, "center"), "center");
            mi.radius = (float)number(member(*sp->second, "radius"), "radius");
            if (sp->second->obj.count("subdiv")) {
                const double sd = number(member(*sp->second, "subdiv"), "subdiv");
                if (!(sd >= 1 && sd <= 2048)) throw std::runtime_error("scene JSON: subdiv must be in 1..2048");
                mi.subdiv = (uint32_t)sd;
            }
        } else {
            mi.mesh.positionBuffer = vec3_list(member(*e, "positions"), "positions");
            mi.mesh.normalBuffer = vec3_list(member(*e, "normals"), "normals");
            if (mi.mesh.normalBuffer.size() != mi.mesh.positionBuffer.size()) throw std::runtime_error("scene JSON: positions and normals differ in length");
            // triangle indices are integers: read them in double (exact up to 2^53), not through the binary32 vec3 path, which
            // would round an index above 2^24 to a neighbouring vertex
            const JValue& idx = member(*e, "indices");
            if (idx.kind != JValue::Array) throw std::runtime_error("scene JSON: \"indices\" must be an array of 3-number arrays");
            const double nverts = (double)mi.mesh.positionBuffer.size();
            for (auto& tri : idx.arr) {
                if (tri->kind != JValue::Array || tri->arr.size() != 3) throw std::runtime_error("scene JSON: \"indices\" must be an array of 3-number arrays");
                for (auto& c : tri->arr) {
                    const double v = number(*c, "indices");
                    if (!(v >= 0 && v < nverts) || v != std::floor(v)) throw std::runtime_error("scene JSON: triangle index out of range");
                    mi.mesh.indexBuffer.push_back((uint32_t)v);
                }
            }
        }
        sc.meshes.push_back(std::move(mi));
    }
}

Scene load_scene_json(const std::string& text)
{
    Parser p(text);
    JPtr root = p.parse();
    Scene sc;
    if (root->kind != JValue::Object) throw std::runtime_error("scene JSON: expected an object");
    load_meshes(*root, sc);
    static const JValue no_spheres = [] { JValue v; v.kind = JValue::Array; return v; }();
    const JValue& spheres = (sc.meshes.empty() || root->obj.count("spheres")) ? member(*root, "spheres") : no_spheres;
    if (spheres.kind != JValue::Array) throw std::runtime_error("scene JSON: \"spheres\" must be an array");
    for (auto& e : spheres.arr) {
        const Refl_t refl = parse_refl(member(*e, "refl"));
        sc.spheres.emplace_back((float)number(member(*e, "radius"), "radius"), vec3(member(*e, "center"), "center"),
                                vec3(member(*e, "emission"), "emission"), vec3(member(*e, "color"), "color"), refl);
    }
    auto cit = root->obj.find("camera");
    if (cit != root->obj.end() && cit->second->kind == JValue::Object) {
        const JValue& c = *cit->second;
        sc.camera.present = true;
        if (c.obj.count("origin")) sc.camera.origin = vec3(member(c, "origin"), "origin");
        if (c.obj.count("direction")) sc.camera.direction = vec3(member(c, "direction"), "direction");
        if (c.obj.count("fov")) sc.camera.fov = number(member(c, "fov"), "fov");
        if (c.obj.count("push")) sc.camera.push = (float)number(member(c, "push"), "push");
    }
    return sc;
}

// render request of the viewer's queue, smallpt.cpp:909-916: {"action": "update_camera", "org": [x, y, z]}
bool parse_update_camera_request(const std::string& text, float3* org)
{
    Parser p(text);
    JPtr root = p.parse();
    if (root->kind != JValue::Object) throw std::runtime_error("render request: expected a JSON object");
    auto a = root->obj.find("action");
    if (a == root->obj.end() || a->second->kind != JValue::String || a->second->str != "update_camera") return false;
    *org = vec3(member(*root, "org"), "org");
    return true;
}

Scene load_scene_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open scene file " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return load_scene_json(ss.str());
}

std::string scene_to_json(const Scene& scene)
{
    std::ostringstream o;
    o.precision(9);   // 9 significant digits round-trip every binary32 value
    auto v3 = [&](float3 v) { o << "[" << v.x << ", " << v.y << ", " << v.z << "]"; };
    static const char* names[] = {"DIFF", "SPEC", "REFR"};
    o << "{\"camera\": {\"origin\": "; v3(scene.camera.origin);
    o << ", \"direction\": "; v3(scene.camera.direction);
    o.precision(17);
    o << ", \"fov\": " << scene.camera.fov;
    o.precision(9);
    o << ", \"push\": " << scene.camera.push << "}, \"spheres\": [";
    for (size_t i = 0; i < scene.spheres.size(); ++i) {
        const Sphere& s = scene.spheres[i];
        o << (i ? ", " : "") << "{\"radius\": " << s.radius << ", \"center\": "; v3(s.center);
        o << ", \"emission\": "; v3(s.material.emission);
        o << ", \"color\": "; v3(s.material.color);
        o << ", \"refl\": \"" << names[s.material.refl] << "\"}";
    }
    o << "]";
    if (!scene.meshes.empty()) {
        o << ", \"meshes\": [";
        for (size_t i = 0; i < scene.meshes.size(); ++i) {
            const MeshInstance& m = scene.meshes[i];
            o << (i ? ", " : "") << "{";
            if (m.generator == "sphere") {
                o << "\"sphere\": {\"center\": "; v3(m.center);
                o << ", \"radius\": " << m.radius << ", \"subdiv\": " << m.subdiv << "}";
            } else {
                auto list = [&](const char* key, const std::vector<float3>& v) {
                    o << "\"" << key << "\": [";
                    for (size_t k = 0; k < v.size(); ++k) { if (k) o << ", "; v3(v[k]); }
                    o << "]";
                };
                list("positions", m.mesh.positionBuffer); o << ", "; list("normals", m.mesh.normalBuffer);
                o << ", \"indices\": [";
                for (size_t k = 0; k + 2 < m.mesh.indexBuffer.size(); k += 3)
                    o << (k ? ", " : "") << "[" << m.mesh.indexBuffer[k] << ", " << m.mesh.indexBuffer[k + 1] << ", " << m.mesh.indexBuffer[k + 2] << "]";
                o << "]";
            }
            o << ", \"emission\": "; v3(m.material.emission);
            o << ", \"color\": "; v3(m.material.color);
            o << ", \"refl\": \"" << names[m.material.refl] << "\"}";
        }
        o << "]";
    }
    o << "}";
    return o.str();
}

}  // namespace spt_host
