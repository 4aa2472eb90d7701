// Host side of the MI355X path: everything Image::render does before and after the pixel loop.
// See portrayer.hpp / host_internal.hpp for the reference locations each piece mirrors.
#include <zlib.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

#include "host_internal.hpp"

namespace portrayer {
using namespace math;

// ------------------------------------------------------------------------------------------------
// math (vek 0.9.8 semantics; operation order documented in DESIGN.md and shared with the kernels)
// ------------------------------------------------------------------------------------------------
Mat4::Mat4() {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i][j] = i == j ? 1.0 : 0.0;
}
Mat4 Mat4::scaling_3d(Vec3 s) { Mat4 r; r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z; return r; }
Mat4 Mat4::translation_3d(Vec3 t) { Mat4 r; r.m[0][3] = t.x; r.m[1][3] = t.y; r.m[2][3] = t.z; return r; }
Mat4 Mat4::rotation_x(double a) {
    double c = std::cos(a), s = std::sin(a); Mat4 r;
    r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s; r.m[2][2] = c; return r;
}
Mat4 Mat4::rotation_y(double a) {
    double c = std::cos(a), s = std::sin(a); Mat4 r;
    r.m[0][0] = c; r.m[0][2] = s; r.m[2][0] = -s; r.m[2][2] = c; return r;
}
Mat4 Mat4::rotation_z(double a) {
    double c = std::cos(a), s = std::sin(a); Mat4 r;
    r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s; r.m[1][1] = c; return r;
}
Mat4 Mat4::look_at_rh(Vec3 eye, Vec3 target, Vec3 up) {
    Vec3 f = (target - eye).normalized();
    Vec3 s = f.cross(up).normalized();
    Vec3 u = s.cross(f);
    Mat4 v;
    v.m[0][0] = s.x; v.m[0][1] = s.y; v.m[0][2] = s.z; v.m[0][3] = -s.dot(eye);
    v.m[1][0] = u.x; v.m[1][1] = u.y; v.m[1][2] = u.z; v.m[1][3] = -u.dot(eye);
    v.m[2][0] = -f.x; v.m[2][1] = -f.y; v.m[2][2] = -f.z; v.m[2][3] = f.dot(eye);
    return v;
}
Mat4 Mat4::operator*(const Mat4& o) const {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = ((m[i][0] * o.m[0][j] + m[i][1] * o.m[1][j]) + m[i][2] * o.m[2][j]) + m[i][3] * o.m[3][j];
    return r;
}
Mat4 Mat4::transposed() const {
    Mat4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = m[j][i];
    return r;
}
Mat4 Mat4::inverted() const {  // adjugate / determinant through the 2x2 minors of rows {0,1} and {2,3}
    double s0 = m[0][0] * m[1][1] - m[1][0] * m[0][1];
    double s1 = m[0][0] * m[1][2] - m[1][0] * m[0][2];
    double s2 = m[0][0] * m[1][3] - m[1][0] * m[0][3];
    double s3 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
    double s4 = m[0][1] * m[1][3] - m[1][1] * m[0][3];
    double s5 = m[0][2] * m[1][3] - m[1][2] * m[0][3];
    double c5 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    double c4 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    double c3 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    double c2 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    double c1 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    double c0 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    double det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    double id = 1.0 / det;
    Mat4 r;
    r.m[0][0] = ((m[1][1] * c5 - m[1][2] * c4) + m[1][3] * c3) * id;
    r.m[0][1] = ((-m[0][1] * c5 + m[0][2] * c4) - m[0][3] * c3) * id;
    r.m[0][2] = ((m[3][1] * s5 - m[3][2] * s4) + m[3][3] * s3) * id;
    r.m[0][3] = ((-m[2][1] * s5 + m[2][2] * s4) - m[2][3] * s3) * id;
    r.m[1][0] = ((-m[1][0] * c5 + m[1][2] * c2) - m[1][3] * c1) * id;
    r.m[1][1] = ((m[0][0] * c5 - m[0][2] * c2) + m[0][3] * c1) * id;
    r.m[1][2] = ((-m[3][0] * s5 + m[3][2] * s2) - m[3][3] * s1) * id;
    r.m[1][3] = ((m[2][0] * s5 - m[2][2] * s2) + m[2][3] * s1) * id;
    r.m[2][0] = ((m[1][0] * c4 - m[1][1] * c2) + m[1][3] * c0) * id;
    r.m[2][1] = ((-m[0][0] * c4 + m[0][1] * c2) - m[0][3] * c0) * id;
    r.m[2][2] = ((m[3][0] * s4 - m[3][1] * s2) + m[3][3] * s0) * id;
    r.m[2][3] = ((-m[2][0] * s4 + m[2][1] * s2) - m[2][3] * s0) * id;
    r.m[3][0] = ((-m[1][0] * c3 + m[1][1] * c1) - m[1][2] * c0) * id;
    r.m[3][1] = ((m[0][0] * c3 - m[0][1] * c1) + m[0][2] * c0) * id;
    r.m[3][2] = ((-m[3][0] * s3 + m[3][1] * s1) - m[3][2] * s0) * id;
    r.m[3][3] = ((m[2][0] * s3 - m[2][1] * s1) + m[2][2] * s0) * id;
    return r;
}
Vec3 math::transformed_point(Vec3 v, const Mat4& t) {
    return {((t.m[0][0] * v.x + t.m[0][1] * v.y) + t.m[0][2] * v.z) + t.m[0][3],
            ((t.m[1][0] * v.x + t.m[1][1] * v.y) + t.m[1][2] * v.z) + t.m[1][3],
            ((t.m[2][0] * v.x + t.m[2][1] * v.y) + t.m[2][2] * v.z) + t.m[2][3]};
}
Vec3 math::transformed_direction(Vec3 v, const Mat4& t) {
    return {(t.m[0][0] * v.x + t.m[0][1] * v.y) + t.m[0][2] * v.z,
            (t.m[1][0] * v.x + t.m[1][1] * v.y) + t.m[1][2] * v.z,
            (t.m[2][0] * v.x + t.m[2][1] * v.y) + t.m[2][2] * v.z};
}

// ------------------------------------------------------------------------------------------------
// primitive
// ------------------------------------------------------------------------------------------------
namespace primitive {

Arc<MeshData> MeshData::create(std::vector<Vec3> positions, std::vector<std::array<uint32_t, 3>> triangles, std::vector<Vec3> normals,
                               std::vector<Uv> tex_coords) {
    if (positions.empty()) throw Panic("Meshes must have at least one vertex");  // mesh.rs:71
    if (!tex_coords.empty() && tex_coords.size() != positions.size())  // mesh.rs:77-79
        throw Panic("If meshes have texture coordinates, they must have enough for all vertices");
    for (const auto& t : triangles)
        for (uint32_t i : t)
            if (i >= positions.size()) throw Panic("index out of bounds: mesh triangle refers to a missing vertex");
    auto md = std::make_shared<MeshData>();
    Vec3 mn = positions[0], mx = positions[0];  // mesh.rs:72-75
    for (size_t i = 1; i < positions.size(); i++) { mn = Vec3::partial_min(mn, positions[i]); mx = Vec3::partial_max(mx, positions[i]); }
    md->positions_ = std::move(positions);
    md->triangles_ = std::move(triangles);
    md->normals_ = std::move(normals);
    md->tex_coords_ = std::move(tex_coords);
    md->min_ = mn; md->max_ = mx;
    return md;
}

Arc<MeshData> MeshData::load_obj(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("could not open OBJ file: " + path);
    std::vector<Vec3> pos, nrm;
    std::vector<Uv> tex;
    std::vector<Vec3> out_pos, out_nrm;
    std::vector<Uv> out_tex;
    std::vector<std::array<uint32_t, 3>> tris;
    std::map<std::tuple<long, long, long>, uint32_t> index_map;
    bool seen_faces = false;
    std::string line;
    auto parse3 = [](std::istringstream& ss) {
        std::string a, b, c;
        ss >> a >> b >> c;
        // tobj parses f32 (mesh.rs:36-53 widens to f64)
        return Vec3((double)std::strtof(a.c_str(), nullptr), (double)std::strtof(b.c_str(), nullptr), (double)std::strtof(c.c_str(), nullptr));
    };
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") pos.push_back(parse3(ss));
        else if (tag == "vn") nrm.push_back(parse3(ss));
        else if (tag == "vt") {
            std::string a, b;
            ss >> a >> b;
            tex.push_back(Uv{(double)std::strtof(a.c_str(), nullptr), (double)std::strtof(b.c_str(), nullptr)});
        }
        else if (tag == "o" || tag == "g") { if (seen_faces) break; }  // models[0] only (mesh.rs:60)
        else if (tag == "f") {
            seen_faces = true;
            std::vector<uint32_t> corner;
            std::string tok;
            while (ss >> tok) {
                long idx[3] = {0, 0, 0};
                int k = 0; size_t start = 0;
                for (size_t i = 0; i <= tok.size() && k < 3; i++)
                    if (i == tok.size() || tok[i] == '/') {
                        if (i > start) idx[k] = std::strtol(tok.substr(start, i - start).c_str(), nullptr, 10);
                        k++; start = i + 1;
                    }
                long v = idx[0] > 0 ? idx[0] - 1 : (long)pos.size() + idx[0];
                long vt = idx[1] > 0 ? idx[1] - 1 : (idx[1] < 0 ? (long)tex.size() + idx[1] : -1);
                long vn = idx[2] > 0 ? idx[2] - 1 : (idx[2] < 0 ? (long)nrm.size() + idx[2] : -1);
                if (v < 0 || (size_t)v >= pos.size()) throw std::runtime_error("OBJ face refers to a missing vertex: " + path);
                auto key = std::make_tuple(v, vt, vn);
                auto it = index_map.find(key);
                if (it == index_map.end()) {
                    it = index_map.emplace(key, (uint32_t)out_pos.size()).first;
                    out_pos.push_back(pos[(size_t)v]);
                    if (vn >= 0 && (size_t)vn < nrm.size()) out_nrm.push_back(nrm[(size_t)vn]);
                    if (vt >= 0 && (size_t)vt < tex.size()) out_tex.push_back(tex[(size_t)vt]);
                }
                corner.push_back(it->second);
            }
            for (size_t k = 1; k + 1 < corner.size(); k++) tris.push_back({corner[0], corner[k], corner[k + 1]});
        }
    }
    if (out_nrm.size() != out_pos.size()) out_nrm.clear();
    if (out_tex.size() != out_pos.size()) out_tex.clear();
    return create(std::move(out_pos), std::move(tris), std::move(out_nrm), std::move(out_tex));
}

Mesh::Mesh(Arc<MeshData> d, Shading s) : data(std::move(d)), shading(s) {
    if (!data) throw Panic("Mesh needs mesh data");
    if (shading == Shading::Smooth && data->positions().size() != data->normals().size())  // mesh.rs:135-138
        throw Panic("Meshes must have a vertex normal for each vertex if they are to be used with smooth shading");
}
KDMesh::KDMesh(const Arc<MeshData>& d, Shading s) : data(d), shading(s) {
    if (!data) throw Panic("KDMesh needs mesh data");
    if (shading == Shading::Smooth && data->positions().size() != data->normals().size())
        throw Panic("index out of bounds: smooth shading needs a vertex normal for each vertex");
}
}  // namespace primitive

// ------------------------------------------------------------------------------------------------
// texture
// ------------------------------------------------------------------------------------------------
namespace texture {
RgbImageBuffer RgbImageBuffer::open(const std::string& path) {
    RgbImageBuffer b;
    if (!detail::image_read(path, &b.width, &b.height, &b.rgb)) throw std::runtime_error("could not open texture image: " + path);
    return b;
}
RgbImageBuffer RgbImageBuffer::from_pixels(size_t width, size_t height, const uint8_t* rgb) {
    RgbImageBuffer b;
    b.width = width; b.height = height;
    b.rgb.assign(rgb, rgb + width * height * 3);
    return b;
}
}  // namespace texture

// ------------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------------
namespace scene {
SceneNode SceneNode::from(Geometry g) { SceneNode n; n.geometry_ = std::move(g); return n; }
SceneNode SceneNode::from(std::vector<Arc<SceneNode>> children) { SceneNode n; n.children_ = std::move(children); return n; }
SceneNode SceneNode::from(Arc<SceneNode> child) { SceneNode n; n.children_.push_back(std::move(child)); return n; }
SceneNode& SceneNode::with_child(Arc<SceneNode> c) { children_.push_back(std::move(c)); return *this; }
SceneNode& SceneNode::with_children(const std::vector<Arc<SceneNode>>& cs) { children_.insert(children_.end(), cs.begin(), cs.end()); return *this; }
SceneNode& SceneNode::scaled(Vec3 s) { set_transform(trans_.scaled_3d(s)); return *this; }
SceneNode& SceneNode::translated(Vec3 t) { set_transform(trans_.translated_3d(t)); return *this; }
SceneNode& SceneNode::rotated_xzy(Radians x, Radians y, Radians z) { return rotated_x(x).rotated_z(z).rotated_y(y); }
SceneNode& SceneNode::rotated_x(Radians a) { set_transform(trans_.rotated_x(a.get())); return *this; }
SceneNode& SceneNode::rotated_y(Radians a) { set_transform(trans_.rotated_y(a.get())); return *this; }
SceneNode& SceneNode::rotated_z(Radians a) { set_transform(trans_.rotated_z(a.get())); return *this; }
void SceneNode::set_transform(const Mat4& t) {
    trans_ = t;
    invtrans_ = t.inverted();
    normal_trans_ = invtrans_.transposed();
}
}  // namespace scene

void reporter::RenderProgress::report_finished_pixels(uint64_t pixels) {
    done_ += pixels;
    std::fprintf(stderr, "rendered %llu / %llu pixels (%.0f %%)\n", (unsigned long long)done_, (unsigned long long)total_,
                 total_ ? 100.0 * (double)done_ / (double)total_ : 100.0);
}

// ------------------------------------------------------------------------------------------------
// detail: bounding boxes, flattening, k-d build, camera
// ------------------------------------------------------------------------------------------------
namespace detail {

BoundingBox BoundingBox::create(Vec3 min, Vec3 max) {
    if (!(min.x <= max.x && min.y <= max.y && min.z <= max.z)) throw Panic("bounding box min must be less than max");  // bounding_box.rs:58
    BoundingBox b;
    b.min = min; b.max = max;
    Vec3 size = Vec3::partial_max(max - min, Vec3(EPSILON));
    Vec3 center = (min + max) / 2.0;
    Mat4 trans = Mat4::scaling_3d(size).translated_3d(center);
    b.invtrans = trans.inverted();
    return b;
}

BoundingBox operator*(const Mat4& m, const BoundingBox& rhs) {
    Vec3 mn(INFINITY), mx(-INFINITY);
    const double xs[2] = {rhs.min.x, rhs.max.x}, ys[2] = {rhs.min.y, rhs.max.y}, zs[2] = {rhs.min.z, rhs.max.z};
    for (double x : xs) for (double y : ys) for (double z : zs) {
        Vec3 v = transformed_point(Vec3(x, y, z), m);
        mn = Vec3::partial_min(mn, v);
        mx = Vec3::partial_max(mx, v);
    }
    return BoundingBox::create(mn, mx);
}

BoundingBox primitive_bounds(const primitive::Primitive& p) {
    using primitive::Primitive;
    switch (p.kind) {
    case Primitive::SphereK: return BoundingBox::create(Vec3(-1.0), Vec3(1.0));                      // sphere.rs:18-24
    case Primitive::PlaneK: return BoundingBox::create(Vec3(-0.5, 0.0, -0.5), Vec3(0.5, 0.0, 0.5));  // plane.rs:17-23
    case Primitive::TriangleK: {                                                                     // triangle.rs:29-36
        const auto& t = p.triangle;
        return BoundingBox::create(Vec3::partial_min(t.a, Vec3::partial_min(t.b, t.c)), Vec3::partial_max(t.a, Vec3::partial_max(t.b, t.c)));
    }
    case Primitive::MeshK: return BoundingBox::create(p.mesh->bounds_min(), p.mesh->bounds_max());    // mesh.rs:125-129
    case Primitive::KDMeshK: {  // kdmesh.rs:26-30: bounds of the tree = union of the triangle bounds
        const auto& pos = p.mesh->positions();
        const auto& tris = p.mesh->triangles();
        if (tris.empty()) return BoundingBox::create(Vec3::zero(), Vec3::zero());  // bounding_box.rs:26-28
        Vec3 mn(0.0), mx(0.0);
        for (size_t i = 0; i < tris.size(); i++) {
            Vec3 a = pos[tris[i][0]], b = pos[tris[i][1]], c = pos[tris[i][2]];
            Vec3 tmn = Vec3::partial_min(a, Vec3::partial_min(b, c)), tmx = Vec3::partial_max(a, Vec3::partial_max(b, c));
            if (i == 0) { mn = tmn; mx = tmx; } else { mn = Vec3::partial_min(mn, tmn); mx = Vec3::partial_max(mx, tmx); }
        }
        return BoundingBox::create(mn, mx);
    }
    default: return BoundingBox::create(Vec3(-0.5), Vec3(0.5));  // cube.rs:30-36, cylinder.rs:18-24, cone.rs:18-24
    }
}

FlatSceneNode::FlatSceneNode(scene::Geometry g, const Mat4& t) : geometry(std::move(g)), trans(t) {
    invtrans = trans.inverted();
    normal_trans = invtrans.transposed();
}

FlatScene FlatScene::from(const scene::HierScene& hier) {
    FlatScene out;
    if (!hier.root) throw Panic("scene has no root node");
    struct Pending {
        Mat4 parent;
        Arc<scene::SceneNode> node;
        std::vector<const scene::SceneNode*> chain;  // ancestors, root first
        std::vector<uint32_t> path;
    };
    std::deque<Pending> remaining;  // flat_scene.rs:24-26
    remaining.push_back(Pending{Mat4::identity(), hier.root, {}, {}});
    size_t visited = 0;
    while (!remaining.empty()) {
        Pending cur = std::move(remaining.front());
        remaining.pop_front();
        if (++visited > (size_t(1) << 26)) throw Panic("scene graph is not a tree (cycle?)");  // the reference would never terminate
        Mat4 total = cur.parent * cur.node->trans();
        cur.chain.push_back(cur.node.get());
        if (cur.node->geometry()) {
            out.root.emplace_back(*cur.node->geometry(), total);
            out.root.back().chain = cur.chain;
            out.root.back().path = cur.path;
        }
        uint32_t k = 0;
        for (const auto& child : cur.node->children()) {
            std::vector<uint32_t> p = cur.path;
            p.push_back(k++);
            remaining.push_back(Pending{total, child, cur.chain, std::move(p)});
        }
    }
    out.lights = hier.lights;
    out.ambient = hier.ambient;
    return out;
}

// The hierarchy as PT_TRAVERSE_HIER needs it (scene.rs:80-120): every SceneNode on a path gets an index and its OWN
// three matrices; a flattened node's chain names them root first; equal hits go to whoever comes first depth-first, a
// node before its children - which is the lexicographic order of the child-index paths (a prefix sorts first).
GraphPacking pack_graph(const FlatScene& flat) {
    GraphPacking g;
    std::map<const scene::SceneNode*, uint32_t> graph_id;
    auto append = [](std::vector<double>& dst, const Mat4& m) { const double* p = &m.m[0][0]; dst.insert(dst.end(), p, p + 16); };
    g.chain_off.push_back(0);
    for (const FlatSceneNode& fn : flat.root) {
        for (const scene::SceneNode* sn : fn.chain) {
            auto it = graph_id.find(sn);
            if (it == graph_id.end()) {
                it = graph_id.emplace(sn, (uint32_t)graph_id.size()).first;
                append(g.trans, sn->trans()); append(g.invtrans, sn->inverse_trans()); append(g.normal_trans, sn->normal_trans());
            }
            g.chain.push_back(it->second);
        }
        g.chain_off.push_back((uint32_t)g.chain.size());
    }
    const size_t n = flat.root.size();
    std::vector<uint32_t> order(n);
    g.dfs_rank.resize(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return flat.root[a].path < flat.root[b].path; });
    for (uint32_t r = 0; r < n; r++) g.dfs_rank[order[r]] = r;
    g.n_graph_nodes = (uint32_t)graph_id.size();
    return g;
}

namespace {
struct InfinitePlane {  // infinite_plane.rs:16-35
    Vec3 normal, point;
    bool front(Vec3 p) const { return (p - point).dot(normal) >= 0.0; }
};
enum class Part { Front, Back, Shared };
Part partition_node(const BoundingBox& b, const InfinitePlane& sep) {  // leaf.rs:115-130
    bool fmin = sep.front(b.min), fmax = sep.front(b.max);
    if (fmin && fmax) return Part::Front;
    if (!fmin && !fmax) return Part::Back;
    return Part::Shared;
}
void list_bounds(const std::vector<BoundingBox>& all, const std::vector<uint32_t>& ids, Vec3* mn, Vec3* mx) {  // bounding_box.rs:24-37
    if (ids.empty()) { *mn = Vec3::zero(); *mx = Vec3::zero(); return; }
    Vec3 a = all[ids[0]].min, b = all[ids[0]].max;
    for (size_t i = 1; i < ids.size(); i++) { a = Vec3::partial_min(a, all[ids[i]].min); b = Vec3::partial_max(b, all[ids[i]].max); }
    *mn = a; *mx = b;
}
struct KdBuilder {
    const std::vector<BoundingBox>& all;
    PartitionConfig conf;
    KdTree tree;
    int deepest = 0;
    int32_t add_leaf(const std::vector<uint32_t>& ids) {
        int32_t me = (int32_t)tree.axis.size();
        tree.axis.push_back(-1); tree.plane.push_back(0.0); tree.front.push_back(-1); tree.back.push_back(-1);
        tree.first.push_back((int32_t)tree.items.size()); tree.count.push_back((int32_t)ids.size());
        for (uint32_t i : ids) tree.items.push_back((int32_t)i);
        return me;
    }
    int32_t partitioned(std::vector<uint32_t> ids, Vec3 bmin, Vec3 bmax, Vec3 axis, size_t max_depth, int level) {  // leaf.rs:89-231
        if (max_depth == 0 || ids.size() <= conf.target_max_nodes) return add_leaf(ids);
        deepest = std::max(deepest, level + 1);
        Vec3 min_axis = axis * bmin, max_axis = axis * bmax;
        InfinitePlane sep{axis, min_axis + (max_axis - min_axis) / 2.0};
        Vec3 plane_min = min_axis, plane_max = max_axis;
        for (size_t tries = 0; tries < conf.max_tries; tries++) {
            long front = 0, back = 0, shared = 0;
            for (uint32_t i : ids) {
                Part p = partition_node(all[i], sep);
                if (p == Part::Front) front++; else if (p == Part::Back) back++; else shared++;
            }
            long merit = std::labs(front - back) + shared;
            if (merit <= conf.target_max_merit) break;
            if (front > back) {
                plane_min = sep.point;
                sep.point = sep.point + (plane_max - sep.point) / 2.0;
            } else {
                plane_max = sep.point;
                sep.point = plane_min + (sep.point - plane_min) / 2.0;
            }
        }
        std::vector<uint32_t> fi, bi;
        for (uint32_t i : ids) {
            Part p = partition_node(all[i], sep);
            if (p == Part::Front) fi.push_back(i);
            else if (p == Part::Back) bi.push_back(i);
            else { fi.push_back(i); bi.push_back(i); }
        }
        ids.clear(); ids.shrink_to_fit();
        Vec3 next(axis.z, axis.x, axis.y);  // leaf.rs:97-103
        int ax = axis.x != 0.0 ? 0 : (axis.y != 0.0 ? 1 : 2);
        int32_t me = (int32_t)tree.axis.size();
        tree.axis.push_back(ax);
        tree.plane.push_back(ax == 0 ? sep.point.x : (ax == 1 ? sep.point.y : sep.point.z));
        tree.front.push_back(-1); tree.back.push_back(-1); tree.first.push_back(0); tree.count.push_back(0);
        Vec3 fmn, fmx, kmn, kmx;
        list_bounds(all, fi, &fmn, &fmx);
        list_bounds(all, bi, &kmn, &kmx);
        int32_t f = partitioned(std::move(fi), fmn, fmx, next, max_depth - 1, level + 1);
        int32_t b = partitioned(std::move(bi), kmn, kmx, next, max_depth - 1, level + 1);
        tree.front[me] = f; tree.back[me] = b;
        return me;
    }
};
}  // namespace

KdTree kd_partition(const std::vector<BoundingBox>& bounds, size_t max_depth, PartitionConfig conf) {
    KdBuilder b{bounds, conf, KdTree(), 0};
    std::vector<uint32_t> ids(bounds.size());
    for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint32_t)i;
    Vec3 mn, mx;
    list_bounds(bounds, ids, &mn, &mx);
    b.tree.root_min = mn; b.tree.root_max = mx;
    b.partitioned(std::move(ids), mn, mx, Vec3::unit_x(), max_depth, 0);
    b.tree.max_depth = b.deepest;
    return std::move(b.tree);
}

KdTree kd_scene_tree(const FlatScene& flat, size_t max_depth) {  // kdscene.rs:19-43
    std::vector<BoundingBox> bounds;
    bounds.reserve(flat.root.size());
    for (const auto& n : flat.root) bounds.push_back(n.bounds());
    return kd_partition(bounds, max_depth, PartitionConfig{3, 3, 10});
}

Camera::Camera(const camera::CameraSettings& cam, double w, double h) {  // camera.rs:34-45
    eye = cam.eye;
    view_to_world = Mat4::look_at_rh(cam.eye, cam.center, cam.up).inverted();
    fov_factor = std::tan(cam.fovy.get() / 2.0);
    aspect_ratio = w / h;
    width = w; height = h;
}
pt_camera Camera::to_abi() const {
    pt_camera c;
    c.eye[0] = eye.x; c.eye[1] = eye.y; c.eye[2] = eye.z;
    std::memcpy(c.view_to_world, view_to_world.m, sizeof c.view_to_world);
    c.fov_factor = fov_factor; c.aspect_ratio = aspect_ratio; c.width = width; c.height = height;
    return c;
}

// ------------------------------------------------------------------------------------------------
// Renderer: pack the flattened scene into the C ABI's arrays and upload it
// ------------------------------------------------------------------------------------------------
static void check(pt_context* ctx, int rc, const char* what) {
    if (rc == PT_OK) return;
    std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + (ctx ? pt_last_error(ctx) : "no context");
    if (rc == PT_ERR_SLICE || rc == PT_ERR_SCENE) throw Panic(msg);
    throw std::runtime_error(msg);
}

// PORTRAYER_GPUS = N | all: tile-partition every render over N GPUs of the node (pt_node_*, one RCCL gather);
// PORTRAYER_DEVICES = "0,1,..." names them (ranks may share a device: tests on a 1-GPU box). Default: one GPU.
static std::vector<int> node_devices(int device) {
    std::vector<int> devs;
    if (const char* e = std::getenv("PORTRAYER_DEVICES")) {
        std::stringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) devs.push_back(std::atoi(tok.c_str()));
        return devs;
    }
    const char* g = std::getenv("PORTRAYER_GPUS");
    int n = !g ? 1 : (std::string(g) == "all" ? pt_device_count() : std::atoi(g));
    if (n <= 1) return {device};
    for (int i = 0; i < n; i++) devs.push_back(i);
    return devs;
}

static double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

Renderer::Renderer(const scene::HierScene& hier, render::Traversal traversal, int kd_depth, int device) {
    auto t_flat = std::chrono::steady_clock::now();
    flat_ = FlatScene::from(hier);
    prep_.flatten = ms_since(t_flat);
    auto t_pack = std::chrono::steady_clock::now();
    const size_t n = flat_.root.size();
    std::vector<double> trans(16 * n), inv(16 * n), nrm(16 * n);
    std::vector<int32_t> type(n), data(n), flags(n), mat(n);
    std::map<const primitive::MeshData*, int32_t> mesh_id;
    std::vector<const primitive::MeshData*> meshes;
    std::map<const material::Material*, int32_t> mat_id;
    std::vector<double> materials, tri_v, tri_n, tri_uv, uv_trans;
    std::vector<uint8_t> tri_has_uv;
    std::vector<int32_t> mat_tex, mat_nmap;
    std::map<const texture::RgbImageBuffer*, int32_t> tex_id;
    std::vector<const texture::RgbImageBuffer*> textures;
    auto texture_index = [&](const texture::RgbImageBuffer* b) -> int32_t {
        auto it = tex_id.find(b);
        if (it == tex_id.end()) { it = tex_id.emplace(b, (int32_t)textures.size()).first; textures.push_back(b); }
        return it->second;
    };
    bool any_tri_normals = false, any_tri_uv = false;
    for (size_t i = 0; i < n; i++) {
        const FlatSceneNode& fn = flat_.root[i];
        std::memcpy(&trans[16 * i], fn.trans.m, 128);
        std::memcpy(&inv[16 * i], fn.invtrans.m, 128);
        std::memcpy(&nrm[16 * i], fn.normal_trans.m, 128);
        const auto& p = fn.geometry.primitive;
        type[i] = (int32_t)p.kind; data[i] = 0; flags[i] = 0;
        if (p.kind == primitive::Primitive::MeshK || p.kind == primitive::Primitive::KDMeshK) {
            auto it = mesh_id.find(p.mesh.get());
            if (it == mesh_id.end()) { it = mesh_id.emplace(p.mesh.get(), (int32_t)meshes.size()).first; meshes.push_back(p.mesh.get()); }
            data[i] = it->second;
            flags[i] = p.shading == primitive::Shading::Smooth ? 1 : 0;
        } else if (p.kind == primitive::Primitive::TriangleK) {
            data[i] = (int32_t)(tri_v.size() / 9);
            const auto& t = p.triangle;
            const Vec3 vs[3] = {t.a, t.b, t.c};
            for (const Vec3& v : vs) { tri_v.push_back(v.x); tri_v.push_back(v.y); tri_v.push_back(v.z); }
            for (int k = 0; k < 3; k++) {
                Vec3 nn = t.normals ? (*t.normals)[k] : Vec3::zero();
                tri_n.push_back(nn.x); tri_n.push_back(nn.y); tri_n.push_back(nn.z);
            }
            if (t.normals) { flags[i] = 1; any_tri_normals = true; }
            for (int k = 0; k < 3; k++) { tri_uv.push_back(t.tex_coords ? (*t.tex_coords)[k].u : 0.0); tri_uv.push_back(t.tex_coords ? (*t.tex_coords)[k].v : 0.0); }
            tri_has_uv.push_back(t.tex_coords ? 1 : 0);
            if (t.tex_coords) any_tri_uv = true;
        }
        const material::Material* m = fn.geometry.material.get();
        if (!m) throw Panic("geometry without a material");
        auto mit = mat_id.find(m);
        if (mit == mat_id.end()) {
            mit = mat_id.emplace(m, (int32_t)(materials.size() / 10)).first;
            const double row[10] = {m->diffuse.r, m->diffuse.g, m->diffuse.b, m->specular.r, m->specular.g, m->specular.b,
                                    m->shininess, m->reflectivity, m->glossy_side_length, m->refraction_index};
            materials.insert(materials.end(), row, row + 10);
            mat_tex.push_back(m->texture ? texture_index(&m->texture->image.buffer) : -1);
            mat_nmap.push_back(m->normals ? texture_index(&m->normals->buffer) : -1);
            for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) uv_trans.push_back(m->uv_trans.m[r][k]);
        }
        mat[i] = mit->second;
    }
    std::vector<uint64_t> vert_off{0}, tri_off{0};
    std::vector<double> positions, normals, mesh_bounds_inv, texcoords;
    std::vector<uint8_t> has_normals, has_texcoords;
    std::vector<uint32_t> indices;
    for (const primitive::MeshData* m : meshes) {
        for (const Vec3& p : m->positions()) { positions.push_back(p.x); positions.push_back(p.y); positions.push_back(p.z); }
        bool hn = m->normals().size() == m->positions().size();
        for (size_t v = 0; v < m->positions().size(); v++) {
            Vec3 nn = hn ? m->normals()[v] : Vec3::zero();
            normals.push_back(nn.x); normals.push_back(nn.y); normals.push_back(nn.z);
        }
        has_normals.push_back(hn ? 1 : 0);
        bool ht = m->tex_coords().size() == m->positions().size();
        for (size_t v = 0; v < m->positions().size(); v++) { texcoords.push_back(ht ? m->tex_coords()[v].u : 0.0); texcoords.push_back(ht ? m->tex_coords()[v].v : 0.0); }
        has_texcoords.push_back(ht ? 1 : 0);
        for (const auto& t : m->triangles()) { indices.push_back(t[0]); indices.push_back(t[1]); indices.push_back(t[2]); }
        vert_off.push_back(vert_off.back() + m->positions().size());
        tri_off.push_back(tri_off.back() + m->triangles().size());
        BoundingBox bb = BoundingBox::create(m->bounds_min(), m->bounds_max());  // mesh.rs:82
        const double* im = &bb.invtrans.m[0][0];
        mesh_bounds_inv.insert(mesh_bounds_inv.end(), im, im + 16);
    }
    // KDMesh::new (kdmesh.rs:37-58): a k-d tree over the triangles of every mesh a KDMesh primitive uses
    std::vector<int32_t> mesh_kd_root(meshes.size(), -1), mesh_kd_depth(meshes.size(), 0);
    std::vector<double> mesh_kd_bounds(6 * meshes.size(), 0.0), mesh_kd_bounds_inv(16 * meshes.size(), 0.0);
    std::vector<int32_t> kdm_axis, kdm_front, kdm_back, kdm_first, kdm_count, kdm_items;
    std::vector<double> kdm_plane;
    {
        long kd_mesh_depth = 10;  // env KD_MESH_DEPTH, kdmesh.rs:51-53
        if (const char* e = std::getenv("KD_MESH_DEPTH")) { char* end = nullptr; long v = std::strtol(e, &end, 10); if (end && *end == 0 && v >= 0) kd_mesh_depth = v; }
        // PORTRAYER_KDMESH_AS_MESH=1: walk KDMesh primitives like Mesh. The reference's KDMesh classifies the sides of a split
        // with a ray segment of length extent() = the SQUARED diagonal of the mesh's bounds (bounding_box.rs:95-99, "HACK"), so
        // a KDMesh smaller than the distance to the camera loses most of its triangles (the script robot-alarm-clock.rs says
        // "KDMesh doesn't work for this for some reason" about exactly such parts). Reproduced by default - this switch is for
        // the picture the scene's author meant.
        const bool kdmesh_as_mesh = std::getenv("PORTRAYER_KDMESH_AS_MESH") != nullptr;
        for (size_t i = 0; i < n && !kdmesh_as_mesh; i++) {
            const auto& p = flat_.root[i].geometry.primitive;
            if (p.kind != primitive::Primitive::KDMeshK) continue;
            size_t mi = (size_t)mesh_id[p.mesh.get()];
            if (mesh_kd_root[mi] >= 0) continue;
            const auto& pos = p.mesh->positions();
            std::vector<BoundingBox> tb;
            tb.reserve(p.mesh->triangles().size());
            for (const auto& t : p.mesh->triangles()) {  // triangle.rs:29-36
                Vec3 a = pos[t[0]], b = pos[t[1]], c = pos[t[2]];
                tb.push_back(BoundingBox::create(Vec3::partial_min(a, Vec3::partial_min(b, c)), Vec3::partial_max(a, Vec3::partial_max(b, c))));
            }
            KdTree t = kd_partition(tb, (size_t)kd_mesh_depth, PartitionConfig{3, 3, 10});
            int32_t base = (int32_t)kdm_axis.size(), ibase = (int32_t)kdm_items.size();
            for (size_t k = 0; k < t.axis.size(); k++) {
                kdm_axis.push_back(t.axis[k]); kdm_plane.push_back(t.plane[k]);
                kdm_front.push_back(t.axis[k] >= 0 ? t.front[k] + base : -1); kdm_back.push_back(t.axis[k] >= 0 ? t.back[k] + base : -1);
                kdm_first.push_back(t.axis[k] < 0 ? t.first[k] + ibase : 0); kdm_count.push_back(t.count[k]);
            }
            kdm_items.insert(kdm_items.end(), t.items.begin(), t.items.end());
            mesh_kd_root[mi] = base; mesh_kd_depth[mi] = t.max_depth;
            const double b6[6] = {t.root_min.x, t.root_min.y, t.root_min.z, t.root_max.x, t.root_max.y, t.root_max.z};
            std::memcpy(&mesh_kd_bounds[6 * mi], b6, sizeof b6);
            BoundingBox rb = BoundingBox::create(t.root_min, t.root_max);  // kdmesh.rs:26-30 bounds(): the root node's bounds
            std::memcpy(&mesh_kd_bounds_inv[16 * mi], rb.invtrans.m, 128);
        }
    }
    std::vector<double> lights;
    for (const auto& l : flat_.lights) {
        const double row[15] = {l.position.x, l.position.y, l.position.z, l.color.r, l.color.g, l.color.b, l.falloff.c0, l.falloff.c1, l.falloff.c2,
                                l.area.a.x, l.area.a.y, l.area.a.z, l.area.b.x, l.area.b.y, l.area.b.z};
        lights.insert(lights.end(), row, row + 15);
    }
    pt_scene s;
    std::memset(&s, 0, sizeof s);
    s.n_nodes = (uint32_t)n;
    s.trans = trans.data(); s.invtrans = inv.data(); s.normal_trans = nrm.data();
    s.prim_type = type.data(); s.prim_data = data.data(); s.prim_flags = flags.data(); s.material = mat.data();
    s.n_meshes = (uint32_t)meshes.size();
    s.mesh_vert_off = vert_off.data(); s.mesh_tri_off = tri_off.data();
    s.mesh_positions = positions.data(); s.mesh_normals = normals.data(); s.mesh_has_normals = has_normals.data();
    s.mesh_indices = indices.data(); s.mesh_bounds_invtrans = mesh_bounds_inv.data();
    s.n_triangles = (uint32_t)(tri_v.size() / 9);
    s.tri_vertices = tri_v.data(); s.tri_normals = any_tri_normals ? tri_n.data() : nullptr;
    s.n_materials = (uint32_t)(materials.size() / 10); s.materials = materials.data();
    s.n_lights = (uint32_t)flat_.lights.size(); s.lights = lights.data();
    s.ambient[0] = flat_.ambient.r; s.ambient[1] = flat_.ambient.g; s.ambient[2] = flat_.ambient.b;
    if (!kdm_axis.empty()) {
        s.mesh_kd_root = mesh_kd_root.data(); s.mesh_kd_depth = mesh_kd_depth.data();
        s.mesh_kd_bounds = mesh_kd_bounds.data(); s.mesh_kd_bounds_invtrans = mesh_kd_bounds_inv.data();
        s.n_kdm_nodes = (uint32_t)kdm_axis.size();
        s.kdm_axis = kdm_axis.data(); s.kdm_plane = kdm_plane.data(); s.kdm_front = kdm_front.data(); s.kdm_back = kdm_back.data();
        s.kdm_first = kdm_first.data(); s.kdm_count = kdm_count.data();
        s.n_kdm_items = (uint32_t)kdm_items.size(); s.kdm_items = kdm_items.data();
    }
    std::vector<uint32_t> tex_size;
    std::vector<uint64_t> tex_off;
    std::vector<uint8_t> tex_rgb;
    if (!textures.empty()) {
        for (const texture::RgbImageBuffer* b : textures) {
            tex_size.push_back((uint32_t)b->width); tex_size.push_back((uint32_t)b->height);
            tex_off.push_back(tex_rgb.size());
            tex_rgb.insert(tex_rgb.end(), b->rgb.begin(), b->rgb.end());
        }
        s.mesh_texcoords = texcoords.data(); s.mesh_has_texcoords = has_texcoords.data();
        s.tri_texcoords = any_tri_uv ? tri_uv.data() : nullptr; s.tri_has_texcoords = any_tri_uv ? tri_has_uv.data() : nullptr;
        s.material_texture = mat_tex.data(); s.material_normal_map = mat_nmap.data(); s.material_uv_trans = uv_trans.data();
        s.n_textures = (uint32_t)textures.size(); s.texture_size = tex_size.data(); s.texture_offset = tex_off.data(); s.texture_rgb = tex_rgb.data();
    }

    const bool verbose = std::getenv("PORTRAYER_VERBOSE") != nullptr;
    prep_.pack = ms_since(t_pack);
    auto t_ctx = std::chrono::steady_clock::now();
    const std::vector<int> devs = node_devices(device);
    int rc;
    if (devs.size() > 1) {
        rc = pt_node_create((int)devs.size(), devs.data(), &node_);
        if (rc == PT_OK) ctx_ = pt_node_context(node_, 0);
    } else {
        rc = pt_context_create(devs.empty() ? device : devs[0], &ctx_);
    }
    prep_.context = ms_since(t_ctx);
    if (rc != PT_OK) throw std::runtime_error("pt_context_create failed (" + std::to_string(rc) + "): no usable MI355X; this path has no CPU fallback");
    auto upload = [&](int mode, const pt_kdtree* kd) {
        auto t_up = std::chrono::steady_clock::now();
        if (node_) {
            int urc = pt_node_scene_upload(node_, &s, mode, kd);
            if (urc != PT_OK) {
                std::string msg = std::string("pt_node_scene_upload failed (") + std::to_string(urc) + "): " + pt_node_last_error(node_);
                if (urc == PT_ERR_SLICE || urc == PT_ERR_SCENE) throw Panic(msg);
                throw std::runtime_error(msg);
            }
        } else {
            check(ctx_, pt_scene_upload(ctx_, &s, mode, kd), "pt_scene_upload");
        }
        prep_.upload = ms_since(t_up);
    };
    try {
        if (traversal == render::Traversal::KdTree) {
            auto t_kd = std::chrono::steady_clock::now();
            KdTree t = kd_scene_tree(flat_, kd_depth < 0 ? 10 : (size_t)kd_depth);
            prep_.kd_build = ms_since(t_kd);
            pt_kdtree kd;
            std::memset(&kd, 0, sizeof kd);
            kd.n_nodes = (uint32_t)t.axis.size();
            kd.axis = t.axis.data(); kd.plane = t.plane.data(); kd.front = t.front.data(); kd.back = t.back.data();
            kd.first = t.first.data(); kd.count = t.count.data();
            kd.n_items = (uint32_t)t.items.size(); kd.leaf_items = t.items.data();
            kd.root_min[0] = t.root_min.x; kd.root_min[1] = t.root_min.y; kd.root_min[2] = t.root_min.z;
            kd.root_max[0] = t.root_max.x; kd.root_max[1] = t.root_max.y; kd.root_max[2] = t.root_max.z;
            kd.max_depth = t.max_depth;
            upload(PT_TRAVERSE_KD, &kd);
        } else if (traversal == render::Traversal::Hier) {
            // scene.rs:80-120: the hierarchy itself. Every SceneNode on a path gets an index; a flattened node's chain
            // names them root first; equal hits go to whoever comes first depth-first, a node before its children -
            // which is the lexicographic order of the child-index paths (a prefix sorts first).
            GraphPacking gp = pack_graph(flat_);
            std::vector<double>&g_trans = gp.trans, &g_inv = gp.invtrans, &g_nrm = gp.normal_trans;
            std::vector<uint32_t>&chain_off = gp.chain_off, &chain = gp.chain, &rank = gp.dfs_rank;
            s.n_graph_nodes = gp.n_graph_nodes;
            s.graph_trans = g_trans.data(); s.graph_invtrans = g_inv.data(); s.graph_normal_trans = g_nrm.data();
            s.node_chain_off = chain_off.data(); s.node_chain = chain.data(); s.node_dfs_rank = rank.data();
            upload(PT_TRAVERSE_HIER, nullptr);
        } else {
            upload(PT_TRAVERSE_FLAT, nullptr);
        }
    } catch (...) {
        if (node_) pt_node_destroy(node_); else pt_context_destroy(ctx_);
        node_ = nullptr; ctx_ = nullptr;
        throw;
    }
    if (verbose)
        std::fprintf(stderr, "[Renderer] flatten %.2f ms, pack %.2f ms, context %.2f ms, k-d build %.2f ms, upload (incl. device trees) %.2f ms\n", prep_.flatten,
                     prep_.pack, prep_.context, prep_.kd_build, prep_.upload);
}

Renderer::~Renderer() {
    if (node_) pt_node_destroy(node_);
    else if (ctx_) pt_context_destroy(ctx_);
}

void Renderer::render(const camera::CameraSettings& cam, uint32_t width, uint32_t height, const double* background, bool background_rows,
                      pt_rect slice, uint32_t samples, uint64_t seed, int sample_mode, bool collect_stats, uint8_t* rgb, double* linear,
                      pt_stats* stats) {
    Camera c(cam, (double)width, (double)height);
    pt_camera pc = c.to_abi();
    pt_render_params p;
    std::memset(&p, 0, sizeof p);
    p.width = width; p.height = height; p.slice = slice; p.samples = samples; p.seed = seed; p.sample_mode = sample_mode;
    p.background_rows = background_rows ? 1 : 0; p.tile_rank = 0; p.tile_ranks = 1; p.collect_stats = collect_stats ? 1 : 0;
    if (node_) {
        if (linear) throw std::runtime_error("the linear (pre-gamma) output is a single-GPU debugging aid: unset PORTRAYER_GPUS / PORTRAYER_DEVICES");
        int rc = pt_node_render(node_, &pc, background, &p, rgb, stats);
        if (rc != PT_OK) {
            std::string msg = std::string("pt_node_render failed (") + std::to_string(rc) + "): " + pt_node_last_error(node_);
            if (rc == PT_ERR_SLICE || rc == PT_ERR_SCENE) throw Panic(msg);
            throw std::runtime_error(msg);
        }
        return;
    }
    check(ctx_, pt_render(ctx_, &pc, background, &p, rgb, linear, stats), "pt_render");
}

// ------------------------------------------------------------------------------------------------
// PNG (8-bit RGB / RGBA / grey, non-interlaced)
// ------------------------------------------------------------------------------------------------
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool image_read(const std::string& path, size_t* width, size_t* height, std::vector<uint8_t>* rgb) {
    unsigned char sig[2] = {0, 0};
    {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        in.read(reinterpret_cast<char*>(sig), 2);
    }
    return (sig[0] == 0xFF && sig[1] == 0xD8) ? jpeg_read(path, width, height, rgb) : png_read(path, width, height, rgb);
}

bool png_read(const std::string& path, size_t* width, size_t* height, std::vector<uint8_t>* rgb) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path);
    size_t pos = 8, w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (pos + 12 <= f.size()) {
        uint32_t len = be32(&f[pos]);
        std::string tag((const char*)&f[pos + 4], 4);
        if (pos + 12 + len > f.size()) throw std::runtime_error("truncated PNG: " + path);
        const uint8_t* d = &f[pos + 8];
        if (tag == "IHDR") { w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (tag == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (tag == "IEND") break;
        pos += 12 + len;
    }
    int channels = ctype == 2 ? 3 : (ctype == 6 ? 4 : (ctype == 0 ? 1 : (ctype == 4 ? 2 : 0)));
    if (depth != 8 || channels == 0 || interlace != 0) throw std::runtime_error("unsupported PNG format: " + path);
    size_t stride = w * channels;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), idat.size()) != Z_OK || out_len != raw.size()) throw std::runtime_error("corrupt PNG data: " + path);
    std::vector<uint8_t> img(stride * h);
    for (size_t y = 0; y < h; y++) {
        uint8_t ft = raw[y * (stride + 1)];
        const uint8_t* src = &raw[y * (stride + 1) + 1];
        uint8_t* dst = &img[y * stride];
        const uint8_t* up = y ? &img[(y - 1) * stride] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)channels ? dst[x - channels] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)channels) ? up[x - channels] : 0;
            int v = src[x];
            switch (ft) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: break;
            }
            dst[x] = (uint8_t)v;
        }
    }
    rgb->resize(w * h * 3);
    for (size_t i = 0; i < w * h; i++)
        for (int k = 0; k < 3; k++) (*rgb)[3 * i + k] = channels >= 3 ? img[i * channels + k] : img[i * channels];
    *width = w; *height = h;
    return true;
}

void png_write(const std::string& path, size_t width, size_t height, const std::vector<uint8_t>& rgb) {
    std::vector<uint8_t> raw((width * 3 + 1) * height);
    for (size_t y = 0; y < height; y++) {
        raw[y * (width * 3 + 1)] = 0;
        std::memcpy(&raw[y * (width * 3 + 1) + 1], &rgb[y * width * 3], width * 3);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) throw std::runtime_error("PNG compression failed");
    comp.resize(clen);
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error("could not write " + path);
    auto put32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); };
    auto chunk = [&](const char* tag, const std::vector<uint8_t>& data) {
        std::vector<uint8_t> c;
        put32(c, (uint32_t)data.size());
        c.insert(c.end(), tag, tag + 4);
        c.insert(c.end(), data.begin(), data.end());
        uint32_t crc = (uint32_t)crc32(0L, c.data() + 4, (uInt)(c.size() - 4));
        put32(c, crc);
        out.write((const char*)c.data(), (std::streamsize)c.size());
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.write((const char*)sig, 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr);
    chunk("IDAT", comp);
    chunk("IEND", {});
}

}  // namespace detail

// ------------------------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------------------------
namespace render {
namespace {
Traversal g_traversal = Traversal::Hier;
bool g_traversal_set = false;
long env_long(const char* name, long fallback) {
    const char* v = std::getenv(name);
    if (!v || !*v) return fallback;
    char* end = nullptr;
    long x = std::strtol(v, &end, 10);
    return (end && *end == 0) ? x : fallback;
}
}  // namespace

void set_traversal(Traversal t) { g_traversal = t; g_traversal_set = true; }
Traversal traversal() {
    if (g_traversal_set) return g_traversal;
    const char* v = std::getenv("PORTRAYER_TRAVERSAL");
    if (v && (std::string(v) == "kdtree" || std::string(v) == "kd")) return Traversal::KdTree;
    if (v && (std::string(v) == "flat" || std::string(v) == "flat_scene")) return Traversal::Flat;
    // Default = the crate built with NO features: SceneNode::ray_cast on the hierarchy (scene.rs:80-120). FLAT and KD
    // correspond to `--features flat_scene` / `--features kdtree` and are faster (big-scene: 14.2 vs 11.8 Gray/s), but
    // FLAT is not image-equivalent to the default where a refractive primitive sits under a transformed group
    // (water-glass: 1.9 % of the pixels), and a drop-in must not change the picture silently.
    return Traversal::Hier;
}

ImageSliceMut::ImageSliceMut(Image& image, std::pair<size_t, size_t> tl, std::pair<size_t, size_t> br) : image_(image), top_left_(tl), bottom_right_(br) {
    size_t w = image.width(), h = image.height();
    if (tl.first >= w || tl.second >= h || br.first >= w || br.second >= h) {  // render.rs:79-90
        std::ostringstream ss;
        ss << "The positions {x: " << tl.first << ", y: " << tl.second << "} and/or {x: " << br.first << ", y: " << br.second
           << "} are not within an image with width = " << w << " and height = " << h;
        throw Panic(ss.str());
    }
}

uint64_t ImageSliceMut::total_pixels() const { return (uint64_t)image_.width() * image_.height(); }

void ImageSliceMut::render_impl(const scene::HierScene& scene, camera::CameraSettings cam, const Background& background, reporter::Reporter& rep) {
    const size_t w = image_.width(), h = image_.height();
    long samples = env_long("SAMPLES", 100);  // render.rs:107-113
    if (samples <= 0) samples = 100;
    long kd_depth = env_long("KD_DEPTH", 10);  // kdscene.rs:36-38
    const char* sm = std::getenv("PORTRAYER_SAMPLE_MODE");
    int sample_mode = (sm && std::string(sm) == "centre") ? PT_SAMPLE_CENTRE : PT_SAMPLE_RNG;
    uint64_t seed = (uint64_t)env_long("PORTRAYER_SEED", 0);
    int device = (int)env_long("PORTRAYER_DEVICE", 0);

    // render.rs:31-34: the background is sampled once per INTEGER pixel; rows that are constant
    // in x (every example's vertical gradient) are sent as one colour per row.
    std::vector<double> bg(w * h * 3);
    bool rows = true;
    for (size_t y = 0; y < h; y++)
        for (size_t x = 0; x < w; x++) {
            Rgb c = background(Uv{(double)x / (double)w, (double)y / (double)h});
            double* o = &bg[3 * (y * w + x)];
            o[0] = c.r; o[1] = c.g; o[2] = c.b;
            if (x && (o[0] != o[-3] || o[1] != o[-2] || o[2] != o[-1])) rows = false;
        }
    std::vector<double> bg_rows;
    if (rows) {
        bg_rows.resize(h * 3);
        for (size_t y = 0; y < h; y++) std::memcpy(&bg_rows[3 * y], &bg[3 * y * w], 24);
    }
    detail::Renderer r(scene, traversal(), (int)kd_depth, device);  // render.rs:121-126
    pt_rect slice{(uint32_t)top_left_.first, (uint32_t)top_left_.second, (uint32_t)bottom_right_.first, (uint32_t)bottom_right_.second};
    pt_stats st;
    r.render(cam, (uint32_t)w, (uint32_t)h, rows ? bg_rows.data() : bg.data(), rows, slice, (uint32_t)samples, seed, sample_mode,
             std::getenv("PORTRAYER_STATS") != nullptr, image_.buffer().data(), nullptr, &st);
    RenderStats& rs = image_.stats_;
    rs.primary = st.primary; rs.shadow = st.shadow; rs.reflect = st.reflect; rs.refract = st.refract; rs.hits = st.hits;
    rs.n_inner = st.n_inner; rs.n_leaf = st.n_leaf; rs.n_analytic = st.n_analytic; rs.n_tri = st.n_tri; rs.n_bbox = st.n_bbox;
    rs.kernel_ms = st.kernel_ms; rs.total_ms = st.total_ms;
    uint64_t sw = bottom_right_.first >= top_left_.first ? bottom_right_.first - top_left_.first + 1 : 0;
    uint64_t sh = bottom_right_.second >= top_left_.second ? bottom_right_.second - top_left_.second + 1 : 0;
    rep.report_finished_pixels(sw * sh);  // render.rs:149 reports per pixel; here once, after the kernel
}

Image Image::create(const std::string& path, size_t width, size_t height) {  // render.rs:165-188
    Image img;
    img.path_ = path;
    img.width_ = width; img.height_ = height;
    size_t w = 0, h = 0;
    std::vector<uint8_t> existing;
    if (detail::png_read(path, &w, &h, &existing) && w == width && h == height) img.buffer_ = std::move(existing);
    else img.buffer_.assign(width * height * 3, 0);
    return img;
}

void Image::save_as(const std::string& path) const { detail::png_write(path, width_, height_, buffer_); }

}  // namespace render
}  // namespace portrayer
