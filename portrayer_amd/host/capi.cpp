// C entry points of the host library (include/portrayer_host.h).
#include <cstring>
#include <map>
#include <string>

#include "../../examples/examples.hpp"
#include "../../include/portrayer_host.h"
#include "host_internal.hpp"

using namespace portrayer;
using namespace portrayer::math;

struct ph_scene {
    scene::HierScene hier;
};
struct ph_renderer {
    std::unique_ptr<detail::Renderer> r;
};

static thread_local std::string g_error;

template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const Panic& e) {
        g_error = std::string("panic: ") + e.what();
        return PH_ERR_PANIC;
    } catch (const std::exception& e) {
        g_error = e.what();
        return PH_ERR_RUNTIME;
    }
}
static int bad(const char* msg) { g_error = msg; return PH_ERR_ARGUMENT; }

extern "C" const char* ph_last_error(void) { return g_error.c_str(); }

extern "C" int ph_scene_create(const ph_scene_desc* d, ph_scene** out) {
    if (!d || !out) return bad("null argument");
    return guarded([&]() -> int {
        *out = nullptr;
        const uint32_t n = d->n_nodes;
        if (n == 0 || d->root >= n) return bad("scene needs a root node");
        std::vector<Arc<texture::Texture>> textures;     // one of each per image: a material picks the view it needs
        std::vector<Arc<texture::NormalMap>> normal_maps;
        for (uint32_t t = 0; t < d->n_textures; t++) {
            auto buf = texture::RgbImageBuffer::from_pixels(d->texture_size[2 * t], d->texture_size[2 * t + 1], d->texture_rgb + d->texture_offset[t]);
            textures.push_back(std::make_shared<texture::Texture>(texture::Texture{texture::ImageTexture{buf}}));
            normal_maps.push_back(std::make_shared<texture::NormalMap>(texture::NormalMap{std::move(buf)}));
        }
        std::vector<Arc<material::Material>> mats;
        for (uint32_t i = 0; i < d->n_materials; i++) {
            const double* m = d->materials + 10 * (size_t)i;
            auto mat = std::make_shared<material::Material>(material::Material{Rgb{m[0], m[1], m[2]}, Rgb{m[3], m[4], m[5]}, m[6], m[7], m[8], m[9]});
            if (d->material_texture && d->material_texture[i] >= 0) {
                if ((uint32_t)d->material_texture[i] >= d->n_textures) throw std::runtime_error("texture index out of range");
                mat->texture = textures[d->material_texture[i]];
            }
            if (d->material_normal_map && d->material_normal_map[i] >= 0) {
                if ((uint32_t)d->material_normal_map[i] >= d->n_textures) throw std::runtime_error("texture index out of range");
                mat->normals = normal_maps[d->material_normal_map[i]];
            }
            if (d->material_uv_trans) for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) mat->uv_trans.m[r][k] = d->material_uv_trans[9 * (size_t)i + 3 * r + k];
            mats.push_back(std::move(mat));
        }
        std::vector<Arc<primitive::MeshData>> meshes;
        for (uint32_t i = 0; i < d->n_meshes; i++) {
            uint64_t v0 = d->mesh_vert_off[i], v1 = d->mesh_vert_off[i + 1], t0 = d->mesh_tri_off[i], t1 = d->mesh_tri_off[i + 1];
            std::vector<Vec3> pos, nrm;
            for (uint64_t v = v0; v < v1; v++) pos.emplace_back(d->mesh_positions[3 * v], d->mesh_positions[3 * v + 1], d->mesh_positions[3 * v + 2]);
            if (d->mesh_has_normals && d->mesh_has_normals[i] && d->mesh_normals)
                for (uint64_t v = v0; v < v1; v++) nrm.emplace_back(d->mesh_normals[3 * v], d->mesh_normals[3 * v + 1], d->mesh_normals[3 * v + 2]);
            std::vector<std::array<uint32_t, 3>> tris;
            for (uint64_t t = t0; t < t1; t++) tris.push_back({d->mesh_indices[3 * t], d->mesh_indices[3 * t + 1], d->mesh_indices[3 * t + 2]});
            std::vector<Uv> uvs;
            if (d->mesh_has_texcoords && d->mesh_has_texcoords[i] && d->mesh_texcoords)
                for (uint64_t v = v0; v < v1; v++) uvs.push_back(Uv{d->mesh_texcoords[2 * v], d->mesh_texcoords[2 * v + 1]});
            meshes.push_back(primitive::MeshData::create(std::move(pos), std::move(tris), std::move(nrm), std::move(uvs)));
        }
        // Children must exist before their parents are finished: build in reverse topological order by
        // memoised recursion (the description is a DAG; shared nodes become shared Arcs).
        std::vector<Arc<scene::SceneNode>> built(n);
        std::vector<int> state(n, 0);
        std::function<Arc<scene::SceneNode>(uint32_t)> build = [&](uint32_t i) -> Arc<scene::SceneNode> {
            if (state[i] == 2) return built[i];
            if (state[i] == 1) throw Panic("scene graph has a cycle");
            state[i] = 1;
            scene::SceneNode node;
            int t = d->prim_type[i];
            if (t >= 0) {
                if (d->material[i] < 0 || (uint32_t)d->material[i] >= d->n_materials) throw std::runtime_error("material index out of range");
                Arc<material::Material> mat = mats[d->material[i]];
                primitive::Shading sh = (d->prim_flags[i] & 1) ? primitive::Shading::Smooth : primitive::Shading::Flat;
                auto mesh_at = [&](int32_t k) -> Arc<primitive::MeshData> {
                    if (k < 0 || (uint32_t)k >= d->n_meshes) throw std::runtime_error("mesh index out of range");
                    return meshes[k];
                };
                switch (t) {
                case PT_PRIM_SPHERE: node = scene::SceneNode::from(scene::Geometry::create(primitive::Sphere{}, mat)); break;
                case PT_PRIM_CUBE: node = scene::SceneNode::from(scene::Geometry::create(primitive::Cube{}, mat)); break;
                case PT_PRIM_PLANE: node = scene::SceneNode::from(scene::Geometry::create(primitive::Plane{}, mat)); break;
                case PT_PRIM_CYLINDER: node = scene::SceneNode::from(scene::Geometry::create(primitive::Cylinder{}, mat)); break;
                case PT_PRIM_CONE: node = scene::SceneNode::from(scene::Geometry::create(primitive::Cone{}, mat)); break;
                case PT_PRIM_MESH: node = scene::SceneNode::from(scene::Geometry::create(primitive::Mesh::create(mesh_at(d->prim_data[i]), sh), mat)); break;
                case PT_PRIM_KDMESH: node = scene::SceneNode::from(scene::Geometry::create(primitive::KDMesh::create(mesh_at(d->prim_data[i]), sh), mat)); break;
                case PT_PRIM_TRIANGLE: {
                    int32_t k = d->prim_data[i];
                    if (k < 0 || (uint32_t)k >= d->n_triangles) throw std::runtime_error("triangle index out of range");
                    const double* v = d->tri_vertices + 9 * (size_t)k;
                    primitive::Triangle tri = primitive::Triangle::flat(Vec3(v[0], v[1], v[2]), Vec3(v[3], v[4], v[5]), Vec3(v[6], v[7], v[8]));
                    if (d->tri_has_normals && d->tri_has_normals[k] && d->tri_normals) {
                        const double* q = d->tri_normals + 9 * (size_t)k;
                        tri.normals = std::array<Vec3, 3>{Vec3(q[0], q[1], q[2]), Vec3(q[3], q[4], q[5]), Vec3(q[6], q[7], q[8])};
                    }
                    if (d->tri_has_texcoords && d->tri_has_texcoords[k] && d->tri_texcoords) {
                        const double* q = d->tri_texcoords + 6 * (size_t)k;
                        tri.tex_coords = std::array<Uv, 3>{Uv{q[0], q[1]}, Uv{q[2], q[3]}, Uv{q[4], q[5]}};
                    }
                    node = scene::SceneNode::from(scene::Geometry::create(tri, mat));
                    break;
                }
                default: throw std::runtime_error("unknown primitive type");
                }
            }
            const double* a = d->args + d->args_off[i];
            for (uint32_t k = d->ops_off[i]; k < d->ops_off[i + 1]; k++) {
                switch (d->ops[k]) {
                case 's': node.scaled(Vec3(a[0], a[1], a[2])); a += 3; break;
                case 't': node.translated(Vec3(a[0], a[1], a[2])); a += 3; break;
                case 'x': node.rotated_x(Radians::from_radians(a[0])); a += 1; break;
                case 'y': node.rotated_y(Radians::from_radians(a[0])); a += 1; break;
                case 'z': node.rotated_z(Radians::from_radians(a[0])); a += 1; break;
                default: throw std::runtime_error("unknown builder op");
                }
            }
            for (uint32_t k = d->child_off[i]; k < d->child_off[i + 1]; k++) {
                if (d->children[k] >= n) throw std::runtime_error("child index out of range");
                node.with_child(build(d->children[k]));
            }
            built[i] = node.into();
            state[i] = 2;
            return built[i];
        };
        auto s = std::make_unique<ph_scene>();
        s->hier.root = build(d->root);
        for (uint32_t i = 0; i < d->n_lights; i++) {
            const double* l = d->lights + 15 * (size_t)i;
            s->hier.lights.push_back(light::Light{Vec3(l[0], l[1], l[2]), Rgb{l[3], l[4], l[5]}, light::Falloff{l[6], l[7], l[8]},
                                                  light::Parallelogram{Vec3(l[9], l[10], l[11]), Vec3(l[12], l[13], l[14])}});
        }
        s->hier.ambient = Rgb{d->ambient[0], d->ambient[1], d->ambient[2]};
        *out = s.release();
        return PH_OK;
    });
}

static examples::Example make_example(const std::string& name, const std::string& assets, int n) {
    if (name == "single-triangle") return examples::single_triangle();
    if (name == "primitives-simple") return examples::primitives_simple();
    if (name == "macho-cows") return examples::macho_cows(assets);
    if (name == "entering-the-mirror-dimension") return examples::entering_the_mirror_dimension(assets);
    if (name == "big-scene") return examples::big_scene(n > 1 ? n : 10);
    if (name == "synthetic:big-mesh") return examples::synthetic_big_mesh(assets, n > 1 ? n : 6);
    if (name == "synthetic:big-soup") return examples::synthetic_big_soup(assets, n > 1 ? n : 6);
    if (name == "smooth-shading") return examples::smooth_shading(assets);
    if (name == "glossy-reflection") return examples::glossy_reflection();
    if (name == "soft-shadows") return examples::soft_shadows(assets);
    if (name == "hier") return examples::hier(assets);
    if (name == "instance") return examples::instance(assets);
    if (name == "antialiasing") return examples::antialiasing(assets);
    if (name == "fish") return examples::fish(assets);
    if (name == "normal-mapping") return examples::normal_mapping(assets);
    if (name == "transmission-refraction") return examples::transmission_refraction(assets);
    if (name == "water-glass") return examples::water_glass(assets);
    if (name == "simple") return examples::simple();
    if (name == "nonhier") return examples::nonhier(assets);
    if (name == "nonhier2") return examples::nonhier2(assets);
    if (name == "four-shapes") return examples::four_shapes();
    if (name == "graphics-poster") return examples::graphics_poster(assets);
    if (name == "simple-cows") return examples::simple_cows(assets);
    if (name == "primitives") return examples::primitives(assets);
    if (name == "texture-mapping") return examples::texture_mapping(assets);
    if (name == "cube-mapping") return examples::cube_mapping(assets);
    if (name == "graphics-castle") return examples::graphics_castle(assets);
    if (name == "graphics-temple") return examples::graphics_temple(assets);
    if (name == "monkeys-making-monkeys") return examples::monkeys_making_monkeys(assets);
    if (name == "robot-alarm-clock") return examples::robot_alarm_clock(assets);
    throw std::runtime_error("unknown example scene: " + name);
}

extern "C" int ph_example_scene(const char* name, const char* assets_dir, int n, ph_scene** out, double camera[10], uint32_t size[2]) {
    if (!name || !out) return bad("null argument");
    return guarded([&]() -> int {
        examples::Example ex = make_example(name, assets_dir ? assets_dir : "assets", n);
        auto s = std::make_unique<ph_scene>();
        s->hier = std::move(ex.scene);
        if (camera) {
            const Vec3 v[3] = {ex.cam.eye, ex.cam.center, ex.cam.up};
            for (int k = 0; k < 3; k++) { camera[3 * k] = v[k].x; camera[3 * k + 1] = v[k].y; camera[3 * k + 2] = v[k].z; }
            camera[9] = ex.cam.fovy.get();
        }
        if (size) { size[0] = (uint32_t)ex.width; size[1] = (uint32_t)ex.height; }
        *out = s.release();
        return PH_OK;
    });
}

extern "C" void ph_scene_destroy(ph_scene* s) { delete s; }

namespace {
struct Linear {  // unique nodes in DFS pre-order, materials / meshes in order of first use
    std::vector<const scene::SceneNode*> nodes;
    std::map<const scene::SceneNode*, uint32_t> node_id;
    std::vector<const material::Material*> mats;
    std::map<const material::Material*, int32_t> mat_id;
    std::vector<const primitive::MeshData*> meshes;
    std::map<const primitive::MeshData*, int32_t> mesh_id;
    std::vector<const primitive::Triangle*> tris;
    size_t n_children = 0;
    void visit(const scene::SceneNode* n) {
        if (node_id.count(n)) return;
        node_id[n] = (uint32_t)nodes.size();
        nodes.push_back(n);
        for (const auto& c : n->children()) visit(c.get());
    }
    explicit Linear(const scene::HierScene& h) {
        visit(h.root.get());
        for (const scene::SceneNode* n : nodes) {
            n_children += n->children().size();
            if (!n->geometry()) continue;
            const auto& g = *n->geometry();
            if (!mat_id.count(g.material.get())) { mat_id[g.material.get()] = (int32_t)mats.size(); mats.push_back(g.material.get()); }
            if (g.primitive.mesh && !mesh_id.count(g.primitive.mesh.get())) { mesh_id[g.primitive.mesh.get()] = (int32_t)meshes.size(); meshes.push_back(g.primitive.mesh.get()); }
            if (g.primitive.kind == primitive::Primitive::TriangleK) tris.push_back(&g.primitive.triangle);
        }
    }
};
}  // namespace

// Textures, normal maps and texture coordinates of the scene, in the numbering of ph_scene_export (materials / meshes /
// triangles in order of first use). Two calls: with texture_rgb == NULL it only fills counts = {n_textures, texel bytes}.
extern "C" int ph_scene_export_textures(const ph_scene* s, uint64_t counts[2], int32_t* material_texture, int32_t* material_normal_map,
                                        double* material_uv_trans, uint32_t* texture_size, uint64_t* texture_offset, uint8_t* texture_rgb,
                                        double* mesh_texcoords, uint8_t* mesh_has_texcoords, double* tri_texcoords, uint8_t* tri_has_texcoords) {
    if (!s || !counts) return bad("null argument");
    return guarded([&]() -> int {
        Linear lin(s->hier);
        std::vector<const texture::RgbImageBuffer*> textures;
        std::map<const texture::RgbImageBuffer*, int32_t> tex_id;
        auto index = [&](const texture::RgbImageBuffer* b) {
            auto it = tex_id.find(b);
            if (it == tex_id.end()) { it = tex_id.emplace(b, (int32_t)textures.size()).first; textures.push_back(b); }
            return it->second;
        };
        for (size_t i = 0; i < lin.mats.size(); i++) {
            const material::Material* m = lin.mats[i];
            int32_t a = m->texture ? index(&m->texture->image.buffer) : -1, b = m->normals ? index(&m->normals->buffer) : -1;
            if (material_texture) material_texture[i] = a;
            if (material_normal_map) material_normal_map[i] = b;
            if (material_uv_trans) for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) material_uv_trans[9 * i + 3 * r + k] = m->uv_trans.m[r][k];
        }
        uint64_t off = 0;
        for (size_t t = 0; t < textures.size(); t++) {
            if (texture_size) { texture_size[2 * t] = (uint32_t)textures[t]->width; texture_size[2 * t + 1] = (uint32_t)textures[t]->height; }
            if (texture_offset) texture_offset[t] = off;
            if (texture_rgb) std::memcpy(texture_rgb + off, textures[t]->rgb.data(), textures[t]->rgb.size());
            off += textures[t]->rgb.size();
        }
        counts[0] = textures.size(); counts[1] = off;
        uint64_t vo = 0;
        for (size_t m = 0; m < lin.meshes.size(); m++) {
            const primitive::MeshData* md = lin.meshes[m];
            const bool ht = md->tex_coords().size() == md->positions().size();
            if (mesh_has_texcoords) mesh_has_texcoords[m] = ht ? 1 : 0;
            if (mesh_texcoords)
                for (size_t v = 0; v < md->positions().size(); v++) {
                    mesh_texcoords[2 * (vo + v)] = ht ? md->tex_coords()[v].u : 0.0;
                    mesh_texcoords[2 * (vo + v) + 1] = ht ? md->tex_coords()[v].v : 0.0;
                }
            vo += md->positions().size();
        }
        for (size_t t = 0; t < lin.tris.size(); t++) {
            const primitive::Triangle* tr = lin.tris[t];
            if (tri_has_texcoords) tri_has_texcoords[t] = tr->tex_coords ? 1 : 0;
            if (tri_texcoords)
                for (int k = 0; k < 3; k++) {
                    tri_texcoords[6 * t + 2 * k] = tr->tex_coords ? (*tr->tex_coords)[k].u : 0.0;
                    tri_texcoords[6 * t + 2 * k + 1] = tr->tex_coords ? (*tr->tex_coords)[k].v : 0.0;
                }
        }
        return PH_OK;
    });
}

extern "C" int ph_scene_counts(const ph_scene* s, uint64_t c[8]) {
    if (!s || !c) return bad("null argument");
    return guarded([&]() -> int {
        Linear lin(s->hier);
        uint64_t verts = 0, mtris = 0;
        for (auto* m : lin.meshes) { verts += m->positions().size(); mtris += m->triangles().size(); }
        c[0] = lin.nodes.size(); c[1] = lin.n_children; c[2] = lin.meshes.size(); c[3] = verts; c[4] = mtris;
        c[5] = lin.tris.size(); c[6] = lin.mats.size(); c[7] = s->hier.lights.size();
        return PH_OK;
    });
}

extern "C" int ph_scene_export(const ph_scene* s, double* node_trans, int32_t* prim_type, int32_t* prim_data, int32_t* prim_flags,
                               int32_t* material, uint32_t* child_off, uint32_t* children, uint32_t* root,
                               uint64_t* mesh_vert_off, uint64_t* mesh_tri_off, double* mesh_positions, double* mesh_normals,
                               uint8_t* mesh_has_normals, uint32_t* mesh_indices, double* tri_vertices, double* tri_normals,
                               uint8_t* tri_has_normals, double* materials, double* lights, double ambient[3]) {
    if (!s) return bad("null argument");
    return guarded([&]() -> int {
        Linear lin(s->hier);
        uint32_t ck = 0, tk = 0;
        for (size_t i = 0; i < lin.nodes.size(); i++) {
            const scene::SceneNode* n = lin.nodes[i];
            if (node_trans) std::memcpy(node_trans + 16 * i, n->trans().m, 128);
            int32_t t = -1, data = 0, flags = 0, mat = 0;
            if (n->geometry()) {
                const auto& g = *n->geometry();
                t = (int32_t)g.primitive.kind;
                mat = lin.mat_id[g.material.get()];
                if (g.primitive.mesh) { data = lin.mesh_id[g.primitive.mesh.get()]; flags = g.primitive.shading == primitive::Shading::Smooth ? 1 : 0; }
                if (g.primitive.kind == primitive::Primitive::TriangleK) data = (int32_t)tk++;
            }
            if (prim_type) prim_type[i] = t;
            if (prim_data) prim_data[i] = data;
            if (prim_flags) prim_flags[i] = flags;
            if (material) material[i] = mat;
            if (child_off) child_off[i] = ck;
            for (const auto& c : n->children()) { if (children) children[ck] = lin.node_id[c.get()]; ck++; }
        }
        if (child_off) child_off[lin.nodes.size()] = ck;
        if (root) *root = 0;
        uint64_t vo = 0, to = 0;
        for (size_t m = 0; m < lin.meshes.size(); m++) {
            const primitive::MeshData* md = lin.meshes[m];
            if (mesh_vert_off) mesh_vert_off[m] = vo;
            if (mesh_tri_off) mesh_tri_off[m] = to;
            bool hn = md->normals().size() == md->positions().size();
            if (mesh_has_normals) mesh_has_normals[m] = hn ? 1 : 0;
            for (size_t v = 0; v < md->positions().size(); v++) {
                const Vec3& p = md->positions()[v];
                if (mesh_positions) { mesh_positions[3 * (vo + v)] = p.x; mesh_positions[3 * (vo + v) + 1] = p.y; mesh_positions[3 * (vo + v) + 2] = p.z; }
                if (mesh_normals) {
                    Vec3 q = hn ? md->normals()[v] : Vec3::zero();
                    mesh_normals[3 * (vo + v)] = q.x; mesh_normals[3 * (vo + v) + 1] = q.y; mesh_normals[3 * (vo + v) + 2] = q.z;
                }
            }
            for (size_t t = 0; t < md->triangles().size(); t++)
                if (mesh_indices) for (int k = 0; k < 3; k++) mesh_indices[3 * (to + t) + k] = md->triangles()[t][k];
            vo += md->positions().size(); to += md->triangles().size();
        }
        if (mesh_vert_off) mesh_vert_off[lin.meshes.size()] = vo;
        if (mesh_tri_off) mesh_tri_off[lin.meshes.size()] = to;
        for (size_t t = 0; t < lin.tris.size(); t++) {
            const primitive::Triangle* tr = lin.tris[t];
            const Vec3 v[3] = {tr->a, tr->b, tr->c};
            for (int k = 0; k < 3; k++) {
                if (tri_vertices) { tri_vertices[9 * t + 3 * k] = v[k].x; tri_vertices[9 * t + 3 * k + 1] = v[k].y; tri_vertices[9 * t + 3 * k + 2] = v[k].z; }
                if (tri_normals) {
                    Vec3 q = tr->normals ? (*tr->normals)[k] : Vec3::zero();
                    tri_normals[9 * t + 3 * k] = q.x; tri_normals[9 * t + 3 * k + 1] = q.y; tri_normals[9 * t + 3 * k + 2] = q.z;
                }
            }
            if (tri_has_normals) tri_has_normals[t] = tr->normals ? 1 : 0;
        }
        for (size_t i = 0; i < lin.mats.size() && materials; i++) {
            const material::Material* m = lin.mats[i];
            const double row[10] = {m->diffuse.r, m->diffuse.g, m->diffuse.b, m->specular.r, m->specular.g, m->specular.b,
                                    m->shininess, m->reflectivity, m->glossy_side_length, m->refraction_index};
            std::memcpy(materials + 10 * i, row, sizeof row);
        }
        for (size_t i = 0; i < s->hier.lights.size() && lights; i++) {
            const light::Light& l = s->hier.lights[i];
            const double row[15] = {l.position.x, l.position.y, l.position.z, l.color.r, l.color.g, l.color.b, l.falloff.c0, l.falloff.c1, l.falloff.c2,
                                    l.area.a.x, l.area.a.y, l.area.a.z, l.area.b.x, l.area.b.y, l.area.b.z};
            std::memcpy(lights + 15 * i, row, sizeof row);
        }
        if (ambient) { ambient[0] = s->hier.ambient.r; ambient[1] = s->hier.ambient.g; ambient[2] = s->hier.ambient.b; }
        return PH_OK;
    });
}

extern "C" int ph_scene_flatten(const ph_scene* s, uint32_t cap, double* trans, double* invtrans, double* normal_trans,
                                int32_t* prim_type, int32_t* material, double* bounds) {
    if (!s) return bad("null argument");
    return guarded([&]() -> int {
        detail::FlatScene flat = detail::FlatScene::from(s->hier);
        std::map<const material::Material*, int32_t> mat_id;
        for (size_t i = 0; i < flat.root.size(); i++) {
            const auto& fn = flat.root[i];
            const material::Material* m = fn.geometry.material.get();
            if (!mat_id.count(m)) { int32_t id = (int32_t)mat_id.size(); mat_id[m] = id; }
            if (i >= cap) continue;
            if (trans) std::memcpy(trans + 16 * i, fn.trans.m, 128);
            if (invtrans) std::memcpy(invtrans + 16 * i, fn.invtrans.m, 128);
            if (normal_trans) std::memcpy(normal_trans + 16 * i, fn.normal_trans.m, 128);
            if (prim_type) prim_type[i] = (int32_t)fn.geometry.primitive.kind;
            if (material) material[i] = mat_id[m];
            if (bounds) {
                detail::BoundingBox b = fn.bounds();
                double* o = bounds + 6 * i;
                o[0] = b.min.x; o[1] = b.min.y; o[2] = b.min.z; o[3] = b.max.x; o[4] = b.max.y; o[5] = b.max.z;
            }
        }
        return (int)flat.root.size();
    });
}

extern "C" int ph_scene_graph(const ph_scene* s, uint32_t node_cap, uint32_t chain_cap, uint32_t graph_cap, uint32_t* chain_off, uint32_t* chain,
                              uint32_t* dfs_rank, double* graph_trans, double* graph_invtrans, double* graph_normal_trans, uint32_t counts[2]) {
    if (!s || !counts) return bad("null argument");
    return guarded([&]() -> int {
        detail::FlatScene flat = detail::FlatScene::from(s->hier);
        detail::GraphPacking g = detail::pack_graph(flat);
        const size_t n = flat.root.size();
        counts[0] = (uint32_t)g.chain.size(); counts[1] = g.n_graph_nodes;
        if (chain_off && node_cap >= n) std::memcpy(chain_off, g.chain_off.data(), (n + 1) * 4);
        if (dfs_rank && node_cap >= n) std::memcpy(dfs_rank, g.dfs_rank.data(), n * 4);
        if (chain && chain_cap >= g.chain.size()) std::memcpy(chain, g.chain.data(), g.chain.size() * 4);
        if (graph_cap >= g.n_graph_nodes) {
            if (graph_trans) std::memcpy(graph_trans, g.trans.data(), g.trans.size() * 8);
            if (graph_invtrans) std::memcpy(graph_invtrans, g.invtrans.data(), g.invtrans.size() * 8);
            if (graph_normal_trans) std::memcpy(graph_normal_trans, g.normal_trans.data(), g.normal_trans.size() * 8);
        }
        return (int)n;
    });
}

extern "C" int ph_scene_kdtree(const ph_scene* s, int kd_depth, uint32_t node_cap, uint32_t item_cap, int32_t* axis, double* plane,
                               int32_t* front, int32_t* back, int32_t* first, int32_t* count, int32_t* leaf_items, uint32_t* n_items,
                               double root_bounds[6], int32_t* max_depth) {
    if (!s) return bad("null argument");
    return guarded([&]() -> int {
        detail::FlatScene flat = detail::FlatScene::from(s->hier);
        detail::KdTree t = detail::kd_scene_tree(flat, kd_depth < 0 ? 10 : (size_t)kd_depth);
        if (t.axis.size() > node_cap || t.items.size() > item_cap) { g_error = "output capacity too small"; return PH_ERR_SMALL; }
        for (size_t i = 0; i < t.axis.size(); i++) {
            if (axis) axis[i] = t.axis[i];
            if (plane) plane[i] = t.plane[i];
            if (front) front[i] = t.front[i];
            if (back) back[i] = t.back[i];
            if (first) first[i] = t.first[i];
            if (count) count[i] = t.count[i];
        }
        for (size_t i = 0; i < t.items.size(); i++) if (leaf_items) leaf_items[i] = t.items[i];
        if (n_items) *n_items = (uint32_t)t.items.size();
        if (root_bounds) {
            root_bounds[0] = t.root_min.x; root_bounds[1] = t.root_min.y; root_bounds[2] = t.root_min.z;
            root_bounds[3] = t.root_max.x; root_bounds[4] = t.root_max.y; root_bounds[5] = t.root_max.z;
        }
        if (max_depth) *max_depth = t.max_depth;
        return (int)t.axis.size();
    });
}

static camera::CameraSettings camera_from(const double c[10]) {
    return camera::CameraSettings{Vec3(c[0], c[1], c[2]), Vec3(c[3], c[4], c[5]), Vec3(c[6], c[7], c[8]), Radians::from_radians(c[9])};
}

extern "C" int ph_camera(const double c[10], double width, double height, pt_camera* out) {
    if (!c || !out) return bad("null argument");
    return guarded([&]() -> int {
        *out = detail::Camera(camera_from(c), width, height).to_abi();
        return PH_OK;
    });
}

extern "C" int ph_obj_load(const char* path, uint64_t counts[3], double* positions, double* normals, uint32_t* indices, uint64_t vert_cap, uint64_t tri_cap) {
    if (!path || !counts) return bad("null argument");
    return guarded([&]() -> int {
        auto md = primitive::MeshData::load_obj(path);
        bool hn = md->normals().size() == md->positions().size();
        counts[0] = md->positions().size(); counts[1] = md->triangles().size(); counts[2] = hn ? 1 : 0;
        for (size_t v = 0; v < md->positions().size() && v < vert_cap; v++) {
            if (positions) { positions[3 * v] = md->positions()[v].x; positions[3 * v + 1] = md->positions()[v].y; positions[3 * v + 2] = md->positions()[v].z; }
            if (normals && hn) { normals[3 * v] = md->normals()[v].x; normals[3 * v + 1] = md->normals()[v].y; normals[3 * v + 2] = md->normals()[v].z; }
        }
        for (size_t t = 0; t < md->triangles().size() && t < tri_cap; t++)
            if (indices) for (int k = 0; k < 3; k++) indices[3 * t + k] = md->triangles()[t][k];
        return PH_OK;
    });
}

extern "C" int ph_renderer_create(const ph_scene* s, int traverse, int kd_depth, int device, ph_renderer** out) {
    if (!s || !out) return bad("null argument");
    if (traverse != PT_TRAVERSE_FLAT && traverse != PT_TRAVERSE_KD && traverse != PT_TRAVERSE_HIER) return bad("traverse must be PT_TRAVERSE_FLAT, PT_TRAVERSE_KD or PT_TRAVERSE_HIER");
    return guarded([&]() -> int {
        auto r = std::make_unique<ph_renderer>();
        r->r = std::make_unique<detail::Renderer>(s->hier, traverse == PT_TRAVERSE_KD ? render::Traversal::KdTree : (traverse == PT_TRAVERSE_HIER ? render::Traversal::Hier : render::Traversal::Flat), kd_depth, device);
        *out = r.release();
        return PH_OK;
    });
}
extern "C" void ph_renderer_destroy(ph_renderer* r) { delete r; }
extern "C" pt_context* ph_renderer_context(ph_renderer* r) { return r ? r->r->context() : nullptr; }
extern "C" pt_node* ph_renderer_node(ph_renderer* r) { return r ? r->r->node() : nullptr; }
extern "C" int ph_renderer_ranks(ph_renderer* r) { return !r ? 0 : (r->r->node() ? pt_node_ranks(r->r->node()) : 1); }
extern "C" int ph_renderer_prepare_ms(ph_renderer* r, double out[5]) {
    if (!r || !out) return bad("null argument");
    const auto& p = r->r->prepare_ms();
    out[0] = p.flatten; out[1] = p.pack; out[2] = p.context; out[3] = p.kd_build; out[4] = p.upload;
    return PH_OK;
}

extern "C" int ph_renderer_render(ph_renderer* r, const double camera[10], const pt_render_params* p, const double* background,
                                  uint8_t* rgb, double* linear, pt_stats* stats) {
    if (!r || !camera || !p || !background || !rgb) return bad("null argument");
    return guarded([&]() -> int {
        if (p->tile_ranks != 1 || p->tile_rank != 0) {  // multi-GPU callers drive pt_render_device themselves
            pt_camera pc = detail::Camera(camera_from(camera), (double)p->width, (double)p->height).to_abi();
            int rc = pt_render(r->r->context(), &pc, background, p, rgb, linear, stats);
            if (rc != PT_OK) { g_error = pt_last_error(r->r->context()); return rc == PT_ERR_SLICE ? PH_ERR_PANIC : PH_ERR_RUNTIME; }
            return PH_OK;
        }
        r->r->render(camera_from(camera), p->width, p->height, background, p->background_rows != 0, p->slice, p->samples, p->seed,
                     p->sample_mode, p->collect_stats != 0, rgb, linear, stats);
        return PH_OK;
    });
}

extern "C" int ph_example_render_to_png(const char* name, const char* assets_dir, int n, uint32_t width, uint32_t height, const char* png_path) {
    if (!name || !png_path) return bad("null argument");
    return guarded([&]() -> int {
        examples::Example ex = make_example(name, assets_dir ? assets_dir : "assets", n);
        render::Image image = render::Image::create(png_path, width ? width : ex.width, height ? height : ex.height);
        image.render<reporter::NullProgress>(ex.scene, ex.cam, ex.background);
        image.save();
        return PH_OK;
    });
}

extern "C" int ph_png_read(const char* path, uint32_t size[2], uint8_t* rgb, uint64_t cap) {
    if (!path || !size) return bad("null argument");
    return guarded([&]() -> int {
        size_t w = 0, h = 0;
        std::vector<uint8_t> buf;
        if (!detail::png_read(path, &w, &h, &buf)) { g_error = "file not found"; return PH_ERR_RUNTIME; }
        size[0] = (uint32_t)w; size[1] = (uint32_t)h;
        if (rgb) { if (cap < buf.size()) { g_error = "output capacity too small"; return PH_ERR_SMALL; } std::memcpy(rgb, buf.data(), buf.size()); }
        return PH_OK;
    });
}
extern "C" int ph_image_read(const char* path, uint32_t size[2], uint8_t* rgb, uint64_t cap) {
    if (!path || !size) return bad("null argument");
    return guarded([&]() -> int {
        size_t w = 0, h = 0;
        std::vector<uint8_t> buf;
        if (!detail::image_read(path, &w, &h, &buf)) { g_error = "file not found"; return PH_ERR_RUNTIME; }
        size[0] = (uint32_t)w; size[1] = (uint32_t)h;
        if (rgb) {
            if (cap < buf.size()) return bad("buffer too small");
            std::memcpy(rgb, buf.data(), buf.size());
        }
        return PH_OK;
    });
}

extern "C" int ph_png_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb) {
    if (!path || !rgb) return bad("null argument");
    return guarded([&]() -> int {
        detail::png_write(path, width, height, std::vector<uint8_t>(rgb, rgb + (size_t)width * height * 3));
        return PH_OK;
    });
}

// examples/*.cpp main(): Image::new(..)?; image.render::<RenderProgress, _>(&scene, cam, sky); image.save()
int portrayer::examples::run_main(Example ex) {
    try {
        render::Image image = render::Image::create(ex.output, ex.width, ex.height);
        image.render<reporter::RenderProgress>(ex.scene, ex.cam, ex.background);
        image.save();
        const auto& st = image.last_stats();
        std::fprintf(stderr, "%s: kernel %.2f ms, total %.2f ms\n", ex.output.c_str(), st.kernel_ms, st.total_ms);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
