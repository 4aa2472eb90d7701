//! A very large (resource intensive) scene with some miscellaneous geometry.
//! Used to test the k-d tree acceleration mechanism (scene data: examples/big-scene.rs:26-109).
#include "../portrayer_amd/host/rand07.hpp"
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Primitive;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

static Example big_scene_of(int n, const std::vector<Primitive>& primitives, double scale_divisor);

Example big_scene(int n) {
    const std::vector<Primitive> primitives = {
        primitive::Sphere{},
        primitive::Cube{},
        primitive::Cone{},
        primitive::Cylinder{},
    };
    return big_scene_of(n, primitives, 1.0);
}

// SURVEY 8(d) synthetic variants (NOT reference scenes; the reference has no scene beyond ~17 k triangles):
// "big-mesh": the generator above with the primitive list replaced by Mesh(cow.obj) - n = 6: 216 instances of one
// 5,804-triangle mesh = 1,253,664 instanced triangles (cow.obj spans about 5 units, so the generator's 30..60 scale is
// divided by 4 to keep the instances apart). "big-soup": the same instances baked into ONE world-space triangle mesh
// (90 MB of vertex records): the input where the scene no longer fits the caches.
Example synthetic_big_mesh(const std::string& assets, int n) {
    auto cow = primitive::MeshData::load_obj(assets + "/cow.obj");
    Example ex = big_scene_of(n, {Primitive(primitive::Mesh::create(cow, primitive::Shading::Flat))}, 4.0);
    ex.output = "big-mesh.png";
    return ex;
}

Example synthetic_big_soup(const std::string& assets, int n) {
    auto cow = primitive::MeshData::load_obj(assets + "/cow.obj");
    Example inst = big_scene_of(n, {Primitive(primitive::Mesh::create(cow, primitive::Shading::Flat))}, 4.0);
    std::vector<Vec3> positions;
    std::vector<std::array<uint32_t, 3>> triangles;
    uint32_t k = 0;
    for (const auto& node : inst.scene.root->children()) {
        const Mat4& m = node->trans();
        for (const Vec3& p : cow->positions()) positions.push_back(transformed_point(p, m));
        const uint32_t base = k * (uint32_t)cow->positions().size();
        for (const auto& t : cow->triangles()) triangles.push_back({t[0] + base, t[1] + base, t[2] + base});
        k++;
    }
    auto soup = primitive::MeshData::create(std::move(positions), std::move(triangles), {});
    auto mat = std::make_shared<Material>(Material{.diffuse = Rgb{0.7, 0.6, 0.5}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{SceneNode::from(Geometry::create(primitive::Mesh::create(soup, primitive::Shading::Flat), mat)).into()}).into(),
        .lights = inst.scene.lights,
        .ambient = inst.scene.ambient,
    };
    return Example{std::move(scene), inst.cam, inst.width, inst.height, "big-soup.png"};
}

static Example big_scene_of(int n, const std::vector<Primitive>& primitives, double scale_divisor) {
    // Want the result to be random but also completely reproducible
    auto rng = rand07::StdRng::seed_from_u64(1234939301);

    std::vector<Arc<Material>> materials;
    for (int i = 0; i < 15; i++) {
        double r = rng.gen_f64(), g = rng.gen_f64(), b = rng.gen_f64();
        materials.push_back(std::make_shared<Material>(Material{
            .diffuse = Rgb{r, g, b},
            .specular = Rgb{0.3, 0.3, 0.3},
            .shininess = 25.0,
        }));
    }

    const double width = 800.0, length = 800.0, height = 800.0;

    std::vector<Arc<SceneNode>> nodes;
    for (int i = 0; i < n; i++) {
        double x = (double)i / (double)(n - 1) * width - width / 2.0;
        for (int j = 0; j < n; j++) {
            double y = (double)j / (double)(n - 1) * length - length / 2.0;
            for (int k = 0; k < n; k++) {
                double z = (double)k / (double)(n - 1) * height - height / 2.0;

                const Primitive& prim = rng.choose(primitives);
                const Arc<Material>& mat = rng.choose(materials);

                const double scale_base = 30.0;
                const double scale_increase = 30.0;

                Geometry geo = Geometry::create(prim, mat);
                double scale = scale_increase * rng.gen_f64() + scale_base;
                Radians angle = Radians::from_degrees(360.0 * rng.gen_f64());
                double yy = y + rng.gen_f64() * 50.0;
                nodes.push_back(SceneNode::from(geo)
                    .scaled(scale_divisor == 1.0 ? scale : scale / scale_divisor)
                    .rotated_xzy(angle)
                    .translated(Vec3{x, yy, z})
                    .into());
            }
        }
    }

    HierScene scene{
        .root = SceneNode::from(nodes).into(),
        .lights = {
            // white_light
            Light{.position = Vec3{-100.0, 150.0, 400.0}, .color = Rgb{0.9, 0.9, 0.9}},
            Light{.position = Vec3{100.0, -150.0, 800.0}, .color = Rgb{0.7, 0.7, 0.7}},
            // magenta_light
            Light{.position = Vec3{400.0, 100.0, 150.0}, .color = Rgb{0.7, 0.0, 0.7}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 0.0, 1200.0},
        .center = Vec3{0.0, 0.0, 0.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(50.0),
    };

    return Example{std::move(scene), cam, 1980, 1020, "big-scene.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::big_scene()); }
#endif
