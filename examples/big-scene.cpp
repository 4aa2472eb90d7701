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

Example big_scene(int n) {
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

    const std::vector<Primitive> primitives = {
        primitive::Sphere{},
        primitive::Cube{},
        primitive::Cone{},
        primitive::Cylinder{},
    };

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
                    .scaled(scale)
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
