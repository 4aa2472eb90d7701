//! A simple scene that demonstrates some of the extra supported primitives
//! (scene data: examples/primitives-simple.rs:17-76)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cone;
using primitive::Cylinder;
using primitive::Plane;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example primitives_simple() {
    auto mat_grass = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.173224, 0.8, 0.226505},
    });
    auto mat_cylinder = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.139339, 0.435762, 0.8},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
    });
    auto mat_cone = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.8, 0.047361, 0.04305},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
    });

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Cylinder{}, mat_cylinder))
                .scaled(2.0)
                .translated({-2.0, 1.0, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cone{}, mat_cone))
                .scaled(2.0)
                .translated({2.0, 1.0, 0.0})
                .into(),

            // Floor
            SceneNode::from(Geometry::create(Plane{}, mat_grass))
                .scaled(10.0)
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 10.0, 9.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.760838, 8.095396, 10.50759},
        .center = Vec3{-0.41716, -3.477774, -5.761218},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(25.0),
    };

    return Example{std::move(scene), cam, 910, 512, "primitives-simple.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::primitives_simple()); }
#endif
