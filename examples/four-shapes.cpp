//! A simple scene with four shapes on a white background (scene data: examples/four-shapes.rs:17-89)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using primitive::Cone;
using primitive::Cube;
using primitive::Cylinder;
using primitive::Sphere;

Example four_shapes() {
    const Material mat_glass_base{.specular = Rgb{0.3, 0.3, 0.3}, .shininess = 100.0};
    auto with_diffuse = [&](Rgb d) { Material m = mat_glass_base; m.diffuse = d; return std::make_shared<Material>(m); };  // Material {diffuse, ..mat_glass_base.clone()}
    auto mat_sphere = with_diffuse(Rgb{0.8, 0.0, 0.0});
    auto mat_cube = with_diffuse(Rgb{0.0, 0.158481, 0.8});
    auto mat_cone = with_diffuse(Rgb{0.064785, 0.8, 0.174433});
    auto mat_cylinder = with_diffuse(Rgb{0.127564, 0.016029, 0.8});

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Sphere{}, mat_sphere))
                .translated({-4.0, 0.0, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, mat_cube))
                .scaled(1.6)
                .rotated_y(Radians::from_degrees(-17.5411))
                .translated({-1.1, 0.0, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cone{}, mat_cone))
                .scaled(1.8)
                .translated({1.5, 0.2, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cylinder{}, mat_cylinder))
                .scaled(1.6)
                .translated({4.0, 0.0, 0.0})
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 3.0, 11.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.1, 0.1, 0.1},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 6.473007, 15.607252},
        .center = Vec3{0.0, -2.181935, -5.702181},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(10.0),
    };

    return Example{std::move(scene), cam, 1920, 512, "four-shapes.png", white};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::four_shapes()); }
#endif
