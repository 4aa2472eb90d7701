//! Demonstrates the glossy reflection feature: two mirror-like balls, the right one with its
//! reflected rays spread over a square (scene data: examples/glossy-reflection.rs:17-87)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cube;
using primitive::Sphere;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example glossy_reflection() {
    Material ball{.diffuse = Rgb{0.146505, 0.314666, 0.170564}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 100.0, .reflectivity = 0.4};
    auto non_glossy_ball = std::make_shared<Material>(ball);
    ball.glossy_side_length = 2.0;  // ..(*non_glossy_ball).clone()
    auto glossy_ball = std::make_shared<Material>(ball);
    auto center_ball = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.0, 0.023362}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto table = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 0.6, 0.1}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Sphere{}, non_glossy_ball))
                .translated({-1.1, 1.3, 0.0})
                .into(),
            SceneNode::from(Geometry::create(Sphere{}, glossy_ball))
                .translated({1.1, 1.3, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Sphere{}, center_ball))
                .scaled(0.5)
                .translated({0.0, 0.8, 1.8})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, table))
                .scaled({10.0, 0.6, 5.0})
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 6.0, 3.0}, .color = Rgb{0.9, 0.9, 0.9}},
            Light{.position = Vec3{0.0, 1.0, 12.0}, .color = Rgb{0.7, 0.7, 0.7}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 2.562834, 8.863271},
        .center = Vec3{0.0, -1.083779, -11.817695},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(20.0),
    };

    return Example{std::move(scene), cam, 910, 512, "glossy-reflection.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::glossy_reflection()); }
#endif
