//! A simple scene with a single triangle (scene data: examples/single-triangle.rs:17-58)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Triangle;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example single_triangle() {
    auto mat1 = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.541, 0.169, 0.886},
        .specular = Rgb{0.5, 0.7, 0.5},
        .shininess = 25.0,
    });

    auto triangle = Triangle::flat(Vec3{-1.0, 0.0, 0.0}, Vec3{1.0, 0.0, 0.0}, Vec3{0.0, 1.5, 0.0});

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(triangle, mat1)).into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{1.0, 1.0, 10.0}, .color = Rgb{0.5, 0.5, 0.5}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 0.5, 4.0},
        .center = Vec3{0.0, 0.5, 0.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(50.0),
    };

    return Example{std::move(scene), cam, 640, 480, "single-triangle.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::single_triangle()); }
#endif
