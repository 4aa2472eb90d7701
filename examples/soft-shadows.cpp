//! Demonstrates the soft shadows feature with area lights: a point light on the left, a
//! parallelogram light on the right (scene data: examples/soft-shadows.rs:17-95)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using light::Parallelogram;
using primitive::Cube;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Plane;
using primitive::Shading;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example soft_shadows(const std::string& assets) {
    auto mat_cow = std::make_shared<Material>(Material{.diffuse = Rgb{0.37168, 0.236767, 0.692066}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_wall_floor = std::make_shared<Material>(Material{.diffuse = Rgb{0.627459, 0.8, 0.589836}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto cow_mesh = MeshData::load_obj(assets + "/cow.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            // Walls + Floor
            SceneNode::from(Geometry::create(Plane{}, mat_wall_floor))
                .scaled(30.0)
                .into(),
            SceneNode::from(Geometry::create(Cube{}, mat_wall_floor))
                .scaled({0.2, 20.0, 20.0})
                .translated({0.0, 8.0, 8.0})
                .into(),
            SceneNode::from(Geometry::create(Cube{}, mat_wall_floor))
                .scaled({30.0, 30.0, 0.4})
                .translated({0.0, 8.0, -2.0})
                .into(),

            // Objects
            SceneNode::from(Geometry::create(Mesh::create(cow_mesh, Shading::Smooth), mat_cow))
                .scaled(0.5)
                .rotated_y(Radians::from_degrees(-15.0))
                .translated({-4.2, 1.8, 4.0})
                .into(),
            SceneNode::from(Geometry::create(Mesh::create(cow_mesh, Shading::Smooth), mat_cow))
                .scaled(0.5)
                .rotated_y(Radians::from_degrees(195.0))
                .translated({4.2, 1.8, 4.0})
                .into(),
        }).into(),
        .lights = {
            // Left - Point Light
            Light{.position = Vec3{-2.0, 2.0, 16.0}, .color = Rgb{0.5, 0.5, 0.5}},
            // Right - Area Light
            Light{.position = Vec3{2.0, 2.0, 16.0}, .color = Rgb{0.5, 0.5, 0.5}, .area = Parallelogram{Vec3{0.0, 0.5, 0.0}, Vec3{0.5, 0.0, 0.0}}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 5.04746, 24.827951},
        .center = Vec3{0.012231, -0.459716, -15.800501},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(25.0),
    };

    return Example{std::move(scene), cam, 910, 512, "soft-shadows.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::soft_shadows("assets")); }
#endif
