//! A simple scene with some spheres, a cube and a mesh (scene data: examples/nonhier.rs)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using primitive::Cube;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Shading;
using primitive::Sphere;

Example nonhier(const std::string& assets) {
    auto mat1 = std::make_shared<Material>(Material{.diffuse = Rgb{0.7, 1.0, 0.7}, .specular = Rgb{0.5, 0.7, 0.5}, .shininess = 25.0});
    auto mat2 = std::make_shared<Material>(Material{.diffuse = Rgb{0.5, 0.5, 0.5}, .specular = Rgb{0.5, 0.7, 0.5}, .shininess = 25.0});
    auto mat3 = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 0.6, 0.1}, .specular = Rgb{0.5, 0.7, 0.5}, .shininess = 25.0});
    auto mat4 = std::make_shared<Material>(Material{.diffuse = Rgb{0.7, 0.6, 1.0}, .specular = Rgb{0.5, 0.4, 0.8}, .shininess = 25.0});

    auto monkey = MeshData::load_obj(assets + "/monkey.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Sphere{}, mat1))
                .scaled(100.0)
                .translated({0.0, 0.0, -400.0})
                .into(),

            SceneNode::from(Geometry::create(Sphere{}, mat1))
                .scaled(150.0)
                .translated({200.0, 50.0, -100.0})
                .into(),

            SceneNode::from(Geometry::create(Sphere{}, mat2))
                .scaled(1000.0)
                .translated({0.0, -1200.0, -500.0})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, mat4))
                .scaled(100.0)
                .translated({-150.0, -75.0, 50.0})
                .into(),

            SceneNode::from(Geometry::create(Sphere{}, mat3))
                .scaled(50.0)
                .translated({-100.0, 25.0, -300.0})
                .into(),

            SceneNode::from(Geometry::create(Sphere{}, mat1))
                .scaled(25.0)
                .translated({0.0, 100.0, -250.0})
                .into(),

            SceneNode::from(Geometry::create(Mesh::create(monkey, Shading::Flat), mat3))
                .scaled(100.0)
                .translated({-150.0, 200.0, -100.0})
                .into(),
        }).into(),
        .lights = {
            // white_light
            Light{.position = Vec3{-100.0, 150.0, 400.0}, .color = Rgb{0.9, 0.9, 0.9}},
            // magenta_light
            Light{.position = Vec3{400.0, 100.0, 150.0}, .color = Rgb{0.7, 0.0, 0.7}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 0.0, 800.0},
        .center = Vec3{0.0, 0.0, 0.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(50.0),
    };

    return Example{std::move(scene), cam, 256, 256, "nonhier.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::nonhier("assets")); }
#endif
