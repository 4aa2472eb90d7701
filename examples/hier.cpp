//! A hierarchical scene: one arch, a floor and a dodecahedron under a rotated, translated root
//! (scene data: examples/hier.rs:17-101)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cube;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Shading;
using primitive::Sphere;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example hier(const std::string& assets) {
    auto gold = std::make_shared<Material>(Material{.diffuse = Rgb{0.9, 0.8, 0.4}, .specular = Rgb{0.8, 0.8, 0.4}, .shininess = 25.0});
    auto grass = std::make_shared<Material>(Material{.diffuse = Rgb{0.1, 0.7, 0.1}, .specular = Rgb{0.0, 0.0, 0.0}, .shininess = 0.0});
    auto blue = std::make_shared<Material>(Material{.diffuse = Rgb{0.7, 0.6, 1.0}, .specular = Rgb{0.5, 0.4, 0.8}, .shininess = 25.0});

    auto plane = MeshData::load_obj(assets + "/plane.obj");
    auto dodeca = MeshData::load_obj(assets + "/dodeca.obj");

    // The arc
    Arc<SceneNode> arc = SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cube{}, gold))
            .scaled({0.8, 4.0, 0.8})
            .translated({-2.0, 2.0, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, gold))
            .scaled({0.8, 4.0, 0.8})
            .translated({2.0, 2.0, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Sphere{}, gold))
            .scaled({4.0, 0.6, 0.6})
            .translated({0.0, 4.0, 0.0})
            .into(),
    }).translated({0.0, 0.0, -10.0}).rotated_y(Radians::from_degrees(60.0)).into();

    // The floor
    Arc<SceneNode> floor = SceneNode::from(Geometry::create(Mesh::create(plane, Shading::Flat), grass))
        .scaled(30.0)
        .into();

    // Central "sphere"
    Arc<SceneNode> poly = SceneNode::from(Geometry::create(Mesh::create(dodeca, Shading::Flat), blue))
        .translated({-2.0, 1.618034, 0.0})
        .into();

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{arc, floor, poly})
            .rotated_x(Radians::from_degrees(23.0))
            .translated({6.0, -2.0, -15.0})
            .into(),
        .lights = {
            Light{.position = Vec3{200.0, 200.0, 400.0}, .color = Rgb{0.8, 0.8, 0.8}},  // l1
            Light{.position = Vec3{0.0, 5.0, -20.0}, .color = Rgb{0.4, 0.4, 0.8}},      // l2
        },
        .ambient = Rgb{0.4, 0.4, 0.4},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 0.0, 0.0},
        .center = Vec3{0.0, 0.0, -1.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(50.0),
    };

    return Example{std::move(scene), cam, 256, 256, "hier.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::hier("assets")); }
#endif
