//! Test for "instancing" (having multiple of the same node in different parts of the hierarchy)
//! (scene data: examples/instance.rs:17-95)
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

Example instance(const std::string& assets) {
    auto stone = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.7, 0.7}, .specular = Rgb{0.0, 0.0, 0.0}, .shininess = 0.0});
    auto grass = std::make_shared<Material>(Material{.diffuse = Rgb{0.1, 0.7, 0.1}, .specular = Rgb{0.0, 0.0, 0.0}, .shininess = 0.0});

    auto plane = MeshData::load_obj(assets + "/plane.obj");

    // The arc
    Arc<SceneNode> arc = SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cube{}, stone))
            .scaled({0.8, 4.0, 0.8})
            .translated({-2.0, 2.0, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, stone))
            .scaled({0.8, 4.0, 0.8})
            .translated({2.0, 2.0, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Sphere{}, stone))
            .scaled({4.0, 0.6, 0.6})
            .translated({0.0, 4.0, 0.0})
            .into(),
    }).translated({0.0, 0.0, -10.0}).into();

    // Instancing
    std::vector<Arc<SceneNode>> nodes;
    for (int i = 1; i <= 6; i++)
        nodes.push_back(SceneNode::from(arc)
            .rotated_y(Radians::from_degrees(60.0 * (double)i))
            .into());

    // The floor
    nodes.push_back(SceneNode::from(Geometry::create(Mesh::create(plane, Shading::Flat), grass))
        .scaled(30.0)
        .into());

    // Central sphere
    nodes.push_back(SceneNode::from(Geometry::create(Sphere{}, stone))
        .scaled(2.5)
        .into());

    HierScene scene{
        .root = SceneNode::from(nodes)
            .rotated_x(Radians::from_degrees(23.0))
            .into(),
        .lights = {
            Light{.position = Vec3{200.0, 202.0, 430.0}, .color = Rgb{0.8, 0.8, 0.8}},
        },
        .ambient = Rgb{0.4, 0.4, 0.4},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 2.0, 30.0},
        .center = Vec3{0.0, 2.0, 29.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(50.0),
    };

    return Example{std::move(scene), cam, 256, 256, "instance.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::instance("assets")); }
#endif
