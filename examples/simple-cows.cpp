//! Stonehenge-like arches with cows that are assumed to be spheres (scene data: examples/simple-cows.rs:18-161)
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

Example simple_cows(const std::string& assets) {
    auto stone = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.7, 0.7}, .specular = Rgb{0.0, 0.0, 0.0}, .shininess = 0.0});
    auto grass = std::make_shared<Material>(Material{.diffuse = Rgb{0.1, 0.7, 0.1}, .specular = Rgb{0.0, 0.0, 0.0}, .shininess = 0.0});
    auto cow_hide = std::make_shared<Material>(Material{.diffuse = Rgb{0.84, 0.6, 0.53}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 20.0});

    auto plane = MeshData::load_obj(assets + "/plane.obj");
    auto buckyball = MeshData::load_obj(assets + "/buckyball.obj");

    // The arch
    Arc<SceneNode> arc = SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cube{}, stone))
            .translated({-1.9, 0.5, 0.1})
            .scaled({0.8, 4.0, 0.8})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, stone))
            .translated({2.1, 0.5, 0.1})
            .scaled({0.8, 4.0, 0.8})
            .into(),

        SceneNode::from(Geometry::create(Sphere{}, stone))
            .scaled({4.0, 0.6, 0.6})
            .translated({0.0, 4.0, 0.0})
            .into(),
    }).translated({0.0, 0.0, -10.0}).into();

    // Instancing the arc
    std::vector<Arc<SceneNode>> nodes;
    for (int i = 1; i <= 6; i++)
        nodes.push_back(SceneNode::from(arc)
            .rotated_y(Radians::from_degrees(60.0 * (double)(i - 1)))
            .into());

    // Let's assume that cows are spheres
    auto part = [&](double scale, Vec3 at) {
        return SceneNode::from(Geometry::create(Sphere{}, cow_hide))
            .scaled(scale)
            .translated(at)
            .into();
    };
    Arc<SceneNode> cow = SceneNode::from(std::vector<Arc<SceneNode>>{
        part(1.0, {0.0, 0.0, 0.0}),      // body
        part(0.6, {0.9, 0.3, 0.0}),      // head
        part(0.2, {-0.94, 0.34, 0.0}),   // tail
        part(0.3, {0.7, -0.7, -0.7}),    // lfleg
        part(0.3, {-0.7, -0.7, -0.7}),   // lrleg
        part(0.3, {0.7, -0.7, 0.7}),     // rfleg
        part(0.3, {-0.7, -0.7, 0.7}),    // rrleg
    }).into();

    // Use instancing on the cow model to place some actual cows in the scene
    const std::pair<Vec3, Radians> cows[] = {
        {Vec3{1.0, 1.3, 14.0}, Radians::from_degrees(20.0)},
        {Vec3{5.0, 1.3, -11.0}, Radians::from_degrees(180.0)},
        {Vec3{-5.5, 1.3, -3.0}, Radians::from_degrees(-60.0)},
    };
    for (const auto& [cow_pos, cow_rot] : cows)
        nodes.push_back(SceneNode::from(cow)
            .scaled(1.4)
            .rotated_y(cow_rot)
            .translated(cow_pos)
            .into());

    // The floor
    nodes.push_back(SceneNode::from(Geometry::create(Mesh::create(plane, Shading::Flat), grass))
        .scaled(30.0)
        .into());

    // Construct a central altar in the shape of a buckyball.  The
    // buckyball at the centre of the real Stonehenge was destroyed
    // in the great fire of 733 AD.
    nodes.push_back(SceneNode::from(Geometry::create(Mesh::create(buckyball, Shading::Flat), stone))
        .scaled(1.5)
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

    return Example{std::move(scene), cam, 256, 256, "simple-cows.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::simple_cows("assets")); }
#endif
