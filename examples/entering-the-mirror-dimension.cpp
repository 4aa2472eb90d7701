//! Title: "Entering The Mirror Dimension" (scene data: examples/entering-the-mirror-dimension.rs:17-188)
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

Example entering_the_mirror_dimension(const std::string& assets) {
    // Materials
    auto mat_mirror_frame = std::make_shared<Material>(Material{.diffuse = Rgb{0.29, 0.204, 0.145}, .specular = Rgb{0.0, 0.0, 0.0}, .shininess = 1.0});
    auto mat_mirror = std::make_shared<Material>(Material{.diffuse = Rgb{0.0, 0.0, 0.0}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 1000.0, .reflectivity = 1.0});
    auto mat_floor = std::make_shared<Material>(Material{.diffuse = Rgb{0.016, 0.384, 0.0}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 25.0});
    auto mat_body = std::make_shared<Material>(Material{.diffuse = Rgb{0.906, 0.22, 0.282}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 25.0});
    auto mat_head = std::make_shared<Material>(Material{.diffuse = Rgb{0.086, 0.671, 0.906}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 50.0});
    auto mat_eyes = std::make_shared<Material>(Material{.diffuse = Rgb{0.3, 0.3, 0.3}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 1000.0, .reflectivity = 0.9});
    auto mat_arms = std::make_shared<Material>(Material{.diffuse = Rgb{0.345, 0.588, 0.906}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 1.0});

    auto monkey = MeshData::load_obj(assets + "/monkey.obj");
    auto plane = MeshData::load_obj(assets + "/plane.obj");
    auto deg = [](double d) { return Radians::from_degrees(d); };

    Arc<SceneNode> mirror = SceneNode::from(std::vector<Arc<SceneNode>>{
        // mirror frame
        SceneNode::from(Geometry::create(Cube{}, mat_mirror_frame))
            .scaled({3.96, 5.5, 0.4})
            .translated({0.0, 2.75, 0.0})
            .into(),

        // mirror glass
        SceneNode::from(Geometry::create(Cube{}, mat_mirror))
            .scaled({3.6, 5.0, 0.1})
            .translated({0.0, 2.75, 0.2})
            .into(),
    }).translated({0.0, 0.0, -1.3}).into();

    Arc<SceneNode> monkey_character = SceneNode::from(std::vector<Arc<SceneNode>>{
        // torso
        SceneNode::from(Geometry::create(Cube{}, mat_body))
            .scaled({0.545055, 2.6, 0.545055})
            .translated({0.0, 1.3, 0.0})
            .into(),

        // head
        SceneNode::from(Geometry::create(Mesh::create(monkey, Shading::Flat), mat_head))
            .scaled({1.0, 1.0, 1.0})
            .rotated_y(deg(180.0))
            .translated({0.0, 2.7, 0.0})
            .with_children({
                // left eye
                SceneNode::from(Geometry::create(Sphere{}, mat_eyes))
                .scaled({0.1, 0.1, 0.05})
                .translated({0.35, 0.24, 0.8})
                .into(),

                // right eye
                SceneNode::from(Geometry::create(Sphere{}, mat_eyes))
                .scaled({0.1, 0.1, 0.05})
                .translated({-0.35, 0.24, 0.8})
                .into(),
            })
            .into(),

        // left upper arm
        SceneNode::from(Geometry::create(Sphere{}, mat_arms))
            .scaled({0.2, 0.63, 0.2})
            .rotated_xzy(deg(161.156), deg(107.062), deg(-133.944))
            .translated({-0.388703, 1.715599, -0.2})
            .into(),
        // left lower arm
        SceneNode::from(Geometry::create(Sphere{}, mat_arms))
            .scaled({0.2, 0.56, 0.2})
            .rotated_xzy(deg(127.221), deg(42.0695), deg(-104.823))
            .translated({-0.711297, 1.284401, -1.0})
            .into(),
        // left mirror bubble
        SceneNode::from(Geometry::create(Sphere{}, mat_mirror))
            .scaled({0.5, 0.5, 0.3})
            .translated({-0.711297, 1.284401, -1.20})
            .into(),

        // right upper arm
        SceneNode::from(Geometry::create(Sphere{}, mat_arms))
            .scaled({0.2, 0.63, 0.2})
            .rotated_xzy(deg(92.3684), deg(-57.6199), deg(38.2278))
            .translated({0.581161, 1.984976, -0.2})
            .into(),
        // right lower arm
        SceneNode::from(Geometry::create(Sphere{}, mat_arms))
            .scaled({0.2, 0.56, 0.2})
            .rotated_xzy(deg(91.5166), deg(-11.239), deg(28.419))
            .translated({1.118839, 2.015024, -1.0})
            .into(),
        // right mirror bubble
        SceneNode::from(Geometry::create(Sphere{}, mat_mirror))
            .scaled({0.5, 0.5, 0.3})
            .translated({1.118839, 2.015024, -1.20})
            .into(),
    }).into();

    // The floor
    Arc<SceneNode> floor = SceneNode::from(Geometry::create(Mesh::create(plane, Shading::Flat), mat_floor))
        .scaled(20.0)
        .into();

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{mirror, floor, monkey_character}).into(),
        .lights = {
            // face_light
            Light{.position = Vec3{2.5, 3.5, -1.0}, .color = Rgb{0.9, 0.9, 0.9}},
            // white_light
            Light{.position = Vec3{10.0, 10.0, 0.0}, .color = Rgb{0.9, 0.9, 0.9}},
            // blue_light
            Light{.position = Vec3{-9.0, 4.0, 0.0}, .color = Rgb{0.406471, 0.901283, 1.0}},
        },
        .ambient = Rgb{0.2, 0.2, 0.2},
    };

    camera::CameraSettings cam{
        .eye = Vec3{5.545485, 2.966984, 1.795613},
        .center = Vec3{-4.348584, 2.148794, -3.057839},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(30.0),
    };

    return Example{std::move(scene), cam, 800, 600, "entering-the-mirror-dimension.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::entering_the_mirror_dimension("assets")); }
#endif
