//! Demonstrates the smooth (Phong) shading feature: the same three models flat on the left,
//! with interpolated vertex normals on the right (scene data: examples/smooth-shading.rs:17-100)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Shading;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example smooth_shading(const std::string& assets) {
    auto mat_rock = std::make_shared<Material>(Material{.diffuse = Rgb{0.256361, 0.256361, 0.256361}, .specular = Rgb{0.6, 0.6, 0.6}, .shininess = 50.0});
    auto mat_cow = std::make_shared<Material>(Material{.diffuse = Rgb{0.692066, 0.477245, 0.293336}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_monkey = std::make_shared<Material>(Material{.diffuse = Rgb{0.261829, 0.8, 0.310477}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto monkey_mesh = MeshData::load_obj(assets + "/monkey.obj");
    auto cow_mesh = MeshData::load_obj(assets + "/cow.obj");
    auto flat_rock_mesh = MeshData::load_obj(assets + "/flat_rock.obj");
    auto smooth_rock_mesh = MeshData::load_obj(assets + "/smooth_rock.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            // Flat objects
            SceneNode::from(Geometry::create(Mesh::create(monkey_mesh, Shading::Flat), mat_monkey))
                .rotated_y(Radians::from_degrees(45.0))
                .translated({-1.904434, 1.4, 0.0})
                .into(),
            SceneNode::from(Geometry::create(Mesh::create(cow_mesh, Shading::Flat), mat_cow))
                .scaled(0.5)
                .rotated_y(Radians::from_degrees(-15.0))
                .translated({-4.2, 1.8, 4.0})
                .into(),
            SceneNode::from(Geometry::create(Mesh::create(flat_rock_mesh, Shading::Flat), mat_rock))
                .translated({-3.396987, -1.4, 2.286671})
                .into(),

            // Smooth objects
            SceneNode::from(Geometry::create(Mesh::create(monkey_mesh, Shading::Smooth), mat_monkey))
                .rotated_y(Radians::from_degrees(-45.0))
                .translated({1.242585, 1.4, 0.0})
                .into(),
            SceneNode::from(Geometry::create(Mesh::create(cow_mesh, Shading::Smooth), mat_cow))
                .scaled(0.5)
                .rotated_y(Radians::from_degrees(205.0))
                .translated({3.8, 1.8, 4.0})
                .into(),
            SceneNode::from(Geometry::create(Mesh::create(smooth_rock_mesh, Shading::Smooth), mat_rock))
                .translated({3.271008, -1.406423, 2.372513})
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 5.0, 10.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{1.062382, 0.54746, 22.827951},
        .center = Vec3{-0.813817, 0.424462, -8.112782},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(24.0),
    };

    return Example{std::move(scene), cam, 910, 512, "smooth-shading.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::smooth_shading("assets")); }
#endif
