//! The poster on the wall of the "Monkeys Making Better Monkeys" scene (scene data: examples/graphics-poster.rs:17-79)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Shading;

Example graphics_poster(const std::string& assets) {
    auto mat_glass = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.003638, 0.017153, 0.048247},
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .reflectivity = 0.8,
        .glossy_side_length = 0.5,
        .refraction_index = material::OPTICAL_GLASS_REFRACTION_INDEX,
    });
    auto mat_cow = std::make_shared<Material>(Material{.diffuse = Rgb{0.725682, 0.501253, 0.8}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto dodeca_model = MeshData::load_obj(assets + "/dodeca.obj");
    auto cow_model = MeshData::load_obj(assets + "/cow.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Mesh::create(dodeca_model, Shading::Flat), mat_glass))
                .rotated_y(Radians::from_degrees(90.0))
                .into(),

            SceneNode::from(Geometry::create(Mesh::create(cow_model, Shading::Smooth), mat_cow))
                .scaled(0.24)
                .rotated_y(Radians::from_degrees(-60.0))
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{1.33223, 4.297232, 3.473453}, .color = Rgb{0.9, 0.9, 0.9}},
            // Need a light inside the mesh to illuminate the cow
            Light{.position = Vec3{0.8, 0.806596, 0.9}, .color = Rgb{0.3, 0.3, 0.3}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{4.482203, 3.038775, 4.350142},
        .center = Vec3{-7.387217, -4.572944, -6.838186},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(35.0),
    };

    // let mut image = Image::new("graphics-poster.png", 1080, 1080)?;
    return Example{std::move(scene), cam, 256, 256, "graphics-poster.png", white};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::graphics_poster("assets")); }
#endif
