//! Two textured fish: a smooth-shaded mesh whose texture coordinates come from the OBJ file's `vt`
//! records, with a PNG image texture (scene data: examples/fish.rs:17-63)
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
using texture::ImageTexture;
using texture::Texture;

Example fish(const std::string& assets) {
    auto fish_skin = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/fish.png")));
    auto mat_fish = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.8, 0.8, 0.8},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .texture = fish_skin,
    });

    auto fish_model = MeshData::load_obj(assets + "/fish.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Mesh::create(fish_model, Shading::Smooth), mat_fish))
                .rotated_y(Radians::from_degrees(30.0))
                .into(),

            SceneNode::from(Geometry::create(Mesh::create(fish_model, Shading::Smooth), mat_fish))
                .rotated_y(Radians::from_degrees(210.0))
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 0.0, 10.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 0.0, 11.0},
        .center = Vec3{0.0, 0.0, 0.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(25.0),
    };

    return Example{std::move(scene), cam, 910, 512, "fish.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::fish("assets")); }
#endif
