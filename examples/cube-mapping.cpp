//! The six faces of a cube map shown by four rotated cubes above a mirror (scene data: examples/cube-mapping.rs)
//! NOTE: the script opens assets/earth_cube.png, which the reference repository does not contain: like the reference's
//! `ImageTexture::open(..)?`, building this scene fails until that file is supplied.
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
using primitive::Plane;
using primitive::Sphere;
using texture::ImageTexture;
using texture::Texture;

Example cube_mapping(const std::string& assets) {
    auto mat_mirror = std::make_shared<Material>(Material{.diffuse = Rgb{0.0, 0.0, 0.0}, .specular = Rgb{0.6, 0.6, 0.6}, .shininess = 1000.0, .reflectivity = 1.0});
    auto mat_wood = std::make_shared<Material>(Material{.diffuse = Rgb{0.545, 0.353, 0.169}, .specular = Rgb{0.5, 0.7, 0.5}, .shininess = 25.0});

    auto earth = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/earth.jpg")));
    auto mat_tex = std::make_shared<Material>(Material{.diffuse = Rgb{0.506, 0.78, 0.518}, .specular = Rgb{0.5, 0.5, 0.5}, .shininess = 25.0, .texture = earth});

    auto earth_cubemap = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/earth_cube.png")));
    auto mat_tex_cube = std::make_shared<Material>(Material{.diffuse = Rgb{0.506, 0.78, 0.518}, .specular = Rgb{0.5, 0.5, 0.5}, .shininess = 25.0, .texture = earth_cubemap});

    Arc<SceneNode> mirror = SceneNode::from(Geometry::create(Cube{}, mat_wood))
        .scaled({9.0, 0.5, 6.0})
        .rotated_x(Radians::from_degrees(10.0))
        .with_child(
            SceneNode::from(Geometry::create(Cube{}, mat_mirror))
                .scaled({8.1 / 9.0, 0.05 / 0.5, 5.4 / 6.0})
                .translated({0.0, 0.27 / 0.5, 0.0})
                .into())
        .into();

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            mirror,

            SceneNode::from(Geometry::create(Plane{}, mat_tex))
                .scaled({8.0, 1.0, 2.0})
                .rotated_x(Radians::from_degrees(90.0))
                .translated({0.0, 2.0, -2.0})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, mat_tex_cube))
                .scaled(1.5)
                .translated({-3.75, 2.0, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, mat_tex_cube))
                .scaled(1.5)
                .rotated_y(Radians::from_degrees(-90.0))
                .translated({-1.25, 2.0, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, mat_tex_cube))
                .scaled(1.5)
                .rotated_y(Radians::from_degrees(180.0))
                .translated({1.25, 2.0, 0.0})
                .into(),

            SceneNode::from(Geometry::create(Cube{}, mat_tex_cube))
                .scaled(1.5)
                .rotated_y(Radians::from_degrees(-270.0))
                .translated({3.75, 2.0, 0.0})
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{-6.0, 5.0, 4.0}, .color = Rgb{0.5, 0.5, 0.5}},
            Light{.position = Vec3{6.0, 5.0, 4.0}, .color = Rgb{0.5, 0.5, 0.5}},
            Light{.position = Vec3{0.0, 1.0, -4.0}, .color = Rgb{0.5, 0.5, 0.5}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 10.15667, 11.579666},
        .center = Vec3{0.0, -5.913023, -7.571445},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(25.0),
    };

    return Example{std::move(scene), cam, 910, 512, "cube-mapping.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::cube_mapping("assets")); }
#endif
