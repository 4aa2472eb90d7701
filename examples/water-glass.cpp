//! A glass of water with a straw on a glossy wooden table in front of a brick wall: refraction through a
//! cylinder, glossy reflection on a textured, normal-mapped surface (scene data: examples/water-glass.rs:17-117)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using material::WATER_REFRACTION_INDEX;
using light::Light;
using primitive::Cube;
using primitive::Cylinder;
using primitive::Plane;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using texture::ImageTexture;
using texture::NormalMap;
using texture::Texture;

namespace {
SceneNode wg_room(const std::string& assets) {
    auto brick = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Brick_Wall_013_COLOR.jpg")));
    auto brick_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/Brick_Wall_013_NORM.jpg"));
    auto mat_wall = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .texture = brick,
        .normals = brick_normals,
    });

    auto wood = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Wood_018_basecolor_cubemap.jpg")));
    auto wood_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/Wood_018_normal_cubemap.jpg"));
    auto mat_table = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .reflectivity = 0.2,
        .glossy_side_length = 2.0,
        .texture = wood,
        .normals = wood_normals,
    });

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Plane{}, mat_wall))
            .scaled(10.0)
            .rotated_x(Radians::from_degrees(90.0))
            .translated({0.0, 1.0, -2.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_table))
            .scaled({8.0, 0.4, 4.0})
            .translated({0.0, 0.0, -0.2})
            .into(),
    });
}

SceneNode wg_drink() {
    auto mat_water = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.0, 0.0, 0.1},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .reflectivity = 0.9,
        .refraction_index = WATER_REFRACTION_INDEX,
    });
    auto mat_straw = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.0, 0.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cylinder{}, mat_water))
            .scaled({1.0, 1.4, 1.0})
            .translated({0.0, 0.7, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Cylinder{}, mat_straw))
            .scaled({0.1, 2.0, 0.1})
            .rotated_z(Radians::from_degrees(28.4282))
            .translated({-0.165556, 0.911109, 0.1})
            .into(),
    });
}
}  // namespace

Example water_glass(const std::string& assets) {
    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            wg_room(assets).into(),
            wg_drink().translated({0.0, 0.2, 0.0}).into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 27.0, 5.0}, .color = Rgb{0.5, 0.5, 0.5}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 3.2, 7.151111},
        .center = Vec3{0.0, 0.091525, -5.719519},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(23.0),
    };

    return Example{std::move(scene), cam, 910, 512, "water-glass.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::water_glass("assets")); }
#endif
