//! Texture maps (left) against texture + normal maps (right) on a plane, cubes and spheres; main() renders the
//! scene three times with the light in different places (scene data and main(): examples/normal-mapping.rs:20-177)
#include <cstdio>

#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cube;
using primitive::Plane;
using primitive::Sphere;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using texture::ImageTexture;
using texture::NormalMap;
using texture::Texture;

Example normal_mapping(const std::string& assets, Vec3 light_pos) {
    auto tex_map_plane = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Terracotta_Tiles_002_Base_Color.jpg")));
    auto norm_map_plane = std::make_shared<NormalMap>(NormalMap::open(assets + "/Terracotta_Tiles_002_Normal.jpg"));

    const Rgb diffuse{0.37168, 0.236767, 0.692066};
    auto mat_tex_plane = std::make_shared<Material>(Material{.diffuse = diffuse, .specular = Rgb{0.4, 0.4, 0.4}, .shininess = 25.0, .texture = tex_map_plane});
    auto mat_tex_plane_norm = std::make_shared<Material>(Material{.diffuse = diffuse, .specular = Rgb{0.4, 0.4, 0.4}, .shininess = 25.0,
                                                                   .texture = tex_map_plane, .normals = norm_map_plane});

    auto tex_map_sphere = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Rock_033_baseColor_2.jpg")));
    auto norm_map_sphere = std::make_shared<NormalMap>(NormalMap::open(assets + "/Rock_033_normal_2.jpg"));

    auto mat_tex_sphere = std::make_shared<Material>(Material{.diffuse = diffuse, .specular = Rgb{0.6, 0.6, 0.6}, .shininess = 25.0, .texture = tex_map_sphere});
    auto mat_tex_sphere_norm = std::make_shared<Material>(Material{.diffuse = diffuse, .specular = Rgb{0.6, 0.6, 0.6}, .shininess = 25.0,
                                                                    .texture = tex_map_sphere, .normals = norm_map_sphere});

    auto tex_map_cube = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Stone_Wall_007_COLOR_cubemap.jpg")));
    auto norm_map_cube = std::make_shared<NormalMap>(NormalMap::open(assets + "/Stone_Wall_007_NORM_cubemap.jpg"));

    auto mat_tex_cube = std::make_shared<Material>(Material{.diffuse = diffuse, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0, .texture = tex_map_cube});
    auto mat_tex_cube_norm = std::make_shared<Material>(Material{.diffuse = diffuse, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0,
                                                                  .texture = tex_map_cube, .normals = norm_map_cube});

    auto mat_wall_floor = std::make_shared<Material>(Material{.diffuse = Rgb{0.424858, 0.531206, 0.8}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    Arc<SceneNode> scene_root = SceneNode::from(std::vector<Arc<SceneNode>>{
        // Floor
        SceneNode::from(Geometry::create(Plane{}, mat_wall_floor))
            .scaled(40.0)
            .translated({0.0, -1.0, 0.0})
            .into(),

        // Left - Texture Only
        SceneNode::from(Geometry::create(Plane{}, mat_tex_plane))
            .scaled(6.0)
            .rotated_x(Radians::from_degrees(90.0))
            .translated({-4.0, 2.0, -6.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_tex_cube))
            .scaled(2.0)
            .translated({-7.0, 0.0, -1.0})
            .into(),
        SceneNode::from(Geometry::create(Sphere{}, mat_tex_sphere))
            .translated({-7.0, 2.0, -1.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_tex_cube))
            .scaled(2.0)
            .translated({-2.0, 0.0, 3.0})
            .into(),
        SceneNode::from(Geometry::create(Sphere{}, mat_tex_sphere))
            .translated({-2.0, 2.0, 3.0})
            .into(),

        // Right - Normal + Texture
        SceneNode::from(Geometry::create(Plane{}, mat_tex_plane_norm))
            .scaled(6.0)
            .rotated_x(Radians::from_degrees(90.0))
            .translated({4.0, 2.0, -6.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_tex_cube_norm))
            .scaled(2.0)
            .translated({7.0, 0.0, -1.0})
            .into(),
        SceneNode::from(Geometry::create(Sphere{}, mat_tex_sphere_norm))
            .translated({7.0, 2.0, -1.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_tex_cube_norm))
            .scaled(2.0)
            .translated({2.0, 0.0, 3.0})
            .into(),
        SceneNode::from(Geometry::create(Sphere{}, mat_tex_sphere_norm))
            .translated({2.0, 2.0, 3.0})
            .into(),
    }).into();

    HierScene scene{
        .root = scene_root,
        .lights = {
            Light{.position = light_pos, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.2, 0.2, 0.2},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 8.07551, 23.078941},
        .center = Vec3{0.0, -2.854475, -16.437334},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(22.0),
    };

    return Example{std::move(scene), cam, 910, 512, "normal-mapping.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() {
    using namespace portrayer;
    const struct { const char* path; math::Vec3 light_pos; } image_configs[] = {
        {"normal-mapping.png", math::Vec3{0.0, 8.0, 10.0}},
        {"normal-mapping-left.png", math::Vec3{-8.0, 8.0, 10.0}},
        {"normal-mapping-right.png", math::Vec3{8.0, 8.0, 10.0}},
    };
    for (const auto& cfg : image_configs) {
        std::printf("Generating %s...\n", cfg.path);
        examples::Example ex = examples::normal_mapping("assets", cfg.light_pos);
        ex.output = cfg.path;
        if (int rc = examples::run_main(std::move(ex))) return rc;
    }
    return 0;
}
#endif
