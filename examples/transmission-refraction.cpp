//! Transmission and refraction: a glass pane in front of a tiled water tank with two fish, a wooden table,
//! a glass of water with a straw (scene data: examples/transmission-refraction.rs:20-264)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using material::WATER_REFRACTION_INDEX;
using material::WINDOW_GLASS_REFRACTION_INDEX;
using light::Light;
using primitive::Cube;
using primitive::Cylinder;
using primitive::KDMesh;
using primitive::MeshData;
using primitive::Plane;
using primitive::Shading;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using texture::ImageTexture;
using texture::NormalMap;
using texture::Texture;

namespace {
SceneNode room(const std::string& assets) {
    auto mat_walls = std::make_shared<Material>(Material{.diffuse = Rgb{0.607917, 0.8, 0.551884}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto wood = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Wood_018_basecolor_cubemap.jpg")));
    auto wood_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/Wood_018_normal_cubemap.jpg"));
    auto mat_table = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .texture = wood,
        .normals = wood_normals,
    });

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Wood Table
        SceneNode::from(Geometry::create(Cube{}, mat_table))
            .scaled({20.0, 5.0, 2.5})
            .translated({0.0, -2.0, 1.3})
            .into(),

        // Back wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled({20.0, 1.0, 20.0})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({0.0, 3.0, -10.0})
            .into(),

        // Right tank wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled({20.0, 1.0, 12.0})
            .rotated_z(Radians::from_degrees(90.0))
            .translated({10.0, 3.0, -6.0})
            .into(),

        // Left tank wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled({20.0, 1.0, 12.0})
            .rotated_z(Radians::from_degrees(-90.0))
            .translated({-10.0, 3.0, -6.0})
            .into(),

        // Right wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled({12.1, 1.0, 20.0})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({16.0, 3.0, 0.0})
            .into(),

        // Left wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled({12.1, 1.0, 20.0})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({-16.0, 3.0, 0.0})
            .into(),
    });
}

SceneNode tank(const std::string& assets) {
    auto tiles = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Tiles_017_basecolor_cubemap.jpg")));
    auto tiles_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/Tiles_017_normal_cubemap.jpg"));

    auto mat_tank = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .texture = tiles,
        .normals = tiles_normals,
    });

    std::vector<Arc<SceneNode>> nodes;

    // Using loops and lots of cubes to preserve aspect ratio of texture

    // Add front and back of tank
    for (int i = 0; i < 4; i++) {
        nodes.push_back(SceneNode::from(Geometry::create(Cube{}, mat_tank))
            .scaled({5.0, 5.0, 0.2})
            .translated({(double)i * 5.0 - 7.5, -2.0, -10.0})
            .into());
        nodes.push_back(SceneNode::from(Geometry::create(Cube{}, mat_tank))
            .scaled({5.0, 5.0, 0.2})
            .translated({(double)i * 5.0 - 7.5, -2.0, 0.0})
            .into());
    }

    // Add side walls
    for (int i = 0; i < 2; i++) {
        nodes.push_back(SceneNode::from(Geometry::create(Cube{}, mat_tank))
            .scaled({0.2, 5.0, 5.0})
            .translated({-10.0, -2.0, -((double)i * 5.0 + 2.5)})
            .into());
        nodes.push_back(SceneNode::from(Geometry::create(Cube{}, mat_tank))
            .scaled({0.2, 5.0, 5.0})
            .translated({10.0, -2.0, -((double)i * 5.0 + 2.5)})
            .into());
    }

    // Add bottom
    for (int x = 0; x < 4; x++)
        for (int y = 0; y < 2; y++)
            nodes.push_back(SceneNode::from(Geometry::create(Cube{}, mat_tank))
                .scaled({5.0, 0.2, 5.0})
                .translated({(double)x * 5.0 - 7.5, -4.0, -((double)y * 5.0 + 2.5)})
                .into());

    return SceneNode::from(nodes);
}

SceneNode water(const std::string& assets) {
    auto mat_water = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.0, 0.0, 0.1},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .reflectivity = 0.9,
        .refraction_index = WATER_REFRACTION_INDEX,
    });

    auto fish_skin = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/fish.png")));
    auto mat_fish = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.8, 0.8}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0, .texture = fish_skin});

    auto fish_model = MeshData::load_obj(assets + "/fish.obj");
    KDMesh fish_mesh = KDMesh::create(fish_model, Shading::Smooth);

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Water
        SceneNode::from(Geometry::create(Cube{}, mat_water))
            .scaled({19.799999, 3.8, 9.8})
            .translated({0.0, -2.0, -5.0})
            .into(),

        // Fishes
        SceneNode::from(Geometry::create(fish_mesh, mat_fish))
            .rotated_xzy(Radians::from_degrees(0.0), Radians::from_degrees(-71.8181), Radians::from_degrees(30.8927))
            .translated({-4.798946, -0.970323, -5.246493})
            .into(),
        SceneNode::from(Geometry::create(fish_mesh, mat_fish))
            .rotated_xzy(Radians::from_degrees(0.0), Radians::from_degrees(108.666), Radians::from_degrees(-23.084))
            .translated({3.110451, -2.562474, -6.838645})
            .into(),
    });
}

SceneNode drink() {
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
            .translated({-7.4, 1.2, 1.2})
            .into(),

        SceneNode::from(Geometry::create(Cylinder{}, mat_straw))
            .scaled({0.1, 2.0, 0.1})
            .rotated_z(Radians::from_degrees(28.4282))
            .translated({-7.565556, 1.411109, 1.1})
            .into(),
    });
}
}  // namespace

Example transmission_refraction(const std::string& assets) {
    auto mat_glass = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.0, 0.0, 0.0},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .reflectivity = 1.0,
        .refraction_index = WINDOW_GLASS_REFRACTION_INDEX,
    });

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            // Front glass
            SceneNode::from(Geometry::create(Cube{}, mat_glass))
                .scaled({20.0, 10.0, 0.2})
                .translated({0.0, 5.0, 0.0})
                .into(),

            room(assets).into(),
            tank(assets).into(),
            water(assets).into(),
            drink().into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 27.0, 5.0}, .color = Rgb{0.5, 0.5, 0.5}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 14.658033, 27.19817},
        .center = Vec3{0.0, -6.058867, -24.828854},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(23.0),
    };

    return Example{std::move(scene), cam, 910, 512, "transmission-refraction.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::transmission_refraction("assets")); }
#endif
