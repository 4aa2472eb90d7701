//! "Monkeys Making Better Monkeys": a monkey at a desk with a computer (scene data: examples/monkeys-making-monkeys.rs:18-389)
//! NOTE: the script opens assets/cpu_cubemap.png, which the reference repository does not contain: like the reference's
//! `ImageTexture::open(..)?`, building this scene fails until that file is supplied.
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using light::Parallelogram;
using primitive::Cone;
using primitive::Cube;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Plane;
using primitive::Shading;
using primitive::Sphere;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using texture::ImageTexture;
using texture::NormalMap;
using texture::Texture;

static SceneNode room() {
    auto mat_floor = std::make_shared<Material>(Material{.diffuse = Rgb{0.655758, 0.8, 0.753899}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_walls = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.680366, 0.555109}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 25.0});

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Ground
        SceneNode::from(Geometry::create(Plane{}, mat_floor))
            .scaled(16.0)
            .translated({0.0, 0.0, 3.708507})
            .into(),
        // Left wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled(16.0)
            .rotated_z(Radians::from_degrees(-90.0))
            .translated({-6.340487, 5.0, 4.199467})
            .into(),
        // Right wall
        SceneNode::from(Geometry::create(Plane{}, mat_walls))
            .scaled(16.0)
            .rotated_x(Radians::from_degrees(90.0))
            .translated({0.0, 5.0, -3.2})
            .into(),
    });
}

static SceneNode wall_decor(const std::string& assets) {
    auto mat_poster = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.329194, 0.120657}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 25.0});

    auto painting = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/four-shapes.png")));
    auto mat_painting = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.2, 0.2, 0.2},
        .shininess = 25.0,
        .texture = painting,
    });

    auto mat_canvas = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.8, 0.8}, .specular = Rgb{0.2, 0.2, 0.2}, .shininess = 25.0});

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Poster
        SceneNode::from(Geometry::create(Plane{}, mat_poster))
            .scaled(4.74905)
            .rotated_z(Radians::from_degrees(-90.0))
            .translated({-6.330487, 8.043096, 3.401992})
            .into(),

        // Canvas painting (right wall)
        SceneNode::from(Geometry::create(Plane{}, mat_painting))
            .scaled({6.0, 1.0, 1.6})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({-1.0, 10.2, -3.095})
            .into(),
        SceneNode::from(Geometry::create(Cube{}, mat_canvas))
            .scaled({6.0, 1.6, 0.2})
            .translated({-1.0, 10.2, -3.2})
            .into(),
    });
}

static SceneNode desk(const std::string& assets) {
    auto wood = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Wood_018_basecolor_cubemap.jpg")));
    auto wood_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/Wood_018_normal_cubemap.jpg"));
    auto mat_desk = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .reflectivity = 0.2,
        .glossy_side_length = 2.0,
        .texture = wood,
        .normals = wood_normals,
    });

    std::vector<Arc<SceneNode>> nodes;

    // Table top
    nodes.push_back(
        SceneNode::from(Geometry::create(Cube{}, mat_desk))
            .scaled({8.0, 0.5, 6.0})
            .translated({0.0, 5.0, 0.0})
            .into());

    // Table legs
    for (double x : {-3.5, 3.5}) {
        for (double z : {-2.517656, 2.517656}) {
            const double y = 2.54158;
            nodes.push_back(
                SceneNode::from(Geometry::create(Cube{}, mat_desk))
                    .scaled({0.470548, 4.8, 0.470548})
                    .translated(Vec3{x, y, z})
                    .into());
        }
    }

    return SceneNode::from(nodes);
}

static SceneNode computer(const std::string& assets, const Arc<MeshData>& monkey_mesh) {
    auto cpu = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/cpu_cubemap.png")));
    auto mat_cpu = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .texture = cpu,
    });

    auto mat_computer = std::make_shared<Material>(Material{.diffuse = Rgb{0.043232, 0.043232, 0.043232}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 10.0});
    auto mat_screen = std::make_shared<Material>(Material{.diffuse = Rgb{0.655925, 0.655925, 0.655925}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 10.0});
    auto mat_screen_text = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.8, 0.8}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 10.0});
    auto mat_hologram = std::make_shared<Material>(Material{.diffuse = Rgb{0.479036, 0.8, 0.518124}, .reflectivity = 0.6, .refraction_index = material::WATER_REFRACTION_INDEX});

    auto computer_screen_base_mesh = MeshData::load_obj(assets + "/computer_screen_base.obj");
    auto computer_edge_display_mesh = MeshData::load_obj(assets + "/computer_edge_display.obj");
    auto screen_text_mesh = MeshData::load_obj(assets + "/text_monkey.3d.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // CPU
        SceneNode::from(Geometry::create(Cube{}, mat_cpu))
            .scaled({1.6, 3.0, 2.0})
            .translated({-3.0, 6.74, 0.0})
            .into(),

        // Mouse
        SceneNode::from(Geometry::create(Sphere{}, mat_computer))
            .scaled({0.28, 0.12, 0.4})
            .translated({1.411292, 5.327119, 1.857835})
            .into(),

        // Computer screen
        SceneNode::from(Geometry::create(Mesh::create(computer_screen_base_mesh, Shading::Smooth), mat_computer))
            .translated({0.0, 5.25, 0.0})
            .into(),
        SceneNode::from(Geometry::create(Mesh::create(computer_edge_display_mesh, Shading::Flat), mat_screen))
            .translated({0.0, 7.256888, 0.0})
            .into(),
        SceneNode::from(Geometry::create(Mesh::create(screen_text_mesh, Shading::Flat), mat_screen_text))
            .translated({0.0, 9.081371, 0.01})
            .into(),

        // Holographic monkey
        SceneNode::from(Geometry::create(Mesh::create(monkey_mesh, Shading::Flat), mat_hologram))
            .scaled(1.5)
            .rotated_xzy(Radians::from_degrees(-33.2668), Radians::from_degrees(8.17821), Radians::from_degrees(-8.17821))
            .translated({0.0, 7.0, 0.0})
            .into(),
    });
}

static SceneNode chair() {
    auto mat_chair = std::make_shared<Material>(Material{.diffuse = Rgb{0.032075, 0.032075, 0.032075}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Chair back
        SceneNode::from(Geometry::create(Sphere{}, mat_chair))
            .scaled({1.283107, 1.537732, 0.425492})
            .translated({0.0, 5.334378, 5.404959})
            .into(),
    });
}

static SceneNode character(const std::string& assets, const Arc<MeshData>& monkey_mesh) {
    auto mat_torso = std::make_shared<Material>(Material{.diffuse = Rgb{0.077701, 0.075793, 0.125964}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 25.0});
    auto mat_head = std::make_shared<Material>(Material{.diffuse = Rgb{0.064598, 0.270305, 0.716789}, .specular = Rgb{0.8, 0.8, 0.8}, .shininess = 25.0});

    auto monkey_torso_mesh = MeshData::load_obj(assets + "/monkey_torso.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Head
        SceneNode::from(Geometry::create(Mesh::create(monkey_mesh, Shading::Smooth), mat_head))
            .rotated_y(Radians::from_degrees(180.0))
            .translated({0.0, 7.0, 4.0})
            .into(),
        // Torso
        SceneNode::from(Geometry::create(Mesh::create(monkey_torso_mesh, Shading::Smooth), mat_torso))
            .translated({0.0, 5.148612, 4.23546})
            .into(),
        // Arm
        SceneNode::from(Geometry::create(Sphere{}, mat_torso))
            .scaled({0.282782, 1.299079, 0.282782})
            .rotated_z(Radians::from_degrees(19.0))
            .translated({0.984683, 5.126376, 4.344858})
            .into(),
    });
}

static SceneNode desk_objects(const std::string& assets) {
    auto mat_teapot = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.314666, 0.314666, 0.314666},
        .specular = Rgb{0.8, 0.8, 0.8},
        .shininess = 25.0,
        .reflectivity = 0.3,
        .glossy_side_length = 1.0,
    });
    auto mat_glass = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.0, 0.0, 0.0},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .reflectivity = 1.0,
        .refraction_index = material::OPTICAL_GLASS_REFRACTION_INDEX,
    });
    auto mat_apple = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.0, 0.0}});
    auto mat_golf_ball = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.8, 0.8, 0.8},
        .specular = Rgb{0.8, 0.8, 0.8},
        .shininess = 25.0,
        .reflectivity = 0.3,
        .glossy_side_length = 1.0,
    });
    auto mat_cone = std::make_shared<Material>(Material{.diffuse = Rgb{0.368949, 0.335492, 0.8}});

    auto teapot_mesh = MeshData::load_obj(assets + "/teapot.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Teapot
        SceneNode::from(Geometry::create(Mesh::create(teapot_mesh, Shading::Smooth), mat_teapot))
            .scaled(0.030)
            .translated({2.43888, 5.241134, -0.617814})
            .into(),
        // Glass ball
        SceneNode::from(Geometry::create(Sphere{}, mat_glass))
            .scaled(0.5)
            .translated({2.768083, 5.751237, -1.475317})
            .into(),
        // Apple
        SceneNode::from(Geometry::create(Sphere{}, mat_apple))
            .scaled(0.28)
            .translated({3.369787, 5.538453, -0.782367})
            .into(),
        // Golf Ball
        SceneNode::from(Geometry::create(Sphere{}, mat_golf_ball))
            .scaled(0.14)
            .translated({3.03616, 5.384166, -0.381234})
            .into(),
        // Cone
        SceneNode::from(Geometry::create(Cone{}, mat_cone))
            .scaled({0.64963, 1.106842, 0.64963})
            .translated({3.182365, 5.777666, -2.332999})
            .into(),
    });
}

Example monkeys_making_monkeys(const std::string& assets) {
    auto monkey_mesh = MeshData::load_obj(assets + "/monkey.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            room().into(),
            wall_decor(assets).into(),
            desk(assets).into(),
            desk_objects(assets).into(),
            computer(assets, monkey_mesh).into(),
            chair().into(),
            character(assets, monkey_mesh).into(),
        }).into(),
        .lights = {
            // Overhead light
            Light{.position = Vec3{0.0, 13.0, 1.0}, .color = Rgb{0.9, 0.9, 0.9}, .area = Parallelogram{.a = Vec3{4.0, 0.0, 0.0}, .b = Vec3{0.0, 0.0, 4.0}}},
            // Window
            Light{.position = Vec3{8.0, 8.0, 8.0}, .color = Rgb{0.4, 0.4, 0.4}, .area = Parallelogram{.a = Vec3{0.0, 0.0, 2.5}, .b = Vec3{0.0, 2.5, 0.0}}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{10.626843, 11.525522, 15.875655},
        .center = Vec3{-11.287256, 4.506533, -10.496798},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(23.0),
    };

    return Example{std::move(scene), cam, 1920, 1080, "monkeys-making-monkeys.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::monkeys_making_monkeys("assets")); }
#endif
