//! A robot alarm clock on a wooden table (scene data: examples/robot-alarm-clock.rs:19-442)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using light::Parallelogram;
using primitive::Cube;
using primitive::KDMesh;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Plane;
using primitive::Shading;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using texture::ImageTexture;
using texture::NormalMap;
using texture::Texture;
using Mat = std::shared_ptr<Material>;

static Arc<MeshData> model(const std::string& assets, const char* name) { return MeshData::load_obj(assets + "/robot-alarm-clock/" + name); }

static SceneNode room(const std::string& assets) {
    auto wallpaper = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/robot-alarm-clock/wallpaper.jpg")));
    auto mat_wall = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 25.0,
        .texture = wallpaper,
        .uv_trans = Mat3::scaling_3d(3.0),
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
            .scaled(20.0)
            .rotated_x(Radians::from_degrees(90.0))
            .translated({-2.0, 8.0, -5.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_table))
            .scaled({20.0, 1.0, 10.0})
            .translated({-2.0, 0.0, 0.0})
            .into(),
    });
}

static SceneNode clock(const std::string& assets) {
    auto mat_clock_case = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 1.0, 1.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_time_bg = std::make_shared<Material>(Material{.diffuse = Rgb{0.059252, 0.059252, 0.059252}});
    auto mat_time = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 0.0, 0.0}});

    auto clock_case_model = model(assets, "robot_base_clock_case.obj");
    auto clock_time_model = model(assets, "robot_base_clock_time.obj");

    const double angle = -6.62911;
    return SceneNode::from(std::vector<Arc<SceneNode>>{
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(clock_case_model, Shading::Smooth), mat_clock_case))
            .rotated_x(Radians::from_degrees(angle))
            .translated({0.0, 1.228179, 0.350087})
            .into(),

        SceneNode::from(Geometry::create(Plane{}, mat_time_bg))
            .scaled({2.966855, 1.0, 0.684205})
            .rotated_x(Radians::from_degrees(90.0 + angle))
            .translated({0.0, 1.294323, 0.919223})
            .into(),

        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(clock_time_model, Shading::Flat), mat_time))
            .rotated_x(Radians::from_degrees(83.2518 - 90.0))
            .translated({0.0, 1.535768, 0.921095})
            .into(),
    });
}

static SceneNode clock_buttons(const std::string& assets) {
    auto mat_clock_button = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.103095, 0.086502}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto clock_button_model = model(assets, "robot_base_clock_button.obj");
    //TODO: KDMesh doesn't work for this for some reason...
    Arc<SceneNode> clock_button = SceneNode::from(Geometry::create(Mesh::create(clock_button_model, Shading::Smooth), mat_clock_button)).into();

    std::vector<Arc<SceneNode>> nodes;
    for (double x : {-1.2, -0.4, 0.4, 1.2})
        nodes.push_back(
            SceneNode::from(clock_button)
                .rotated_x(Radians::from_degrees(15.0))
                .translated({x, 1.7, -0.2})
                .into());

    return SceneNode::from(nodes);
}

static SceneNode connectors(const std::string& assets, const char* obj, const Mat& mat_connector, std::initializer_list<double> x_values, int count, double y_offset) {
    const double height = 0.2;

    auto connector_model = model(assets, obj);
    Arc<SceneNode> connector = SceneNode::from(Geometry::create(KDMesh::create(connector_model, Shading::Flat), mat_connector)).into();

    std::vector<Arc<SceneNode>> nodes;
    for (double x : x_values)
        for (int i = 0; i < count; i++) {
            const double y = y_offset + (double)i * height;
            nodes.push_back(
                SceneNode::from(connector)
                    .translated({x, y, -0.712655})
                    .into());
        }

    return SceneNode::from(nodes);
}
static SceneNode base_connectors(const std::string& assets, const Mat& mat_connector) {  // robot-alarm-clock.rs:214-233
    return connectors(assets, "robot_base_connector.obj", mat_connector, {0.0}, 5, 1.960454);
}
static SceneNode torso_connectors(const std::string& assets, const Mat& mat_connector) {  // :331-350
    return connectors(assets, "robot_torso_connector.obj", mat_connector, {0.0}, 4, 4.783508);
}
static SceneNode head_connectors(const std::string& assets, const Mat& mat_connector) {  // :420-442
    return connectors(assets, "robot_head_connector.obj", mat_connector, {-0.6, 0.6}, 3, 6.583508);
}

static SceneNode robot_base(const std::string& assets, const Mat& mat_robot_metal, const Mat& mat_connector) {
    auto robot_base_model = model(assets, "robot_base.obj");
    auto robot_base_sides_model = model(assets, "robot_base_sides.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(KDMesh::create(robot_base_model, Shading::Smooth), mat_robot_metal))
            .translated({0.0, 1.002795, -0.209603})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(robot_base_sides_model, Shading::Flat), mat_robot_metal))
            .translated({0.0, 1.002795, -0.209603})
            .into(),

        clock(assets)
            .into(),
        clock_buttons(assets)
            .into(),
        base_connectors(assets, mat_connector)
            .into(),
    });
}

static SceneNode arm_sockets(const std::string& assets) {
    auto mat_arm_socket = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 1.0, 1.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto arm_socket_model = model(assets, "robot_arm_socket.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(arm_socket_model, Shading::Smooth), mat_arm_socket))
            .translated({2.1, 3.8, -0.7})
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(arm_socket_model, Shading::Smooth), mat_arm_socket))
            .rotated_y(Radians::from_degrees(180.0))
            .translated({-2.1, 3.8, -0.7})
            .into(),
    });
}

static SceneNode arms(const std::string& assets, const Mat& mat_robot_metal) {
    auto mat_hand = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 1.0, 1.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto arm_left_model = model(assets, "robot_arm_left.obj");
    auto arm_right_model = model(assets, "robot_arm_right.obj");
    auto hand_left_model = model(assets, "robot_hand_left.obj");
    auto hand_right_model = model(assets, "robot_hand_right.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(arm_left_model, Shading::Smooth), mat_robot_metal))
            .translated({2.1, 3.8, -0.7})
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(arm_right_model, Shading::Smooth), mat_robot_metal))
            .translated({-2.1, 3.8, -0.7})
            .into(),

        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(hand_left_model, Shading::Smooth), mat_hand))
            .translated({2.95, 5.45, -0.7})
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(hand_right_model, Shading::Smooth), mat_hand))
            .translated({-2.95, 5.45, -0.7})
            .into(),
    });
}

static SceneNode robot_torso(const std::string& assets, const Mat& mat_robot_metal, const Mat& mat_connector) {
    auto robot_torso_model = model(assets, "robot_torso.obj");
    auto robot_torso_sides_model = model(assets, "robot_torso_sides.obj");
    auto robot_torso_display_model = model(assets, "robot_torso_display.obj");
    auto robot_torso_text_model = model(assets, "robot_torso_text.obj");

    auto mat_torso_display = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.204899, 0.066919, 0.086002},
        .reflectivity = 0.1,
        .refraction_index = material::OPTICAL_GLASS_REFRACTION_INDEX,
    });
    auto mat_torso_text = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 0.0, 0.0}});

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(KDMesh::create(robot_torso_model, Shading::Smooth), mat_robot_metal))
            .translated({0.0, 3.781665, -0.7})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(robot_torso_sides_model, Shading::Flat), mat_robot_metal))
            .translated({0.0, 3.781665, -0.7})
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(robot_torso_display_model, Shading::Smooth), mat_torso_display))
            .translated({0.0, 3.828179, -0.255186})
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(robot_torso_text_model, Shading::Flat), mat_torso_text))
            .translated({-0.016937, 3.806762, 0.040324})
            .into(),

        arm_sockets(assets).into(),
        arms(assets, mat_robot_metal).into(),
        torso_connectors(assets, mat_connector).into(),
    });
}

static SceneNode robot_head(const std::string& assets, const Mat& mat_robot_metal, const Mat& mat_connector) {
    auto mat_smile = std::make_shared<Material>(Material{.diffuse = Rgb{0.0, 0.0, 0.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_eyeball = std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 1.0, 1.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_pupil = std::make_shared<Material>(Material{.diffuse = Rgb{0.0, 0.0, 0.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto robot_head_model = model(assets, "robot_head.obj");
    auto robot_head_sides_model = model(assets, "robot_head_sides.obj");
    auto robot_smile_model = model(assets, "robot_smile.obj");
    auto robot_eyeball_model = model(assets, "robot_eyeball.obj");
    auto robot_pupil_model = model(assets, "robot_pupil.obj");

    Arc<SceneNode> eyeball = SceneNode::from(std::vector<Arc<SceneNode>>{
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(robot_eyeball_model, Shading::Smooth), mat_eyeball))
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(robot_pupil_model, Shading::Smooth), mat_pupil))
            .into(),
    }).into();

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(KDMesh::create(robot_head_model, Shading::Smooth), mat_robot_metal))
            .translated({0.0, 5.95, -0.7})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(robot_head_sides_model, Shading::Flat), mat_robot_metal))
            .translated({0.0, 5.95, -0.7})
            .into(),
        //TODO: KDMesh doesn't work for this for some reason...
        SceneNode::from(Geometry::create(Mesh::create(robot_smile_model, Shading::Smooth), mat_smile))
            .translated({0.0, 6.137964, -0.117689})
            .into(),

        head_connectors(assets, mat_connector).into(),

        SceneNode::from(eyeball)
            .translated({-0.6, 7.53, -0.7})
            .into(),
        SceneNode::from(eyeball)
            .translated({0.6, 7.53, -0.7})
            .into(),
    });
}

static SceneNode robot(const std::string& assets) {
    auto mat_robot_metal = std::make_shared<Material>(Material{
        // diffuse: Rgb {r: 0.211857, g: 0.772537, b: 0.8971}, // Cyan
        // diffuse: Rgb {r: 0.006512, g: 0.08022, b: 0.417885}, // Dark blue
        // diffuse: Rgb {r: 0.417885, g: 0.006501, b: 0.006501}, // Red
        .diffuse = Rgb{0.006449, 0.417885, 0.025384},  // Green
        .specular = Rgb{0.8, 0.8, 0.8},
        .shininess = 100.0,
        .reflectivity = 0.3,
        .glossy_side_length = 2.0,
    });
    auto mat_connector = std::make_shared<Material>(Material{.diffuse = Rgb{0.048247, 0.048247, 0.048247}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        robot_base(assets, mat_robot_metal, mat_connector).into(),
        robot_torso(assets, mat_robot_metal, mat_connector).into(),
        robot_head(assets, mat_robot_metal, mat_connector).into(),
    });
}

Example robot_alarm_clock(const std::string& assets) {
    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            room(assets).into(),
            robot(assets).into(),
        }).into(),
        .lights = {
            // Overhead light
            Light{.position = Vec3{-2.0, 15.0, 5.0}, .color = Rgb{0.9, 0.9, 0.9}, .area = Parallelogram{.a = Vec3{5.0, 0.0, 0.0}, .b = Vec3{0.0, 0.0, 5.0}}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{1.914036, 3.826548, 20.213762},
        .center = Vec3{-3.201259, 4.146196, -14.407373},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(23.0),
    };

    return Example{std::move(scene), cam, 1920, 1080, "robot-alarm-clock.png",
                   [](Uv uv) { return Rgb{0.529, 0.808, 0.922} * (1.0 - uv.v) + Rgb{0.086, 0.38, 0.745} * uv.v; }};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::robot_alarm_clock("assets")); }
#endif
