//! A castle of primitives in a forest of cylinder-and-cone trees (scene data: examples/primitives.rs:17-250)
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cone;
using primitive::Cube;
using primitive::Cylinder;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Plane;
using primitive::Shading;
using primitive::Sphere;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

static SceneNode make_castle(const std::string& assets) {
    auto mat_dome = std::make_shared<Material>(Material{.diffuse = Rgb{0.609065, 0.731162, 0.8}, .specular = Rgb{0.5, 0.5, 0.5}, .shininess = 1000.0, .reflectivity = 0.3});
    auto mat_castle = std::make_shared<Material>(Material{.diffuse = Rgb{0.769051, 0.304112, 0.8}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_castle_tower_top = std::make_shared<Material>(Material{.diffuse = Rgb{0.352613, 0.42773, 0.8}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_castle_door = std::make_shared<Material>(Material{.diffuse = Rgb{0.176099, 0.115632, 0.054921}});
    auto mat_road = std::make_shared<Material>(Material{.diffuse = Rgb{0.121484, 0.024035, 0.0}});

    std::vector<Arc<SceneNode>> nodes;

    const double castle_width = 4.0;
    const double castle_length = castle_width;
    const double castle_height = 2.0;
    const double dome_radius = castle_width / 2.0;
    const double tower_height = castle_height * 1.5;
    const double tower_width = 1.5;
    const double tower_roof_height = 2.0;
    const double tower_roof_width = tower_width + 0.1;

    // Main castle body
    nodes.push_back(
        SceneNode::from(Geometry::create(Cube{}, mat_castle))
            .scaled({castle_width, castle_height, castle_length})
            .translated({0.0, castle_height / 2.0, 0.0})
            .into());

    // Castle dome
    nodes.push_back(
        SceneNode::from(Geometry::create(Sphere{}, mat_dome))
            .scaled({dome_radius, castle_height, dome_radius})
            .translated({0.0, castle_height, 0.0})
            .into());

    // Castle door
    auto castle_door_model = MeshData::load_obj(assets + "/prim_castle_door.obj");
    nodes.push_back(
        SceneNode::from(Geometry::create(Mesh::create(castle_door_model, Shading::Smooth), mat_castle_door))
            .translated({0.0, 1.1, castle_length / 2.0 + 0.1})
            .into());

    // Road
    nodes.push_back(
        SceneNode::from(Geometry::create(Cube{}, mat_road))
            .scaled({2.0, 0.01, 4.0})
            .translated({0.0, 0.0, castle_length / 2.0 + 2.0 - 0.3})
            .into());

    // All 4 towers
    Arc<SceneNode> tower = SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cylinder{}, mat_castle))
            .scaled({tower_width, tower_height, tower_width})
            .translated({0.0, tower_height / 2.0, 0.0})
            .into(),
        SceneNode::from(Geometry::create(Cone{}, mat_castle_tower_top))
            .scaled({tower_roof_width, tower_roof_height, tower_roof_width})
            .translated({0.0, tower_height + tower_roof_height / 2.0, 0.0})
            .into(),
    }).into();

    // Castle towers
    for (double x : {-1.0, 1.0}) {
        for (double z : {-1.0, 1.0}) {
            Vec3 tower_pos{castle_width / 2.0 * x, 0.0, castle_length / 2.0 * z};
            nodes.push_back(
                SceneNode::from(tower)
                    .translated(tower_pos)
                    .into());
        }
    }

    return SceneNode::from(nodes);
}

static SceneNode make_trees() {
    auto mat_tree_leaves = std::make_shared<Material>(Material{.diffuse = Rgb{0.289596, 0.8, 0.308959}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
    auto mat_tree_trunk = std::make_shared<Material>(Material{.diffuse = Rgb{0.8, 0.441708, 0.115746}});

    Arc<SceneNode> tree = SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cylinder{}, mat_tree_trunk))
            .scaled({0.3, 2.0, 0.3})
            .translated({0.0, 1.0, 0.0})
            .into(),
        SceneNode::from(Geometry::create(Cone{}, mat_tree_leaves))
            .scaled({1.0, 2.0, 1.0})
            .translated({0.0, 2.9, 0.0})
            .into(),
    }).into();

    const Vec3 tree_positions[] = {
        // Trees to the right of the camera
        Vec3{4.225878, 0.0, 3.695781},
        Vec3{5.225877, 0.0, 2.895781},
        Vec3{4.125877, 0.0, 2.395781},
        Vec3{5.125877, 0.0, 1.595781},
        Vec3{6.525877, 0.0, 0.795781},
        Vec3{5.125877, 0.0, 0.395781},
        Vec3{5.925876, 0.0, -0.704219},
        Vec3{4.725877, 0.0, -1.30422},
        Vec3{3.425877, 0.0, -0.804219},
        Vec3{3.025877, 0.0, -2.204219},
        Vec3{4.225877, 0.0, -2.30422},
        Vec3{5.425877, 0.0, -2.50422},
        Vec3{6.525876, 0.0, -2.00422},
        Vec3{6.925876, 0.0, -3.50422},
        Vec3{5.825876, 0.0, -3.90422},
        Vec3{4.625876, 0.0, -3.70422},
        Vec3{3.425876, 0.0, -3.40422},
        Vec3{3.625876, 0.0, -4.80422},
        Vec3{5.025876, 0.0, -5.10422},
        Vec3{6.825876, 0.0, -5.00422},
        // Trees to the left of the camera
        Vec3{-3.374122, 0.0, 3.79578},
        Vec3{-4.874123, 0.0, 3.29578},
        Vec3{-2.874123, 0.0, 2.39578},
        Vec3{-4.374123, 0.0, 2.19578},
        Vec3{-5.674122, 0.0, 1.79578},
        Vec3{-5.974123, 0.0, 0.195781},
        Vec3{-4.674122, 0.0, 0.395781},
        Vec3{-3.574123, 0.0, 1.09578},
        Vec3{-3.274122, 0.0, -0.204219},
        Vec3{-4.674122, 0.0, -1.00422},
        Vec3{-5.874123, 0.0, -1.20422},
        Vec3{-5.874123, 0.0, -2.40422},
        Vec3{-4.574122, 0.0, -2.40422},
        Vec3{-3.474122, 0.0, -1.70422},
        Vec3{-3.574123, 0.0, -3.30422},
        Vec3{-5.374123, 0.0, -3.60422},
    };

    Arc<SceneNode> fallen_tree = SceneNode::from(tree)
        .rotated_xzy(Radians::from_degrees(0.0), Radians::from_degrees(50.0), Radians::from_degrees(-80.0))
        .translated({2.285154, 0.13965, 2.474418})
        .into();

    std::vector<Arc<SceneNode>> nodes;
    for (const Vec3& tree_pos : tree_positions)
        nodes.push_back(SceneNode::from(tree)
            .translated(tree_pos)
            .into());
    nodes.push_back(fallen_tree);
    return SceneNode::from(nodes);
}

Example primitives(const std::string& assets) {
    auto mat_grass = std::make_shared<Material>(Material{.diffuse = Rgb{0.177353, 0.334328, 0.169638}});

    Arc<SceneNode> castle = make_castle(assets)
        .translated({0.0, 0.0, -1.6})
        .into();
    Arc<SceneNode> trees = make_trees()
        .into();

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            castle,
            trees,

            // Floor
            SceneNode::from(Geometry::create(Plane{}, mat_grass))
                .scaled(30.0)
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 10.0, 9.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 4.311144, 17.370693},
        .center = Vec3{0.0, 2.133119, -7.534255},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(25.0),
    };

    return Example{std::move(scene), cam, 910, 512, "primitives.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::primitives("assets")); }
#endif
