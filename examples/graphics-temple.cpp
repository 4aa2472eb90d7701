//! A (work-in-progress) temple of graphics by a lake (scene data: examples/graphics-temple.rs:23-461)
#include <algorithm>

#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cube;
using primitive::Cylinder;
using primitive::KDMesh;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Shading;
using primitive::Sphere;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

static std::shared_ptr<Material> placeholder_red() {
    //TODO: Replace this material
    return std::make_shared<Material>(Material{.diffuse = Rgb{1.0, 0.0, 0.0}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});
}

static SceneNode hills(const std::string& assets) {
    auto mat_grass = std::make_shared<Material>(Material{.diffuse = Rgb{0.376, 0.502, 0.22}});

    auto grass_model = MeshData::load_obj(assets + "/tog_grass.obj");

    SceneNode node = SceneNode::from(Geometry::create(KDMesh::create(grass_model, Shading::Smooth), mat_grass));
    node.translated({1.958125, 16.093138, -86.113747});
    return node;
}

static SceneNode lake(const std::string& assets) {
    auto mat_water = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.0, 0.0, 0.1},
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .reflectivity = 0.9,
        .glossy_side_length = 1.0,
        .refraction_index = material::WATER_REFRACTION_INDEX,
    });

    auto mat_dirt = std::make_shared<Material>(Material{
        // Color of algae makes the water blue!
        .diffuse = Rgb{0.592, 0.671, 0.055},
    });

    auto underwater_land_model = MeshData::load_obj(assets + "/tog_underwater_land.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cube{}, mat_water))
            .scaled({600.0, 200.0, 600.0})
            .translated({0.0, -107.0, 300.0})
            .into(),

        // Flat shaded to speed up rendering since the normals don't super matter for this (not visible)
        SceneNode::from(Geometry::create(KDMesh::create(underwater_land_model, Shading::Flat), mat_dirt))
            .translated({0.0, -107.0, 300.0})
            .into(),
    });
}

static SceneNode temple_floor_1() {
    // Generates a maze pattern around the entire floor
    const double floor_width = 240.0;
    const double floor_length = 40.0;
    const double floor_height = 20.0;

    // This number MUST evenly divide floor_width AND floor_length AND floor_height
    const double cell_size = 5.0;

    const double total_width = floor_width * 2.0 + floor_length * 2.0;
    const double total_height = floor_height;

    const size_t maze_cols = (size_t)(total_width / cell_size);
    const size_t maze_rows = (size_t)(total_height / cell_size);
    if ((double)maze_cols * cell_size != total_width) throw Panic("bug: cell size should evenly divide floor width");
    if ((double)maze_rows * cell_size != total_height) throw Panic("bug: cell size should evenly divide floor height");

    // The script fills a maze here (generate_maze: a placeholder of random cells from StdRng seed 193920103958) and then
    // loops over it WITHOUT emitting any node (graphics-temple.rs:149-175: the loop bodies only compute x and y): the
    // first floor contributes an empty group to the scene, and so it does here.
    std::vector<Arc<SceneNode>> nodes;
    return SceneNode::from(nodes);
}

/// A cylinderical column with center at its bottom middle
static SceneNode cylinder_column(const std::shared_ptr<Material>& mat_column) {
    SceneNode column = SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(Cube{}, mat_column))
            .scaled({3.2, 1.0, 3.2})
            .translated({0.0, 3.8, 0.0})
            .into(),
        SceneNode::from(Geometry::create(Cube{}, mat_column))
            .scaled({3.2, 1.0, 3.2})
            .translated({0.0, -3.8, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Sphere{}, mat_column))
            .scaled({1.5, 0.5, 1.5})
            .translated({0.0, 3.0, 0.0})
            .into(),
        SceneNode::from(Geometry::create(Sphere{}, mat_column))
            .scaled({1.5, 0.5, 1.5})
            .translated({0.0, -3.0, 0.0})
            .into(),

        SceneNode::from(Geometry::create(Cylinder{}, mat_column))
            .scaled({2.0, 6.0, 2.0})
            .into(),
    });
    column.translated({0.0, 4.3, 0.0});
    return column;
}

static SceneNode temple_floor_2() {
    // Generate a layout with equally spaced sections of a given width. Each section has a column
    // on each side
    const double floor_width = 168.0;
    const double floor_height = 20.0;
    const double floor_length = 32.0;
    const double floor_y_offset = 20.0;
    const double floor_front_z = floor_length / 2.0;

    const size_t sections = 4;
    const double section_width = 30.0;

    const double column_scale = 2.0;
    // The diameter in this case is width == length since the column has cubes at its ends
    const double column_diameter = 3.2 * column_scale;
    const double column_height = 8.6 * column_scale;

    // Compute the amount of space between each section.
    // -1 because there is only spacing *between* the sections, not at the end
    const double section_spacing = (floor_width - column_diameter - (double)sections * section_width) / (double)(sections - 1);

    std::vector<Arc<SceneNode>> nodes;

    // Generate columns to hold up the ceiling
    auto mat_column = placeholder_red();

    Arc<SceneNode> column = cylinder_column(mat_column).into();
    for (size_t i = 0; i < sections * 2; i++) {
        // Add section width on odd i
        const double x = section_width * (double)((i + 1) / 2)
              // Add section spacing on even i
              + section_spacing * (double)(i / 2)
              // Center in the image and column size
              - floor_width / 2.0 + column_diameter / 2.0;

        // Front column
        nodes.push_back(
            SceneNode::from(column)
                .scaled(column_scale)
                .translated({x, floor_y_offset, floor_front_z - column_diameter / 2.0})
                .into());

        // Back column
        nodes.push_back(
            SceneNode::from(column)
                .scaled(column_scale)
                .translated({x, floor_y_offset, -(floor_front_z - column_diameter / 2.0)})
                .into());
    }

    // The ceiling
    const double ceiling_height = floor_height - column_height;
    nodes.push_back(
        SceneNode::from(Geometry::create(Cube{}, mat_column))
            .scaled({floor_width, ceiling_height, floor_length})
            .translated({0.0, floor_y_offset + column_height + ceiling_height / 2.0, 0.0})
            .into());

    // Each section contains an "idol" or "diety" which for this floor represents a cube and the
    // three types of transformations on it
    auto mat_idol = placeholder_red();

    const double extent = std::min(section_width, column_height);
    Arc<SceneNode> base_idol =
        SceneNode::from(Geometry::create(Cube{}, mat_idol))
            .scaled(extent * 0.5)
            .rotated_y(Radians::from_degrees(30.0))
            .into();

    std::vector<SceneNode> idols;
    idols.push_back(SceneNode::from(base_idol));
    idols.push_back(SceneNode::from(base_idol));
    idols.back().scaled({1.0, 0.4, 1.0});
    idols.push_back(SceneNode::from(base_idol));
    idols.back().rotated_z(Radians::from_degrees(80.0));
    idols.push_back(SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(base_idol)
            .scaled(0.5)
            .translated({-extent / 4.0, extent / 8.0, -floor_length / 8.0})
            .into(),
        SceneNode::from(base_idol)
            .scaled(0.5)
            .translated({extent / 4.0, -extent / 8.0, floor_length / 8.0})
            .into(),
    }));
    if (idols.size() != sections) throw Panic("assertion failed: idols.len() == sections");

    for (size_t i = 0; i < idols.size(); i++) {
        const double x = section_width * (double)(i + 1) + section_spacing * (double)i
              // Center in the image and section width
              - floor_width / 2.0 - section_width / 2.0 + column_diameter / 2.0;

        nodes.push_back(
            idols[i].translated({x, floor_y_offset + column_height / 2.0, 0.0}).into());
    }

    return SceneNode::from(nodes);
}

static SceneNode temple_floor_3(const std::string& assets) {
    const double floor_width = 117.6;
    const double floor_length = 25.6;
    const double floor_height = 20.0;
    const double floor_y_offset = 40.0;

    const double puppet_height = 17.2;
    const double puppet_y_offset = 44.083061;

    const double ceiling_height = floor_height - puppet_height;
    const double ceiling_y_offset = floor_y_offset + puppet_height + ceiling_height / 2.0;

    auto mat_puppet = placeholder_red();

    auto puppet_model = MeshData::load_obj(assets + "/tog_puppet.obj");
    Arc<SceneNode> puppet = SceneNode::from(Geometry::create(KDMesh::create(puppet_model, Shading::Smooth), mat_puppet))
        .translated({0.0, puppet_y_offset, 0.0})
        .into();

    auto mat_ceiling = placeholder_red();

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // The ceiling
        SceneNode::from(Geometry::create(Cube{}, mat_ceiling))
            .scaled({floor_width, ceiling_height, floor_length})
            .translated({0.0, ceiling_y_offset, 0.0})
            .into(),

        // Left puppet
        SceneNode::from(puppet)
            .rotated_y(Radians::from_degrees(90.0))
            .translated({-55.1, 0.0, 0.0})
            .into(),

        // Center puppet
        SceneNode::from(puppet)
            .translated({0.0, 0.0, -5.0})
            .into(),

        // Right puppet
        SceneNode::from(puppet)
            .rotated_y(Radians::from_degrees(-90.0))
            .translated({55.1, 0.0, 0.0})
            .into(),
    });
}

static SceneNode temple_floor_4(const std::string& assets) {
    auto mat_crystal = placeholder_red();

    auto monkey_model = MeshData::load_obj(assets + "/monkey.obj");
    auto teapot_model = MeshData::load_obj(assets + "/teapot.obj");
    auto cow_model = MeshData::load_obj(assets + "/cow.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Monkey
        SceneNode::from(Geometry::create(Mesh::create(monkey_model, Shading::Smooth), mat_crystal))
            .scaled(8.0)
            .rotated_xzy(Radians::from_degrees(-34.9072), Radians::from_degrees(25.0), Radians::from_degrees(0.0))
            .translated({-30.0, 64.214905, 1.0})
            .into(),

        // Teapot
        SceneNode::from(Geometry::create(KDMesh::create(teapot_model, Shading::Smooth), mat_crystal))
            .scaled(0.6)
            .rotated_y(Radians::from_degrees(-55.0))
            .translated({0.0, 59.857296, 0.0})
            .into(),

        // Cow
        SceneNode::from(Geometry::create(KDMesh::create(cow_model, Shading::Smooth), mat_crystal))
            .scaled(1.5)
            .rotated_y(Radians::from_degrees(-125.0))
            .translated({30.0, 65.31517, 0.0})
            .into(),
    });
}

Example graphics_temple(const std::string& assets) {
    auto mat_temple_block = std::make_shared<Material>(Material{.diffuse = Rgb{0.913099, 0.913099, 0.715694}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            //TODO: All temple blocks should be removed by the time we're done
            SceneNode::from(Geometry::create(Cube{}, mat_temple_block))
                .scaled({240.0, 20.0, 40.0})
                .translated({0.0, 10.0, 0.0})
                .into(),

            hills(assets).into(),
            lake(assets).into(),
            temple_floor_1().into(),
            temple_floor_2().into(),
            temple_floor_3(assets).into(),
            temple_floor_4(assets).into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{190.0, 98.0, 151.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 61.971188, 546.971191},
        .center = Vec3{0.0, -13.390381, -585.524353},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(25.0),
    };

    // let mut image = Image::new("graphics-temple.png", 1920, 1080)?;
    return Example{std::move(scene), cam, 533, 300, "graphics-temple.png",
                   [](Uv uv) { return Rgb{0.529, 0.808, 0.922} * (1.0 - uv.v) + Rgb{0.086, 0.38, 0.745} * uv.v; }};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::graphics_temple("assets")); }
#endif
