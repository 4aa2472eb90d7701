//! A castle on a hill by a lake, surrounded by a hedge maze (scene data: examples/graphics-castle.rs:23-473)
//! NOTE: the script opens assets/shrub.png, which the reference repository does not contain: like the reference's
//! `ImageTexture::open(..)?`, building this scene fails until that file is supplied.
#include <deque>
#include <optional>
#include <set>

#include "../portrayer_amd/host/rand07.hpp"
#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Cube;
using primitive::Cylinder;
using primitive::KDMesh;
using primitive::MeshData;
using primitive::Shading;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;
using texture::ImageTexture;
using texture::NormalMap;
using texture::Texture;

static SceneNode castle(const std::string& assets) {
    auto mat_castle_walls = std::make_shared<Material>(Material{.diffuse = Rgb{0.25, 0.25, 0.25}});

    auto wood = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/old_planks_02_diff_1k.png")));
    auto wood_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/old_planks_02_nor_1k.png"));
    auto mat_castle_door = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .texture = wood,
        .normals = wood_normals,
    });

    auto mat_castle_window_frames = std::make_shared<Material>(Material{.diffuse = Rgb{0.132866, 0.132866, 0.132866}});

    auto mat_ceiling_glass = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.147337, 0.239555, 0.034547},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 100.0,
        .reflectivity = 0.8,
        .refraction_index = material::WINDOW_GLASS_REFRACTION_INDEX,
    });

    auto mat_window_glass = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.147337, 0.239555, 0.034547},
        .specular = Rgb{0.3, 0.3, 0.3},
        .shininess = 100.0,
        .reflectivity = 1.0,
        .refraction_index = material::WINDOW_GLASS_REFRACTION_INDEX,
    });

    auto mat_stairs_side = std::make_shared<Material>(Material{.diffuse = Rgb{0.132866, 0.132866, 0.132866}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto mat_tapestry = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .texture = wood,
        .normals = wood_normals,
    });

    auto mat_puppet = std::make_shared<Material>(Material{.diffuse = Rgb{0.06998, 0.06998, 0.06998}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto castle_model = MeshData::load_obj(assets + "/castle.obj");
    auto castle_window_frames_model = MeshData::load_obj(assets + "/castle_window_frames.obj");
    auto castle_glass_ceilings_model = MeshData::load_obj(assets + "/castle_glass_ceilings.obj");
    auto castle_door_model = MeshData::load_obj(assets + "/castle_door.obj");
    auto castle_door_arch_model = MeshData::load_obj(assets + "/castle_door_arch.obj");
    auto castle_tapestry_model = MeshData::load_obj(assets + "/castle_tapestry.obj");

    auto castle_stairs_side_model = MeshData::load_obj(assets + "/castle_stairs_side.obj");
    KDMesh castle_stairs_side = KDMesh::create(castle_stairs_side_model, Shading::Flat);

    auto puppet_castle_left_tower_model = MeshData::load_obj(assets + "/puppet_castle_left_tower.obj");
    auto puppet_castle_right_tower_model = MeshData::load_obj(assets + "/puppet_castle_right_tower.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        // Main castle body
        SceneNode::from(Geometry::create(KDMesh::create(castle_model, Shading::Flat), mat_castle_walls))
            .translated({0.0, 30.0, -30.0})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(castle_window_frames_model, Shading::Flat), mat_castle_window_frames))
            .translated({0.0, 83.5746, -2.25})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(castle_glass_ceilings_model, Shading::Flat), mat_ceiling_glass))
            .translated({0.0, 96.0, -23.0})
            .into(),

        // Windows
        SceneNode::from(Geometry::create(Cube{}, mat_window_glass))
            .scaled({9.1, 1.0, 12.7})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({-30.0, 70.7, 12.7})
            .into(),
        SceneNode::from(Geometry::create(Cube{}, mat_window_glass))
            .scaled({9.1, 1.0, 12.7})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({30.0, 70.7, 12.7})
            .into(),
        SceneNode::from(Geometry::create(Cube{}, mat_window_glass))
            .scaled({13.4, 1.0, 18.8})
            .rotated_x(Radians::from_degrees(90.0))
            .translated({0.0, 79.4, -2.9})
            .into(),

        // Door
        SceneNode::from(Geometry::create(KDMesh::create(castle_door_model, Shading::Flat), mat_castle_door))
            .translated({0.0, 21.739681, 10.0})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(castle_door_arch_model, Shading::Flat), mat_castle_door))
            .translated({0.0, 42.0, 9.0})
            .into(),

        // Stairs
        SceneNode::from(Geometry::create(castle_stairs_side, mat_stairs_side))
            .translated({-11.0, 5.0, 19.0})
            .into(),
        SceneNode::from(Geometry::create(castle_stairs_side, mat_stairs_side))
            .translated({11.0, 5.0, 19.0})
            .into(),

        // Statues / Guardians
        SceneNode::from(Geometry::create(KDMesh::create(puppet_castle_left_tower_model, Shading::Smooth), mat_puppet))
            .translated({30.0, 33.6, 19.0})
            .into(),
        SceneNode::from(Geometry::create(Cylinder{}, mat_castle_walls))
            .scaled(10.0)
            .translated({30.0, 5.0, 20.0})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(puppet_castle_right_tower_model, Shading::Smooth), mat_puppet))
            .translated({-30.0, 33.6, 19.0})
            .into(),
        SceneNode::from(Geometry::create(Cylinder{}, mat_castle_walls))
            .scaled(10.0)
            .translated({-30.0, 5.0, 20.0})
            .into(),

        // Tapestries
        SceneNode::from(Geometry::create(KDMesh::create(castle_tapestry_model, Shading::Smooth), mat_tapestry))
            .translated({60.0, 37.0, 10.0})
            .into(),
        SceneNode::from(Geometry::create(KDMesh::create(castle_tapestry_model, Shading::Smooth), mat_tapestry))
            .translated({-60.0, 37.0, 10.0})
            .into(),
    });
}

static SceneNode lake(const std::string& assets) {
    auto mat_water = std::make_shared<Material>(Material{
        .diffuse = Rgb{0.0, 0.0, 0.1},
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .reflectivity = 0.9,
        .glossy_side_length = 0.5,
        .refraction_index = material::WATER_REFRACTION_INDEX,
    });

    auto dock = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/Wood_018_basecolor_cubemap.jpg")));
    auto dock_normals = std::make_shared<NormalMap>(NormalMap::open(assets + "/Wood_018_normal_cubemap.jpg"));
    auto mat_dock = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .specular = Rgb{0.5, 0.5, 0.5},
        .shininess = 100.0,
        .texture = dock,
        .normals = dock_normals,
    });

    auto mat_dirt = std::make_shared<Material>(Material{
        // Color of algae makes the water blue!
        .diffuse = Rgb{0.592, 0.671, 0.055},
    });

    auto castle_water_dirt_model = MeshData::load_obj(assets + "/castle_water_dirt.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(KDMesh::create(castle_water_dirt_model, Shading::Flat), mat_dirt))
            .translated({0.0, -62.0, 125.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_water))
            .scaled({640.0, 125.0, 250.0})
            .translated({0.0, -62.0, 125.0})
            .into(),

        // Dock
        SceneNode::from(Geometry::create(Cube{}, mat_dock))
            .scaled({30.0, 4.0, 36.0})
            .translated({-100.0, 0.0, 18.0})
            .into(),
    });
}

static SceneNode land(const std::string& assets) {
    auto mat_grass = std::make_shared<Material>(Material{.diffuse = Rgb{0.116971, 0.278894, 0.0}});

    auto castle_hill_model = MeshData::load_obj(assets + "/castle_hill.obj");

    return SceneNode::from(std::vector<Arc<SceneNode>>{
        SceneNode::from(Geometry::create(KDMesh::create(castle_hill_model, Shading::Smooth), mat_grass))
            .translated({0.0, 3.75, -15.75})
            .scaled(1.4)
            .translated({0.0, 0.0, -229.0})
            .into(),

        SceneNode::from(Geometry::create(Cube{}, mat_grass))
            .scaled({2560.0, 132.0, 1040.0})
            .translated({0.0, -65.0, -520.0})
            .into(),
    });
}

namespace {
enum class Cell { Empty, Wall };
using Pos = std::pair<size_t, size_t>;

struct Maze {
    /// The rows of the maze, stored row-wise
    std::vector<std::vector<Cell>> cells;

    Maze(size_t rows, size_t cols) : cells(rows, std::vector<Cell>(cols, Cell::Wall)) {
        // Rest of the code relies on these being non-empty
        if (!(rows > 0 && cols > 0)) throw Panic("assertion failed: rows > 0 && cols > 0");
    }

    /// Reserves the given range of cells so that no walls will be placed there.
    ///
    /// The ranges are inclusive on both ends.
    void reserve(Pos a, Pos b) {
        for (size_t row = a.first; row <= b.first; row++)
            for (size_t col = a.second; col <= b.second; col++) cells[row][col] = Cell::Empty;
    }

    /// Generate the maze by filling the cells starting at the given point
    void fill_maze(Pos start) {
        const size_t rows = cells.size();
        const size_t cols = cells[0].size();
        using Adj = std::array<std::optional<Pos>, 4>;

        // Utility function for finding the adjacents of a given cell and storing the result in a
        // pre-allocated array
        auto find_adjacents = [&](Adj& adjacents, size_t row, size_t col) {
            // Leave the first and last row/column untouched
            adjacents[0] = row > 1 ? std::optional<Pos>({row - 1, col}) : std::nullopt;
            adjacents[1] = row < rows - 2 ? std::optional<Pos>({row + 1, col}) : std::nullopt;
            adjacents[2] = col > 1 ? std::optional<Pos>({row, col - 1}) : std::nullopt;
            adjacents[3] = col < cols - 2 ? std::optional<Pos>({row, col + 1}) : std::nullopt;
        };

        // Utility function for finding the diagonal adjacents of a given cell and storing the
        // result in a pre-allocated array
        auto find_diagonal_adjacents = [&](Adj& adjacents, size_t row, size_t col) {
            // Leave the first and last row/column untouched
            adjacents[0] = (row > 1 && col > 1) ? std::optional<Pos>({row - 1, col - 1}) : std::nullopt;
            adjacents[1] = (row < rows - 2 && col > 1) ? std::optional<Pos>({row + 1, col - 1}) : std::nullopt;
            adjacents[2] = (row > 1 && col < cols - 2) ? std::optional<Pos>({row - 1, col + 1}) : std::nullopt;
            adjacents[3] = (row < rows - 2 && col < cols - 2) ? std::optional<Pos>({row + 1, col + 1}) : std::nullopt;
        };
        auto count_empty = [&](const Adj& adjacents) {
            size_t n = 0;
            for (const auto& a : adjacents)
                if (a && cells[a->first][a->second] == Cell::Empty) n++;
            return n;
        };

        // Want a random maze but want the same one every time
        auto rng = rand07::StdRng::seed_from_u64(19392103958ull);

        // Reuse memory to store adjacents
        Adj adjacents{};

        std::deque<Pos> walls;
        std::set<Pos> seen;

        // Set the start cell to empty and explore its adjacents
        cells[start.first][start.second] = Cell::Empty;
        find_adjacents(adjacents, start.first, start.second);
        for (const auto& a : adjacents)
            if (a) walls.push_back(*a);

        while (!walls.empty()) {
            const auto [row, col] = walls.front();
            walls.pop_front();
            if (seen.count({row, col})) continue;
            seen.insert({row, col});

            if (cells[row][col] == Cell::Empty) {
                // Cell is probably reserved
                continue;
            }

            // Diagonal lines of empty cells look ugly, so we filter them out
            find_diagonal_adjacents(adjacents, row, col);
            if (count_empty(adjacents) > 1) continue;

            // Compute adjacents later so we can reuse them
            find_adjacents(adjacents, row, col);
            // Don't want to inadvertantly create any loops
            if (count_empty(adjacents) > 1) continue;

            // Add the cell to the maze
            cells[row][col] = Cell::Empty;

            // Add its adjacent walls to the queue in a random order
            rng.shuffle(adjacents);
            bool first = true;
            for (const auto& a : adjacents) {
                if (!a || cells[a->first][a->second] != Cell::Wall) continue;
                // Go depth first to create longer paths
                if (first) { walls.push_front(*a); first = false; }
                else walls.push_back(*a);
            }
        }
    }
};
}  // namespace

static SceneNode outdoor_maze(const std::string& assets) {
    // Needs to be a size that works proportionally with the rest of the scene
    const double cell_width = 12.0;
    const double cell_length = cell_width;

    // Chosen to be evenly divisible by cell_width
    const double maze_width = 1572.0;
    // Chosen to be evenly divisible by cell_length
    const double maze_length = 1284.0;
    // Constant for all cells / the whole maze
    const double maze_height = 8.0;
    const Vec3 maze_pos{-450.0, maze_height / 2.0 + 1.0, -660.0};

    // Area around the castle
    // Chosen to be evenly divisible by cell_width
    const double castle_area_width = 276.0;
    // Chosen to be evenly divisible by cell_length
    const double castle_area_length = 264.0;
    // Centered at the castle but then offset relative to maze pos (see last line of this function)
    const Vec3 castle_pos{0.0 - maze_pos.x, 0.0, -260.0 - maze_pos.z};

    // Entrance position (assumed to be in the bottom row)
    const double entrance_x = -100.0 - maze_pos.x;

    const size_t maze_cols = (size_t)(maze_width / cell_width);
    const size_t maze_rows = (size_t)(maze_length / cell_length);

    // Assume last row
    const size_t entrance_row = maze_rows - 1;
    const size_t entrance_col = (size_t)((entrance_x + maze_width / 2.0) / cell_width);

    // Find the boundary around the castle
    const size_t back_corner_row = (size_t)((castle_pos.z - castle_area_length / 2.0 + maze_length / 2.0) / cell_length);
    const size_t back_corner_col = (size_t)((castle_pos.x - castle_area_width / 2.0 + maze_width / 2.0) / cell_width);
    const size_t front_corner_row = (size_t)((castle_pos.z + castle_area_length / 2.0 + maze_length / 2.0) / cell_length);
    const size_t front_corner_col = (size_t)((castle_pos.x + castle_area_width / 2.0 + maze_width / 2.0) / cell_width);

    Maze maze(maze_rows, maze_cols);
    maze.reserve({back_corner_row, back_corner_col}, {front_corner_row, front_corner_col});
    maze.fill_maze({entrance_row, entrance_col});

    auto shrub = std::make_shared<Texture>(Texture::from(ImageTexture::open(assets + "/shrub.png")));
    auto mat_maze = std::make_shared<Material>(Material{
        // diffuse comes from texture
        .texture = shrub,
        .uv_trans = Mat3::scaling_3d({1.0, maze_height, 1.0}),
    });

    std::vector<Arc<SceneNode>> nodes;
    for (size_t i = 0; i < maze.cells.size(); i++) {
        const double z = (double)i * cell_length - maze_length / 2.0;
        for (size_t j = 0; j < maze.cells[i].size(); j++) {
            if (maze.cells[i][j] == Cell::Empty) continue;

            const double x = (double)j * cell_width - maze_width / 2.0;
            nodes.push_back(
                SceneNode::from(Geometry::create(Cube{}, mat_maze))
                    .scaled({cell_width, maze_height, cell_length})
                    .translated({x, 0.0, z})
                    .into());
        }
    }

    // Translate the maze to its correct position in the scene
    SceneNode all = SceneNode::from(nodes);
    all.translated(maze_pos);
    return all;
}

Example graphics_castle(const std::string& assets) {
    SceneNode castle_node = castle(assets);
    castle_node.scaled(1.4).translated({0.0, 0.0, -229.0});

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            castle_node.into(),
            lake(assets).into(),
            land(assets).into(),
            outdoor_maze(assets).into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{65.0, 130.0, -120.0}, .color = Rgb{0.9, 0.9, 0.9}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{110.877441, 30.43659, 373.276886},
        .center = Vec3{-412.953094, 65.409714, -1390.236328},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(24.0),
    };

    return Example{std::move(scene), cam, 1920, 1080, "graphics-castle.png",
                   [](Uv uv) { return Rgb{0.529, 0.808, 0.922} * (1.0 - uv.v) + Rgb{0.086, 0.38, 0.745} * uv.v; }};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() { return portrayer::examples::run_main(portrayer::examples::graphics_castle("assets")); }
#endif
