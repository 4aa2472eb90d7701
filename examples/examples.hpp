// The reference's scene scripts (examples/*.rs) transliterated against portrayer.hpp. Each
// function builds the scene exactly as the script's main() does; `main()` itself (render + save)
// is compiled when PORTRAYER_EXAMPLE_MAIN is defined.
#pragma once

#include <string>

#include "../portrayer_amd/host/portrayer.hpp"

namespace portrayer {
namespace examples {

// |uv| Rgb {r: 0.2, g: 0.4, b: 0.6} * (1.0 - uv.v) + Rgb::blue() * uv.v — the closure most scripts pass
inline math::Rgb sky(math::Uv uv) { return math::Rgb{0.2, 0.4, 0.6} * (1.0 - uv.v) + math::Rgb::blue() * uv.v; }
// |_| Rgb::white() (four-shapes.rs:86, graphics-poster.rs:76)
inline math::Rgb white(math::Uv) { return math::Rgb::white(); }

struct Example {
    scene::HierScene scene;
    camera::CameraSettings cam;
    size_t width, height;
    std::string output;
    render::Background background = sky;  // the closure the script passes to image.render
};

Example single_triangle();                                       // examples/single-triangle.rs
Example primitives_simple();                                     // examples/primitives-simple.rs
Example macho_cows(const std::string& assets_dir);               // examples/macho-cows.rs
Example entering_the_mirror_dimension(const std::string& assets_dir);  // examples/entering-the-mirror-dimension.rs
Example big_scene(int n = 10);                                   // examples/big-scene.rs (n = objects per axis)
Example synthetic_big_mesh(const std::string& assets_dir, int n = 6);  // the big-scene generator over Mesh(cow.obj) instances (not a reference scene)
Example synthetic_big_soup(const std::string& assets_dir, int n = 6);  // the same instances baked to one 1.25 M-triangle mesh
Example smooth_shading(const std::string& assets_dir);           // examples/smooth-shading.rs
Example glossy_reflection();                                     // examples/glossy-reflection.rs
Example soft_shadows(const std::string& assets_dir);             // examples/soft-shadows.rs
Example hier(const std::string& assets_dir);                     // examples/hier.rs
Example instance(const std::string& assets_dir);                 // examples/instance.rs
Example fish(const std::string& assets_dir);                     // examples/fish.rs (PNG texture on a mesh with `vt` records)
Example normal_mapping(const std::string& assets_dir, math::Vec3 light_pos = math::Vec3{0.0, 8.0, 10.0});  // examples/normal-mapping.rs (JPEG textures + normal maps; its main() renders three light positions)
Example transmission_refraction(const std::string& assets_dir);  // examples/transmission-refraction.rs (glass, water, textured KDMesh fish, normal-mapped cubes)
Example water_glass(const std::string& assets_dir);              // examples/water-glass.rs (glossy, textured, normal-mapped table; water cylinder)
Example antialiasing(const std::string& assets_dir);             // examples/antialiasing.rs (its main() renders twice)
Example simple();                                                // examples/simple.rs
Example nonhier(const std::string& assets_dir);                  // examples/nonhier.rs
Example nonhier2(const std::string& assets_dir);                 // examples/nonhier2.rs
Example four_shapes();                                           // examples/four-shapes.rs (white background)
Example graphics_poster(const std::string& assets_dir);          // examples/graphics-poster.rs (white background; glossy glass dodecahedron around a cow)
Example simple_cows(const std::string& assets_dir);              // examples/simple-cows.rs
Example primitives(const std::string& assets_dir);               // examples/primitives.rs (render/01b_primitives.png)
Example texture_mapping(const std::string& assets_dir);          // examples/texture-mapping.rs (needs assets/earth_cube.png, absent from the reference repository)
Example cube_mapping(const std::string& assets_dir);             // examples/cube-mapping.rs (needs assets/earth_cube.png, absent from the reference repository)
Example graphics_castle(const std::string& assets_dir);          // examples/graphics-castle.rs (KDMesh castle, StdRng hedge maze; needs assets/shrub.png, absent from the reference repository)
Example graphics_temple(const std::string& assets_dir);          // examples/graphics-temple.rs
Example monkeys_making_monkeys(const std::string& assets_dir);   // examples/monkeys-making-monkeys.rs (area lights; needs assets/cpu_cubemap.png, absent from the reference repository)
Example robot_alarm_clock(const std::string& assets_dir);        // examples/robot-alarm-clock.rs (render/10_robot-alarm-clock.png)

int run_main(Example ex);  // Image::new(..)? ; image.render::<RenderProgress, _>(..) ; image.save()

}  // namespace examples
}  // namespace portrayer
