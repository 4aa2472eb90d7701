//! Generates multiple test images with different numbers of samples: the same scene rendered with
//! SAMPLES=1 and SAMPLES=32, the variable set between the two Image::render calls
//! (scene data and main(): examples/antialiasing.rs:18-64)
#include <cstdio>
#include <cstdlib>

#include "examples.hpp"

namespace portrayer {
namespace examples {
using namespace math;
using material::Material;
using light::Light;
using primitive::Mesh;
using primitive::MeshData;
using primitive::Shading;
using scene::Geometry;
using scene::HierScene;
using scene::SceneNode;

Example antialiasing(const std::string& assets) {
    auto mat_monkey = std::make_shared<Material>(Material{.diffuse = Rgb{0.961, 0.573, 0.259}, .specular = Rgb{0.3, 0.3, 0.3}, .shininess = 25.0});

    auto monkey_mesh = MeshData::load_obj(assets + "/monkey.obj");

    HierScene scene{
        .root = SceneNode::from(std::vector<Arc<SceneNode>>{
            SceneNode::from(Geometry::create(Mesh::create(monkey_mesh, Shading::Flat), mat_monkey))
                .into(),
        }).into(),
        .lights = {
            Light{.position = Vec3{0.0, 0.0, 10.0}, .color = Rgb{0.5, 0.5, 0.5}},
        },
        .ambient = Rgb{0.3, 0.3, 0.3},
    };

    camera::CameraSettings cam{
        .eye = Vec3{0.0, 0.0, 6.5},
        .center = Vec3{0.0, 0.0, 0.0},
        .up = Vec3::up(),
        .fovy = Radians::from_degrees(20.0),
    };

    return Example{std::move(scene), cam, 300, 250, "antialiasing_1.png"};
}
}  // namespace examples
}  // namespace portrayer

#ifdef PORTRAYER_EXAMPLE_MAIN
int main() {
    using namespace portrayer;
    examples::Example ex = examples::antialiasing("assets");
    for (int samples : {1, 32}) {
        try {
            render::Image image = render::Image::create("antialiasing_" + std::to_string(samples) + ".png", ex.width, ex.height);
            std::printf("Rendering with %d samples\n", samples);
            setenv("SAMPLES", std::to_string(samples).c_str(), 1);  // env::set_var("SAMPLES", ..): read by every render call
            image.render<reporter::RenderProgress>(ex.scene, ex.cam, examples::sky);
            image.save();
        } catch (const std::exception& e) {
            std::fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    return 0;
}
#endif
