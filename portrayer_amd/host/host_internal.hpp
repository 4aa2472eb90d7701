// Host-side pieces of Image::render that the reference runs once per render before its pixel loop
// (src/render.rs:93-126): flattening (src/flat_scene.rs), bounding boxes (src/bounding_box.rs), the
// scene k-d tree build (src/kdtree/leaf.rs, src/kdtree/kdscene.rs) and the camera (src/camera.rs).
#pragma once

#include <string>
#include <vector>

#include "../../include/portrayer_hip.h"
#include "portrayer.hpp"

namespace portrayer {
namespace detail {

struct BoundingBox {  // bounding_box.rs:39-117
    math::Vec3 min, max;
    math::Mat4 invtrans;
    static BoundingBox create(math::Vec3 min, math::Vec3 max);  // BoundingBox::new, panics unless min <= max
    double extent() const {  // bounding_box.rs:95-99: squared diagonal
        math::Vec3 d = max - min;
        return d.magnitude_squared();
    }
};
BoundingBox operator*(const math::Mat4& m, const BoundingBox& b);  // bounding_box.rs:123-148
BoundingBox primitive_bounds(const primitive::Primitive& p);       // Bounds for Primitive, primitive.rs:46-53

struct FlatSceneNode {  // flat_scene.rs:50-61
    scene::Geometry geometry;
    math::Mat4 trans, invtrans, normal_trans;
    // not in the reference's FlatSceneNode: where the node sits in the hierarchy, for PT_TRAVERSE_HIER
    std::vector<const scene::SceneNode*> chain;  // root .. the SceneNode that owns the geometry
    std::vector<uint32_t> path;                  // child index taken at every step below the root
    FlatSceneNode(scene::Geometry g, const math::Mat4& t);  // flat_scene.rs:103-108
    BoundingBox bounds() const { return trans * primitive_bounds(geometry.primitive); }  // flat_scene.rs:63-69
};

struct FlatScene {  // flat_scene.rs:16
    std::vector<FlatSceneNode> root;
    std::vector<light::Light> lights;
    math::Rgb ambient;
    static FlatScene from(const scene::HierScene& s);  // flat_scene.rs:18-46
};

struct GraphPacking {  // the ABI-4 scene-graph arrays of pt_scene (include/portrayer_hip.h)
    uint32_t n_graph_nodes = 0;
    std::vector<double> trans, invtrans, normal_trans;  // n_graph_nodes x 16
    std::vector<uint32_t> chain_off, chain, dfs_rank;
};
GraphPacking pack_graph(const FlatScene& flat);

struct PartitionConfig {  // leaf.rs:55-67
    size_t target_max_nodes = 3;
    long target_max_merit = 3;
    size_t max_tries = 10;
};

// KDTreeNode (node.rs:13-25) linearised in pre-order; node 0 is the root.
struct KdTree {
    std::vector<int32_t> axis, front, back, first, count, items;
    std::vector<double> plane;
    math::Vec3 root_min, root_max;
    int max_depth = 0;
};
// KDLeaf::partitioned (leaf.rs:89-231) over a list of cached bounds (NodeBounds, leaf.rs:16-34)
KdTree kd_partition(const std::vector<BoundingBox>& bounds, size_t max_depth, PartitionConfig conf);
KdTree kd_scene_tree(const FlatScene& flat, size_t max_depth);  // KDTreeScene::from, kdscene.rs:19-43

struct Camera {  // camera.rs:17-45
    math::Vec3 eye;
    math::Mat4 view_to_world;
    double fov_factor, aspect_ratio, width, height;
    Camera(const camera::CameraSettings& cam, double width, double height);
    pt_camera to_abi() const;
};

// A scene prepared for the GPU: flattened, packed and uploaded once (pt_scene_upload); render() may
// then be called any number of times. Image::render builds one per call, like render.rs:121-126.
class Renderer {
   public:
    Renderer(const scene::HierScene& scene, render::Traversal traversal, int kd_depth, int device);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    void render(const camera::CameraSettings& cam, uint32_t width, uint32_t height, const double* background, bool background_rows,
                pt_rect slice, uint32_t samples, uint64_t seed, int sample_mode, bool collect_stats, uint8_t* rgb, double* linear,
                pt_stats* stats);
    pt_context* context() const { return ctx_; }  // rank 0's context when the scene is on a node
    pt_node* node() const { return node_; }
    const FlatScene& flat() const { return flat_; }
    struct PrepareMs { double flatten = 0, pack = 0, context = 0, kd_build = 0, upload = 0; };  // where the time before the first pixel goes
    const PrepareMs& prepare_ms() const { return prep_; }

   private:
    FlatScene flat_;
    pt_context* ctx_ = nullptr;
    pt_node* node_ = nullptr;  // PORTRAYER_GPUS > 1: the render is tile-partitioned over the node's GPUs (one RCCL gather)
    PrepareMs prep_;
};

// PNG codec for Image::new / Image::save (render.rs:165-208; the reference uses the `image` crate)
bool png_read(const std::string& path, size_t* width, size_t* height, std::vector<uint8_t>* rgb);
bool jpeg_read(const std::string& path, size_t* width, size_t* height, std::vector<uint8_t>* rgb);   // jpeg.cpp
bool image_read(const std::string& path, size_t* width, size_t* height, std::vector<uint8_t>* rgb);  // PNG or JPEG, by signature
void png_write(const std::string& path, size_t width, size_t height, const std::vector<uint8_t>& rgb);

}  // namespace detail
}  // namespace portrayer
