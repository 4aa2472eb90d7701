// portrayer.hpp — C++ mirror of the portrayer crate's public API for the ray-cast/shade path.
//
// The reference is a Rust crate (no Rust toolchain exists in this build environment), so the host
// side above the C ABI (include/portrayer_hip.h) is C++ with the crate's module / type / method
// names, argument meaning and failure behaviour, so that a scene script reads like the reference's
// examples/*.rs:
//
//   portrayer::scene   -> src/scene.rs      (HierScene, SceneNode builder API, Geometry)
//   portrayer::material-> src/material.rs   (Material, refraction-index constants)
//   portrayer::light   -> src/light.rs      (Light, Falloff, Parallelogram)
//   portrayer::primitive-> src/primitive.rs + src/primitive/*.rs (unit shapes, Triangle, MeshData, Mesh) and src/kdtree/kdmesh.rs (KDMesh)
//   portrayer::camera  -> src/camera.rs     (CameraSettings)
//   portrayer::render  -> src/render.rs     (Image, ImageSliceMut, render)
//   portrayer::reporter-> src/reporter.rs   (Reporter, RenderProgress, NullProgress)
//   portrayer::math    -> src/math.rs + the parts of `vek` the scene scripts use
//
// Image::render() does on the host what render.rs:93-126 does before its pixel loop (camera, SAMPLES,
// flatten, optional k-d tree build) and hands the pixel loop itself (render.rs:127-150) to the
// MI355X through pt_render(). Where Rust would panic, these functions throw portrayer::Panic.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace portrayer {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

template <class T>
using Arc = std::shared_ptr<T>;

// ------------------------------------------------------------------------------------------------
namespace math {  // src/math.rs
constexpr double EPSILON = 0.00001;  // math.rs:15
constexpr double GAMMA = 2.2;        // math.rs:20
constexpr double INFINITY_F64 = INFINITY;

struct Vec3 {
    double x = 0.0, y = 0.0, z = 0.0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    Vec3(double v) : x(v), y(v), z(v) {}  // Vec3::from(f64) broadcasts
    static Vec3 zero() { return {0, 0, 0}; }
    static Vec3 up() { return {0, 1, 0}; }
    static Vec3 down() { return {0, -1, 0}; }
    static Vec3 right() { return {1, 0, 0}; }
    static Vec3 forward_rh() { return {0, 0, -1}; }
    static Vec3 back_rh() { return {0, 0, 1}; }
    static Vec3 unit_x() { return {1, 0, 0}; }
    static Vec3 unit_y() { return {0, 1, 0}; }
    static Vec3 unit_z() { return {0, 0, 1}; }
    double dot(Vec3 o) const { return (x * o.x + y * o.y) + z * o.z; }
    Vec3 cross(Vec3 o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double magnitude_squared() const { return dot(*this); }
    double magnitude() const { return std::sqrt(magnitude_squared()); }
    Vec3 normalized() const { double m = magnitude(); return {x / m, y / m, z / m}; }
    static Vec3 partial_min(Vec3 a, Vec3 b) { return {a.x <= b.x ? a.x : b.x, a.y <= b.y ? a.y : b.y, a.z <= b.z ? a.z : b.z}; }
    static Vec3 partial_max(Vec3 a, Vec3 b) { return {a.x >= b.x ? a.x : b.x, a.y >= b.y ? a.y : b.y, a.z >= b.z ? a.z : b.z}; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(double s, Vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

struct Rgb {
    double r = 0.0, g = 0.0, b = 0.0;
    Rgb() = default;
    Rgb(double r_, double g_, double b_) : r(r_), g(g_), b(b_) {}
    static Rgb black() { return {0, 0, 0}; }
    static Rgb red() { return {1, 0, 0}; }
    static Rgb green() { return {0, 1, 0}; }
    static Rgb blue() { return {0, 0, 1}; }
    static Rgb white() { return {1, 1, 1}; }
};
inline Rgb operator+(Rgb a, Rgb b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Rgb operator*(Rgb a, Rgb b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Rgb operator*(Rgb a, double s) { return {a.r * s, a.g * s, a.b * s}; }
inline Rgb operator*(double s, Rgb a) { return {a.r * s, a.g * s, a.b * s}; }
inline Rgb operator/(Rgb a, double s) { return {a.r / s, a.g / s, a.b / s}; }

struct Uv {
    double u = 0.0, v = 0.0;
};

class Radians {  // math.rs:55-71
    double v_ = 0.0;
    explicit Radians(double v) : v_(v) {}

   public:
    Radians() = default;
    static Radians from_degrees(double deg) { return Radians(deg * (3.14159265358979323846 / 180.0)); }  // f64::to_radians
    static Radians from_radians(double rad) { return Radians(rad); }
    double get() const { return v_; }
};

// 4x4 matrix with vek's builder semantics: m.scaled_3d(s) = S * m etc. (pre-multiplication), pinned
// by the reference's bounding_box.rs:184-195.
struct Mat4 {
    double m[4][4];
    Mat4();  // identity (Mat4::default / Mat4::identity)
    static Mat4 identity() { return Mat4(); }
    static Mat4 scaling_3d(Vec3 s);
    static Mat4 translation_3d(Vec3 t);
    static Mat4 rotation_x(double radians);
    static Mat4 rotation_y(double radians);
    static Mat4 rotation_z(double radians);
    static Mat4 look_at_rh(Vec3 eye, Vec3 target, Vec3 up);
    Mat4 scaled_3d(Vec3 s) const { return scaling_3d(s) * *this; }
    Mat4 translated_3d(Vec3 t) const { return translation_3d(t) * *this; }
    Mat4 rotated_x(double a) const { return rotation_x(a) * *this; }
    Mat4 rotated_y(double a) const { return rotation_y(a) * *this; }
    Mat4 rotated_z(double a) const { return rotation_z(a) * *this; }
    Mat4 inverted() const;
    Mat4 transposed() const;
    Mat4 operator*(const Mat4& o) const;
};
struct Mat3 {  // only what Material::uv_trans needs (material.rs:83)
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    static Mat3 identity() { return Mat3(); }
    static Mat3 scaling_3d(Vec3 s) { Mat3 r; r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z; return r; }
};
Vec3 transformed_point(Vec3 v, const Mat4& m);      // math.rs:45-47
Vec3 transformed_direction(Vec3 v, const Mat4& m);  // math.rs:49-51
}  // namespace math

// ------------------------------------------------------------------------------------------------
namespace texture {  // src/texture.rs
struct RgbImageBuffer {  // texture.rs:74-141: the decoded RGB8 pixels, no colour correction
    size_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
    static RgbImageBuffer open(const std::string& path);  // PNG or JPEG, told apart by the file's first bytes
    static RgbImageBuffer from_pixels(size_t width, size_t height, const uint8_t* rgb);
};
struct ImageTexture {  // texture.rs:143-169: sampled colours go sRGB -> linear with powf(2.2)
    RgbImageBuffer buffer;
    static ImageTexture open(const std::string& path) { return ImageTexture{RgbImageBuffer::open(path)}; }
};
// texture.rs:23-29 `enum Texture {FnTex, Image}`: only image textures can run on the GPU; a
// function texture would have to be baked into an image first.
struct Texture {
    ImageTexture image;
    static Texture from(ImageTexture img) { return Texture{std::move(img)}; }
};
struct NormalMap {  // texture.rs:171-221
    RgbImageBuffer buffer;
    static NormalMap open(const std::string& path) { return NormalMap{RgbImageBuffer::open(path)}; }
};
}  // namespace texture

// ------------------------------------------------------------------------------------------------
namespace material {  // src/material.rs
constexpr double AIR_REFRACTION_INDEX = 1.00;
constexpr double WATER_REFRACTION_INDEX = 1.33;
constexpr double WINDOW_GLASS_REFRACTION_INDEX = 1.51;
constexpr double OPTICAL_GLASS_REFRACTION_INDEX = 1.92;
constexpr double DIAMOND_REFRACTION_INDEX = 2.42;

struct Material {  // material.rs:50-86
    math::Rgb diffuse;
    math::Rgb specular;
    double shininess = 0.0;
    double reflectivity = 0.0;
    double glossy_side_length = 0.0;
    double refraction_index = 0.0;
    Arc<texture::Texture> texture;     // Option<Arc<Texture>>: diffuse colour sampled from an image
    math::Mat3 uv_trans;               // applied to the texture coordinate before sampling
    Arc<texture::NormalMap> normals;   // Option<Arc<NormalMap>>: shading normal sampled from an image
};
}  // namespace material

namespace light {  // src/light.rs
struct Falloff {
    double c0 = 1.0, c1 = 0.0, c2 = 0.0;
};
struct Parallelogram {
    math::Vec3 a, b;
};
struct Light {
    math::Vec3 position;
    math::Rgb color;
    Falloff falloff;
    Parallelogram area;
};
}  // namespace light

// ------------------------------------------------------------------------------------------------
namespace primitive {  // src/primitive.rs, src/primitive/*.rs, src/kdtree/kdmesh.rs
struct Sphere {};
struct Cube {};
struct Plane {};
struct Cylinder {};
struct Cone {};

struct Triangle {  // triangle.rs:8-26
    math::Vec3 a, b, c;
    std::optional<std::array<math::Vec3, 3>> normals;
    std::optional<std::array<math::Uv, 3>> tex_coords;
    static Triangle flat(math::Vec3 a, math::Vec3 b, math::Vec3 c) { return Triangle{a, b, c, std::nullopt, std::nullopt}; }
};

enum class Shading { Flat, Smooth };  // mesh.rs:11-18

class MeshData {  // mesh.rs:21-34
   public:
    // Loads the FIRST model of an OBJ file the way tobj 0.1.7 does (mesh.rs:57-61): f32 parse widened
    // to f64, fan triangulation, vertices de-duplicated per (v, vt, vn).
    static Arc<MeshData> load_obj(const std::string& path);
    static Arc<MeshData> create(std::vector<math::Vec3> positions, std::vector<std::array<uint32_t, 3>> triangles,
                                std::vector<math::Vec3> normals, std::vector<math::Uv> tex_coords = {});
    const std::vector<math::Uv>& tex_coords() const { return tex_coords_; }
    const std::vector<math::Vec3>& positions() const { return positions_; }
    const std::vector<math::Vec3>& normals() const { return normals_; }
    const std::vector<std::array<uint32_t, 3>>& triangles() const { return triangles_; }
    math::Vec3 bounds_min() const { return min_; }
    math::Vec3 bounds_max() const { return max_; }

   private:
    std::vector<math::Vec3> positions_, normals_;
    std::vector<math::Uv> tex_coords_;
    std::vector<std::array<uint32_t, 3>> triangles_;
    math::Vec3 min_, max_;
};

struct Mesh {  // mesh.rs:117-144
    Arc<MeshData> data;
    Shading shading;
    Mesh(Arc<MeshData> d, Shading s);
    static Mesh create(Arc<MeshData> d, Shading s) { return Mesh(std::move(d), s); }
};

struct KDMesh {  // kdmesh.rs:19-58: same geometry, triangles organised in a tree
    Arc<MeshData> data;
    Shading shading;
    KDMesh(const Arc<MeshData>& d, Shading s);
    static KDMesh create(const Arc<MeshData>& d, Shading s) { return KDMesh(d, s); }
};

struct Primitive {  // primitive.rs:67-81
    enum Kind { SphereK = 0, TriangleK = 1, MeshK = 2, KDMeshK = 3, PlaneK = 4, CubeK = 5, CylinderK = 6, ConeK = 7 };
    Kind kind;
    Arc<MeshData> mesh;
    Shading shading = Shading::Flat;
    Triangle triangle;
    Primitive(Sphere) : kind(SphereK) {}
    Primitive(Cube) : kind(CubeK) {}
    Primitive(Plane) : kind(PlaneK) {}
    Primitive(Cylinder) : kind(CylinderK) {}
    Primitive(Cone) : kind(ConeK) {}
    Primitive(Triangle t) : kind(TriangleK), triangle(std::move(t)) {}
    Primitive(Mesh m) : kind(MeshK), mesh(std::move(m.data)), shading(m.shading) {}
    Primitive(KDMesh m) : kind(KDMeshK), mesh(std::move(m.data)), shading(m.shading) {}
};
}  // namespace primitive

// ------------------------------------------------------------------------------------------------
namespace scene {  // src/scene.rs
struct Geometry {  // scene.rs:21-33
    primitive::Primitive primitive;
    Arc<material::Material> material;
    Geometry(primitive::Primitive p, Arc<material::Material> m) : primitive(std::move(p)), material(std::move(m)) {}
    static Geometry create(primitive::Primitive p, Arc<material::Material> m) { return Geometry(std::move(p), std::move(m)); }
};

class SceneNode {  // scene.rs:36-206
   public:
    SceneNode() = default;
    static SceneNode from(Geometry g);
    static SceneNode from(std::vector<Arc<SceneNode>> children);
    static SceneNode from(Arc<SceneNode> child);

    const std::optional<Geometry>& geometry() const { return geometry_; }
    const math::Mat4& trans() const { return trans_; }
    const math::Mat4& inverse_trans() const { return invtrans_; }
    const math::Mat4& normal_trans() const { return normal_trans_; }
    const std::vector<Arc<SceneNode>>& children() const { return children_; }

    SceneNode& with_child(Arc<SceneNode> c);
    SceneNode& with_children(const std::vector<Arc<SceneNode>>& cs);
    SceneNode& scaled(math::Vec3 scale);
    SceneNode& translated(math::Vec3 translation);
    SceneNode& rotated_xzy(math::Radians x, math::Radians y, math::Radians z);  // x, then z, then y (scene.rs:177-180)
    SceneNode& rotated_xzy(math::Radians all) { return rotated_xzy(all, all, all); }
    SceneNode& rotated_x(math::Radians a);
    SceneNode& rotated_y(math::Radians a);
    SceneNode& rotated_z(math::Radians a);
    void set_transform(const math::Mat4& t);  // scene.rs:201-205
    Arc<SceneNode> into() { return std::make_shared<SceneNode>(std::move(*this)); }

   private:
    std::optional<Geometry> geometry_;
    math::Mat4 trans_, invtrans_, normal_trans_;
    std::vector<Arc<SceneNode>> children_;
};

struct HierScene {  // scene.rs:11-18
    Arc<SceneNode> root;
    std::vector<light::Light> lights;
    math::Rgb ambient;
};
}  // namespace scene

namespace camera {  // src/camera.rs:5-14
struct CameraSettings {
    math::Vec3 eye, center, up;
    math::Radians fovy;
};
}  // namespace camera

// ------------------------------------------------------------------------------------------------
namespace reporter {  // src/reporter.rs
struct Reporter {
    virtual ~Reporter() = default;
    virtual void report_finished_pixels(uint64_t pixels) = 0;
};
struct NullProgress : Reporter {  // reporter.rs:87-97
    explicit NullProgress(uint64_t) {}
    void report_finished_pixels(uint64_t) override {}
};
struct RenderProgress : Reporter {  // reporter.rs:15-85; prints a percentage line to stderr
    explicit RenderProgress(uint64_t total) : total_(total) {}
    void report_finished_pixels(uint64_t pixels) override;

   private:
    uint64_t total_, done_ = 0;
};
}  // namespace reporter

// ------------------------------------------------------------------------------------------------
namespace render {  // src/render.rs
// The reference selects the traversal at compile time with cargo features (render.rs:121-126);
// here it is a run-time setting (default Flat; environment variable PORTRAYER_TRAVERSAL=kdtree).
enum class Traversal { Flat = 1, KdTree = 2, Hier = 3 };  // the crate's `flat_scene` feature, its `kdtree` feature, its DEFAULT (neither: SceneNode::ray_cast)
void set_traversal(Traversal t);
Traversal traversal();

using Background = std::function<math::Rgb(math::Uv)>;  // TextureSource for the background (render.rs:31-34)

struct RenderStats {
    uint64_t primary = 0, shadow = 0, reflect = 0, refract = 0, hits = 0;
    uint64_t n_inner = 0, n_leaf = 0, n_analytic = 0, n_tri = 0, n_bbox = 0;
    double kernel_ms = 0.0, total_ms = 0.0;
};

class Image;

class ImageSliceMut {  // render.rs:56-152
   public:
    ImageSliceMut(Image& image, std::pair<size_t, size_t> top_left, std::pair<size_t, size_t> bottom_right);  // throws Panic like render.rs:79-90
    template <class R = reporter::NullProgress>
    void render(const scene::HierScene& scene, camera::CameraSettings camera, const Background& background) {
        R rep(total_pixels());
        render_impl(scene, camera, background, rep);
    }

   private:
    uint64_t total_pixels() const;
    void render_impl(const scene::HierScene&, camera::CameraSettings, const Background&, reporter::Reporter&);
    Image& image_;
    std::pair<size_t, size_t> top_left_, bottom_right_;
};

class Image {  // render.rs:154-224
   public:
    // Image::new: re-opens an existing PNG of the same size so that a slice render keeps the rest
    static Image create(const std::string& path, size_t width, size_t height);
    size_t width() const { return width_; }
    size_t height() const { return height_; }
    void save() const { save_as(path_); }
    void save_as(const std::string& path) const;
    ImageSliceMut slice_mut(std::pair<size_t, size_t> top_left, std::pair<size_t, size_t> bottom_right) { return ImageSliceMut(*this, top_left, bottom_right); }
    template <class R = reporter::NullProgress>
    void render(const scene::HierScene& scene, camera::CameraSettings camera, const Background& background) {
        ImageSliceMut(*this, {0, 0}, {width_ - 1, height_ - 1}).render<R>(scene, camera, background);
    }
    std::vector<uint8_t>& buffer() { return buffer_; }
    const std::vector<uint8_t>& buffer() const { return buffer_; }
    const RenderStats& last_stats() const { return stats_; }

   private:
    friend class ImageSliceMut;
    std::string path_;
    size_t width_ = 0, height_ = 0;
    std::vector<uint8_t> buffer_;  // RGB8, row-major
    RenderStats stats_;
};
}  // namespace render

}  // namespace portrayer
