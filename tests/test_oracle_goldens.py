"""Pins the oracle against the reference's committed renders.

The goldens were rendered with random anti-aliasing (thread_rng, unknown SAMPLES), so a
centre-sample oracle render must agree everywhere except on thin silhouette / shadow / facet
edges; interior pixels and background rows are exact known answers."""
import os

import numpy as np
import pytest
from PIL import Image

from example_scenes import EXAMPLES
from rand07 import StdRng
from scene_dsl import GOLDEN


def golden(name):
    return np.array(Image.open(os.path.join(GOLDEN, "render", name)).convert("RGB"))


def erode_mismatch(mask, px=2):
    """True where a (2*px+1)^2 block of mismatching pixels survives: thin edges vanish."""
    m = mask.copy()
    for _ in range(px):
        n = m.copy()
        n[1:, :] &= m[:-1, :]; n[:-1, :] &= m[1:, :]; n[:, 1:] &= m[:, :-1]; n[:, :-1] &= m[:, 1:]
        n[0, :] = n[-1, :] = False; n[:, 0] = n[:, -1] = False
        m = n
    return m


@pytest.fixture(scope="module")
def renders(oracle):
    out = {}
    for name, png, mode in [("primitives-simple", "01a_primitives-simple.png", oracle.MODE_HIER),
                            ("entering-the-mirror-dimension", "entering-the-mirror-dimension.png", oracle.MODE_HIER),
                            ("big-scene", "09a_kdtree.png", oracle.MODE_KD)]:
        scene, cam, (w, h) = EXAMPLES[name]()
        out[name] = (oracle.render(scene, cam, w, h, mode=mode), golden(png))
    return out


@pytest.mark.parametrize("name,exact_min,within1_min", [
    ("primitives-simple", 0.989, 0.992),             # measured 99.01 % / 99.35 %
    ("entering-the-mirror-dimension", 0.90, 0.96),   # measured 90.3 % / 96.6 % (noisy low-sample golden)
    ("big-scene", 0.93, 0.96),                       # measured 93.3 % / 96.8 %
])
def test_centre_sample_matches_golden(renders, name, exact_min, within1_min):
    r, g = renders[name]
    assert r.rgb.shape == g.shape
    d = np.abs(r.rgb.astype(int) - g.astype(int)).max(axis=2)
    assert (d == 0).mean() >= exact_min
    assert (d <= 1).mean() >= within1_min
    assert not erode_mismatch(d > 8, 2).any(), "a mismatch blob thicker than an anti-aliased edge survived"
    assert r.stats["kd_plane_miss"] == 0


def test_background_rows_are_exact(renders):
    for name, (r, g) in renders.items():
        h = g.shape[0]
        assert tuple(r.rgb[0, 0]) == (122, 168, 202) == tuple(g[0, 0]), name
        assert tuple(r.rgb[h // 2, 0]) == (89, 122, 230) == tuple(g[h // 2, 0]), name


def test_interior_known_answers(renders):  # SURVEY §8c-4 (x, y) -> rgb
    ps = {(455, 450): (117, 235, 132), (100, 300): (112, 225, 126), (250, 260): (98, 165, 218),
          (600, 330): (241, 66, 64), (700, 60): (115, 158, 209), (380, 230): (110, 222, 125)}
    r, g = renders["primitives-simple"]
    for (x, y), rgb in ps.items():
        assert tuple(r.rgb[y, x]) == rgb == tuple(g[y, x]), (x, y)
    md = {(560, 420): (70, 97, 240), (400, 50): (117, 161, 207), (680, 300): (164, 139, 119)}
    r, g = renders["entering-the-mirror-dimension"]
    for (x, y), rgb in md.items():
        assert tuple(r.rgb[y, x]) == rgb == tuple(g[y, x]), (x, y)


Q1_PIXELS = {(944, 653): (115, 75, 233), (1151, 731): (87, 177, 29), (1151, 719): (65, 133, 22), (1125, 734): (78, 159, 26),
             (1160, 754): (54, 111, 18), (1159, 740): (86, 176, 29), (915, 642): (70, 46, 144), (1156, 735): (89, 180, 30)}


def test_cone_quirk_q1_pixels_kd_exact(renders):  # results depend on the k-d tree (cone.rs:64-76)
    r, g = renders["big-scene"]
    for (x, y), rgb in Q1_PIXELS.items():
        assert tuple(g[y, x]) == rgb, (x, y)
        assert tuple(r.rgb[y, x]) == rgb, (x, y)


def test_cone_quirk_q1_pixels_flat_differs(oracle):
    """Flat semantics reproduces only the two '=' pixels of SURVEY §8c-5; the others differ."""
    scene, cam, (w, h) = EXAMPLES["big-scene"]()
    ps = oracle.pack(scene)
    same = 0
    for (x, y), rgb in Q1_PIXELS.items():
        r = oracle.render(ps, cam, w, h, mode=oracle.MODE_FLAT, rect=(x, y, x, y), threads=1)
        same += tuple(r.rgb[y, x]) == rgb
    assert same == 2


def test_rand07_known_answers():
    r = StdRng.seed_from_u64(1234939301)
    assert [f"{w:08x}" for w in r.key] == "bf1e3d47 eb93bac8 4f3b4401 4991de93 388161da 4c7039a4 d09a30e1 803dc311".split()
    assert [f"{w:08x}" for w in r._block()[:4]] == "6df8b5e8 c53e7e25 9b646b73 4fa8aa27".split()
    r = StdRng.seed_from_u64(1234939301)
    assert [r.gen_f64() for _ in range(3)] == [0.770484813821869, 0.3111673685738353, 0.45554167289057856]


def test_big_scene_first_node(oracle):  # first node = Cone, material 13, scale 37.84..., angle 352.12... deg
    from scene_dsl import CONE
    scene, _, _ = EXAMPLES["big-scene"]()
    ps = oracle.pack(scene)
    first = ps.lin.nodes[1]
    assert first.geometry[0].kind == CONE
    rng = StdRng.seed_from_u64(1234939301)
    draws = [rng.gen_f64() for _ in range(45)]
    assert tuple(first.geometry[1].diffuse) == tuple(draws[39:42])  # material 13
    assert first.ops[0] == ("s", (37.8468300601445,) * 3)
    assert abs(first.ops[1][1][0] * 180.0 / np.pi - 352.12624439923894) < 1e-12
    assert first.ops[4][1][1] == -400.0 + 44.85171610632929


def test_kd_tree_shape_big_scene(oracle):  # SURVEY App.C: depth 10 -> 1023 splits, 1024 leaves, 6806 refs, max leaf 12
    scene, _, _ = EXAMPLES["big-scene"]()
    tree = oracle.kd_scene_dump(scene, kd_depth=10)
    kinds = tree["kind"]
    assert (kinds == 0).sum() == 1023 and (kinds == 1).sum() == 1024
    assert len(tree["items"]) == 6806
    assert tree["count"][kinds == 1].max() == 12


def test_textured_golden_normal_mapping(oracle):
    """render/04a_normal-mapping.png pins the texture path (uv / TBN of Plane, Cube, Sphere; nearest-texel
    sampling; sRGB -> linear; normal maps, texture.rs + material.rs:109-144). The golden was rendered with
    many jittered samples and another JPEG decoder, so 8x8 block means are compared."""
    from example_scenes import normal_mapping
    scene, cam, (w, h) = normal_mapping()
    g = golden("04a_normal-mapping.png")
    r = oracle.render(scene, cam, w, h, mode=oracle.MODE_HIER)

    def blk(a, k=8):
        hh, ww = a.shape[0] // k * k, a.shape[1] // k * k
        return a[:hh, :ww].astype(float).reshape(hh // k, k, ww // k, k, 3).mean(axis=(1, 3))

    d = np.abs(blk(r.rgb) - blk(g)).max(axis=2)
    assert d.mean() < 1.5 and (d > 6).mean() < 0.05  # measured 1.11 / 3.1 %
    # a wrong orientation of the maps is far outside that band: flip v of every texture and look again
    textures = {id(n.geometry[1].texture): n.geometry[1].texture for n in scene.root.children if n.geometry[1].texture is not None}
    for t in textures.values():
        t.pixels = np.ascontiguousarray(t.pixels[::-1])
    r2 = oracle.render(scene, cam, w, h, mode=oracle.MODE_HIER)
    d2 = np.abs(blk(r2.rgb) - blk(g)).max(axis=2)
    assert d2.mean() > 1.5 * d.mean()


@pytest.mark.parametrize("name,png,exact_min,within1_min,block_mean_max", [
    ("smooth-shading", "02_smooth-shading.png", 0.915, 0.95, 0.5),        # measured 92.5 % / 95.7 % / 0.27
    ("glossy-reflection", "07_glossy-reflection.png", 0.915, 0.945, 0.4),  # measured 92.5 % / 95.3 % / 0.22
    ("soft-shadows", "08_soft-shadows.png", 0.58, 0.79, 0.6),              # measured 61.3 % / 81.2 % / 0.36 (every lit pixel carries area-light noise)
])
def test_more_goldens(oracle, name, png, exact_min, within1_min, block_mean_max):
    """Reference renders beyond SURVEY §8c's list: vertex-normal interpolation (triangle.rs:82-86), glossy
    reflection (material.rs:221-239) and area lights (light.rs:62-70). The random parts (glossy offsets,
    light samples, anti-aliasing) differ from the reference's thread_rng, so next to the per-pixel
    agreement the 8x8 block means are compared."""
    from example_scenes import MORE_EXAMPLES
    scene, cam, (w, h) = MORE_EXAMPLES[name]()
    g = golden(png)
    r = oracle.render(scene, cam, w, h, mode=oracle.MODE_HIER)
    assert r.rgb.shape == g.shape
    d = np.abs(r.rgb.astype(int) - g.astype(int)).max(axis=2)
    assert (d == 0).mean() >= exact_min
    assert (d <= 1).mean() >= within1_min

    def blk(a, k=8):
        hh, ww = a.shape[0] // k * k, a.shape[1] // k * k
        return a[:hh, :ww].astype(float).reshape(hh // k, k, ww // k, k, 3).mean(axis=(1, 3))

    db = np.abs(blk(r.rgb) - blk(g)).max(axis=2)
    assert db.mean() < block_mean_max and (db > 6).mean() < 0.02


def test_golden_transmission_refraction(oracle):
    """render/06b_transmission-refraction.png pins the dielectric branch of hit_color (glass pane:
    entering / leaving, Schlick mix, internal reflections to depth 10) in the upper 45 % of the image, where the walls
    are seen through the front glass (measured 98.99 % exact, 99.41 % within 1). The water tank below is not a
    per-pixel pin (coincident faces, quirk Q13; JPEG-decoder and anti-aliasing noise on the textured surfaces):
    8x8 block means are compared there."""
    from example_scenes import transmission_refraction
    scene, cam, (w, h) = transmission_refraction()
    g = golden("06b_transmission-refraction.png")
    r = oracle.render(scene, cam, w, h, mode=oracle.MODE_HIER)
    assert r.rgb.shape == g.shape and r.stats["refract"] > 3_000_000 and r.stats["depth11"] > 0
    d = np.abs(r.rgb.astype(int) - g.astype(int)).max(axis=2)
    top = int(h * 0.45)
    assert (d[:top] == 0).mean() >= 0.985 and (d[:top] <= 1).mean() >= 0.99
    assert not erode_mismatch(d[:top] > 8, 2).any()

    def blk(a, k=8):
        hh, ww = a.shape[0] // k * k, a.shape[1] // k * k
        return a[:hh, :ww].astype(float).reshape(hh // k, k, ww // k, k, 3).mean(axis=(1, 3))

    db = np.abs(blk(r.rgb) - blk(g)).max(axis=2)
    assert db.mean() < 1.5 and (db > 6).mean() < 0.04  # measured 0.94 / 2.5 %


def test_golden_water_glass(oracle):
    """render/06a_water-glass.png: glossy reflection on a textured, normal-mapped table, refraction through a water
    cylinder. Glossy offsets, anti-aliasing and the JPEG decoder differ from the reference's run, so 8x8 block means
    of a 4-sample render are compared (measured: mean 0.35 levels, 0.01 % of the blocks above 6)."""
    from example_scenes import water_glass
    scene, cam, (w, h) = water_glass()
    g = golden("06a_water-glass.png")
    r = oracle.render(scene, cam, w, h, mode=oracle.MODE_HIER, samples=4, jitter=oracle.JITTER_RNG, seed=1)
    assert r.rgb.shape == g.shape and r.stats["refract"] > 500_000 and r.stats["reflect"] > 1_000_000

    def blk(a, k=8):
        hh, ww = a.shape[0] // k * k, a.shape[1] // k * k
        return a[:hh, :ww].astype(float).reshape(hh // k, k, ww // k, k, 3).mean(axis=(1, 3))

    db = np.abs(blk(r.rgb) - blk(g)).max(axis=2)
    assert db.mean() < 0.6 and (db > 6).mean() < 0.002


def test_hier_and_flat_differ_only_on_the_refractive_cylinder(oracle):
    """Documents a KNOWN divergence, it does not excuse it: the reference's default traversal is
    hierarchical, the GPU implements FLAT / KD. On water-glass the two oracle modes make different hit decisions for
    rays leaving the water cylinder (a dielectric inside a transformed group): measured 1.94 % of the u8 pixels (up to
    130 levels), ray counts differ. Two tiers: VISIBLE differences are confined to the cylinder's image region; outside
    it no u8 pixel differs and the linear f64 image differs only by rounding (138 pixels, <= 1.2e-12). This test fails
    if visible differences spread beyond that region or grow - i.e. if the divergence stops being explainable as the
    refractive self-intersection case."""
    from example_scenes import water_glass
    scene, cam, (w, h) = water_glass()
    a = oracle.render(scene, cam, w, h, mode=oracle.MODE_HIER)
    b = oracle.render(scene, cam, w, h, mode=oracle.MODE_FLAT)
    d = np.abs(a.rgb.astype(int) - b.rgb.astype(int)).max(axis=2)
    assert a.stats["primary"] == b.stats["primary"]
    assert a.stats["hits"] != b.stats["hits"], "HIER and FLAT now agree on water-glass: update DESIGN.md section 7"
    assert 0.0 < (d > 0).mean() < 0.03
    ys, xs = np.nonzero(d > 0)
    assert xs.min() >= 330 and xs.max() <= 580 and ys.min() >= 180, "differences outside the water cylinder's region"
    outside = np.ones_like(d, dtype=bool); outside[180:, 330:581] = False
    assert not (d[outside] > 0).any(), "a u8 pixel differs outside the cylinder's region"
    # rounding-only there: measured max 1.2e-12; 1e-9 is ~1000x that and ~1e6x below one u8 level (4e-3)
    assert np.abs(a.linear[outside] - b.linear[outside]).max() < 1e-9


ROBOT_COLOURS = {  # examples/robot-alarm-clock.rs:97-101: the commented-out diffuse colours of mat_robot_metal are the other three renders
    "10_robot-alarm-clock_green.png": (0.006449, 0.417885, 0.025384),
    "10_robot-alarm-clock.png": (0.211857, 0.772537, 0.8971),
    "10_robot-alarm-clock_dark_blue.png": (0.006512, 0.08022, 0.417885),
    "10_robot-alarm-clock_red.png": (0.417885, 0.006501, 0.006501),
}


def _block8(img):
    h, w = (img.shape[0] // 8) * 8, (img.shape[1] // 8) * 8
    return img[:h, :w].astype(np.float64).reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))


def _kdmesh_two_ways(oracle, example, w, h, rect, colour=None):
    """The scene script rendered by the oracle (the crate's default traversal, centre samples) over `rect`, KDMesh primitives walked
    (a) through the reference's own triangle k-d tree as the code stands (kdmesh.rs:62-74 + node.rs:112-202 with
    bounding_box.rs:95-99's squared extent: quirk Q3), (b) like Mesh (box, then every triangle)."""
    from portrayer_amd import host
    from scene_dsl import ASSETS
    sc = host.Scene.example(example, assets=ASSETS)
    ex = sc.export()
    if colour is not None:
        mats = ex["materials"].copy()
        green = [i for i in range(len(mats)) if abs(mats[i][1] - 0.417885) < 1e-9 and abs(mats[i][0] - 0.006449) < 1e-9]
        assert len(green) == 1
        mats[green[0], 0:3] = colour
        ex["materials"] = mats
    as_mesh = dict(ex)
    pt = ex["prim_type"].copy(); pt[pt == 3] = 2
    as_mesh["prim_type"] = pt
    out = []
    for arrays in (ex, as_mesh):
        out.append(oracle.render(oracle.pack_arrays(arrays), sc.camera, w, h, samples=1, jitter=oracle.JITTER_CENTRE, mode=oracle.MODE_HIER, rect=rect, threads=8).rgb)
    return out


def test_which_kdmesh_behaviour_the_reference_renders_support(oracle):
    """VERDICT r02 #8. The reference rendered KDMesh objects in five committed images: the fish of 06b_transmission-refraction.png
    and the robot of the four 10_robot-alarm-clock*.png. The code as it stands (HEAD) drops triangles of a KDMesh whose squared
    model-space diagonal is smaller than its distance from the ray's origin (quirk Q3: node.rs:118 takes `t_range.start + extent`
    as the end of the segment that decides which sides of a split are visited, and bounding_box.rs:95-99's extent is the SQUARED
    diagonal - "HACK ... or else the k-d tree will miss points"). This test renders each scene both ways over the region where the
    two behaviours differ and records which one every reference render supports.

    Finding, asserted below: ALL FIVE renders support "KDMesh renders like Mesh" - on the blocks where the behaviours differ the
    as-Mesh picture is several times closer to the reference's. So the committed renders were not made by the code as it stands
    (the script's own comments - "KDMesh doesn't work for this for some reason", robot-alarm-clock.rs:169 ff. - describe the
    breakage for the parts it moved to Mesh). The oracle and the product follow the CODE (the k-d walk restated expression by
    expression, pinned by the reference's unit test kdmesh.rs:99-166 on castle.obj, where both behaviours agree); the product's
    switch PORTRAYER_KDMESH_AS_MESH=1 gives the renders' picture (tests/test_examples_extra.py)."""
    golden_blocks = lambda name: (_block8(golden(name)) if os.path.exists(os.path.join(GOLDEN, "render", name)) else
                                  np.array(Image.open(os.path.join(GOLDEN, "render_blocks8", name)).convert("RGB")).astype(np.float64))
    verdicts = {}
    # the fish: the two behaviours differ in a handful of pixels only (the fish are large against their distance from the rays' origins)
    w, h, rect = 910, 512, (200, 304, 671, 351)
    q3, mesh = _kdmesh_two_ways(oracle, "transmission-refraction", w, h, rect)
    g = golden_blocks("06b_transmission-refraction.png")
    bq, bm = _block8(q3), _block8(mesh)
    differ = np.abs(bq - bm).mean(axis=2) > 1.0
    assert 1 <= differ.sum() <= 40
    verdicts["06b"] = (float(np.abs(bq - g).mean(axis=2)[differ].mean()), float(np.abs(bm - g).mean(axis=2)[differ].mean()))
    # the robot (base, torso, connectors as KDMesh): large holes under Q3
    w, h, rect = 1920, 1080, (752, 144, 1807, 1079)
    for name, colour in ROBOT_COLOURS.items():
        q3, mesh = _kdmesh_two_ways(oracle, "robot-alarm-clock", w, h, rect, colour)
        g = golden_blocks(name)
        bq, bm = _block8(q3), _block8(mesh)
        differ = np.abs(bq - bm).mean(axis=2) > 2.0
        assert differ.sum() > 500
        verdicts[name] = (float(np.abs(bq - g).mean(axis=2)[differ].mean()), float(np.abs(bm - g).mean(axis=2)[differ].mean()))
    print({k: "code as it stands %.2f, as Mesh %.2f levels from the reference's render" % v for k, v in verdicts.items()})
    for k, (as_code, as_mesh) in verdicts.items():
        assert as_mesh < as_code, f"{k}: the reference's render is closer to the code as it stands"
    for k, (as_code, as_mesh) in verdicts.items():
        if k != "06b":
            assert as_mesh < 15.0 and as_code > 3.0 * as_mesh, k  # measured: ~7-8 against ~58
