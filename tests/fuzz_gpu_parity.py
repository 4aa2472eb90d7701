#!/usr/bin/env python3
"""Fuzz parity run (not collected by pytest; run on the GPU box):
   python tests/fuzz_gpu_parity.py <first seed> <count> [width height [samples]]
Random scenes (tests/test_gpu_render_parity.random_scene) with extreme scales / near-degenerate
transforms mixed in, textured scenes, mesh-free scenes without reflection (the wave-uniform walk), scenes with mirrors but no glass (the chain kernel); FLAT, KD and HIER, GPU vs oracle: reports every pixel that differs."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import host_glue
import oracle_lib as O
from portrayer_amd import _hip as H
from portrayer_amd import host
from scene_dsl import Cone, Cube, Cylinder, Light, Material, Node, Plane, Scene, Sphere, default_background
from test_gpu_render_parity import analytic_scene, random_scene


def extreme_scene(seed):
    """Thin / huge / tiny primitives, coincident faces, rays grazing edges."""
    rng = np.random.default_rng(1000 + seed)
    mats = [Material(diffuse=tuple(rng.uniform(0.1, 1, 3)), specular=(0.3, 0.3, 0.3), shininess=20.0) for _ in range(4)]
    mats.append(Material(diffuse=(0.1, 0.1, 0.1), specular=(0.8, 0.8, 0.8), shininess=200.0, reflectivity=0.8))
    kids = []
    prims = [Sphere, Cube, Plane, Cylinder, Cone]
    for i in range(int(rng.integers(6, 14))):
        p = prims[int(rng.integers(0, 5))]()
        s = np.exp(rng.uniform(-4, 2, 3))  # 0.02 .. 7, anisotropic
        if rng.random() < 0.3:
            s[int(rng.integers(0, 3))] *= 1e-3  # nearly flat
        n = Node.geo(p, mats[int(rng.integers(0, len(mats)))]).scaled(tuple(s))
        if rng.random() < 0.8:
            n.rotated_xzy(tuple(rng.uniform(-3.2, 3.2, 3)))
        n.translated(tuple(rng.uniform(-3, 3, 3)))
        kids.append(n)
    # two cubes sharing a face exactly, and a sphere touching a plane: exact ties / grazing hits
    kids.append(Node.geo(Cube(), mats[0]).translated((0.0, 0.0, 0.0)))
    kids.append(Node.geo(Cube(), mats[1]).translated((1.0, 0.0, 0.0)))
    kids.append(Node.geo(Plane(), mats[2]).scaled(20.0).translated((0.0, -0.5, 0.0)))
    kids.append(Node.geo(Sphere(), mats[4]).scaled(0.5).translated((-1.5, 0.0, 1.0)))
    lights = [Light(position=(5.0, 8.0, 6.0), color=(0.8, 0.8, 0.8)), Light(position=(-4.0, 3.0, 8.0), color=(0.4, 0.4, 0.6))]
    from scene_dsl import Camera
    return Scene(root=Node.group(kids), lights=lights, ambient=(0.2, 0.2, 0.2)), Camera(eye=(1.0, 2.5, 9.0), center=(0.0, 0.0, 0.0), fovy_degrees=45.0)


MODES = os.environ.get("FUZZ_MODES", "flat,kd,hier").split(",")  # FUZZ_MODES=kd: only the k-d tree semantics


def run(first, count, w=128, h=96, samples=2, modes=None, out=print):
    """`count` seeds from `first`: five scene families x the traversal semantics in `modes`, each scene rendered by the counting and by the plain
    instantiation and compared with the oracle. Returns (differing pixels, tolerated texel-edge pixels of textured scenes, scenes x modes)."""
    modes = MODES if modes is None else modes
    bad_total = 0
    tex_edge = 0
    n_renders = 0
    for seed in range(first, first + count):
        from test_gpu_textures import textured_scene  # random texels, normal maps, uv transforms on every primitive kind
        for kind, make in (("random", random_scene), ("extreme", extreme_scene), ("textured", textured_scene), ("analytic", analytic_scene),
                           ("mirrors", lambda sd: random_scene(sd, dielectric=False))):  # mirrors / glossy, nothing refracts: the straight-line kernel's depth loop
            scene, cam = make(seed)
            ps = O.pack(scene)
            hs = host_glue.host_scene(scene)
            for mode, tr, om in (("flat", H.TRAVERSE_FLAT, O.MODE_FLAT), ("kd", H.TRAVERSE_KD, O.MODE_KD), ("hier", H.TRAVERSE_HIER, O.MODE_HIER)):
                if mode not in modes:
                    continue
                n_renders += 1
                r = host.Renderer(hs, tr, kd_depth=8)
                rgb, linear, st = r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=samples, seed=seed, sample_mode=H.SAMPLE_RNG, stats=True)
                ref = O.render(ps, cam, w, h, samples=samples, seed=seed, jitter=O.JITTER_RNG, mode=om, kd_depth=8)
                plain, plain_linear, _ = r.render(host_glue.cam10(cam), w, h, default_background(w, h), samples=samples, seed=seed, sample_mode=H.SAMPLE_RNG)  # the timed (non-counting) instantiation
                if not np.array_equal(plain, rgb) or not np.array_equal(plain_linear, linear):
                    bad_total += 1
                    out(f"MISMATCH seed {seed} {kind} {mode}: the counting and the plain instantiation render different images")
                bad = (rgb != ref.rgb).any(axis=2)
                lin_bad = (linear.view(np.uint64) != ref.linear.view(np.uint64)).any(axis=2) & ~bad  # the device pow is glibc's: f64 means identical where the pixel is
                if kind != "textured" and lin_bad.any():
                    bad_total += int(lin_bad.sum())
                    out(f"MISMATCH seed {seed} {kind} {mode}: {int(lin_bad.sum())} pixels whose f64 mean differs in the last bits")
                # sphere uv goes through atan2 / acos: a last-bit difference may move a sample across a texel edge - but only where the oracle saw a lookup within
                # 4096 ulps of one (po_stats.tex_sphere_near_edge, round 5); a scene without such a lookup must match exactly
                if kind == "textured" and ref.stats["tex_sphere_near_edge"] > 0 and bad.sum() <= ref.stats["tex_sphere_near_edge"]:
                    tex_edge += int(bad.sum()); bad[:] = False
                rays_equal = all(st[k] == ref.stats[k] for k in ("primary", "shadow", "reflect", "refract", "hits"))
                if bad.any() or not rays_equal or st["stack_overflow"]:
                    bad_total += int(bad.sum()) + (0 if bad.any() else 1)
                    out(f"MISMATCH seed {seed} {kind} {mode}: {int(bad.sum())} pixels, rays_equal={rays_equal}, first {np.argwhere(bad)[:4].tolist()}, "
                        f"kd_plane_miss gpu {st['kd_plane_miss']} oracle {ref.stats['kd_plane_miss']}")
                r.close()
    return bad_total, tex_edge, n_renders


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (128, 96)
    samples = int(sys.argv[5]) if len(sys.argv) > 5 else 2  # 2: 32 pixels x 2 samples per wavefront; 32: 2 pixels x 4 chunks x 8; 64: one pixel
    bad_total, tex_edge, n_renders = run(first, count, w, h, samples, out=lambda m: print(m, flush=True))
    print(f"fuzz done: seeds {first}..{first + count - 1} ({n_renders} scenes x modes, each the counting and the plain instantiation, {w}x{h}x{samples}), {bad_total} differing pixels in total ({tex_edge} texel-edge pixels in textured scenes tolerated)")


if __name__ == "__main__":
    main()
