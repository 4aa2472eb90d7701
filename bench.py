#!/usr/bin/env python3
"""Benchmark of the MI355X ray-cast/shade path: Mray/s (primary + shadow + secondary rays) at
1920x1080, SAMPLES=64 (BASELINE.json `metric`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload big-scene] [--traversal flat|kd|hier]

A "step" is one full frame of the workload, inputs resident in HBM (scene uploaded once, background
rows on the device) when the timed region starts. With N > 1 (launched by torch.distributed.run, one
rank process per GPU, or by this script itself) the frame's 8x8 tiles are dealt round-robin to the GPUs, every
GPU renders its tiles into a compact device buffer, and ONE gather (RCCL over xGMI) brings them to GPU 0,
which scatters them into the row-major image: fixed total work, so `scaling` is "strong".
  --via node  (default) the product's own multi-GPU path: rank 0 drives pt_node_* (include/portrayer_hip.h: one context
              per GPU in one process, ncclCommInitAll + one grouped ncclGather, pt_untile_device); the other rank
              processes take part in the barriers only. If the node cannot be set up, the run falls back to
  --via torch the same partition with one process per GPU and torch.distributed's gather (backend nccl = RCCL;
              `--backend gloo` on CPU tensors for tests) - the cross-check of the node path.
Rank 0 prints one JSON line.

The line also carries
  roofline     : the kernel is bound by VALU issue, not by HBM or MFMA: `frac` = the time the vector ALUs need
                 for SURVEY 8(d)'s algorithmic operations of one launch (f64 operations and the instructions the shipped
                 tree step issues, one slot each, at 39.3 T slots/s) over the kernel's launch duration measured live with
                 HIP events; `issue_frac` = all the vector units issued (committed PMC profile), `f64_frac` = the f64
                 work alone; HBM-side bytes per launch
                 (`traffic`, from the committed rocprofv3 PMC profile of this very kernel instantiation, else null)
                 beside the bytes the frame needs, so that wasted traffic shows;
  secondary    : big-soup and the mirror scene at the metric's size (N = 1 default run);
  cpu_baseline : the CPU oracle (a C restatement of the reference, kind "port") timed on this box's
                 host cores on a bounded sample of the same workload: median of 3, pixel loop only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VALU_PEAK_TOPS = 39.3  # 256 CUs x 4 SIMDs x 16 f64 lanes x 2.4 GHz lane-operations/s (MI355X_MICROARCH.md: 78.6 TFLOP/s vector f64 counts an fma as 2; f32 issues at twice this rate)
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s

WORKLOADS = {
    # name: (example scene, big-scene n, width, height, samples)
    "big-scene": ("big-scene", 10, 1920, 1080, 64),
    "mirror": ("entering-the-mirror-dimension", 0, 1920, 1080, 64),
    "cows": ("macho-cows", 0, 1280, 720, 16),
    "primitives": ("primitives-simple", 0, 800, 600, 1),
    "triangle": ("single-triangle", 0, 256, 256, 1),
    "aquarium": ("transmission-refraction", 0, 1920, 1080, 16),  # glass + water to depth 10, textured KDMesh fish, normal maps
    "water-glass": ("water-glass", 0, 1920, 1080, 16),  # glossy table, glass, water, textures; FLAT and HIER differ visibly on this one
    # synthetic, not reference scenes: the big-scene generator over cow.obj instances / baked triangles
    "big-mesh": ("synthetic:big-mesh", 6, 1920, 1080, 16),
    "big-soup": ("synthetic:big-soup", 6, 1920, 1080, 16),
}


def algorithmic_bytes(st, n_lights, pixels, traversal):
    """B_ray = 56 + node bytes + 104 n_analytic + 72 n_tri + 48 n_bbox + H (168 + 80 + 120 L),
    summed over all rays of one launch, plus 3 B written and 24 B of background read per pixel. A k-d split
    record is 16 B (plane + two child indices); a node of this build's two-child bounding-volume tree is
    56 B (two f32 boxes + two child references)."""
    rays = st["primary"] + st["shadow"] + st["reflect"] + st["refract"]
    node_bytes = 16 if traversal == "kd" else 56
    return (56 * rays + node_bytes * st["n_inner"] + 104 * st["n_analytic"] + 72 * st["n_tri"] + 48 * st["n_bbox"]
            + st["hits"] * (168 + 80 + 120 * n_lights) + 27 * pixels)


def algorithmic_ops(st, n_lights, traversal):
    """SURVEY §8(d) "algorithmic flops per test" as VALU issue slots (one lane-operation each: f64 mul / add / div / sqrt /
    pow = 1; an f32 fma of the box test = 1 slot, 2 flops). Returns (f64 slots, f32 slots, flops).
    f64: ray -> model transform 30 + an average analytic primitive 60 per primitive test; triangle 50; mesh box test 30 + 96;
    9 per k-d split; per shaded hit 60 (point, normal, normalise) + 43 per light.
    f32: a step of this build's tree = two boxes x (6 fma + 12 min / max)."""
    f64 = (90 * st["n_analytic"] + 50 * st["n_tri"] + 126 * st["n_bbox"] + st["hits"] * (60 + 43 * n_lights)
           + (9 * st["n_inner"] if traversal == "kd" else 0))
    f32 = 0 if traversal == "kd" else 36 * st["n_inner"]
    flops = f64 + (0 if traversal == "kd" else 48 * st["n_inner"])
    return f64, f32, flops


def kernel_name(st):
    """The instantiation pt_stats says ran (ABI 6: kernel_mode / kernel_variant), spelled like rocprofv3's kernel trace spells it."""
    v = st["kernel_variant"]
    waves, interp, park, tex, fork = v & 15, bool(v & 16), bool(v & 32), bool(v & 128), bool(v & 256)
    t = "true" if tex else "false"
    if interp:
        return f"void pt_render_kernel<{st['kernel_mode']}, false, {t}, {3 if fork else (1 if park else (2 if waves == 4 else 0))}>(PtRenderArgs)"
    if v & 512:
        return f"void pt_render_simple_kernel<{st['kernel_mode']}, false, {t}, {waves}, true>(PtRenderArgs)"
    return f"void pt_render_simple_kernel<{st['kernel_mode']}, false, {t}, {waves}, false>(PtRenderArgs)"


def measured_profile(key, kernel):
    """The committed rocprofv3 PMC summary of a workload (profiles/traffic.json, written by profiles/summarise.py from separate
    --pmc passes over this very command): HBM-side bytes per render-kernel launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
    KB -> bytes), lanes active per VALU instruction, VALU busy. None unless a profile of this exact workload AND of the kernel
    instantiation that is running now is committed - a profile of another kernel says nothing about this one."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None
    for k in (key, key + "/waves3", key + "/waves4"):
        e = t.get(k)
        if e is not None and e.get("kernel") == kernel:
            return e
    return None


def needed_hbm_bytes(scene_export, pixels, n_chunks):
    """What one launch of the render kernel has to move through HBM if nothing is read twice: the scene once (node records,
    triangles, trees), at most one background colour per pixel, and one 24-byte chunk sum per (pixel, 8-sample chunk) written
    (the finishing pass, a kernel of its own, reads them and writes the 3 bytes per pixel)."""
    n = int(len(scene_export["prim_type"]))
    tris = int(scene_export["mesh_tri_off"][-1]) + int(scene_export["n_triangles"])
    scene = n * ((12 + 12 + 9) * 8 + 16) + tris * 72 + (2 * n + tris) * 64
    return scene + pixels * 24 + 24 * pixels * n_chunks


def roofline_block(key, at_config_size, st, algorithmic, kernel_s, copy_gbps, counts, total, rays_frame, n_lights, traversal, needed_bytes):
    """The ceiling of this kernel is VALU issue (`bound`): every scene of the reference fits L2, a tree node is fetched once per
    wavefront through the scalar cache, and there is no contraction for the matrix cores. `achieved` = SURVEY 8(d)'s algorithmic
    lane-operations of one launch per second, in f64-rate units (an f32 box-test instruction takes half the issue time of an
    f64 one on gfx950, so it counts 1/2); `peak` = 39.3 T/s (256 CUs x 4 SIMDs x 16 f64 lanes x 2.4 GHz; parity forbids fma,
    so one operation per slot); `frac` = achieved / peak = the time the vector ALUs need for the algorithmic work over the
    launch duration measured live (HIP events on the launch stream). `traffic` = HBM-side bytes per launch from the committed
    PMC profile of THIS kernel instantiation (null when none is committed); `hbm` sets it against what the frame needs."""
    name = kernel_name(st)
    prof = measured_profile(key, name) if at_config_size else None
    traffic = prof.get("hbm_bytes_per_launch") if prof else None
    f64, f32, flops = algorithmic_ops(counts, n_lights, traversal)
    achieved = (f64 + 0.5 * f32) / kernel_s / 1e12
    # The tree step, told straight: `frac` credits a node visit with the TEXTBOOK two-box slab test - 36 f32 operations, counted as 18
    # issue slots (as if every pair of them were one packed instruction). What the shipped step of the wave-uniform walks ISSUES per node visit:
    # 6 packed fmas + 8 min / max + 3 compares = 17 instructions where the wavefront's rays share their direction signs (mesh-free scenes, plain
    # kernel), 6 + 20 + 3 = 29 in the per-lane form (scenes with mesh instances). An instruction is one issue slot whatever its width on gfx950
    # (a plain f32 instruction does not issue faster than an f64 one; only PACKED f32 doubles the work per slot), so the honest figure replaces
    # 18 by 17 / 29; the third figure is what one gets by calling a shipped f32 instruction half a slot.
    step_slots = None if traversal == "kd" else (17 if st["kernel_mode"] in (3, 6) else 29)
    tree_step = None
    if step_slots is not None:
        n_in = counts["n_inner"]
        tree_step = {"textbook_f32_ops_per_node_visit": 36, "counted_in_frac_as_issue_slots": 18, "shipped_instructions_per_node_visit": step_slots,
                     "frac_with_the_shipped_step": (f64 + step_slots * n_in) / kernel_s / 1e12 / VALU_PEAK_TOPS,
                     "frac_if_a_shipped_f32_instruction_were_half_a_slot": (f64 + 0.5 * step_slots * n_in) / kernel_s / 1e12 / VALU_PEAK_TOPS,
                     "frac_f64_work_alone": f64 / kernel_s / 1e12 / VALU_PEAK_TOPS}
    # the top-level `frac` is the SHIPPED-instruction figure (every instruction the tree step really issues = one slot), not the
    # textbook one that flattered by 0.01; `achieved` is restated to match, the textbook figure stays in valu.tree_step. Beside it the two numbers
    # that bracket it: `issue_frac` = what the vector units issued at all (SQ_THREAD_CYCLES_VALU over t x peak, from the committed PMC profile of this
    # kernel instantiation; null without one) and `f64_frac` = SURVEY 8(d)'s f64 work alone (none of the builder's own box arithmetic).
    if tree_step is not None:
        achieved = (f64 + step_slots * counts["n_inner"]) / kernel_s / 1e12
    return {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s (issue slots: one per f64 operation, one per shipped tree-step instruction)",
            "frac": achieved / VALU_PEAK_TOPS,
            "issue_frac": prof.get("valu_issue_frac") if prof else None, "f64_frac": f64 / kernel_s / 1e12 / VALU_PEAK_TOPS,
            "traffic": traffic, "kernel": name, "kernel_ms": kernel_s * 1e3,
            "valu": {"f64_ops_per_launch": f64, "f32_ops_per_launch": f32, "flops_per_launch_fma_as_2": flops, "TFLOP_per_s_fma_as_2": flops / kernel_s / 1e12,
                     "tree_step": tree_step,
                     "issued": ({"lanes_active_of_64": prof.get("lanes_active"), "valu_busy": prof.get("valu_busy"),
                                 "thread_cycles_valu_over_peak": prof.get("valu_issue_frac")} if prof else None)},
            "hbm": {"needed_bytes": needed_bytes, "measured_bytes": traffic, "waste_ratio": (traffic / needed_bytes) if traffic else None,
                    "GBps": (traffic / kernel_s / 1e9) if traffic else None, "measured_copy_GBps": copy_gbps, "spec_GBps": HBM_PEAK_GBPS,
                    "frac_of_measured_copy": (traffic / kernel_s / 1e9 / copy_gbps) if traffic else None,
                    "profile": prof.get("source") if prof else "no PMC profile of this kernel instantiation on this workload is committed"},
            "algorithmic": {"bytes_per_launch": algorithmic, "GBps": algorithmic / kernel_s / 1e9,
                            "note": "SURVEY 8(d) per-ray operand bytes x the kernel's own counters; served by the scalar cache / L2, NOT HBM traffic"},
            "per_ray": {"inner_nodes": total["n_inner"] / rays_frame, "primitive_tests": total["n_analytic"] / rays_frame,
                        "triangle_tests": total["n_tri"] / rays_frame}}


def timed_frame(host, H, scene, traverse, device, w, h, s, bg, repeats=3):
    """One scene through pt_render: its own ray count (an untimed counting pass) and the best kernel time of `repeats` frames."""
    import numpy as np
    r = host.Renderer(scene, traverse, device=device)
    img = np.zeros((h, w, 3), dtype=np.uint8)
    _, _, c = r.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False, stats=True)
    rays = c["primary"] + c["shadow"] + c["reflect"] + c["refract"]
    best = None
    st = None
    for _ in range(repeats):
        _, _, st = r.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False)
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    prep = r.prepare_ms()
    r.close()
    return {"kernel_ms_per_frame": best, "rays_per_frame": rays, "Mray_per_s": rays / best / 1e3, "kernel": kernel_name(st), "counters": c, "prepare_ms": prep}


def row_background(h):
    """the example scripts' background closure (one colour per row), e.g. examples/big-scene.rs:94"""
    import numpy as np
    v = np.arange(h, dtype=np.float64) / float(h)
    return np.ascontiguousarray(np.array([0.2, 0.4, 0.6])[None, :] * (1.0 - v)[:, None] + np.array([0.0, 0.0, 1.0])[None, :] * v[:, None])


def secondary_workload(host, H, lib, name, device, copy_gbps):
    """big-soup (1.25 M baked triangles: the one input beyond the caches) and the mirror scene (reflection recursion) at the
    metric's size, 1920x1080 SAMPLES=64, flat_scene semantics, kernel time from HIP events."""
    example, n, _, _, _ = WORKLOADS[name]
    w, h, s = 1920, 1080, 64
    scene = host.Scene.example(example, n=n or 10)
    t = timed_frame(host, H, scene, H.TRAVERSE_FLAT, device, w, h, s, row_background(h), repeats=2)
    c = t.pop("counters")
    prof = measured_profile(name + "@1920x1080x64/flat/gpus1", t["kernel"])
    traffic = prof.get("hbm_bytes_per_launch") if prof else None
    gbps = traffic / (t["kernel_ms_per_frame"] * 1e-3) / 1e9 if traffic else None
    f64, f32, _ = algorithmic_ops(c, scene.export()["n_lights"], "flat")
    return dict(t, workload=f"{example} 1920x1080 SAMPLES=64, flat", valu_frac=(f64 + 0.5 * f32) / (t["kernel_ms_per_frame"] * 1e-3) / 1e12 / VALU_PEAK_TOPS,
                traffic=traffic, hbm_GBps=gbps, frac_of_measured_copy_bandwidth=(gbps / copy_gbps) if gbps else None,
                lanes_active_of_64=prof.get("lanes_active") if prof else None,
                per_ray={"inner_nodes": c["n_inner"] / t["rays_per_frame"], "triangle_tests": c["n_tri"] / t["rays_per_frame"]})


def spawn_ranks(n):
    """Starts `n` copies of this script as ranks 0..n-1 of one job (the environment torch.distributed.run
    would give them, rendezvous on 127.0.0.1) and waits for them. Rank 0's stdout is passed through."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: rank(s) failed: " + ", ".join(f"rank {r} rc {rc}" for r, rc in bad), file=sys.stderr)
        return max(abs(rc) for _, rc in bad) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="big-scene", choices=sorted(WORKLOADS))
    ap.add_argument("--traversal", default="flat", choices=["flat", "kd", "hier"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--via", default="node", choices=["node", "torch"], help="N > 1: the product's pt_node_* on rank 0 (default), or torch.distributed's gather with one process per GPU")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="--via torch: the collective's backend")
    ap.add_argument("--same-device", action="store_true", help="testing on a 1-GPU box: every rank uses GPU 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="profiling: only the headline kernel (no host-buffer pass, no default-semantics pass)")
    ap.add_argument("--check", action="store_true", help="compare rank 0's assembled image with a single-GPU render")
    ap.add_argument("--overlap", action="store_true", help="N = 1 measurements: the timed frames two at a time on the context's two streams (the next frame's wavefronts start "
                    "in the places this frame's tail frees); kernel times of overlapped launches cover each other, so `roofline` is not to be read from such a line")
    ap.add_argument("--no-pipeline", action="store_true", help="N > 1: wait for each frame (its gather, its image) before queueing the next")
    ap.add_argument("--force-dist", action="store_true", help="testing: take the torch.distributed path even with one rank")
    ap.add_argument("--share", type=int, default=1, help="testing: render only one rank's tiles of an N-rank partition (no gather)")
    ap.add_argument("--share-rank", type=int, default=0, help="testing: which rank's tiles --share renders")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: this process becomes a launcher. It has not touched the GPU (no
        # torch, no HIP call so far) and never will: it starts N fresh rank processes, one per GPU, relays
        # rank 0's stdout (the JSON line) and exits with the worst return code.
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    dist = torch = None
    use_dist = world > 1 or args.force_dist
    via_node = world > 1 and args.via == "node"
    # the process group: with --via node it only carries the barriers and the fall-back decision (gloo: the rank processes other than 0
    # never touch a GPU, rank 0's GPUs belong to pt_node alone); with --via torch it carries the frame's gather (nccl = RCCL)
    pg_backend = "gloo" if via_node else args.backend
    if use_dist:
        # torch first: its bundled HIP runtime and ours share a SONAME; loaded in this order the process ends up with ONE runtime.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=pg_backend, rank=rank, world_size=world)

    import ctypes as C

    import numpy as np

    from portrayer_amd import _hip as H
    from portrayer_amd import host

    example, n, w, h, s = WORKLOADS[args.workload]
    w, h, s = args.width or w, args.height or h, args.samples or s
    lib = H.lib()
    bg = row_background(h)
    traverse = {"kd": H.TRAVERSE_KD, "hier": H.TRAVERSE_HIER}.get(args.traversal, H.TRAVERSE_FLAT)
    keys = ["primary", "shadow", "reflect", "refract", "hits", "n_inner", "n_leaf", "n_analytic", "n_tri", "n_bbox"]

    # ---------------------------------------------------------------- the product's multi-GPU path: rank 0 drives pt_node_*
    node_note = None
    if via_node:
        state = {}
        if rank == 0:
            try:
                devices = [0] * world if args.same_device else list(range(world))
                os.environ["PORTRAYER_DEVICES"] = ",".join(map(str, devices))  # the C++ Renderer puts the scene on a pt_node of these GPUs
                t0 = time.perf_counter()
                scene = host.Scene.example(example, n=n or 10)
                t1 = time.perf_counter()
                renderer = host.Renderer(scene, traverse, kd_depth=10, device=devices[0])
                t2 = time.perf_counter()
                node = renderer.node
                if not node or lib.pt_node_ranks(node) != world:
                    raise RuntimeError("the renderer did not come up on a pt_node of %d ranks" % world)
                cam = host.camera(scene.camera, w, h)
                pp = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), s, 0, H.SAMPLE_RNG, 1, 0, 1, 0)
                if lib.pt_node_upload_background(node, bg.ctypes.data_as(H._dp), C.byref(pp), None) != 0:
                    raise RuntimeError("pt_node_upload_background: " + lib.pt_node_last_error(node).decode())

                def node_step(stats=False):
                    q = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), s, 0, H.SAMPLE_RNG, 1, 0, 1, 1 if stats else 0)
                    st = H.PtStats()
                    if lib.pt_node_render_resident(node, C.byref(cam), C.byref(q), C.byref(st)) != 0:
                        raise RuntimeError("pt_node_render_resident: " + lib.pt_node_last_error(node).decode())
                    return st.as_dict()
                counts = node_step(stats=True)  # also the proof that the whole path runs before anything is timed

                def node_frames(k, pipelined):
                    """k frames through pt_node_frame_begin / pt_node_frame_end: two frames open at a time (frame i + 1 renders while frame i is
                    gathered and untiled, the host's launches hidden behind the GPUs' work) or one (every frame waited for before the next)."""
                    q = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), s, 0, H.SAMPLE_RNG, 1, 0, 1, 0)
                    st, kms, host_ms = H.PtStats(), [], []
                    hm = (C.c_double * 5)()
                    rk = (C.c_double * world)()
                    state.setdefault("rank_kernel_ms", {})[pipelined] = []

                    def end():
                        if lib.pt_node_frame_end(node, C.byref(st)) != 0:
                            raise RuntimeError("pt_node_frame_end: " + lib.pt_node_last_error(node).decode())
                        lib.pt_node_last_frame_host_ms(node, C.byref(hm))
                        lib.pt_node_last_frame_rank_kernel_ms(node, rk, world)
                        state["rank_kernel_ms"][pipelined].append([float(x) for x in rk])
                        kms.append(st.kernel_ms); host_ms.append([float(x) for x in hm])
                    for i in range(k):
                        if lib.pt_node_frame_begin(node, C.byref(cam), C.byref(q)) != 0:
                            raise RuntimeError("pt_node_frame_begin: " + lib.pt_node_last_error(node).decode())
                        if not pipelined or i > 0:
                            end()
                    if pipelined and k > 0:
                        end()
                    return kms, host_ms
                state.update(scene=scene, renderer=renderer, node=node, cam=cam, step=node_step, frames=node_frames, counts=counts, prep=(t0, t1, t2), devices=devices)
            except Exception as e:  # noqa: BLE001 - whatever went wrong, the run falls back to the torch path and says so
                node_note = f"{type(e).__name__}: {e}"
                os.environ.pop("PORTRAYER_DEVICES", None)
        flag = [node_note]
        dist.broadcast_object_list(flag, src=0)
        node_note = flag[0]
        if node_note is None:
            out = None
            if rank == 0:
                state["frames"](args.warmup, not args.no_pipeline)
            dist.barrier()
            t0 = time.perf_counter()
            kernel_ms = []
            if rank == 0:  # K frames; the last one is closed (its image assembled on GPU 0) before the clock stops
                kernel_ms, _ = state["frames"](args.steps, not args.no_pipeline)
            dist.barrier()
            elapsed = time.perf_counter() - t0
            if rank == 0:  # untimed: the same frames one at a time, for what the host adds to a frame that nothing overlaps
                t1 = time.perf_counter()
                lat_kernel, lat_host = state["frames"](max(args.steps, 3), False)
                state["latency"] = ((time.perf_counter() - t1) * 1e3 / max(args.steps, 3), lat_kernel, lat_host)
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            if rank == 0:
                out = node_report(args, H, host, lib, state, world, w, h, s, example, elapsed, kernel_ms, bg)
            dist.barrier()
            dist.destroy_process_group()
            if out is not None:
                sys.stderr.flush()
                C.CDLL(None).fflush(None)  # RCCL's banner goes through C stdio: flush it before the JSON line
                print(json.dumps(out), flush=True)
            return
        # fall back: a fresh process group for the collective itself (the gloo one carried the decision)
        dist.destroy_process_group()
        os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + 1)
        pg_backend = args.backend
        dist.init_process_group(backend=pg_backend, rank=rank, world_size=world)

    # ---------------------------------------------------------------- one GPU, or one process per GPU with torch.distributed's gather
    device = 0 if args.same_device else local_rank
    t_prep0 = time.perf_counter()
    scene = host.Scene.example(example, n=n or 10)  # the product's C++ scene scripts (examples/*.cpp), synthetic variants included
    t_prep1 = time.perf_counter()
    renderer = host.Renderer(scene, traverse, kd_depth=10, device=device)  # flatten + build + upload: once, outside the timed region
    t_prep2 = time.perf_counter()
    ctx = renderer.context
    cam = host.camera(scene.camera, w, h)
    export = scene.export()
    n_lights = export["n_lights"]

    def check(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with {rc}: {lib.pt_last_error(ctx).decode()}")

    d_bg = C.c_void_p()
    check(lib.pt_device_alloc(ctx, bg.nbytes, C.byref(d_bg)), "pt_device_alloc")
    check(lib.pt_copy_to_device(ctx, d_bg, bg.ctypes.data_as(C.c_void_p), bg.nbytes), "pt_copy_to_device")

    def params(stats):
        return H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), s, 0, H.SAMPLE_RNG, 1, rank if args.share == 1 else args.share_rank, world if args.share == 1 else args.share, 1 if stats else 0)

    p = params(False)
    compact_bytes = int(lib.pt_compact_bytes(C.byref(p)))
    if use_dist:
        dev = torch.device("cpu") if pg_backend == "gloo" else torch.device(f"cuda:{device}")
        if pg_backend == "nccl":
            torch.cuda.set_device(device)
        # two sets of buffers: frame k+1 is rendered while frame k's tiles travel (nccl path)
        mine_ts = [torch.empty(compact_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        gathered_ts = [torch.empty(compact_bytes * world, dtype=torch.uint8, device=dev) if rank == 0 else None for _ in range(2)]
        gather_lists = [list(g.chunk(world)) if rank == 0 else None for g in gathered_ts]  # views: the gather lands rank-major in one buffer
        mine_t, gathered_t, gather_list = mine_ts[0], gathered_ts[0], gather_lists[0]
    # renders go to a stream of their own on the RCCL path, so that waiting for torch's current stream (the gather's
    # hand-over) never waits for the render of the next frame
    render_stream = torch.cuda.Stream(device=dev) if (use_dist and pg_backend == "nccl") else None
    hip_stream = C.c_void_p(render_stream.cuda_stream) if render_stream is not None else None
    d_mine = C.c_void_p()
    if use_dist and pg_backend == "gloo":
        check(lib.pt_device_alloc(ctx, compact_bytes, C.byref(d_mine)), "pt_device_alloc")
    d_full = C.c_void_p(); d_gath = C.c_void_p()
    if rank == 0:
        check(lib.pt_device_alloc(ctx, w * h * 3, C.byref(d_full)), "pt_device_alloc")
        if use_dist and pg_backend == "gloo":
            check(lib.pt_device_alloc(ctx, compact_bytes * world, C.byref(d_gath)), "pt_device_alloc")

    def sync_all():
        if use_dist:
            dist.barrier()
            if pg_backend == "nccl":
                torch.cuda.synchronize()

    def step(stats=False):
        """One frame: render own tiles -> (gather -> untile on rank 0). Returns this rank's pt_stats."""
        pp = params(stats)
        st = H.PtStats()
        if not use_dist:
            target, compact = d_full, 0
        elif pg_backend == "nccl":
            target, compact = C.c_void_p(mine_t.data_ptr()), 1  # render straight into the tensor RCCL sends from
        else:
            target, compact = d_mine, 1
        check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(pp), compact, target, hip_stream), "pt_render_device")
        check(lib.pt_render_finish(ctx, C.byref(st)), "pt_render_finish")  # waits for the kernel (HIP event)
        if use_dist:
            if pg_backend == "gloo":
                check(lib.pt_copy_from_device(ctx, C.c_void_p(mine_t.data_ptr()), d_mine, compact_bytes), "pt_copy_from_device")
            dist.gather(mine_t, gather_list, dst=0)  # the single collective of the frame
            if rank == 0:
                if pg_backend == "nccl":
                    torch.cuda.synchronize()
                    src = C.c_void_p(gathered_t.data_ptr())
                else:
                    check(lib.pt_copy_to_device(ctx, d_gath, C.c_void_p(gathered_t.data_ptr()), compact_bytes * world), "pt_copy_to_device")
                    src = d_gath
                check(lib.pt_untile_device(ctx, C.byref(pp), src, d_full, None), "pt_untile_device")
                check(lib.pt_synchronize(ctx), "pt_synchronize")
        return st.as_dict()

    # Pipelined frames (nccl): the render of frame k runs while frame k-1's gather is in flight and is
    # followed in the stream by frame k-1's untile; nothing waits for a collective except the drain.
    pending = []  # (async gather handle, buffer index, params) of the frame whose tiles are still travelling

    def finish_pending():
        if not pending:
            return
        work, b, pp = pending.pop()
        work.wait()  # the current torch stream waits for the gather ...
        torch.cuda.current_stream().synchronize()  # ... and the host for that stream only; the render in flight is on render_stream
        if rank == 0:
            check(lib.pt_untile_device(ctx, C.byref(pp), C.c_void_p(gathered_ts[b].data_ptr()), d_full, None), "pt_untile_device")

    def step_pipelined(k):
        b = k & 1
        pp = params(False)
        st = H.PtStats()
        check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(pp), 1, C.c_void_p(mine_ts[b].data_ptr()), hip_stream), "pt_render_device")
        finish_pending()  # frame k-1: its untile queues behind frame k's render
        check(lib.pt_render_finish(ctx, C.byref(st)), "pt_render_finish")
        pending.append((dist.gather(mine_ts[b], gather_lists[b], dst=0, async_op=True), b, pp))
        return st.as_dict()

    def drain():
        finish_pending()
        check(lib.pt_synchronize(ctx), "pt_synchronize")

    # counting pass (untimed): ray / node / test counters of this rank's share of one frame
    counts = step(stats=True)
    if counts["stack_overflow"]:
        raise RuntimeError("traversal stack overflow")
    total = dict(counts)
    if use_dist:
        t = torch.tensor([counts[k] for k in keys], dtype=torch.int64, device=mine_t.device)
        dist.all_reduce(t)
        for k, x in zip(keys, t.tolist()):
            total[k] = int(x)
    rays_frame = total["primary"] + total["shadow"] + total["reflect"] + total["refract"]
    if os.environ.get("PT_DUMP_COUNTERS") and rank == 0:  # kernel experiments (profiles/ab.sh builds with -DPT_PHASE_TIMING)
        print("counters", json.dumps({k: int(v) if isinstance(v, (int, np.integer)) else v for k, v in counts.items()}), file=sys.stderr)

    pipelined = use_dist and pg_backend == "nccl" and not args.no_pipeline
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms = []
    last = None
    overlapped = args.overlap and not use_dist
    if overlapped:
        d_alt = C.c_void_p()
        check(lib.pt_device_alloc(ctx, w * h * 3, C.byref(d_alt)), "pt_device_alloc")
        pp2, st2 = params(False), H.PtStats()
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            slot = int(lib.pt_context_next_slot(ctx))
            check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(pp2), 0, d_full if slot == 0 else d_alt, C.c_void_p(lib.pt_context_stream(ctx, slot))), "pt_render_device")
            if k > 0:
                check(lib.pt_render_finish(ctx, C.byref(st2)), "pt_render_finish")
                kernel_ms.append(st2.kernel_ms)
        check(lib.pt_render_finish(ctx, C.byref(st2)), "pt_render_finish")
        kernel_ms.append(st2.kernel_ms)
        check(lib.pt_synchronize(ctx), "pt_synchronize")
        last = st2.as_dict()
    for k in range(0 if overlapped else args.steps):
        last = step_pipelined(k) if pipelined else step()
        kernel_ms.append(last["kernel_ms"])
    if pipelined:
        drain()  # every frame assembled on rank 0 before the clock stops
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=mine_t.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = None
    if args.check and rank == 0:
        img = np.zeros((h, w, 3), dtype=np.uint8)
        check(lib.pt_copy_from_device(ctx, img.ctypes.data_as(C.c_void_p), d_full, img.nbytes), "pt_copy_from_device")
        one = np.zeros((h, w, 3), dtype=np.uint8)
        renderer.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=one, want_linear=False)
        ok = bool(np.array_equal(img, one))

    if rank == 0:
        mean_kernel_s = float(np.mean(kernel_ms)) * 1e-3
        own_pixels = compact_bytes // 3 if use_dist else w * h
        mine_bytes = algorithmic_bytes(counts, n_lights, own_pixels, args.traversal)
        at_config_size = (w, h, s) == WORKLOADS[args.workload][2:] and args.share == 1
        copy_gbps = C.c_double(0.0)
        check(lib.pt_measure_copy_bandwidth(ctx, 1 << 30, 5, C.byref(copy_gbps)), "pt_measure_copy_bandwidth")  # this box's copy bandwidth, beside the 8 TB/s of the data sheet
        prep = renderer.prepare_ms()
        out = headline(args, example, w, h, s, world, rays_frame, elapsed, total)
        out["config"]["collective"] = ({"via": "torch.distributed gather, one process per GPU" + (f" (fell back from pt_node: {node_note})" if node_note else ""),
                                        "backend": "nccl = RCCL over xGMI" if pg_backend == "nccl" else pg_backend, "ranks_in_group": dist.get_world_size(),
                                        "per_frame": "one gather of %d B per rank" % compact_bytes} if use_dist else None)
        # what a drop-in pays before the first pixel (the reference converts the scene inside every render call, render.rs:115-126)
        out["config"]["prepare_ms"] = dict(prep, scene_script=(t_prep1 - t_prep0) * 1e3, renderer_total=(t_prep2 - t_prep1) * 1e3)
        out["roofline"] = roofline_block(f"{args.workload}/{args.traversal}/gpus{world}", at_config_size, last, mine_bytes, mean_kernel_s, float(copy_gbps.value),
                                         counts, total, rays_frame, n_lights, args.traversal, needed_hbm_bytes(export, own_pixels, (s + 7) // 8))
        if ok is not None:
            out["config"]["assembled_image_equals_single_gpu_render"] = ok
        if world == 1 and not args.no_extras:
            # Two frames in flight on the context's two streams (ABI 8, pt_context_stream): the next frame's wavefronts start where this frame's tail frees
            # places. Reported beside `value`, never as it: `value` keeps one frame at a time, so that the kernel time under the roofline stays one launch's own.
            d_alt = C.c_void_p()
            check(lib.pt_device_alloc(ctx, w * h * 3, C.byref(d_alt)), "pt_device_alloc")
            nfr = max(args.steps, 4)
            pp2 = params(False)
            st2 = H.PtStats()
            check(lib.pt_synchronize(ctx), "pt_synchronize")
            t_o = time.perf_counter()
            for k in range(nfr):
                slot = int(lib.pt_context_next_slot(ctx))
                check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(pp2), 0, d_full if slot == 0 else d_alt, C.c_void_p(lib.pt_context_stream(ctx, slot))), "pt_render_device")
                if k > 0:
                    check(lib.pt_render_finish(ctx, C.byref(st2)), "pt_render_finish")
            check(lib.pt_render_finish(ctx, C.byref(st2)), "pt_render_finish")
            check(lib.pt_synchronize(ctx), "pt_synchronize")
            ms_o = (time.perf_counter() - t_o) * 1e3 / nfr
            out["config"]["two_frames_in_flight"] = {"ms_per_frame": ms_o, "Mray_per_s": rays_frame / ms_o / 1e3, "frames": nfr,
                                                     "note": "pt_render_device on pt_context_stream(ctx, k & 1), the older frame closed after the newer one is queued"}
            lib.pt_device_free(ctx, d_alt)
        if world == 1 and not args.no_extras:
            # the same frame through pt_render (host buffers in and out: background upload, image
            # round trip over PCIe): reported for reference, never used as `value`
            img = np.zeros((h, w, 3), dtype=np.uint8)
            _, _, hst = renderer.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False)
            out["config"]["host_buffer_path"] = {"ms_per_frame": hst["total_ms"], "Mray_per_s": rays_frame / hst["total_ms"] / 1e3}
        if world == 1 and not args.no_extras:
            # The same preparation once more in this process ("warm": HIP is initialised, the library's code objects are loaded, the allocator has its pools) and the
            # first frame of that renderer: what a caller of Image::render pays on every call AFTER its first (the reference converts its scene inside every
            # render call, render.rs:115-126); `prepare_ms` above is the process's first, cold.
            tw0 = time.perf_counter()
            again = host.Renderer(scene, traverse, kd_depth=10, device=device)
            tw1 = time.perf_counter()
            img = np.zeros((h, w, 3), dtype=np.uint8)
            _, _, wst = again.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False)
            tw2 = time.perf_counter()
            out["config"]["prepare_ms_warm"] = dict(again.prepare_ms(), renderer_total=(tw1 - tw0) * 1e3, first_frame_host_buffers_ms=(tw2 - tw1) * 1e3, its_kernel_ms=wst["kernel_ms"])
            again.close()
        if world == 1 and args.traversal == "flat" and args.share == 1 and not args.no_extras:
            # `value` is measured in the flat_scene semantics north_star prescribes. The crate's DEFAULT feature set is the
            # hierarchical traversal, which is not image-equivalent to it on every scene and costs more: the same
            # frame in those semantics, measured here so that the disclosure travels with the number (never used as `value`)
            hier = timed_frame(host, H, scene, H.TRAVERSE_HIER, device, w, h, s, bg)
            hier.pop("counters"); hier.pop("prepare_ms")
            out["config"]["default_semantics"] = dict(hier, traversal="hier (the crate built without features, scene.rs:80-120)")
        if world == 1 and args.workload == "big-scene" and at_config_size and not args.no_extras:
            # the workloads where memory / ray incoherence matter, timed by the same run at the metric's size
            out["secondary"] = [secondary_workload(host, H, lib, name, device, float(copy_gbps.value)) for name in ("big-soup", "mirror")]
        if not args.no_cpu_baseline and world == 1 and not example.startswith("synthetic:"):
            out["cpu_baseline"] = cpu_baseline(example, n, w, h, args.traversal)
    else:
        out = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        sys.stderr.flush()
        # RCCL writes its version banner through C stdio, which is flushed at exit - after this line -
        # when stdout is a pipe or a file: flush it now so that the JSON is the last line on stdout
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)  # the one JSON line, last thing on rank 0's stdout


def headline(args, example, w, h, s, world, rays_frame, elapsed, total):
    """The fields of the JSON line every path shares."""
    return {
        "metric": "Mray/s (primary+shadow+secondary) at 1920x1080 SAMPLES=64" if (w, h, s) == (1920, 1080, 64) else f"Mray/s (primary+shadow+secondary) at {w}x{h} SAMPLES={s}",
        "value": rays_frame * args.steps / elapsed / 1e6,
        "unit": "Mray/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{example} (the big-scene generator over cow.obj: 216 instances, 1,253,664 triangles; SURVEY 8d synthetic) {w}x{h} SAMPLES={s}"
                                if example.startswith("synthetic:") else
                                f"{example} ({'1000 analytic primitives, 3 point lights, ' if example == 'big-scene' else ''}reference scene script) {w}x{h} SAMPLES={s}"),
                   "width": w, "height": h, "samples": s, "traversal": args.traversal, "sampling": "counter-based jitter, seed 0",
                   "partition": f"8x8 tiles round-robin over {world} GPU(s), one gather" if world > 1 else "single GPU",
                   "rays_per_frame": rays_frame,
                   "rays": {k: total[k] for k in ("primary", "shadow", "reflect", "refract")}},
    }


def node_report(args, H, host, lib, state, world, w, h, s, example, elapsed, kernel_ms, bg):
    """Rank 0's JSON line of the --via node path: the frame went through pt_node_render_resident."""
    import ctypes as C

    import numpy as np
    counts, node, scene, renderer = state["counts"], state["node"], state["scene"], state["renderer"]
    if counts["stack_overflow"]:
        raise RuntimeError("traversal stack overflow")
    rays_frame = counts["primary"] + counts["shadow"] + counts["reflect"] + counts["refract"]
    export = scene.export()
    out = headline(args, example, w, h, s, world, rays_frame, elapsed, counts)
    ranks = int(lib.pt_node_ranks(node))
    pp = H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), s, 0, H.SAMPLE_RNG, 1, 0, ranks, 0)
    per = int(lib.pt_compact_bytes(C.byref(pp)))
    out["config"]["collective"] = {"via": "pt_node_render_resident (include/portrayer_hip.h): one process, one context per GPU",
                                   "backend": "RCCL: ncclCommInitAll + one grouped ncclGather over xGMI" if lib.pt_node_uses_rccl(node) else "device-to-device copies (ranks share a GPU: RCCL needs distinct devices)",
                                   "uses_rccl": bool(lib.pt_node_uses_rccl(node)), "ranks_in_group": ranks,
                                   "devices": [int(lib.pt_node_device(node, r)) for r in range(ranks)],
                                   "per_frame": "one gather of %d B per rank" % per, "rank_processes": world,
                                   "frames": "pt_node_frame_begin / _end, one frame at a time (--no-pipeline)" if args.no_pipeline else
                                             "pt_node_frame_begin / _end, two frames open on two streams per rank: frame k + 1 starts in the wavefront slots frame k's tail frees and renders while frame k is gathered and untiled"}
    if state.get("latency"):
        frame_ms, lat_kernel, lat_host = state["latency"]
        hm = np.median(np.array(lat_host), axis=0)
        out["config"]["collective"]["one_frame_at_a_time"] = {
            "frame_ms": frame_ms, "slowest_rank_kernel_ms": float(np.median(lat_kernel)), "sum_of_rank_kernels_ms": float(hm[4]),
            "host_begin_ms": float(hm[0]), "host_slowest_rank_launch_ms": float(hm[3]), "host_blocked_on_gpu_ms": float(hm[1]), "host_finish_ms": float(hm[2]),
            "host_overhead_ms": float(hm[0] + hm[2]),
            "frame_minus_sum_of_rank_kernels_ms": frame_ms - float(hm[4]),
            "note": "untimed pass after the timed one, median of the frames; host_overhead_ms = what pt_node_frame_begin and the tail of pt_node_frame_end spend on the host. "
                    "Ranks that share a GPU run their kernels side by side, so frame - sum of kernels says little there; with one GPU per rank the frame is the slowest kernel + gather + this overhead"}
    t0, t1, t2 = state["prep"]
    out["config"]["prepare_ms"] = dict(renderer.prepare_ms(), scene_script=(t1 - t0) * 1e3, renderer_total=(t2 - t1) * 1e3)
    if args.check:
        img = np.zeros((h, w, 3), dtype=np.uint8)
        if lib.pt_node_download_image(node, C.byref(pp), img.ctypes.data_as(H._u8p)) != 0:
            raise RuntimeError("pt_node_download_image: " + lib.pt_node_last_error(node).decode())
        os.environ.pop("PORTRAYER_DEVICES", None)
        one = host.Renderer(scene, {"kd": H.TRAVERSE_KD, "hier": H.TRAVERSE_HIER}.get(args.traversal, H.TRAVERSE_FLAT), kd_depth=10, device=state["devices"][0])
        ref = np.zeros((h, w, 3), dtype=np.uint8)
        one.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=ref, want_linear=False)
        one.close()
        out["config"]["assembled_image_equals_single_gpu_render"] = bool(np.array_equal(img, ref))
    # the roofline of the slowest rank's kernel: the counters are the node's totals, a rank's share is 1 / ranks of them
    share = {k: v / ranks if isinstance(v, (int, float)) and k not in ("kernel_ms", "total_ms", "kernel_mode", "kernel_variant") else v for k, v in counts.items()}
    # The kernel time of a rank is taken from the frames rendered ONE AT A TIME (the untimed pass after the timed one): with two frames open a frame's
    # launch starts in the wavefront slots the previous frame's tail frees, and the HIP events around it then cover a stretch of that frame too.
    seq = np.array(state.get("rank_kernel_ms", {}).get(False) or [[k] * ranks for k in kernel_ms], dtype=np.float64)  # frames x ranks
    per_rank = np.median(seq, axis=0)
    mean_kernel_s = float(per_rank.max()) * 1e-3
    out["config"]["collective"]["per_rank_kernel_ms"] = {"one_frame_at_a_time": [float(x) for x in per_rank], "max_over_mean": float(per_rank.max() / per_rank.mean()),
                                                         "slowest_rank_kernel_ms_in_the_timed_frames": float(np.mean(kernel_ms)) if len(kernel_ms) else None}
    copy_gbps = C.c_double(0.0)
    lib.pt_measure_copy_bandwidth(lib.pt_node_context(node, 0), 1 << 30, 5, C.byref(copy_gbps))
    n_lights = export["n_lights"]
    out["roofline"] = roofline_block(f"{args.workload}/{args.traversal}/gpus{world}", False, counts, algorithmic_bytes(share, n_lights, w * h // ranks, args.traversal), mean_kernel_s,
                                     float(copy_gbps.value), share, counts, rays_frame, n_lights, args.traversal, needed_hbm_bytes(export, w * h // ranks, (s + 7) // 8))
    out["roofline"]["note"] = "per GPU: the slowest rank's kernel time (frames rendered one at a time) against 1 / ranks of the frame's counters"
    return out


def cpu_baseline(example, n, w, h, traversal):
    """The reference's rayon CPU path as far as it can be had here: the oracle (oracle/portrayer_oracle.c, a C restatement
    of the reference - kind "port"; the Rust binary cannot be built on this box), one task per image row on all host
    cores, the same scene at the same resolution at a reduced sample count (every pixel covered). BASELINE.md section 3:
    three runs, the median, scene preparation (flatten, k-d build: render.rs:115-126) outside the timed region - only the
    pixel loop (render.rs:127-150) is timed - in both of the reference's accelerated features (kdtree = its fastest,
    flat_scene = the semantics of `value`). `value` is the k-d figure."""
    import platform
    import oracle_lib as O
    from portrayer_amd import host
    O.build()
    sc = host.Scene.example(example, n=n or 10)
    ps = O.pack_arrays(sc.export())
    cores = os.cpu_count() or 1
    model = platform.processor() or "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), model)
    except OSError:
        pass
    out = {}
    for name, mode, budget in (("kdtree", O.MODE_KD, 4.0), ("flat_scene", O.MODE_FLAT, 2.0)):
        O.render(ps, sc.camera, w, h, samples=1, seed=0, jitter=O.JITTER_RNG, mode=mode, threads=cores)  # calibration pass
        t1 = O.last_render_ms()[1] * 1e-3
        cs = int(max(1, min(16, round(budget / max(t1, 1e-3)))))  # ~`budget` seconds per run
        runs = []
        for _ in range(3):
            r = O.render(ps, sc.camera, w, h, samples=cs, seed=0, jitter=O.JITTER_RNG, mode=mode, threads=cores)
            prep_ms, loop_ms = O.last_render_ms()
            rays = r.stats["primary"] + r.stats["shadow"] + r.stats["reflect"] + r.stats["refract"]
            runs.append((rays / (loop_ms * 1e-3) / 1e6, loop_ms, prep_ms, rays))
        runs.sort()
        med = runs[1]
        out[name] = {"Mray_per_s": med[0], "runs_Mray_per_s": [x[0] for x in runs], "pixel_loop_s": med[1] * 1e-3, "scene_preparation_s_excluded": med[2] * 1e-3,
                     "samples": cs, "rays": med[3]}
    kd = out["kdtree"]
    return {"value": kd["Mray_per_s"], "unit": "Mray/s", "cores": cores, "kind": "port", "cpu_model": model,
            "sample": f"one {w}x{h} frame of the same scene at SAMPLES={kd['samples']} (every pixel covered), reference k-d tree mode (KD_DEPTH=10), median of 3 runs, "
                      f"pixel loop only ({kd['rays']} rays in {kd['pixel_loop_s']:.2f} s on {cores} threads, one task per image row; scene preparation "
                      f"{kd['scene_preparation_s_excluded']:.2f} s excluded); C restatement of the reference, not its Rust binary",
            "modes": out}


if __name__ == "__main__":
    main()
