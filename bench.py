#!/usr/bin/env python3
"""Benchmark of the MI355X ray-cast/shade path: Mray/s (primary + shadow + secondary rays) at
1920x1080, SAMPLES=64 (BASELINE.json `metric`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload big-scene] [--traversal flat|kd]

A "step" is one full frame of the workload, inputs resident in HBM (scene uploaded once, background
rows on the device) when the timed region starts. With N > 1 (launched by torch.distributed.run, one
rank per GPU) the frame's 8x8 tiles are dealt round-robin to the ranks, every rank renders its tiles
into a compact device buffer, and ONE gather (RCCL over xGMI; `--backend gloo` on CPU tensors for
tests) brings them to rank 0, which scatters them into the row-major image: fixed total work, so
`scaling` is "strong". Rank 0 prints one JSON line.

`python bench.py --gpus N` without a launcher starts its own N rank processes (before anything touches the GPU).

The line also carries
  roofline     : HBM-side bytes per launch of the render kernel (committed rocprofv3 PMC profile of this command)
                 over its launch duration measured live with HIP events, against the copy bandwidth measured
                 on this box; SURVEY 8(d)'s algorithmic bytes and the f64-VALU ceiling under their own names;
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

VALU_PEAK_TFLOPS = 39.3  # 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz scalar-per-lane VALU ops/s (the guide's 157.3 TF FP32 vector peak = this x 2 for fma x 2 for packed f32)
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s

WORKLOADS = {
    # name: (example scene, big-scene n, width, height, samples)
    "big-scene": ("big-scene", 10, 1920, 1080, 64),
    "mirror": ("entering-the-mirror-dimension", 0, 1920, 1080, 64),
    "cows": ("macho-cows", 0, 1280, 720, 16),
    "primitives": ("primitives-simple", 0, 800, 600, 1),
    "triangle": ("single-triangle", 0, 256, 256, 1),
    "aquarium": ("transmission-refraction", 0, 1920, 1080, 16),  # glass + water to depth 10, textured KDMesh fish, normal maps
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


def algorithmic_flops(st, n_lights, traversal):
    """SURVEY §8(d) "algorithmic flops per test" (f64 mul / add / div / sqrt / pow each = 1; an fma of
    the f32 box test = 2): ray->model transform 30 + an average analytic primitive 60 per primitive
    test; triangle 50; mesh box test 30 + 96; tree step 48 (two boxes x (6 fma + 12 min/max)) in this
    build's tree or 9 per k-d split; per shaded hit 60 (point, normal, normalise) + 43 per light."""
    node = 9 if traversal == "kd" else 48
    return (node * st["n_inner"] + 90 * st["n_analytic"] + 50 * st["n_tri"] + 126 * st["n_bbox"]
            + st["hits"] * (60 + 43 * n_lights))


def measured_profile(workload, traversal, n_gpus):
    """The committed rocprofv3 PMC summary of this workload (profiles/traffic.json, written by profiles/summarise.py
    from separate --pmc passes over this very command): HBM-side bytes per render-kernel launch (FETCH_SIZE x 2 on
    gfx950 + WRITE_SIZE, KB -> bytes), lanes active per VALU instruction, VALU busy. None when no profile of this
    exact workload is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        key = f"{workload}/{traversal}/gpus{n_gpus}"
        entry = t.get(key)
        # the kernel variant is part of what was profiled (profiles/r02/notes.md section 16): instantiation "..., 2>" = 4 waves per SIMD
        four = entry is not None and ", 2>" in entry.get("kernel", "")
        waves = os.environ.get("PORTRAYER_WAVES")
        if waves == "3":
            return t.get(key + "/waves3") if four else entry
        if waves == "4":
            return entry if four else None
        return entry
    except (OSError, ValueError):
        return None


def roofline_block(workload, traversal, world, at_config_size, algorithmic, kernel_s, copy_gbps, counts, total, rays_frame, n_lights):
    """`achieved` / `frac`: HBM-side bytes per launch of the render kernel as the rocprofv3 PMC passes committed under
    profiles/ measured them for this workload (FETCH_SIZE x 2 + WRITE_SIZE, per the MI355X guide's gfx950 correction)
    over the kernel's launch duration measured live (HIP events on the launch stream), against `peak` = the copy
    bandwidth measured on THIS box just now; the 8 TB/s datasheet figure is `spec_peak`. SURVEY 8(d)'s ALGORITHMIC
    bytes are served by L1 / L2 on every reference scene (0.3 MB scenes), so that figure is reported under its own
    name and never as a fraction of the HBM roofline. What bounds the kernel is the f64 VALU: `valu`."""
    prof = measured_profile(workload, traversal, world) if at_config_size else None
    traffic = prof.get("hbm_bytes_per_launch") if prof else None
    achieved = traffic / kernel_s / 1e9 if traffic else None
    flops = algorithmic_flops(counts, n_lights, traversal)
    return {"bound": "hbm", "achieved": achieved, "peak": copy_gbps, "unit": "GB/s", "frac": (achieved / copy_gbps) if achieved else None,
            "traffic": traffic, "spec_peak": HBM_PEAK_GBPS,
            "basis": ("HBM-side bytes per launch from %s" % prof.get("source", "profiles/traffic.json")) if traffic else "no PMC profile of this exact workload is committed: achieved / frac are null",
            "kernel": "pt_render_kernel", "kernel_ms": kernel_s * 1e3,
            "algorithmic": {"bytes_per_launch": algorithmic, "GBps": algorithmic / kernel_s / 1e9,
                            "over_measured_copy_bandwidth": algorithmic / kernel_s / 1e9 / copy_gbps,
                            "note": "SURVEY 8(d) per-ray operand bytes x the kernel's own counters; cache-served, NOT HBM traffic (a ratio above 1 is what that means)"},
            # MI355X vector f64 = 78.6 TFLOP/s counting an fma as 2; parity forbids contraction, so 39.3 T mul-or-add/s
            "valu": {"achieved": flops / kernel_s / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / kernel_s / 1e12 / VALU_PEAK_TFLOPS,
                     "lanes_active_of_64": prof.get("lanes_active") if prof else None, "valu_busy": prof.get("valu_busy") if prof else None},
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
    for _ in range(repeats):
        _, _, st = r.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False)
        best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
    prep = r.prepare_ms()
    r.close()
    return {"kernel_ms_per_frame": best, "rays_per_frame": rays, "Mray_per_s": rays / best / 1e3, "counters": c, "prepare_ms": prep}


def secondary_workload(host, H, lib, name, device, copy_gbps):
    """big-soup (1.25 M baked triangles: the one input beyond the caches) and the mirror scene (reflection recursion) at the
    metric's size, 1920x1080 SAMPLES=64, flat_scene semantics, kernel time from HIP events."""
    import numpy as np
    example, n, _, _, _ = WORKLOADS[name]
    w, h, s = 1920, 1080, 64
    scene = host.Scene.example(example, n=n or 10)
    v = np.arange(h, dtype=np.float64) / float(h)
    bg = np.ascontiguousarray(np.array([0.2, 0.4, 0.6])[None, :] * (1.0 - v)[:, None] + np.array([0.0, 0.0, 1.0])[None, :] * v[:, None])
    t = timed_frame(host, H, scene, H.TRAVERSE_FLAT, device, w, h, s, bg, repeats=2)
    c = t.pop("counters")
    prof = measured_profile(name + "@1920x1080x64", "flat", 1)
    traffic = prof.get("hbm_bytes_per_launch") if prof else None
    gbps = traffic / (t["kernel_ms_per_frame"] * 1e-3) / 1e9 if traffic else None
    return dict(t, workload=f"{example} 1920x1080 SAMPLES=64, flat", traffic=traffic, hbm_GBps=gbps, frac_of_measured_copy_bandwidth=(gbps / copy_gbps) if gbps else None,
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
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true", help="testing: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="profiling: only the headline kernel (no host-buffer pass, no default-semantics pass)")
    ap.add_argument("--check", action="store_true", help="compare rank 0's assembled image with a single-GPU render")
    ap.add_argument("--no-pipeline", action="store_true", help="nccl path: wait for each frame's gather before rendering the next")
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
    if use_dist:
        # torch first: its bundled HIP runtime and ours share a SONAME; loaded in this order the
        # process ends up with ONE runtime.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    import ctypes as C

    import numpy as np

    from portrayer_amd import _hip as H
    from portrayer_amd import host

    example, n, w, h, s = WORKLOADS[args.workload]
    w, h, s = args.width or w, args.height or h, args.samples or s
    device = 0 if args.same_device else local_rank
    t_prep0 = time.perf_counter()
    scene = host.Scene.example(example, n=n or 10)  # the product's C++ scene scripts (examples/*.cpp), synthetic variants included
    traverse = {"kd": H.TRAVERSE_KD, "hier": H.TRAVERSE_HIER}.get(args.traversal, H.TRAVERSE_FLAT)
    t_prep1 = time.perf_counter()
    renderer = host.Renderer(scene, traverse, kd_depth=10, device=device)  # flatten + build + upload: once, outside the timed region
    t_prep2 = time.perf_counter()
    ctx = renderer.context
    lib = H.lib()
    cam = host.camera(scene.camera, w, h)
    n_lights = scene.export()["n_lights"]

    def check(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with {rc}: {lib.pt_last_error(ctx).decode()}")

    v = np.arange(h, dtype=np.float64) / float(h)  # the example scripts' background closure, one colour per row
    bg = np.ascontiguousarray(np.array([0.2, 0.4, 0.6])[None, :] * (1.0 - v)[:, None] + np.array([0.0, 0.0, 1.0])[None, :] * v[:, None])
    d_bg = C.c_void_p()
    check(lib.pt_device_alloc(ctx, bg.nbytes, C.byref(d_bg)), "pt_device_alloc")
    check(lib.pt_copy_to_device(ctx, d_bg, bg.ctypes.data_as(C.c_void_p), bg.nbytes), "pt_copy_to_device")

    def params(stats):
        return H.PtRenderParams(w, h, H.PtRect(0, 0, w - 1, h - 1), s, 0, H.SAMPLE_RNG, 1, rank if args.share == 1 else args.share_rank, world if args.share == 1 else args.share, 1 if stats else 0)

    p = params(False)
    compact_bytes = int(lib.pt_compact_bytes(C.byref(p)))
    if use_dist:
        dev = torch.device("cpu") if args.backend == "gloo" else torch.device(f"cuda:{device}")
        if args.backend == "nccl":
            torch.cuda.set_device(device)
        # two sets of buffers: frame k+1 is rendered while frame k's tiles travel (nccl path)
        mine_ts = [torch.empty(compact_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        gathered_ts = [torch.empty(compact_bytes * world, dtype=torch.uint8, device=dev) if rank == 0 else None for _ in range(2)]
        gather_lists = [list(g.chunk(world)) if rank == 0 else None for g in gathered_ts]  # views: the gather lands rank-major in one buffer
        mine_t, gathered_t, gather_list = mine_ts[0], gathered_ts[0], gather_lists[0]
    # renders go to a stream of their own on the RCCL path, so that waiting for torch's current stream (the gather's
    # hand-over) never waits for the render of the next frame
    render_stream = torch.cuda.Stream(device=dev) if (use_dist and args.backend == "nccl") else None
    hip_stream = C.c_void_p(render_stream.cuda_stream) if render_stream is not None else None
    d_mine = C.c_void_p()
    if use_dist and args.backend == "gloo":
        check(lib.pt_device_alloc(ctx, compact_bytes, C.byref(d_mine)), "pt_device_alloc")
    d_full = C.c_void_p(); d_gath = C.c_void_p()
    if rank == 0:
        check(lib.pt_device_alloc(ctx, w * h * 3, C.byref(d_full)), "pt_device_alloc")
        if use_dist and args.backend == "gloo":
            check(lib.pt_device_alloc(ctx, compact_bytes * world, C.byref(d_gath)), "pt_device_alloc")

    def sync_all():
        if use_dist:
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    def step(stats=False):
        """One frame: render own tiles -> (gather -> untile on rank 0). Returns this rank's pt_stats."""
        pp = params(stats)
        st = H.PtStats()
        if not use_dist:
            target, compact = d_full, 0
        elif args.backend == "nccl":
            target, compact = C.c_void_p(mine_t.data_ptr()), 1  # render straight into the tensor RCCL sends from
        else:
            target, compact = d_mine, 1
        check(lib.pt_render_device(ctx, C.byref(cam), d_bg, C.byref(pp), compact, target, hip_stream), "pt_render_device")
        check(lib.pt_render_finish(ctx, C.byref(st)), "pt_render_finish")  # waits for the kernel (HIP event)
        if use_dist:
            if args.backend == "gloo":
                check(lib.pt_copy_from_device(ctx, C.c_void_p(mine_t.data_ptr()), d_mine, compact_bytes), "pt_copy_from_device")
            dist.gather(mine_t, gather_list, dst=0)  # the single collective of the frame
            if rank == 0:
                if args.backend == "nccl":
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
    keys = ["primary", "shadow", "reflect", "refract", "hits", "n_inner", "n_leaf", "n_analytic", "n_tri", "n_bbox"]
    total = dict(counts)
    if use_dist:
        t = torch.tensor([counts[k] for k in keys], dtype=torch.int64, device=mine_t.device)
        dist.all_reduce(t)
        for k, x in zip(keys, t.tolist()):
            total[k] = int(x)
    rays_frame = total["primary"] + total["shadow"] + total["reflect"] + total["refract"]
    if os.environ.get("PT_DUMP_COUNTERS") and rank == 0:  # kernel experiments (profiles/ab.sh builds with -DPT_PHASE_TIMING)
        print("counters", json.dumps({k: int(v) if isinstance(v, (int, np.integer)) else v for k, v in counts.items()}), file=sys.stderr)

    pipelined = use_dist and args.backend == "nccl" and not args.no_pipeline
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms = []
    for k in range(args.steps):
        kernel_ms.append((step_pipelined(k) if pipelined else step())["kernel_ms"])
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
        mine_bytes = algorithmic_bytes(counts, n_lights, compact_bytes // 3 if use_dist else w * h, args.traversal)
        at_config_size = (w, h, s) == WORKLOADS[args.workload][2:] and args.share == 1
        copy_gbps = C.c_double(0.0)
        check(lib.pt_measure_copy_bandwidth(ctx, 1 << 30, 5, C.byref(copy_gbps)), "pt_measure_copy_bandwidth")  # this box's HBM roofline
        prep = renderer.prepare_ms()
        out = {
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
                       "partition": f"8x8 tiles round-robin over {world} rank(s), one gather" if world > 1 else "single GPU",
                       "collective": ({"backend": "nccl = RCCL over xGMI" if args.backend == "nccl" else args.backend, "ranks_in_group": dist.get_world_size(),
                                       "per_frame": "one gather of %d B per rank" % compact_bytes} if use_dist else None),
                       "rays_per_frame": rays_frame,
                       # what a drop-in pays before the first pixel (the reference converts the scene inside every render call, render.rs:115-126)
                       "prepare_ms": dict(prep, scene_script=(t_prep1 - t_prep0) * 1e3, renderer_total=(t_prep2 - t_prep1) * 1e3),
                       "rays": {k: total[k] for k in ("primary", "shadow", "reflect", "refract")}},
            "roofline": roofline_block(args.workload, args.traversal, world, at_config_size, mine_bytes, mean_kernel_s, float(copy_gbps.value),
                                       counts, total, rays_frame, n_lights),
        }
        if ok is not None:
            out["config"]["assembled_image_equals_single_gpu_render"] = ok
        if world == 1 and not args.no_extras:
            # the same frame through pt_render (host buffers in and out: background upload, image
            # round trip over PCIe): reported for reference, never used as `value`
            img = np.zeros((h, w, 3), dtype=np.uint8)
            _, _, hst = renderer.render(scene.camera, w, h, bg, samples=s, seed=0, sample_mode=H.SAMPLE_RNG, into=img, want_linear=False)
            out["config"]["host_buffer_path"] = {"ms_per_frame": hst["total_ms"], "Mray_per_s": rays_frame / hst["total_ms"] / 1e3}
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
