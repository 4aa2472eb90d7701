"""The multi-GPU protocol of bench.py / DESIGN.md §5 with world size 2 on CPU (gloo): every rank
owns the 8x8 tiles t with t % world == rank, fills a compact tile-major buffer of pt_compact_bytes(),
ONE gather brings them to rank 0, pt_untile_host() scatters them into the row-major image.

No GPU here, so the ranks' pixels come from the oracle (test stand-in for the kernel); what is under
test is the product's partition / compact layout / untile code and the collective."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, w, h, rect, out_path):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import ctypes as C

    import torch
    import torch.distributed as dist

    import oracle_lib as O
    from example_scenes import EXAMPLES
    from portrayer_amd import _hip as H

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = H.lib()
    scene, cam, _ = EXAMPLES["primitives-simple"]()
    full = O.render(scene, cam, w, h, mode=O.MODE_FLAT, threads=2, rect=rect).rgb  # stand-in for the GPU render
    p = H.PtRenderParams(w, h, H.PtRect(*rect), 1, 0, H.SAMPLE_CENTRE, 1, rank, world, 0)
    nbytes = int(lib.pt_compact_bytes(C.byref(p)))
    mine = np.zeros(nbytes, dtype=np.uint8)
    x, y = C.c_uint32(), C.c_uint32()
    owned = 0
    for slot in range(nbytes // 3):
        ok = lib.pt_tile_slot_pixel(C.byref(p), rank, slot, C.byref(x), C.byref(y))
        assert ok >= 0
        if ok:
            mine[3 * slot:3 * slot + 3] = full[y.value, x.value]
            owned += 1
    t = torch.from_numpy(mine)
    gathered = torch.empty(nbytes * world, dtype=torch.uint8) if rank == 0 else None
    dist.gather(t, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
    counts = torch.tensor([owned], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        img = np.full((h, w, 3), 5, dtype=np.uint8)
        g = gathered.numpy()
        assert lib.pt_untile_host(C.byref(p), g.ctypes.data_as(H._u8p), img.ctypes.data_as(H._u8p)) == 0
        ref = np.full((h, w, 3), 5, dtype=np.uint8)
        x0, y0, x1, y1 = rect
        ref[y0:y1 + 1, x0:x1 + 1] = full[y0:y1 + 1, x0:x1 + 1]
        np.save(out_path, np.array([int(np.array_equal(img, ref)), int(counts.item()), (x1 - x0 + 1) * (y1 - y0 + 1)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h,rect", [(100, 60, (0, 0, 99, 59)), (77, 45, (5, 3, 70, 40))])
def test_tile_partition_gather_untile_world2(tmp_path, w, h, rect):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "r.npy")
    mp.spawn(_worker, args=(2, port, w, h, rect, out), nprocs=2, join=True)
    equal, owned, pixels = np.load(out)
    assert equal == 1, "assembled image differs from the single-process image"
    assert owned == pixels, "every pixel of the slice must belong to exactly one rank"
