"""A bounded slice of the fuzz harness (tests/fuzz_gpu_parity.py) inside the `-m gpu` suite, so that the driver's run sees it too: fixed seeds, the five
scene families (random with meshes / KDMesh / mirrors / glass, extreme scales and coincident faces, textured, mesh-free analytic, mirrors without glass) in all
three traversal semantics, every scene rendered by the counting AND the plain instantiation of its kernel, image, f64 means and ray counts against the
oracle. The long runs (tens of thousands of renders per round) stay a script."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first,count,w,h,samples", [(9000, 30, 128, 96, 2), (9100, 8, 64, 48, 64), (9200, 6, 96, 64, 13)])
def test_fuzz_slice(oracle, first, count, w, h, samples):
    """2 samples: 32 pixels of a tile per wavefront; 64: one pixel per wavefront (the timed layout); 13: two ragged chunks."""
    import fuzz_gpu_parity as F
    msgs = []
    bad, tex_edge, n = F.run(first, count, w, h, samples, modes=["flat", "kd", "hier"], out=msgs.append)
    assert n == count * 15
    assert bad == 0, msgs[:5]
