"""Host-side geometry predicates of the kernel library (pure host code: they run without a GPU)."""
from __future__ import annotations

import pytest


def _lib():
    try:
        from alphatriangle_amd import native

        return native.hip()
    except Exception as exc:  # the library is not built / its HIP runtime is not loadable on this machine
        pytest.skip(f"kernel library not loadable here: {exc}")


@pytest.mark.parametrize("h,w,cin,cout,halo,flat", [
    (8, 15, 128, 128, 1, 1),    # default board: the halo-tile kernel serves it (the engine prefers it), the flat one could
    (12, 21, 128, 128, 0, 1),   # large preset: 12 x 22 = 264 flat positions = 17 tiles
    (12, 21, 32, 64, 0, 1),
    (16, 16, 128, 128, 0, 1),   # 272 positions, 256 cells
    (17, 16, 128, 128, 0, 0),   # 272 cells: more than a thread's eight staging chunks cover
    (64, 4, 128, 128, 0, 1),    # 64 x 5 = 320 positions: the largest tile count (20)
    (65, 4, 128, 128, 0, 0),    # 325 positions
    (3, 40, 64, 64, 0, 1),
    (12, 21, 16, 128, 0, 0),    # unsupported widths
    (12, 21, 128, 256, 0, 0),
])
def test_conv_kernel_geometry_predicates(h, w, cin, cout, halo, flat):
    lib = _lib()
    assert lib.atri_rowconv_supported(h, w, cin, cout) == halo
    assert lib.atri_flatconv_supported(h, w, cin, cout) == flat


@pytest.mark.parametrize("batch,live,expect", [(1024, 1024, (16, 4, 124)), (1024, 900, (15, 4, 124)), (1024, 512, (8, 8, 64)),
                                               (1024, 100, (2, 16, 32)), (1024, 1, (1, 16, 32)), (1024, 0, (16, 4, 124)),
                                               (200, 64, (1, 7, 4))])
def test_fp32_shared_fc_plan_follows_the_live_count(batch, live, expect):
    """The (row blocks, K splits, steps per split) the FC's workgroups deal themselves for a live count: never more
    workgroups than the launch holds, whole pixel rows per split, no empty split, the static plan for a full batch."""
    import ctypes

    lib = _lib()
    total = 482 if batch == 1024 else 25
    sk, steps = (4, 124) if batch == 1024 else (2, 16)
    out = (ctypes.c_int * 3)()
    lib.atri_split_fc_plan(live, batch, sk, steps, total, 16, out)
    blocks, splits, per = out[0], out[1], out[2]
    assert (blocks, splits, per) == expect
    assert blocks * splits <= (batch + 63) // 64 * sk and per % 4 == 0
    assert (splits - 1) * per < total <= splits * per
