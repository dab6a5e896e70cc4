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
