"""Host logic of the streamed pack kernels (``csrc/icv_kernel_pack.hpp``: PackRing / FillRing through
``icv_pack_geometry``; no GPU): how the 160 KB of LDS of a CU are split for rows of W windows, and where the streamed
kernels hand over to the per-row kernels.  The constraints checked here are the ones the kernels rely on: the whole
layout fits the LDS, a loader wavefront never has more LDS-DMA loads outstanding than the hardware counter (vmcnt, 6
bits) can hold, every ring has a slot to read while the others are in flight."""
import ctypes as C

import pytest

LDS = 160 * 1024


def _info(W):
    from infercnvpy_amd import _lib

    info = _lib.PackInfo()
    _lib.check(_lib.load().icv_pack_geometry(W, C.byref(info)))
    return {n: getattr(info, n) for n, _ in info._fields_}


def test_every_width_is_either_streamed_within_the_limits_or_left_to_the_per_row_kernels():
    streamed = []
    for W in list(range(1, 4200)) + [4801, 8191, 12288, 17822, 20480, 100_000]:
        g = _info(W)
        assert g["rows_per_round"] == 4
        for kind in ("mask", "fill"):
            if g[f"{kind}_streamed"]:
                assert 2 <= g[f"{kind}_ring_slots"] <= 16, (W, g)
                assert 0 < g[f"{kind}_lds_bytes"] <= LDS, (W, g)
                assert 0 < g[f"{kind}_loads_in_flight"] <= 60, (W, g)
                # the ring alone: slots x rows x the row rounded to 16 bytes
                assert g[f"{kind}_ring_slots"] * 4 * ((4 * W + 15) // 16 * 16) < g[f"{kind}_lds_bytes"], (W, g)
            else:
                assert all(g[k] == 0 for k in g if k.startswith(kind + "_")), (W, g)
        if g["fill_streamed"]:
            assert g["fill_stage_entries"] >= 1024 and g["fill_stage_entries"] % 64 == 0, (W, g)
            assert g["mask_streamed"], W  # (the fill never streams where the mask does not)
        if g["mask_streamed"]:
            streamed.append(W)
    # rows of 1 KB .. 8 KB: 256 .. 2 048 windows, without holes
    assert streamed == list(range(256, 2049))


def test_geometry_of_the_benchmark_configurations():
    g = _info(1802)  # config 2 / 3: 20 000 genes, window 100, step 10
    assert (g["mask_ring_slots"], g["mask_loads_in_flight"]) == (5, 32)  # four rounds = 115 KB of x_res in flight
    assert (g["fill_ring_slots"], g["fill_loads_in_flight"]) == (4, 39)
    assert g["fill_stage_entries"] >= 4 * 1802 * 0.35  # a usual round (~20 % kept) fits one staging block
    g4 = _info(1790)  # config 4: window 250
    assert g4["mask_streamed"] and g4["fill_streamed"] and g4["fill_ring_slots"] == 4


def test_bad_arguments():
    from infercnvpy_amd import _lib

    with pytest.raises(ValueError):
        _lib.check(_lib.load().icv_pack_geometry(0, C.byref(_lib.PackInfo())))
    with pytest.raises(ValueError):
        _lib.check(_lib.load().icv_pack_geometry(100, None))
