"""The ctypes stub printed in INTEGRATION.md section 2 -- what a maintainer of infercnvpy would paste into
``infercnvpy/tl/_hip.py`` -- executed VERBATIM (only the library path is substituted) and compared with the oracle
(reference ``_infercnv_chunk``, tl/_infercnv.py:411-457) and with the package's own binding."""
import os
import re

import numpy as np
import pandas as pd
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "import ctypes as C" in b]
    assert len(stub) == 1
    return stub[0]


def test_stub_declares_every_function_it_calls():
    src = _stub_source()
    called = set(re.findall(r"_lib\.(icv_[a-z_0-9]+)\(", src))
    declared = set(re.findall(r"_lib\.(icv_[a-z_0-9]+)\.(?:argtypes|restype)", src))
    assert called and called <= declared, called - declared
    header = open(os.path.join(ROOT, "include", "infercnv_hip.h")).read()
    for name in called:
        assert re.search(r"\b%s\(" % name, header), name


@pytest.mark.gpu
def test_stub_runs_and_matches_the_oracle():
    import torch

    import infercnvpy_amd as cnv
    from infercnvpy_amd import _lib
    from infercnvpy_amd._compat import SimpleAnnData
    from infercnvpy_amd._plan import GenePlan
    from oracle import infercnv_oracle as O

    _lib.load()
    src = _stub_source().replace('"libinfercnv_hip.so"', repr(_lib.LIB_PATH))
    ns = {}
    exec(compile(src, "INTEGRATION.md:stub", "exec"), ns)

    v = cases.synthetic_var([300, 120, 101, 60], extra=(("chrX", 9), (None, 2)))
    X = cases.synthetic_expr(150, len(v["names"]), seed=11)
    host = GenePlan(v["chromosome"], v["start"], window_size=20, step=5)  # host side only: col_pos, offsets, names
    plan, n_windows, first = ns["make_plan"](host.col_pos, host.chrom_offsets, 20, 5)
    try:
        assert n_windows == host.n_windows
        ref = X.mean(axis=0).astype(np.float32)
        out = ns["infercnv_rows"](plan, n_windows, torch.as_tensor(X, device="cuda"), torch.as_tensor(ref, device="cuda"),
                                  3.0, 1.5, 64)
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
    finally:
        ns["_lib"].icv_plan_destroy(plan)
    e_pos, e_res, _, _ = O.infercnv(X, v["chromosome"], v["start"], reference=ref, window_size=20, step=5, chunksize=64)
    exp = e_res.toarray()
    assert dict(zip(host.chromosomes, (int(p) for p in first))) == {k: int(p) for k, p in e_pos.items()}
    np.testing.assert_array_equal(got == 0, exp == 0)
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-6)
    # and the package's binding gives the same bits
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    _, res, _ = cnv.tl.infercnv(SimpleAnnData(X, var=var), reference=ref, window_size=20, step=5, chunksize=64,
                                inplace=False)
    np.testing.assert_array_equal(res.toarray(), got)
    # an error comes back as the exception the reference raises
    with pytest.raises(ValueError):
        ns["make_plan"](host.col_pos, host.chrom_offsets, 0, 5)
