"""MI355X-native CNV inference behind infercnvpy's scanpy-style tool API.

    import infercnvpy_amd as cnv
    cnv.tl.infercnv(adata, reference_key="cell_type", reference_cat=[...])
    cnv.tl.cnv_score(adata, "cnv_leiden")
    cnv.pl.chromosome_heatmap(adata, groupby="cell_type")
"""
from . import io, pl, tl  # noqa: F401
from ._compat import SimpleAnnData  # noqa: F401  (duck-typed container; holds device objects, unlike anndata)
from ._engine import DeviceMatrix, PackedCsr  # noqa: F401  (HBM-resident input / output of tl.infercnv)

__version__ = "0.1.0"
