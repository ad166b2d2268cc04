from ._infercnv import infercnv, infercnv_device
from ._linkage import cell_linkage, leaves_list, ward_linkage
from ._scores import cnv_score, ithcna, ithgex

__all__ = ["infercnv", "infercnv_device", "cnv_score", "ithcna", "ithgex", "cell_linkage", "ward_linkage", "leaves_list"]
