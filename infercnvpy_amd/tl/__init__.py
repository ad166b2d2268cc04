from ._infercnv import infercnv
from ._scores import cnv_score, ithcna, ithgex

__all__ = ["infercnv", "cnv_score", "ithcna", "ithgex"]
