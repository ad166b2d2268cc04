from ._infercnv import infercnv
from ._scores import cnv_score

__all__ = ["infercnv", "cnv_score"]
