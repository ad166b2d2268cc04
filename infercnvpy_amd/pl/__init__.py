from ._chromosome_heatmap import chromosome_heatmap, chromosome_heatmap_summary

__all__ = ["chromosome_heatmap", "chromosome_heatmap_summary"]
