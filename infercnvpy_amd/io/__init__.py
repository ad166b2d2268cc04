from ._genepos import genomic_position_from_gtf, read_gtf_genes

__all__ = ["genomic_position_from_gtf", "read_gtf_genes"]
