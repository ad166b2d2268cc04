"""End-to-end walk through the public API on synthetic data (needs an MI355X):

    python tools/demo.py

GTF annotation -> tl.infercnv -> tl.cnv_score -> tl.ithcna -> tl.cell_linkage -> pl.chromosome_heatmap.
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

import infercnvpy_amd as cnv  # noqa: E402  (deliberately before any `import torch`)
from infercnvpy_amd._compat import SimpleAnnData  # noqa: E402


def main():
    import matplotlib

    matplotlib.use("Agg")
    gtf = os.path.join(ROOT, "tests", "data", "chr21_gencode_genes.gtf")
    genes = cnv.io.read_gtf_genes(gtf)
    names = list(genes["gene_name"].drop_duplicates(keep=False))
    rng = np.random.RandomState(0)
    n_cells = 600
    X = rng.gamma(0.4, 1.0, size=(n_cells, len(names))).astype(np.float32)
    X[X < 0.4] = 0
    groups = np.repeat(["normal", "tumorA", "tumorB"], n_cells // 3)
    X[groups == "tumorA", 40:110] *= 1.8   # a gained segment
    X[groups == "tumorB", 150:210] *= 0.3  # a lost segment
    adata = SimpleAnnData(X, obs=pd.DataFrame({"cell_type": groups}, index=[f"cell{i}" for i in range(n_cells)]),
                          var=pd.DataFrame(index=names))
    cnv.io.genomic_position_from_gtf(gtf, adata)
    cnv.tl.infercnv(adata, reference_key="cell_type", reference_cat="normal", window_size=30, step=5)
    print("X_cnv", adata.obsm["X_cnv"].shape, "nnz fraction %.3f" % (adata.obsm["X_cnv"].nnz / np.prod(adata.obsm["X_cnv"].shape)),
          "chr_pos", adata.uns["cnv"]["chr_pos"])
    print("cnv_score", cnv.tl.cnv_score(adata, "cell_type", inplace=False))
    print("ithcna", cnv.tl.ithcna(adata, "cell_type", inplace=False))
    cnv.tl.cell_linkage(adata)
    Z = adata.uns["cnv_linkage"]["linkage"]
    print("linkage", Z.shape, "top merge height %.3f" % Z[-1, 2])
    out = os.path.join(ROOT, "gpurun_out", "demo_heatmap.png")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    axes = cnv.pl.chromosome_heatmap(adata, groupby="cell_type", cell_order="ward", dendrogram=True, show=False, save=out)
    print("heatmap axes", sorted(axes), "->", out)
    score = cnv.tl.cnv_score(adata, "cell_type", inplace=False)
    assert score["tumorA"] > 1.5 * score["normal"] and score["tumorB"] > 1.3 * score["normal"]
    print("demo OK")


if __name__ == "__main__":
    main()
