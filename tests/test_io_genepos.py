"""genomic_position_from_gtf (host code; reference io/_genepos.py:94-179, tests/test_io.py:10-28).

The reference's own test needs scanpy's pbmc3k download; the properties it asserts are checked here on
the reference's GTF data (reduced to the gene records, tests/golden/make_gtf_fixtures.py) with synthetic
``var`` tables.
"""
import gzip
import os
import shutil

import numpy as np
import pandas as pd
import pytest

import infercnvpy_amd as cnv
from infercnvpy_amd._compat import SimpleAnnData

DATA = os.path.join(os.path.dirname(__file__), "data")
GENCODE = os.path.join(DATA, "chr21_gencode_genes.gtf")
ENSEMBL = os.path.join(DATA, "chr1_ensembl_genes.gtf")


def _adata(names, **cols):
    var = pd.DataFrame(cols, index=pd.Index(names, name="symbol"))
    return SimpleAnnData(np.zeros((3, len(names)), dtype=np.float32), var=var)


def test_read_gtf_genes_gencode():
    g = cnv.io.read_gtf_genes(GENCODE)
    assert list(g.columns) == ["chromosome", "start", "end", "gene_id", "gene_name"]
    assert len(g) == 272 and (g["chromosome"] == "chr21").all()  # 272 genes
    assert not g["gene_id"].str.contains(r"\.\d+$").any()         # version suffix removed
    first = g.iloc[0]
    assert (first["gene_id"], first["gene_name"], first["start"], first["end"]) == (
        "ENSG00000270533", "CR382285.1", 9975017, 10119309)
    assert not g["start"].duplicated().any()


def test_annotate_by_gene_name_keeps_order_and_marks_missing():
    g = cnv.io.read_gtf_genes(GENCODE)
    names = list(g["gene_name"].iloc[[5, 100, 17, 250]]) + ["NOT_A_GENE", "ALSO_MISSING"]
    ad = _adata(names, keep=np.arange(6))
    cnv.io.genomic_position_from_gtf(GENCODE, ad)
    v = ad.var
    assert list(v.index) == names and v.index.name == "symbol" and list(v["keep"]) == list(range(6))
    assert {"chromosome", "start", "end", "gene_id", "gene_name"} <= set(v.columns)
    assert all(v["chromosome"].dropna().str.startswith("chr"))
    np.testing.assert_array_equal(v["start"].isnull().values, v["end"].isnull().values)
    assert int((~v["start"].isnull()).sum()) == 4
    for i, row in zip([5, 100, 17, 250], range(4)):
        assert v["start"].iloc[row] == g["start"].iloc[i] and v["end"].iloc[row] == g["end"].iloc[i]
        assert v["gene_id"].iloc[row] == g["gene_id"].iloc[i]


def test_annotate_by_ensembl_id_column_and_not_inplace():
    g = cnv.io.read_gtf_genes(GENCODE)
    ids = list(g["gene_id"].iloc[:50]) + ["ENSG00000000000"]
    ad = _adata([f"g{i}" for i in range(51)], gene_ids=ids)
    out = cnv.io.genomic_position_from_gtf(GENCODE, ad, adata_gene_id="gene_ids", gtf_gene_id="gene_id",
                                           inplace=False)
    assert "chromosome" not in ad.var.columns  # untouched
    assert int((~out["start"].isnull()).sum()) == 50
    np.testing.assert_array_equal(out["start"].values[:50], g["start"].values[:50].astype(float))


def test_ensembl_style_file_gets_chr_prefix_and_gzip(tmp_path):
    g = cnv.io.read_gtf_genes(ENSEMBL)
    assert len(g) > 0 and not g["chromosome"].str.startswith("chr").any()
    gz = tmp_path / "genes.gtf.gz"
    with open(ENSEMBL, "rb") as src, gzip.open(gz, "wb") as dst:
        shutil.copyfileobj(src, dst)
    names = [n for n in g["gene_name"].dropna().unique()[:10]]
    ad = _adata(names)
    cnv.io.genomic_position_from_gtf(gz, ad)
    assert set(ad.var["chromosome"]) == {"chr1", "chrMT"}


def test_duplicate_identifiers_are_skipped_and_duplicate_keys_fail(tmp_path):
    lines = [l for l in open(GENCODE) if l.split("\t")[2:3] == ["gene"]][:6]
    dup = lines[0].replace("9975017", "1234").replace("ENSG00000270533", "ENSG00000999999")  # same gene_name again
    f = tmp_path / "dup.gtf"
    f.write_text("".join(lines) + dup)
    g = cnv.io.read_gtf_genes(f)
    names = list(g["gene_name"].unique())
    ad = _adata(names)
    cnv.io.genomic_position_from_gtf(f, ad)
    assert np.isnan(ad.var.loc["CR382285.1", "start"])        # ambiguous symbol: skipped
    assert int((~ad.var["start"].isnull()).sum()) == len(names) - 1
    bad = _adata(["a", "b"], gene_ids=[g["gene_id"].iloc[1]] * 2)
    with pytest.raises(ValueError):
        cnv.io.genomic_position_from_gtf(f, bad, adata_gene_id="gene_ids", gtf_gene_id="gene_id")


@pytest.mark.gpu
def test_infercnv_runs_on_gtf_annotated_var():
    """The reference's test ends with `infercnv(adata)` on the annotated object (tests/test_io.py:28)."""
    g = cnv.io.read_gtf_genes(GENCODE)
    names = list(g["gene_name"].drop_duplicates(keep=False))[:200] + [f"unk{i}" for i in range(20)]
    rng = np.random.RandomState(0)
    ad = SimpleAnnData(rng.gamma(0.5, 1.0, (40, len(names))).astype(np.float32),
                       var=pd.DataFrame(index=names), obs=pd.DataFrame(index=[f"c{i}" for i in range(40)]))
    cnv.io.genomic_position_from_gtf(GENCODE, ad)
    cnv.tl.infercnv(ad, window_size=20, step=5)
    assert ad.obsm["X_cnv"].shape[0] == 40 and list(ad.uns["cnv"]["chr_pos"]) == ["chr21"]
