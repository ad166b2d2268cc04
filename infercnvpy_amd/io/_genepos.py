"""Genomic positions for ``adata.var`` from a GTF file -- the step before ``tl.infercnv``.

Host-side counterpart of ``infercnvpy.io.genomic_position_from_gtf`` (reference
``src/infercnvpy/io/_genepos.py:94-179``).  The reference reads the file through
the optional ``gtfparse`` package; here the ``gene`` records are parsed directly (plain or gzip text), so
there is no extra dependency.  Semantics kept from the reference:

* only ``feature == "gene"`` records, identical records dropped (:134-141);
* Ensembl version suffixes are stripped from ``gene_id`` (:143);
* records are matched on ``gtf_gene_id`` (``gene_name`` by default) against ``adata.var_names`` or the
  ``adata_gene_id`` column; a warning counts the genes of ``adata`` the file does not annotate (:145-150);
* identifiers that occur more than once among the matched records are skipped altogether, with a warning
  (:152-155);
* ``adata.var`` gains ``chromosome, start, end, gene_id, gene_name`` in its original row order; unmatched
  genes get NaN (:157-170); the result is one row per gene of ``adata`` (a duplicated key on either side is
  an error, ``validate="one_to_one"``);
* if no annotated chromosome starts with ``"chr"`` (Ensembl style), the prefix is added (:172-174).

``genomic_position_from_biomart`` needs network access (ENSEMBL Biomart through scanpy) and is not provided.
"""
from __future__ import annotations

import gzip
import logging
import re
from pathlib import Path

import numpy as np
import pandas as pd

log = logging.getLogger("infercnvpy_amd")

_ATTR = {key: re.compile(r'(?:^|;)\s*' + key + r'\s+"?([^";]*)"?') for key in ("gene_id", "gene_name")}
_VERSION = re.compile(r"\.\d+$")


def read_gtf_genes(gtf_file) -> pd.DataFrame:
    """The ``gene`` records of a GTF file as a frame ``chromosome, start, end, gene_id, gene_name``.

    ``gene_name`` falls back to NaN when the attribute is absent; version suffixes are removed from
    ``gene_id``; exact duplicate records are dropped.
    """
    path = Path(gtf_file)
    opener = gzip.open if path.suffix == ".gz" else open
    rows = []
    with opener(path, "rt") as fh:
        for line in fh:
            if not line or line[0] == "#":
                continue
            fields = line.rstrip("\n").split("\t")
            if len(fields) < 9 or fields[2] != "gene":
                continue
            attrs = fields[8]
            gid = _ATTR["gene_id"].search(attrs)
            gname = _ATTR["gene_name"].search(attrs)
            rows.append((fields[0], int(fields[3]), int(fields[4]),
                         _VERSION.sub("", gid.group(1)) if gid else np.nan,
                         gname.group(1) if gname else np.nan))
    genes = pd.DataFrame(rows, columns=["chromosome", "start", "end", "gene_id", "gene_name"])
    return genes.drop_duplicates().reset_index(drop=True)


def genomic_position_from_gtf(gtf_file, adata=None, *, gtf_gene_id: str = "gene_name", adata_gene_id: str | None = None,
                              inplace: bool = True):
    """Add ``chromosome``, ``start``, ``end`` (and ``gene_id``, ``gene_name``) from a GTF file to ``adata.var``.

    Same parameters and return value as the reference function: with ``inplace=False`` the annotated copy of
    ``adata.var`` is returned instead of being assigned.
    """
    if gtf_gene_id not in ("gene_id", "gene_name"):
        raise ValueError("gtf_gene_id must be 'gene_id' or 'gene_name'")
    genes = read_gtf_genes(gtf_file)

    var = adata.var
    keys = pd.Index(adata.var_names if adata_gene_id is None else var[adata_gene_id].values)
    genes = genes.loc[genes[gtf_gene_id].isin(keys)]

    n_missing = len(set(keys) - set(genes[gtf_gene_id].values))
    if n_missing:
        log.warning(f"GTF file misses annotation for {n_missing} genes in adata.")
    n_dup = int(genes["gene_name"].duplicated().sum())
    if n_dup:
        log.warning(f"Skipped {n_dup} genes because of duplicate identifiers in GTF file.")
        genes = genes.loc[~genes[gtf_gene_id].duplicated(keep=False)]
    if genes[gtf_gene_id].duplicated().any() or keys.duplicated().any():
        # the reference's merge(validate="one_to_one") raises pandas' MergeError, a ValueError
        raise ValueError("Merge keys are not unique: gene identifiers must be unique in adata and in the GTF file")

    # left join in the row order of adata.var
    lookup = genes.set_index(gtf_gene_id, drop=False)
    annotated = var.copy()
    pos = lookup.index.get_indexer(keys)
    hit = pos >= 0
    for col in ("chromosome", "start", "end", "gene_id", "gene_name"):
        values = lookup[col].to_numpy()
        if col in ("start", "end"):
            out = np.full(len(keys), np.nan)
        else:
            out = np.full(len(keys), np.nan, dtype=object)
        out[hit] = values[pos[hit]]
        annotated[col] = out

    chrom = annotated["chromosome"].dropna()
    if np.all(~chrom.str.startswith("chr")):  # not a GENCODE-style file: add the prefix tl.infercnv filters on
        annotated["chromosome"] = "chr" + annotated["chromosome"]

    if inplace:
        adata.var = annotated
    else:
        return annotated
