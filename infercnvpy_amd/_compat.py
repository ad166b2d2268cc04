"""Minimal AnnData stand-in.

``anndata`` is not a dependency of this package: the tool functions only touch the attributes
``X, layers, obs, var, var_names, obsm, uns, shape`` and work with a real ``anndata.AnnData`` as well
as with this container (used by ``bench.py``, ``__graft_entry__.smoke()`` and the tests, where anndata
is not installed).
"""
from __future__ import annotations

import pandas as pd


class SimpleAnnData:
    def __init__(self, X, obs=None, var=None, layers=None, obsm=None, uns=None):
        self.X = X
        n_obs, n_var = X.shape
        self.obs = obs if obs is not None else pd.DataFrame(index=[str(i) for i in range(n_obs)])
        self.var = var if var is not None else pd.DataFrame(index=[str(i) for i in range(n_var)])
        self.layers = dict(layers or {})
        self.obsm = dict(obsm or {})
        self.uns = dict(uns or {})

    @property
    def shape(self):
        return self.X.shape

    @property
    def n_obs(self):
        return self.X.shape[0]

    @property
    def n_vars(self):
        return self.X.shape[1]

    @property
    def var_names(self):
        return self.var.index

    @property
    def obs_names(self):
        return self.obs.index

    def obsm_keys(self):
        return list(self.obsm.keys())

    def uns_keys(self):
        return list(self.uns.keys())
