"""Batched tensor AUC / AP vs scikit-learn (the reference's metric code, base.py:248-249,274-277),
with heavy ties, constant scores and tiny inputs.  CPU."""
import numpy as np
import pytest
import torch
from sklearn.metrics import average_precision_score, roc_auc_score

from gnndelete_amd.framework.metrics import batched_average_precision, batched_roc_auc


@pytest.mark.parametrize('seed,levels', [(0, None), (1, 7), (2, 2), (3, 50), (4, 1)])
def test_matches_sklearn_including_ties(seed, levels):
    g = torch.Generator().manual_seed(seed)
    b, m = 6, 257
    s = torch.rand(b, m, generator=g, dtype=torch.float64)
    if levels:
        s = torch.floor(s * levels) / levels             # many / all tied scores
    y = (torch.rand(b, m, generator=g) < 0.4).double()
    y[:, 0], y[:, 1] = 1, 0                               # both classes present
    auc = batched_roc_auc(s, y)
    ap = batched_average_precision(s, y)
    for i in range(b):
        assert abs(float(auc[i]) - roc_auc_score(y[i].numpy(), s[i].numpy())) < 1e-12
        assert abs(float(ap[i]) - average_precision_score(y[i].numpy(), s[i].numpy())) < 1e-12


def test_shared_labels_and_1d_inputs():
    s = torch.tensor([0.1, 0.9, 0.4, 0.4, 0.7])
    y = torch.tensor([0, 1, 0, 1, 1])
    assert abs(float(batched_roc_auc(s, y)[0]) - roc_auc_score(y.numpy(), s.numpy())) < 1e-12
    assert abs(float(batched_average_precision(s, y)[0]) - average_precision_score(y.numpy(), s.numpy())) < 1e-12
    sb = torch.stack([s, s.flip(0)])
    got = batched_roc_auc(sb, y)
    assert abs(float(got[1]) - roc_auc_score(y.numpy(), s.flip(0).numpy())) < 1e-12
