"""Ranking objective description (reference: models/objectives.py:30-69).

The reference builds a Theano expression; here the loss is evaluated inside the
fused HIP training step, so the "compiled objective" is the parameter record
that step consumes.  Only the contrastive cosine loss is on the hot path
(SURVEY.md section 2, row 4)."""
from __future__ import annotations


class ContrastiveCosLoss(object):
    """L = weight * mean_{i != j} clip(gamma - lv1_i.lv2_i + lv1_i.lv2_j, 0, 1000)
    (models/objectives.py:36-50); symmetric adds the transposed direction (:53-65)."""

    def __init__(self, weight, gamma, symmetric=False):
        self.weight, self.gamma, self.symmetric = float(weight), float(gamma), bool(symmetric)

    def __repr__(self):
        return "ContrastiveCosLoss(weight=%g, gamma=%g, symmetric=%s)" % (self.weight, self.gamma, self.symmetric)


def get_contrastive_cos_loss(weight, gamma, symmetric=False):
    return ContrastiveCosLoss(weight, gamma, symmetric)
