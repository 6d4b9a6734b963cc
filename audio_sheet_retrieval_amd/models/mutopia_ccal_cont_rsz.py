#!/usr/bin/env python
"""Model module `mutopia_ccal_cont_rsz` - same module protocol as the reference's
audio_sheet_retrieval/models/mutopia_ccal_cont_rsz.py (constants :23-51, build_model
:61-149, objectives :152-155, compute_updates :158-162, update_learning_rate
:165-167, prepare :170-190, batch iterators :193-204), with the Lasagne graph
replaced by handles onto the HIP library (audio_sheet_retrieval_amd/network.py)."""
from __future__ import annotations

from ._common import make_build_model, prepare_rsz

SPEC_CONTEXT = 42            # utils/mutopia_data.py (imported at :20)

INI_LEARNING_RATE = 0.002
REFINEMENT_STEPS = 5
LR_MULTIPLIER = 0.5
BATCH_SIZE = 100
MOMENTUM = 0.9               # unused by the reference too (adam)
MAX_EPOCHS = 1000
PATIENCE = 30
INPUT_SHAPE_1 = [1, 160, 200]   # raw snippet (:32); the InputLayer is 80x100 (:68)
INPUT_SHAPE_2 = [1, 92, SPEC_CONTEXT]

DIM_LATENT = 32

L1 = None
L2 = 0.00001
GRAD_NORM = None

r1 = r2 = 1e-3
rT = 1e-3

FIT_CCA = False
ALPHA = 1.0
WEIGHT_TNO = 0.0
USE_CCAL = True
GAMMA = 0.7

EXP_NAME = "mutopia_ccal_cont_rsz"

build_model = make_build_model("mutopia_ccal_cont_rsz", [1, 80, 100], INPUT_SHAPE_2, raw_shape_1=INPUT_SHAPE_1,
                               r1=r1, r2=r2, rT=rT, alpha=ALPHA, gamma=GAMMA, l2=L2)


def objectives():
    """get_contrastive_cos_loss(1 - WEIGHT_TNO, GAMMA) (:152-155): returns the
    (weight, gamma) description the HIP training step implements
    (models/objectives.py:30-69)."""
    from .objectives import get_contrastive_cos_loss
    return get_contrastive_cos_loss(1.0 - WEIGHT_TNO, GAMMA)


def compute_updates(all_grads, all_params, learning_rate):
    """lasagne.updates.adam(all_grads, all_params, learning_rate) (:158-162):
    the update rule is part of the fused training step; this returns its
    description."""
    return dict(rule="adam", beta1=0.9, beta2=0.999, epsilon=1e-8, learning_rate=learning_rate)


def update_learning_rate(lr, epoch=None):
    """(:165-167)"""
    return lr


prepare = prepare_rsz


def valid_batch_iterator():
    """(:193-197)"""
    from ..utils.batch_iterators import MultiviewPoolIteratorUnsupervised
    return MultiviewPoolIteratorUnsupervised(batch_size=BATCH_SIZE, prepare=prepare, shuffle=False)


def train_batch_iterator(batch_size=BATCH_SIZE):
    """(:200-204)"""
    from ..utils.batch_iterators import MultiviewPoolIteratorUnsupervised
    return MultiviewPoolIteratorUnsupervised(batch_size=batch_size, prepare=prepare, k_samples=10000)
