"""CPU oracle for the audio<->sheet retrieval hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy (float32/float64, CPU) restatement of the arithmetic
the reference (CPJKU/audio_sheet_retrieval) runs on its hot path:

    twin-CNN forward -> CCA projection -> L2-norm ->
        eval : all-pairs cosine distance -> rank / Recall@k
        train: pairwise ranking loss + backward + Adam
    plus the 25 000-sample CCA re-estimation of refine_cca.py.

Every function cites the reference file:line it follows.  Semantics that live
in un-vendored third-party code (Lasagne 0.2.dev1, Theano 1.0.1, SciPy cdist)
are restated from their published behaviour and marked
"third-party semantic, unverified offline".

PARITY PARTLY PINNED.  The reference ships no tests, golden vectors or numeric
notebook outputs, and most of it cannot be executed in the build container
(Python-2 source on theano / lasagne / cv2 / madmom / msmd, none installed).
  PINNED by outputs of the reference's own code, run in the build container by
  tests/golden/make_reference_golden.py (-> tests/golden/reference_golden.npz,
  checked in tests/test_reference_golden.py against this oracle AND the HIP
  library): CCA.fit('svd') (utils/cca.py), eval_retrieval
  (utils/train_dcca_pool.py:28-82), dtw_by_dist (utils/dtw_by_dist.py),
  align_baseline / align_pydtw / compute_alignment / estimate_alignment_error
  (utils/alignment.py:112-190), the server's detect_score / detect_performance
  / _retrieve_* methods (audio_sheet_server.py:213-300, :530-563; slicing,
  top-n retrieval and vote - the network between them replaced by a fixed
  projection) and the AudioScoreRetrievalPool class (utils/data_pools.py:36-228
  without its cv2 rescaling branch, including the order of the random draws).
  UNPINNED ("parity unpinned"): everything that lives in Theano graphs - the CNN
  forward, BatchNorm, CCALayer, the ranking loss, gradients and Adam - plus
  the pool's cv2.resize(INTER_NEAREST) branch and the spectrogram front-end
  (madmom).  What pins those instead:
  * independent re-derivations run in tests/ (torch-CPU conv2d / batch_norm /
    elu / max_pool2d / linalg.eigh autograd, scipy.spatial.distance.cdist,
    numpy.linalg float64 CCA);
  * algebraic invariants (rot-180 equivariance of conv-vs-correlation, row
    independence in deterministic mode, unit-norm outputs, rank-by-counting ==
    argsort position on tie-free inputs ...);
  * consistency checks of the reference's shipped parameter pickle (build
    container only; nothing from /root/reference travels to the GPU box).
The other golden vectors under tests/golden/ (hotpath_golden.npz) are produced
BY this oracle (script committed next to them): they pin HIP <-> oracle, not
HIP <-> Theano.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package, and only as the checker / reported CPU baseline.  The
product path (audio_sheet_retrieval_amd) never imports it and fails loudly
when the HIP library is missing.
"""
