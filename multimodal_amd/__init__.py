"""multimodal_amd -- MI355X-native KL-divergence NMF behind the API of
omangin/multimodal's `KLdivNMF` / `MultimodalLearner`.

Layout mirrors the reference package for the hot path only:

    multimodal_amd.lib.nmf          <-> multimodal/lib/nmf.py
    multimodal_amd.lib.metrics      <-> multimodal/lib/metrics.py (generalized_KL)
    multimodal_amd.lib.array_utils  <-> multimodal/lib/array_utils.py (normalize_sum, safe_hstack)
    multimodal_amd.lib.sklearn_utils<-> multimodal/lib/sklearn_utils.py (input contract)
    multimodal_amd.learner          <-> multimodal/learner.py
    multimodal_amd.distributed      row-sharded multi-GPU driver (new; RCCL all-reduce)

All arithmetic of the path runs in hand-written HIP kernels (csrc/) reached
through the C-ABI of include/klnmf.h; there is no CPU fallback.
"""

__version__ = '0.1.0'
