"""`ttts` -- the reference's import names, served by the MI355X-native build.

adelacvg/ttts is imported as `ttts.gpt.train`, `ttts.vqvae.train`, `ttts.gpt.model`, `ttts.vqvae.vq2`, ... (its README runs
`python ttts/gpt/train.py` / `python ttts/vqvae/train.py`).  These modules re-export the same names from `ttts_amd`, so code
written against the reference (`from ttts.gpt.model import UnifiedVoice`, `python -m ttts.gpt.train`,
`from ttts.vqvae.train import run, train_and_evaluate`) runs on the HIP kernels unchanged.  Only the training hot path and
the SURVEY 8f rows exist here; everything else of the reference's tree is out of scope.
"""
