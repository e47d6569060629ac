"""ttts/vqvae/dataset.py -> ttts_amd.vqvae.dataset (sampler + collater semantics; audio decoding is injected)."""
from ttts_amd.vqvae.dataset import DistributedBucketSampler, VQVAECollater  # noqa: F401
