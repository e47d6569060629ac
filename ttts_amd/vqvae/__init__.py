from .quantize import EuclideanCodebook, ResidualVectorQuantization, ResidualVectorQuantizer, VectorQuantization  # noqa: F401
