class LatentCodec:                  # imported but never used by the reference (scene/gaussian_model.py:27)
    pass


class HyperLatentCodec(LatentCodec):
    pass
