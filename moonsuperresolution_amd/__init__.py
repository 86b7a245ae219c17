"""moonsuperresolution_amd — MI355X (gfx950) implementation of the tiled DEM super-resolution inference path of
AntoineRichard/MoonSuperResolution (process_full_tiles.py over a SPADE / pix2pix generator).

Public surface (mirrors the reference's names where it has them):
    Generator            the ``model`` callable of process_full_tiles.py:338 on hand-written HIP kernels
    DSRConfig            process_full_tiles.py:53-66
    DEMSuperResolution   process_full_tiles.py:129-587 (tiling + stitching on the GPU)
    HaloShardedSuperResolution   the patch-row-sharded multi-GPU mode with a neighbour exchange of overlap halos (halo.py)
    load_GAN_model / load_CNN_model   process_full_tiles.py:13-51 (Keras SavedModel variables, read without TensorFlow)
"""
from .weights import make_weights, make_latent_noise, synthetic_patches, weight_shapes  # noqa: F401


def __getattr__(name):
    # torch / the HIP library are only needed by the device-facing classes
    if name == "Generator":
        from .generator import Generator
        return Generator
    if name in ("DSRConfig", "DEMSuperResolution"):
        from . import tiler
        return getattr(tiler, name)
    if name == "HaloShardedSuperResolution":
        from .halo import HaloShardedSuperResolution
        return HaloShardedSuperResolution
    if name in ("load_GAN_model", "load_CNN_model"):
        from . import loaders
        return getattr(loaders, name)
    raise AttributeError(name)
