"""``load_GAN_model`` / ``load_CNN_model`` — the reference's model loaders (process_full_tiles.py:13-51) on the HIP path.

Same names, arguments and error behaviour: ``path`` is the directory holding the ``generator/`` and ``encoder/``
Keras SavedModel folders written by ``gaugan.save`` (spade/models/model.py:569-605); like the reference the
sub-folders are found by string concatenation ``path+'generator'`` (a trailing slash is required, run_GAN.sh:23).
No TensorFlow is needed: the variables are read by ``tf_checkpoint`` (see its header for what that reader is and is
not validated against).  The discriminator (and the VGG19 download ``gaugan.compile()`` triggers, losses.py:67) is
not needed for inference and is skipped.
"""
from __future__ import annotations

import os

from .generator import Generator
from .tf_checkpoint import keras_to_weights


def load_GAN_model(path: str, image_size: int, batch_size: int, **kwargs) -> Generator:
    """GauGAN(image_size, batch_size, latent_dim=256) with the weights under ``path`` (process_full_tiles.py:13-31)."""
    assert os.path.exists(path), "The path to the neural-network weight is invalid. Please ensure you gave a valid path."
    weights = keras_to_weights(path + "generator", path + "encoder", image_size, latent_dim=256)
    return Generator(image_size, batch_size, latent_dim=256, variant="gaugan", weights=weights, **kwargs)


def load_CNN_model(path: str, image_size: int, batch_size: int, **kwargs) -> Generator:
    """CNNSpade(image_size, batch_size, latent_dim=256) with the weights under ``path`` (process_full_tiles.py:33-51)."""
    assert os.path.exists(path), "The path to the neural-network weight is invalid. Please ensure you gave a valid path."
    weights = keras_to_weights(path + "generator", path + "encoder", image_size, latent_dim=256)
    return Generator(image_size, batch_size, latent_dim=256, variant="cnn", weights=weights, **kwargs)
