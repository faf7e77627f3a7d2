"""`rawvae.model` -- the reference's import path (train.py:11, tutorial.ipynb) bound
to the MI355X implementation in rawaudiovae_kelsey_amd.model."""
from rawaudiovae_kelsey_amd.model import VAE, Decoder, Encoder, loss_function

# pickles written by torch.save(model) name the class by this module path, as the
# reference's do (train.py:244)
VAE.__module__ = __name__

__all__ = ["VAE", "Encoder", "Decoder", "loss_function"]
