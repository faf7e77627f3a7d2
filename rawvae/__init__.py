"""Drop-in module path of the reference package (`from rawvae.model import VAE,
loss_function`, train.py:11).  The reference ships an empty `rawvae/init.py` and
works as a namespace package; this is a regular package so `torch.save(model)`
pickles resolve `rawvae.model.VAE` (train.py:244,298)."""
