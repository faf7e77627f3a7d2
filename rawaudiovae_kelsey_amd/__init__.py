"""MI355X-native (gfx950) training path for the raw-audio VAE of
kelseyicotton/rawaudiovae_kelsey: hand-written HIP kernels behind a C ABI
(`librawvae_hip.so`, `include/rawvae_hip.h`) and the Python host mirror of the
reference surface (`rawvae.model.VAE`, `loss_function`)."""
__version__ = "0.1.0"
