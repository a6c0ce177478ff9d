"""MI355X-native (gfx950) beta-VAE-GAN training engine: the CelebA 64x64 conv
encoder / decoder / discriminator hot path of RicoFio/disentangle_mlp behind the
reference's own module API, computed by hand-written HIP kernels
(libvaegan_hip.so, C ABI in include/vaegan_hip.h)."""
__version__ = "0.1.0"
