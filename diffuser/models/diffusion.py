from madiff_amd.diffusion import GaussianDiffusion  # noqa: F401
