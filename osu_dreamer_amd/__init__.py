"""osu_dreamer_amd — MI355X-native denoiser train + sample hot path of jaswon/osu-dreamer.

Python host code over hand-written HIP kernels (csrc/, C ABI in include/osu_dreamer_hip.h).
Mirrors the reference's `DiffusionModel` / `DiffusionTrainer` surface
(osu_dreamer/models/diffusion/{model,train}.py) so it drops in for that path.
"""
__all__ = ["__version__"]
__version__ = "0.1.0"
