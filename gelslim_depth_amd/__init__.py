"""gelslim_depth_amd -- MI355X-native U-Net train/inference step of MMintLab/gelslim_depth.

Importing the package does not load the HIP library (so CPU-only tooling can import `synth`);
`gelslim_depth_amd.models.unet`, `.engine` and `.train` do, and fail loudly if it is missing.
"""
__all__ = ["synth"]
