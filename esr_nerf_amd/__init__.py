"""esr-hip: MI355X-native (gfx950) implementation of ESR-NeRF's volumetric-rendering hot path.

Drop-in renderers with the reference's API: ``voxurfc.VoxurfC`` (coarse stage), ``voxurff.VoxurfF`` (fine stage),
``esrnerf.ESRNeRF`` (lts / pdra stages); ``render_utils`` mirrors the reference's pybind extension modules;
``optimizer`` its Adam; ``trainer`` holds the autograd-free training steps with data parallelism.  All arithmetic on
the path runs in ``libesr_hip.so`` (``include/esr_hip.h``); ``python -m esr_nerf_amd.build`` compiles it.
"""
