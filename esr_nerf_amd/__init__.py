"""esr-hip: MI355X-native (gfx950) implementation of ESR-NeRF's volumetric-rendering hot path.

Drop-in renderers with the reference's API: ``voxurfc.VoxurfC`` (coarse stage), ``voxurff.VoxurfF`` (fine stage),
``esrnerf.ESRNeRF`` (lts / pdra stages); ``render_utils`` mirrors the reference's pybind extension modules;
``optimizer`` its Adam; ``trainer`` holds the autograd-free training steps with data parallelism.  All arithmetic on
the path runs in ``libesr_hip.so`` (``include/esr_hip.h``); ``python -m esr_nerf_amd.build`` compiles it.
"""
import os as _os

# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With an RCCL communicator alive its
# streams take queues too and the engine's side stream (weight gradients beside the grid scatters, packing beside the
# march) lands on the main stream's in-order queue: everything serialises (+0.12 ms per C2 step).  The
# variable is read when HIP initialises, so it is set here, at package import, unless the deployment chose a value;
# a process that touched the GPU before importing this package must export it itself (trainer.py warns).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
