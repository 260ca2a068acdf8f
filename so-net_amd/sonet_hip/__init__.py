"""sonet_hip -- host side of the MI355X-native SO-Net hot path.

``ops``   tensor-level wrappers over the C ABI of libsonet_hip.so (include/sonet_hip.h)
``synth`` synthetic ModelNet40-shaped inputs and seeded weights (bench / smoke / fixtures)
``dp``    one-process-per-GPU data-parallel helpers (RCCL gradient all-reduce, batch sharding)

Importing the package does not load the library; the first op does, and raises if it is missing
or if the device is not a gfx950 -- there is no CPU path.
"""
__all__ = ["ops", "synth", "dp"]
