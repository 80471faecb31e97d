"""zeroshotsemanticsegmentation_amd -- MI355X-native SZN pixel-embedding training path.

Host-side mirror of the reference's module surface (models, utils, trainer_fcn, trainer_seenmask, train,
configs) over hand-written HIP kernels (csrc/, C-ABI in include/szn.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
