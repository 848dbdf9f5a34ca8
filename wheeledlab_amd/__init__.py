"""wheeledlab_amd -- MI355X-native vectorised env.step() for the WheeledLab tasks (drift / elevation / visual).

Layout: ``csrc/`` hand-written HIP kernels for gfx950 + the C ABI (include/wheeledlab_amd.h); ``_abi.py`` the ctypes
binding; ``params.py`` task constants restated from the reference configs; ``envs/`` the ManagerBasedRLEnv-shaped
host surface.  There is no CPU path in this package.
"""
__version__ = "0.1.0"
