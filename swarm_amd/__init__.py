"""swarm_amd — MI355X-native (gfx950) amplicon neighbour-finding path of swarm, behind a C ABI.

  include/swarm_amd.h      the C ABI (the drop-in boundary)
  swarm_amd/csrc/          hand-written HIP kernels + the C-ABI implementation
  swarm_amd/capi.py        ctypes binding used by tests / bench / smoke

The product never imports oracle/ and has no CPU fallback.
"""
from .capi import Context, MultiContext, D0Clusters, D1Clusters, DnClusters, derep, HostDb, SwaError, d1_write_uclust, reduced_penalties, build_library, load_library  # noqa: F401
