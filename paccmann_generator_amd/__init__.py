"""MI355X-native REINFORCE hot path of paccmann_generator (decode + policy-gradient BPTT).

Importing this package loads lib/libpmg.so (hand-written gfx950 HIP kernels behind the
C ABI of include/pmg.h).  There is no CPU fallback.
"""
from . import _ffi  # noqa: F401  (fails loudly when the HIP library is missing)
from .decoder import Generator, GreedySearch, SamplingSearch, StackGRUDecoder  # noqa: F401
from . import ops  # noqa: F401
from .hostpool import HostPool  # noqa: F401
from .reinforce import Reinforce, ReinforceMultiModalSets, ReinforceOmic, ReinforceProtein  # noqa: F401

__all__ = ['Generator', 'GreedySearch', 'SamplingSearch', 'StackGRUDecoder', 'ops', 'Reinforce', 'ReinforceOmic',
           'ReinforceProtein', 'ReinforceMultiModalSets', 'HostPool']
