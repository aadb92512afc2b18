"""ft_mpc_amd -- host-side mirror of the reference's ft_mpc package for the MPC QP-step path.

Only what the hot path needs lives here: the ctypes binding to the gfx950
C-ABI library (include/ftmpc.h), the batched solver front-end, and `ft_mpc`-shaped model /
controller / simulation classes so that reference scripts (examples/sim.py) run unchanged
against the HIP path.  Nothing in this package imports oracle/.
"""
from ._lib import FtmpcError, load_library, library_path  # noqa: F401
from .batch import BatchedMPC, MPCConfig, make_synthetic_batch  # noqa: F401

__version__ = "0.1.0"
