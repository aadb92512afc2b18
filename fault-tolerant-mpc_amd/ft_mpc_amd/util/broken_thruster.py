"""Mirror of ft_mpc/util/broken_thruster.py: a broken thruster is (index, intensity in [0,1]);
intensity*f_max is the force it keeps producing."""
from dataclasses import dataclass


@dataclass
class BrokenThruster:
    index: int
    intensity: float
