"""Mirror of ft_mpc/controllers/tools/control_allocator.py:6-94 (thruster allocation) on the HIP path.

Same constructor and `get_physical_input(u_simple) -> ndarray[16]` as the reference's
`ControlAllocator`; the min-energy allocation QP (cvxpy/OSQP there, `:27-40,86`) runs in
`libftmpc_hip.so` (`ftmpc_allocate_batch`).  `get_physical_input_batch` is the batched form.

Differences, both deliberate:
  * the reference exit()s the process when the QP is not solved (`:88-93`); here a `ValueError`
    is raised for a single request and the per-instance status is returned for a batch;
  * `clip_generalized_input` (`:42-63`) projects onto the attainable hull with a 3x3 identity for a
    6-D variable and an unpinned solver (known reference defect); only its feasibility
    test is reproduced: an unattainable wrench is reported, not silently reshaped.
"""
import numpy as np

from ...batch import BatchedMPC, MPCConfig


class ControlAllocator:
    def __init__(self, model, bounds=None, device_id: int = 0):
        self.model = model
        self.faulty_force_generalized = np.asarray(model.faulty_force_generalized, float).flatten()
        self.input_bounds = bounds
        D = np.asarray(model.D, float)
        self._nt = D.shape[1]
        # the allocation only needs D from the handle; the horizon is irrelevant (smallest legal one)
        self._mpc = BatchedMPC(MPCConfig(N=1, NT=self._nt, D=D, device_id=device_id))

    def get_physical_input_batch(self, u_simple, u_ub=None):
        """u_simple [B,6] desired generalized forces (faulty wrench already removed, as the reference's
        controller passes them); u_ub [B,NT] or None (the model's current bounds for every row)."""
        tau = np.asarray(u_simple, float).reshape(-1, 6)
        if u_ub is None:
            u_ub = np.tile(np.asarray(self.model.u_ub_physical, float).flatten(), (tau.shape[0], 1))
        return self._mpc.allocate(tau, u_ub)

    def get_physical_input(self, u_simple):
        out = self.get_physical_input_batch(np.asarray(u_simple, float).reshape(1, 6))
        if int(out["status"][0]) != 0:
            raise ValueError(f"allocation not solved (status {int(out['status'][0])}): u_des={np.asarray(u_simple).flatten()}, "
                             f"u_ub={np.asarray(self.model.u_ub_physical).flatten()}")
        return out["u"][0]

    def close(self):
        self._mpc.close()
