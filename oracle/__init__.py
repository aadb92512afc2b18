"""CPU oracle for the MPC QP-step path.  TEST INFRASTRUCTURE ONLY -- see refmath.py."""
