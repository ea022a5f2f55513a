"""simulator package of the reference (simulator/solver.py): the HIP-backed Simulator under the same import path."""
