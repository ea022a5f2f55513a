"""Mirror of the reference's ``simulator`` package: ``from simulator.solver import Simulator``."""
