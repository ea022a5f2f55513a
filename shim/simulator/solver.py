from pienerf_amd.simulator.solver import Simulator, npfloat, torchfloat  # noqa: F401  (main_gui.py:8, main_sim.py: from simulator.solver import Simulator)
