"""MI355X-native simulate-and-render path of PIE-NeRF."""
import os

# The pipelined harness keeps 3 render streams + 1 simulator stream busy at once.  ROCm maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4, shared with the null stream and torch's side streams); two of our streams
# landing on one queue serialise a ~30-launch substep behind a whole render.  Give every stream its own queue.  Read by the
# HIP runtime when it initialises (first device call), so this must be set before that; an explicit user setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
