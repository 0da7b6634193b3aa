"""MI355X-native hot path of half-potato/nmf (model=microfacet_tensorf2); see DESIGN.md."""
import os as _os

# The training pass overlaps up to seven streams (main + six roles + the table rebuilds); the HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue serialise.  Eight queues: 1.61 -> 1.59 ms
# per step.  Read by the runtime when it initialises, so it has to be in the environment before the first HIP
# call of the process; an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
