"""The synthetic workload shapes live with the product's bench-side code (sirius_amd/workloads.py) since r04; the tests keep importing
them under the old name."""
from sirius_amd.workloads import *          # noqa: F401,F403
from sirius_amd.workloads import gates_for, make_structure_inputs, make_support_inputs, rand_fe, sangria_shape, support_gate, trace_like   # noqa: F401
