"""mpc4quantum_amd - MI355X-native engine for the MPC hot path of andgoldschmidt/MPC4quantum.

Same public names as the reference package (mpc4quantum/__init__.py:3-7 star-exports experiment,
linearize, model, mpc, vectorize), backed by hand-written HIP kernels in libm4q_hip.so."""
from .experiment import (Experiment, LExperiment, QCoupledExperiment, QExperiment, QExperiment32, isqrt,  # noqa: F401
                         plant_step_batch, split_blocks)
from .library import (create_library, create_library_from_list, create_power_list, diff_library, krtimes,  # noqa: F401
                      multinomial_powers, size_of_library)
from .linearize import WrapModel  # noqa: F401
from .model import DMDc, DiscrepDMDc, OnlineDMDc  # noqa: F401
from .mpc import (StepClock, complex_to_real, complex_to_real_op, iqp_line_search, isinf_warning, mpc, mpc_batch,  # noqa: F401
                  real_to_complex, real_to_complex_op, shift_guess, val_to_str)
from .optimize import quad_program, quad_program_batch  # noqa: F401
from .session import EnsembleSession  # noqa: F401
from .vectorize import discretize_homogeneous, discretize_homogeneous_batch, liouvillian, vectorize_me  # noqa: F401

__version__ = "0.1.0"
