"""lfpsqp.jl_amd -- MI355X-native hot path of LFPSQP (projected CG + retractions).

Host-side mirror of the reference's operator interface over the C ABI of
``liblfpsqp_hip.so`` (include/lfpsqp_hip.h).  Import as ``lfpsqp_jl_amd``.
"""
from ._capi import LfpsqpError, load_library, header_functions  # noqa: F401
from .device import (Context, DeviceMatrix, DeviceVector, SparseMatrix, amax, axpby, dot, gemv_n, gemv_t, nrm2, spmv_n,  # noqa: F401
                     spmv_t, vmul, waxpby)
from .projcg import DeviceBasis, DiagOperator, LowRankOperator, ProjCGWork, TridiagonalOperator, projcg_  # noqa: F401
from .factorize import gram, gram_rhs, ksvd_, orthonormalize_, rmul, small_svd_  # noqa: F401
from .inequality import (InequalityData, InequalityDecomp, InequalityDecompOp, InequalityDecompProject, StackedVector,  # noqa: F401
                         augmented_hess_diag_, calculate_h_, calculate_lambda_kkt_, generate_initial_y_, half_stride,
                         inequality_gradient_, y_retract_)
from .params import DeviceOptions, DisplayOption, LFPSQPParams, LinesearchOption, TerminationCondition, TerminationInfo  # noqa: F401
from .retractions import (NR, DeviceConstraints, ElementwiseConstraints, Euclidean, NRWork, YRetract, retract_, retract_nr_batch_, retract_nr_batch_width_,  # noqa: F401
                          sin_system_constraints, sphere_system_constraints)
from .projpenalty import ProjPenalty, ProjPenaltyWork, ProjPrecondition, no_precondition, pcg_, proj_precondition_  # noqa: F401
from .linesearch import ArmijoWork, ExactLinesearchWork, armijo_, exact_linesearch_  # noqa: F401
from .optimize import optimize_core  # noqa: F401
from .problems import ChainSeparableLinear, Derivatives, QuadLinearBallBox, SeparableElementwiseBox, SeparableLinearBallBox, optimize  # noqa: F401
