"""LFPSQPParams, enums and TerminationInfo -- a 1:1 mirror of reference src/LFPSQP.jl:27-81
(Greek field names transliterated: alpha, beta, t_beta, sigma, eps_c, ..., mu0, tn_kappa)."""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Callable, Optional


class DisplayOption(enum.Enum):
    off = 0
    iter = 1


class LinesearchOption(enum.Enum):
    armijo = 0
    exact = 1


class TerminationCondition(enum.Enum):
    f_tol = 0
    x_tol = 1
    kkt_tol = 2
    max_iter = 3
    armijo_error = 4


@dataclass
class TerminationInfo:
    condition: TerminationCondition
    f_diff: float
    step_diff: float
    kkt_diff: float
    iter: int

    def __str__(self):
        return (f"TerminationInfo:\ncondition = {self.condition.name}\n       Δf = {self.f_diff!r}\n"
                f"   ||Δx|| = {self.step_diff!r}\n||P(∇f)|| = {self.kkt_diff!r}\n    iters = {self.iter}")


@dataclass
class LFPSQPParams:
    alpha: float = 1.0
    beta: float = 0.0
    t_beta: int = 0
    s: float = 0.5
    sigma: float = 1e-4
    eps_c: float = 1e-6
    eps_f: float = 1e-6
    eps_x: float = 0.0
    eps_kkt: float = 1e-6
    eps_rank: float = 1e-10
    maxiter: int = 10000
    maxiter_retract: int = 100
    maxiter_pcg: int = 100
    mu0: float = 1e-2
    disable_linesearch: bool = False
    do_project_retract: bool = True
    disp: DisplayOption = DisplayOption.iter
    callback: Optional[Callable] = None
    callback_period: int = 100
    linesearch: LinesearchOption = LinesearchOption.armijo
    do_newton: bool = True
    tn_maxiter: int = 10000
    tn_kappa: float = 0.5


@dataclass
class DeviceOptions:
    """Options of the DEVICE implementation that have no counterpart in the reference -- kept out of LFPSQPParams, which mirrors
    src/LFPSQP.jl:57-81 field for field.  One instance per Context (``ctx.options``)."""
    # After the first failed retraction of an Armijo search (or in the shrinking phase of the exact search), the next `ls_batch` trial steps
    # (alpha*s, alpha*s^2, ...) are retracted together: they share every pass over the constraint gradients (lfpsqp_retract_nr_batch); the
    # search consumes them in the reference's order.  1 = off.  Only the Newton retraction with device-resident constraints batches;
    # everything else ignores it.
    # 0 (default) = automatic: as many as the previous search's failed retractions suggest (at least 4), up to what one pass takes for
    # the problem's shape and batch mode (lfpsqp_retract_nr_batch_width); k > 1 = at most k.
    ls_batch: int = 0
    # How the batched retractions are computed (lfpsqp_ctx_set_nr_batch_mode; applied at the start of `optimize`):
    #   False (default) = the EXACT batch: up to 4 trials per pass, every sum of a trial formed in the order of the single-trial step --
    #     flags, counts and iterates are BIT FOR BIT those of retracting one by one, so the search IS the one-by-one search, also where it
    #     is chaotic (config 4's failing searches: a trial that ends in its 100th Newton step or just inside it decides the accepted step).
    #   True (opt-in) = the matrix-core batch: up to 16 trials per pass (8 from 133 to 528 columns), faster (config 4 at full size: 9.5 s),
    #     equal to the one-by-one retractions up to rounding only -- in a chaotic search it MAY accept another step than the reference's
    #     order of arithmetic does and the run then forks onto another (equally valid) trajectory to the same optimum.
    ls_batch_matrix_cores: bool = False
    # ProjPenalty's inner pcg! solves with the EXACT preconditioner of their operator (lfpsqp_pcg_pre; the reference's proj_precondition!,
    # src/retractions.jl:248-257 -- its call is commented out at :374 -- generalised to the bound operator): one or two inner iterations
    # per Gauss-Newton step instead of hundreds to thousands, at one Gram pass per step.  False (default) = the reference's live path
    # (no_precondition): same iterates as the reference, iteration for iteration.
    pp_precondition: bool = False
    # candidate allocations per placement-tuned buffer (lfpsqp_ctx_set_placement; 1 = off)
    placement_tries: int = 3
    # the tangent basis stays in factored form U = Jct W (no n x m basis matrix, no basis-forming product in the tangent setup) whenever the
    # fused projected-CG iteration applies (diagonal Lagrangian Hessian, 4 .. 1024 constraints, dense constraint gradients); False: always
    # materialise Z = Jct W as rounds 1-2 did
    factored_basis: bool = True
    # the small eigenproblem of the tangent setup starts from the previous outer iteration's eigenvectors (lfpsqp_factorize_hint): one or two
    # Jacobi sweeps instead of eight.  Same Sigma / rank / span; the basis may differ from the cold call's by a rotation inside clusters of
    # equal singular values, to which every use is invariant.  False: every factorisation starts cold
    warm_factorize: bool = True
    # the tangent step of an outer iteration with a plain basis in factored form takes two passes over the constraint gradients fewer than the
    # statement-by-statement sequence (three fewer for the nonlinear class with streamed gradients): Jct'd rides with the Gram pass
    # (lfpsqp_factorize_rhs), and ONE pass projects the step, completes the Hessian diagonal and forms projcg!'s first U'r
    # (lfpsqp_tangent_step, LFPSQP_PROJCG_START_GIVEN).  False: src/optimize.jl:305-343 and src/projcg.jl:55-59 statement by statement
    fused_tangent_step: bool = True
    # a problem class with a TRIDIAGONAL Lagrangian Hessian (an ``offdiag`` vector next to ``diag_``) gets its truncated-Newton solves on the one-pass
    # iteration (lfpsqp_projcg_tridiag: U'AU by two or three Gram passes per solve, then 1.85 instead of 3.3 ms per iteration at (1e7, 128) -- ahead
    # from six to nine iterations per solve on).  False: the same operator through the callback path (lfpsqp_projcg_op), identical iterates
    tridiagonal_one_pass: bool = True
