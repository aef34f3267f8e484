"""Algorithm constants of the crossover methods.

Same names and values as the reference's ``smart_crossover/parameters.py`` (lines 7-28); the HIP
kernels hard-code the ones that live inside device arithmetic (1e-6 floor, 1e-2 divisor, 1e6 cap,
see include/sxhip.h K3) and tests assert that both agree.
"""

# --- solution accuracy --------------------------------------------------------------------------
TOLERANCE_FOR_ARTIFICIAL_VARS = 1e-8      # an artificial arc counts as "unused" below this flow
TOLERANCE_FOR_REDUCED_COSTS = 1e-6        # dual feasibility tolerance of the pricing test

# --- network crossover (CNET / TNET) ------------------------------------------------------------
COLUMN_GENERATION_RATIO = 2               # growth factor of the released-column budget per round

# --- perturbation crossover ---------------------------------------------------------------------
OPTIMAL_FACE_ESTIMATOR = 1e-3             # gamma = gamma_dual at the first attempt
OPTIMAL_FACE_ESTIMATOR_UPDATE_RATIO = 1e-5
PERTURB_THRESHOLD = 1e-6                  # floor on the distance-to-bound used in the perturbation
CONSTANT_SCALE_FACTOR = 1e-2
PRIMAL_DUAL_GAP_THRESHOLD = 1e-8
PROJECTOR_THRESHOLD = 1e-8
PERTURB_UPPER_BOUND = 1e6
