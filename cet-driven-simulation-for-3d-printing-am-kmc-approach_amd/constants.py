"""Physical / numerical constants of the W-Re-C CET kinetic Monte Carlo model.

Drop-in for the reference's ``constants.py`` (same names, same values: they are part
of the parity contract -- reference constants.py:34-148).  Written fresh; grouped by
role rather than in the reference's order.
"""

# --- lattice-site identifiers (reference constants.py:34-41) -----------------------
STATES = {"Empty": 0, "W": 1, "Re": 2, "C": 3, "Defect": 4}
DEFECT_ID = STATES["Defect"]

COLORS = {
    STATES["Empty"]: (1.0, 1.0, 1.0),
    STATES["W"]: (0.2, 0.6, 1.0),
    STATES["Re"]: (0.8, 0.5, 0.2),
    STATES["C"]: (0.1, 0.8, 0.1),
    STATES["Defect"]: (1.0, 0.0, 0.0),
}

# --- run shape (constants.py:54-59) -------------------------------------------------
LATTICE_SIZE = 30
VOXEL_SIZE = 5e-6            # [m]
N_STEPS = 20000
METRIC_UPDATE_STEP = 200
VISUAL_UPDATE_STEP = 2000
N_SEEDS = 20

# --- thermodynamics (constants.py:63-67) --------------------------------------------
K_T = 8.617333262e-5         # [eV/K]
T_MELT = 3695                # [K]
T_SUB = 2800                 # [K]
ATOMIC_SPACING_W = 2.74e-10  # [m]

# --- attempt frequencies and energies (constants.py:70-90) --------------------------
NU = 1e13
NU_DEP = 2e13
E_B_W, E_DIFF_W = 3.8, 0.35
E_B_RE, E_DIFF_RE = 4.2, 0.50
IMPURITY_RE = 0.10
E_B_C, E_DIFF_C = 3.2, 0.30
CARBON_SOLUTION_ENERGY = 0.25
CARBON_MIGRATION_ENERGY = 0.15
IMPURITY_C = 0.20
MAX_IMP_FRACTION = 1.0

# --- derived growth scales (constants.py:96-102) ------------------------------------
G = (T_MELT - T_SUB) / (LATTICE_SIZE * VOXEL_SIZE)
R_VOX = (NU_DEP * ATOMIC_SPACING_W) / VOXEL_SIZE
R_SI = NU_DEP * ATOMIC_SPACING_W
ANISOTROPY_FACTOR = 0.25

# --- CET classification (constants.py:107-112) --------------------------------------
CET_EQ_THRESHOLD = 0.50
CET_AR_THRESHOLD = 3.0
CET_GR_THRESHOLD = 5e6
CRITICAL_GRAIN_DENSITY = 1e5
CET_CHECK_INTERVAL = 100
PLOT_INTERVAL = 500

# --- demo kinetics / nucleation model (constants.py:117-131) ------------------------
NU_EVAP = 1e3
NU_DIFF = 5e2
NU_NUCLEATION = 5e-3
EVAP_ACT_ENERGY = 0.70
DIFF_ACT_ENERGY = 0.40
NUCLEATION_ENERGY = 0.80
DELTA_T_C = 10
E_STICK = 0.10
S0_STICK = 0.50
DELTA_T_REF = 50.0
I0 = 5e13
K_NUC = 500
BETA_IMP_NUC = 0.4

# --- defect model (constants.py:136-140) --------------------------------------------
DEFECT_FORMATION_RATE = 1e-4
DEFECT_PROB = 3e-3
DEFECT_PROB_BASE = 0.12
DEFECT_DEP_REDUCTION = 0.8
DEFECT_EVAP_BOOST = 2.5

# --- guards (constants.py:145-147) --------------------------------------------------
RATE_THRESHOLD = 1e-30
EPSILON = 1e-10
RANDOM_SEED = 42
