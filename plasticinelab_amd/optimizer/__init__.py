from .optim import Adam, Momentum, Optimizer  # noqa: F401
from .solver import Solver, solve_action      # noqa: F401
from .solver_nn import SolverNN, Observation   # noqa: F401
