from .optim import Adam, Momentum, Optimizer  # noqa: F401
from .solver import Solver, solve_action      # noqa: F401
