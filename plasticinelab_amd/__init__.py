"""MI355X-native differentiable MPM engine behind the PlasticineLab simulator interface."""
__version__ = "0.1.0"
