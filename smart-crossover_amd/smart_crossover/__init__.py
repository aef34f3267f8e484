"""smart_crossover on MI355X: drop-in mirror of the reference package's hot-path API.

Module names follow the reference (formats, output, parameters, timer, lp_methods.*,
network_methods.*, solver_caller.*); ``smart_crossover.hip`` holds the binding of libsxhip.so.
"""
__version__ = "0.1.0"
