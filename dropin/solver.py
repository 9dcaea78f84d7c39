"""Drop-in for the reference's ``solver.py``: put this directory first on PYTHONPATH and the reference's main.py
(which does ``from solver import ...``) runs against the MI355X engine unchanged."""
from speechsplit_amd.solver import *  # noqa: F401,F403
from speechsplit_amd.solver import Solver  # noqa: F401,E402
