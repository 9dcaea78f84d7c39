"""Drop-in for the reference's ``model.py``: put this directory first on PYTHONPATH and the reference's main.py
(which does ``from model import ...``) runs against the MI355X engine unchanged."""
from speechsplit_amd.model import *  # noqa: F401,F403
from speechsplit_amd.model import Generator_3, Generator_6, InterpLnr  # noqa: F401,E402
