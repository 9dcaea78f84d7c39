"""Drop-in for the reference's ``data_loader.py``: put this directory first on PYTHONPATH and the reference's main.py
(which does ``from data_loader import ...``) runs against the MI355X engine unchanged."""
from speechsplit_amd.data_loader import *  # noqa: F401,F403
from speechsplit_amd.data_loader import get_loader  # noqa: F401,E402
