"""Drop-in for the reference's ``utils.py``: put this directory first on PYTHONPATH and the reference's main.py
(which does ``from utils import ...``) runs against the MI355X engine unchanged."""
from speechsplit_amd.utils import *  # noqa: F401,F403
