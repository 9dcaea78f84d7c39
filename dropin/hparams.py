"""Drop-in for the reference's ``hparams.py``: put this directory first on PYTHONPATH and the reference's main.py
(which does ``from hparams import ...``) runs against the MI355X engine unchanged."""
from speechsplit_amd.hparams import *  # noqa: F401,F403
from speechsplit_amd.hparams import hparams, hparams_debug_string  # noqa: F401,E402
