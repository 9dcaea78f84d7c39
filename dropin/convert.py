"""The conversion step of the reference's demo.ipynb (cell 0) on the MI355X engine: see speechsplit_amd/convert.py."""
from speechsplit_amd.convert import *  # noqa: F401,F403
from speechsplit_amd.convert import CONDITIONS, convert_f0, demo_conversion  # noqa: F401,E402
