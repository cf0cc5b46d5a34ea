"""CPU oracle for the VidCom2 hot path -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (vidcom2_amd/) never imports this.
"""
from .oracle import *  # noqa: F401,F403
