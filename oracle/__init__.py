"""CPU oracle for the FewBit quantized-activation path -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``fewbit_amd``) never does.
"""
from .oracle import *  # noqa: F401,F403
