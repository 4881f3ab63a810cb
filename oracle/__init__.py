"""CPU oracle for the CVAEMemRand hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.
"""
import os as _os
import sys as _sys

# the oracle borrows the synthetic config / seeded-weight helpers of the package (vsv/synth.py: no model
# arithmetic); the package never imports the oracle
_PKG = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vol-surface-vae-pub_amd")
if _PKG not in _sys.path:
    _sys.path.append(_PKG)          # appended, never first: make_goldens.py must keep importing the REFERENCE's `vae`
