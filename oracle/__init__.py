"""CPU oracle for the MuopDB ANN hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (muopdb_amd/) must never do so.  See muopdb_oracle.cpp's header.
"""
from .oracle import *  # noqa: F401,F403
