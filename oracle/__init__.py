"""CPU oracle (test infrastructure only) -- see oracle/spec.py header.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports it.
"""
