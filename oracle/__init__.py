"""ORACLE -- test infrastructure only (CPU restatement of the reference's algorithms).

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, and only as the checker.  Nothing under drone-sim-python_amd/ imports it.
"""
